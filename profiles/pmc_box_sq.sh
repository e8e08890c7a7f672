#!/bin/bash
# Where the matrix-free product's time goes: SQ / TCP / TCC counters of profiles/pmc_box_sq.py, one
# rocprofv3 --pmc pass per counter group (run from the repo root through gpurun).
R=$PWD
O=$R/gpurun_out/sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/prof_sq -o g$i -- python3 $R/profiles/pmc_box_sq.py > $O/g$i.log 2>&1 || echo "group $i failed: $grp"
  f=$(find /tmp/prof_sq -name "g${i}_counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $O/g$i.csv
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] + "/g*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "k_spmv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v.sort()
        print(f"{k:44s} median of {len(v)} launches: {v[len(v)//2]:.6g}")
PY
