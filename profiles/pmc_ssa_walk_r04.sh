#!/bin/bash
# Round 4: hardware counters of the SSA walk kernel over the 272 expansions of the resident Goutsias T = 300 run
# (KFSP_SSA_STREAMS=1).  One rocprofv3 --pmc pass per counter group, kernels restricted to k_ssa_walk.
O=$PWD/gpurun_out/r04
mkdir -p $O/pmc_walk
cd tests/golden/models
export TMPDIR=/tmp KFSP_CASE_CAPACITY=2097169 KFSP_SSA_STREAMS=1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "k_ssa_walk" --output-format csv -d $O/pmc_walk -o g$i -- ../../../krylovfspssa_amd/fortran/_build/kfsp_dump solve goutsias_input $O/pmc_walk.bin 300.0 > $O/pmc_walk/g$i.log 2>&1 || echo "group $i failed: $grp"
  echo "group $i done: $grp"
done
rm -f $O/pmc_walk.bin $O/pmc_walk.bin.in
cd $O/pmc_walk
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
calls = 0
for f in sorted(glob.glob('**/*counter_collection.csv', recursive=True) + glob.glob('*counter_collection.csv')):
    seen = set()
    for r in csv.DictReader(open(f)):
        tot[r['Counter_Name']] += float(r['Counter_Value'])
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.4e}")
PY
