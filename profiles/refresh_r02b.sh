#!/bin/bash
# Round-2 evidence for the matrix-free product and the SELL product on a real FSP: rocprofv3 kernel
# stats of `bench.py --workload c3x --matrix-free`, PMC FETCH_SIZE / WRITE_SIZE of profiles/pmc_box_fsp.py.
set -e
R=$PWD
O=$R/gpurun_out/r02
B=$R/krylovfspssa_amd/fortran/_build
mkdir -p $O
(cd /tmp && /opt/rocm/lib/llvm/bin/flang -O3 -fopenmp -I$B $R/profiles/statespace_bench.f90 $B/libkfsp_fortran.a -L$R/krylovfspssa_amd/lib -lkfsp_hip -Wl,-rpath,$R/krylovfspssa_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib -o /tmp/ssb)
KFSP_HOST_THREADS=16 /tmp/ssb 2.0 38 /tmp/fsp.bin | tail -1
python bench.py > $O/bench_c3_final.json 2> $O/bench_c3_final.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o mf -- python3 $R/bench.py --workload c3x --matrix-free --no-expv --no-cpu > $O/bench_c3x_mf.json 2> $O/bench_c3x_mf.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof -o pf -- python3 $R/profiles/pmc_box_fsp.py > $O/pmc_box_fsp.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof -o pw -- python3 $R/profiles/pmc_box_fsp.py >> $O/pmc_box_fsp.log 2>&1
cd $R
cp $(find /tmp/prof -name "mf_kernel_stats.csv" | head -1) $O/c3x_matrix_free_kernel_stats.csv
cp $(find /tmp/prof -name "pf_counter_collection.csv" | head -1) $O/pmc_fetch_box_fsp.csv
cp $(find /tmp/prof -name "pw_counter_collection.csv" | head -1) $O/pmc_write_box_fsp.csv
python3 - $O/pmc_fetch_box_fsp.csv $O/pmc_write_box_fsp.csv <<'PY' | tee $O/pmc_box_fsp_summary.txt
import csv, sys
def by_dispatch(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_spmv" in r["Kernel_Name"]:
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:48], float(r["Counter_Value"])))
    return sorted(out)
f, w = by_dispatch(sys.argv[1], "FETCH_SIZE"), by_dispatch(sys.argv[2], "WRITE_SIZE")
# three groups of 5 launches (box, fsp discovery order, fsp internal order), in dispatch order; medians
def groups(v):
    ks = [x for x in v]
    return [ks[i:i + 5] for i in range(0, len(ks), 5)]
for name, gf, gw in zip(("matrix-free c3x (1.0e7 states)", "SELL-64, Goutsias FSP in discovery order", "SELL-64, same FSP in the internal order"), groups(f), groups(w)):
    rd = sorted(x[2] for x in gf)[len(gf) // 2] * 1024 * 2.0      # FETCH_SIZE in KiB, x2 on gfx950 (guides/MI355X_MICROARCH.md)
    wr = sorted(x[2] for x in gw)[len(gw) // 2] * 1024
    print(f"{name}: kernel {gf[0][1]}  read {rd/1e6:.1f} MB  write {wr/1e6:.1f} MB  total {(rd+wr)/1e6:.1f} MB per launch")
PY
cat $O/pmc_box_fsp.log | grep -v "^\s*$" | tail -8
head -3 $O/c3x_matrix_free_kernel_stats.csv
