#!/bin/bash
# Round 4, closing batch: the whole GPU suite, the resident Goutsias run, the default bench line.
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > $O/k_tests.log 2>&1
echo "tests rc=$?"; tail -22 $O/k_tests.log | cut -c1-200
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
KFSP_SSA_STREAMS=1 timeout -k 10 120 $D solve goutsias_input /tmp/p1.bin 300.0 > $O/k_goutsias.log 2>&1
grep -E "KFSP WALL|KFSP RESIDENT REBUILDS|FINAL" $O/k_goutsias.log | cut -c1-250
export TMPDIR=/tmp
KFSP_SSA_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- $D solve goutsias_input /tmp/p2.bin 300.0 > $O/k_prof.log 2>&1
cp $(find /tmp/prof_k -name "k_kernel_stats.csv" | head -1) $O/k_resident_goutsias_kernel_stats.csv
head -12 $O/k_resident_goutsias_kernel_stats.csv | cut -c1-140
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/k_smoke.log 2>&1; tail -1 $O/k_smoke.log
timeout -k 10 300 python bench.py > $O/k_bench.json 2> $O/k_bench.err
echo "bench rc=$?"; cut -c1-600 $O/k_bench.json
