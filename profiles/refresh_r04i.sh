#!/bin/bash
# Round 4: the speculative rebuild of the resident FSP (option build_speculate): its tests, then the resident Goutsias run
# (T = 300) with the option off and on (kernel statistics of the run with it on: profiles/r04_e2e_resident_goutsias_kernel_stats.csv).
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_expand.py tests/test_gpu_drop.py tests/test_gpu_edge_cases.py -m gpu -x -q > $O/i_tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/i_tests.log
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for rep in 1 2; do
for s in 0 1; do
  KFSP_SSA_STREAMS=1 KFSP_OPTIONS="build_speculate=$s" timeout -k 10 120 $D solve goutsias_input /tmp/p$s.bin 300.0 > $O/spec_$s.log 2>&1
  echo "== resident Goutsias T=300, build_speculate=$s"; grep -E "KFSP WALL|KFSP RESIDENT REBUILDS" $O/spec_$s.log | cut -c1-250
done
done
cmp /tmp/p0.bin /tmp/p1.bin && echo "dumps identical"
cd $R
