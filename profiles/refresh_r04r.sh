#!/bin/bash
# Round 4: the bit-map filter in front of the walk's table (option ssa_filter): its tests, then the resident Goutsias run with it off / on
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_expand.py tests/test_fortran_host.py -m gpu -x -q -k "not full_horizon" > $O/p_tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/p_tests.log | cut -c1-200
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for rep in 1 2; do
for s in 0 1; do
  KFSP_SSA_STREAMS=1 KFSP_OPTIONS="ssa_filter=$s" timeout -k 10 120 $D solve goutsias_input /tmp/p$s.bin 300.0 > $O/filter_$s.log 2>&1
  echo "== resident Goutsias T=300, ssa_filter=$s"; grep -E "KFSP WALL" $O/filter_$s.log | cut -c1-250
done
done
cmp /tmp/p0.bin /tmp/p1.bin && echo "dumps identical"
cd $R
