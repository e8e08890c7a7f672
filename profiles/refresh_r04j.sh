#!/bin/bash
# Round 4: speculation under the row partition (loop-back ranks): the group / partition / expansion tests, then the resident
# Goutsias run over 2 loop-back ranks.
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_expand.py tests/test_gpu_drop.py tests/test_gpu_group.py tests/test_gpu_loopback.py tests/test_fortran_host.py -m gpu -x -q -k "not full_horizon and not digest" > $O/j_tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/j_tests.log
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for s in 0 1; do
  KFSP_SSA_STREAMS=1 KFSP_NRANKS=2 KFSP_OPTIONS="build_speculate=$s" timeout -k 10 200 $D solve goutsias_input /tmp/q$s.bin 300.0 > $O/spec2_$s.log 2>&1
  echo "== resident Goutsias T=300 over 2 loop-back ranks, build_speculate=$s"; grep -E "KFSP WALL|KFSP RESIDENT REBUILDS" $O/spec2_$s.log | cut -c1-250
done
cmp /tmp/q0.bin /tmp/q1.bin && echo "dumps identical"
cd $R
