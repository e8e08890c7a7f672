#!/bin/bash
# Round 4: one launch clears H and the breakdown flag of an Arnoldi pass - parity tests, the c2 exp(tA)v recipe, the resident run
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_loopback.py tests/test_lockstep.py tests/test_gpu_expand.py tests/test_gpu_drop.py tests/test_gpu_sell_code.py tests/test_gpu_state_order.py tests/test_gpu_group.py -m gpu -x -q > $O/q_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/q_tests.log | cut -c1-200
timeout -k 10 300 python profiles/expv_c2.py 40 > $O/q_expv_c2.log 2>&1; tail -3 $O/q_expv_c2.log | cut -c1-200
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for rep in 1 2; do
  KFSP_SSA_STREAMS=1 timeout -k 10 120 $D solve goutsias_input /tmp/p$rep.bin 300.0 > $O/q_$rep.log 2>&1
  echo "== resident Goutsias T=300"; grep -E "KFSP WALL|FINAL" $O/q_$rep.log | cut -c1-250
done
cd $R
