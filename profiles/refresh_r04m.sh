#!/bin/bash
# Round 4: the host Pade with the substitution of its solve dealt to the pool's threads - the resident Goutsias run (T = 300)
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_padm_bits.py tests/test_abi_symbols.py -q > $O/m_tests.log 2>&1
echo "tests rc=$?"; tail -2 $O/m_tests.log
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for t in 1 4 6 8; do
  KFSP_SSA_STREAMS=1 KFSP_PADE_THREADS=$t timeout -k 10 120 $D solve goutsias_input /tmp/p$t.bin 300.0 > $O/m_$t.log 2>&1
  echo "== resident Goutsias T=300, KFSP_PADE_THREADS=$t"; grep -E "KFSP WALL|KFSP HOST PADE|FINAL" $O/m_$t.log | cut -c1-250
done
cmp /tmp/p1.bin /tmp/p4.bin && cmp /tmp/p1.bin /tmp/p6.bin && cmp /tmp/p1.bin /tmp/p8.bin && echo "dumps identical for 1 / 4 / 6 / 8 threads"
cd $R
