#!/bin/bash
# Round 4: how long the host Pade's workers spin before they sleep (KFSP_PADE_SPIN_US) - the resident Goutsias run (T = 300)
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for rep in 1 2; do
for u in 0 800 3000 10000; do
  KFSP_SSA_STREAMS=1 KFSP_PADE_SPIN_US=$u timeout -k 10 120 $D solve goutsias_input /tmp/p$u.bin 300.0 > $O/n_$u.log 2>&1
  echo "== resident Goutsias T=300, KFSP_PADE_SPIN_US=$u"; grep -E "KFSP WALL|KFSP HOST PADE" $O/n_$u.log | cut -c1-250
done
done
cmp /tmp/p0.bin /tmp/p3000.bin && echo "dumps identical"
cd $R
