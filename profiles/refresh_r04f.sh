#!/bin/bash
# Round 4: the host Pade inside the resident Goutsias run for 1 / 4 / 8 worker threads (KFSP HOST PADE PARTS line).
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for t in 1 4 8; do
  t0=$(date +%s.%N); KFSP_SSA_STREAMS=1 KFSP_PADE_THREADS=$t $D solve goutsias_input /tmp/p.bin 300.0 > $O/pade_threads_$t.log 2>&1; t1=$(date +%s.%N)
  echo "== KFSP_PADE_THREADS=$t (process wall $(python3 -c "print(round($t1 - $t0, 2))") s)"; grep -E "KFSP WALL|KFSP HOST PADE" $O/pade_threads_$t.log
done
for t in 1 4; do
  KFSP_PADE_THREADS=$t $D solve toggle_example /tmp/p.bin 100.0 > $O/pade_toggle_$t.log 2>&1
  echo "== toggle_example T=100 KFSP_PADE_THREADS=$t"; grep -E "KFSP WALL|KFSP HOST PADE" $O/pade_toggle_$t.log
done
