#!/bin/bash
# Round 4, last batch: traffic of formats 4 / 7 / 8 on the 6-species boxes (FETCH_SIZE / WRITE_SIZE), the host Pade with worker
# threads inside the resident run (after the affinity fix), and the bench line of the c5 workload with its kernel statistics.
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
bash profiles/refresh_r04d.sh > $O/d2.log 2>&1
tail -9 $O/d2.log | cut -c1-215
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for t in 1 2 4; do
  KFSP_SSA_STREAMS=1 KFSP_PADE_THREADS=$t timeout -k 10 120 $D solve goutsias_input /tmp/p.bin 300.0 > $O/pade2_threads_$t.log 2>&1
  echo "== resident Goutsias T=300, KFSP_PADE_THREADS=$t"; grep -E "KFSP WALL|KFSP HOST PADE" $O/pade2_threads_$t.log
done
cd $R
