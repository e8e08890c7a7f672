#!/bin/bash
# Round 4: the product's grid (grid_blocks) on the resident Goutsias run: times AND what the run did (the dot products' partial sums
# change with the grid, and with them - in the last bits - H and every decision that hangs on it)
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for o in "grid_blocks=0" "grid_blocks=1024" "grid_blocks=768" "grid_blocks=512" "grid_blocks=640" "grid_blocks=896"; do
  KFSP_SSA_STREAMS=1 KFSP_OPTIONS="$o" timeout -k 10 120 $D solve goutsias_input /tmp/p.bin 300.0 > $O/t_run.log 2>&1
  echo "== resident Goutsias T=300, $o"; grep -E "KFSP WALL|KFSP STATS|FINAL|UNKNOWN" $O/t_run.log | cut -c1-250
done
cd $R
