#!/bin/bash
# Round 4: grid of the product (option grid_blocks) and of the vector kernels (k_ortho2, k_combine: vec_grid_blocks) on the c2
# recipe (banded, 10^6 rows), and on the resident Goutsias run (SELL, N -> 10^6) through KFSP_OPTIONS
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
for o in "" "grid_blocks=256" "grid_blocks=384" "grid_blocks=512" "grid_blocks=768" "vec_grid_blocks=256" "vec_grid_blocks=512" "grid_blocks=512 vec_grid_blocks=512" "grid_blocks=512 vec_grid_blocks=256"; do
  timeout -k 10 200 python profiles/expv_c2.py 40 $o 2>&1 | tail -1 | cut -c1-215
done
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for o in "grid_blocks=0" "grid_blocks=512" "grid_blocks=768" "grid_blocks=512,vec_grid_blocks=512" "vec_grid_blocks=512" "grid_blocks=256"; do
  KFSP_SSA_STREAMS=1 KFSP_OPTIONS="$o" timeout -k 10 120 $D solve goutsias_input /tmp/p.bin 300.0 > $O/s_run.log 2>&1
  echo "== resident Goutsias T=300, $o"; grep -E "KFSP WALL|UNKNOWN" $O/s_run.log | cut -c1-250
done
cd $R
