#!/bin/bash
# Round 3: the reference's heaviest example workload (models/goutsias_model.input, T = 300, FSPTOL 1e-6, KRYTOL 1e-8,
# N -> 1e6; 2186 s on one CPU core for the reference) through the drop-in Fortran host (kfsp_dump solve), where the
# wall time goes (the solver's own KFSP lines):
#   default     the reference's sampling order: the sequential SSA walk on the host
#   streams     KFSP_SSA_STREAMS=1, walked by the host's thread team
#   streams_dev KFSP_SSA_STREAMS=1, the paths walked on the device (kfsp_ssa_streams), propensity columns from the
#               device program; the same states in the same order as `streams` (tests/test_fortran_host.py)
#   ranks2      default mode over 2 loop-back ranks (KFSP_NRANKS=2)
O=$PWD/gpurun_out/r03
mkdir -p $O
cd tests/golden/models
D=../../../krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
run() { name=$1; shift; t0=$(date +%s.%N); env "$@" $D solve goutsias_input $O/e2e_$name.bin 300.0 > $O/e2e_$name.log 2>&1; t1=$(date +%s.%N); \
        echo "== $name: $@  (process wall $(python3 -c "print(round($t1 - $t0, 2))") s)"; grep -E "KFSP|FINAL" $O/e2e_$name.log; }
run default KFSP_NOTHING=1
run streams KFSP_SSA_STREAMS=1 KFSP_DEVICE_SSA=0
run streams_dev KFSP_SSA_STREAMS=1
run ranks2 KFSP_NRANKS=2
rm -f $O/e2e_*.bin $O/e2e_*.bin.in
