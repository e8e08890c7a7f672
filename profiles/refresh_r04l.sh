#!/bin/bash
# Round 4: where the HOST spends the resident Goutsias run - HIP API statistics (rocprofv3 --hip-trace --stats, no counters)
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169 TMPDIR=/tmp
KFSP_SSA_STREAMS=1 timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/hip_prof -o hip -- $D solve goutsias_input /tmp/p2.bin 300.0 > $O/hip_prof.log 2>&1
grep -E "KFSP WALL" $O/hip_prof.log | cut -c1-250
ls $O/hip_prof | head
f=$(ls $O/hip_prof/*hip_api_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -30 $f | cut -c1-200
cd $R
