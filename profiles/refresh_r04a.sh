#!/bin/bash
# Round 4, first measurements: where a c2 exp(tA)v step (10^6 states, m = 30: the launch-sensitive regime)
# spends its time - kernels vs the gaps between dependent launches (rocprofv3 --kernel-trace -> profiles/arnoldi_gaps.py).
# Run from the repo root through gpurun.
set -e
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
python3 profiles/expv_c2.py 40 > $O/expv_c2.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o c2 -- python3 $R/profiles/expv_c2.py 40 > $O/expv_c2_prof.log 2>&1
cd $R
cp $(find /tmp/prof_c2 -name "c2_kernel_stats.csv" | head -1) $O/expv_c2_kernel_stats.csv
python3 profiles/arnoldi_gaps.py $(find /tmp/prof_c2 -name "c2_kernel_trace.csv" | head -1) 0.5 > $O/expv_c2_gaps.txt
cat $O/expv_c2.log $O/expv_c2_gaps.txt
