#!/bin/bash
# Round 3, first measurement pass: HIP-event timings, rocprofv3 kernel stats and FETCH_SIZE / WRITE_SIZE traffic of
# the generator product on the box workloads (stored / matrix-free, c3x .. c5 at full size) and on the 1.0e7-state
# NON-BOX Goutsias FSP (plain SELL in search order, plain SELL and coded SELL in the internal order).
# Run from the repo root through gpurun; summaries land in gpurun_out/r03/ and are copied to profiles/r03_*.
set -e
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
CASES="${CASES:-c3x:stored c3x:mf c5s:stored c5s:mf c5:stored c5:mf fsp:sell_search fsp:sell fsp:coded}"
python3 profiles/pmc_target.py $CASES > $O/timing_$1.log 2>&1
grep -E "CASE|fsp generated" $O/timing_$1.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o ks -- python3 $R/profiles/pmc_target.py --no-time $CASES > $O/ks_$1.log 2>&1
cp $(find /tmp/prof_k -name "ks_kernel_stats.csv" | head -1) $O/kernel_stats_$1.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o pf -- python3 $R/profiles/pmc_target.py --calib --no-time $CASES > $O/pf_$1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o pw -- python3 $R/profiles/pmc_target.py --calib --no-time $CASES > $O/pw_$1.log 2>&1
cd $R
cp $(find /tmp/prof_f -name "pf_counter_collection.csv" | head -1) $O/pmc_fetch_$1.csv
cp $(find /tmp/prof_w -name "pw_counter_collection.csv" | head -1) $O/pmc_write_$1.csv
python3 profiles/pmc_reduce_r03.py $O/pmc_fetch_$1.csv $O/pmc_write_$1.csv $O/timing_$1.log $CASES > $O/pmc_summary_$1.txt
cat $O/pmc_summary_$1.txt
