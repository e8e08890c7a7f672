#!/bin/bash
# Round 4: the pencil product (format 7) against format 4 on the 6-species boxes: HIP-event timing, FETCH_SIZE / WRITE_SIZE.
set -e
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
CASES="${CASES:-c5s:mf4 c5s:mf c5s:mf8 c5:mf4 c5:mf c5:mf8}"
python3 profiles/pmc_target.py $CASES > $O/timing_pencil.log 2>&1
grep -E "CASE" $O/timing_pencil.log | cut -c1-120
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f2 -o pf -- python3 $R/profiles/pmc_target.py --calib --no-time $CASES > $O/pf_pencil.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w2 -o pw -- python3 $R/profiles/pmc_target.py --calib --no-time $CASES > $O/pw_pencil.log 2>&1
cd $R
cp $(find /tmp/prof_f2 -name "pf_counter_collection.csv" | head -1) $O/pmc_fetch_pencil.csv
cp $(find /tmp/prof_w2 -name "pw_counter_collection.csv" | head -1) $O/pmc_write_pencil.csv
python3 profiles/pmc_reduce_r04.py $O/pmc_fetch_pencil.csv $O/pmc_write_pencil.csv $O/timing_pencil.log $CASES > $O/pmc_summary_pencil.txt
cat $O/pmc_summary_pencil.txt
