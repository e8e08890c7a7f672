#!/bin/bash
# Round 4: does a small-tile trip order cut the fabric traffic of the matrix-free product on the 6-species boxes, and does the
# time follow?  HIP-event timing + FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_target.py form@cut,B) reduced by
# profiles/pmc_reduce_r03.py.  Run from the repo root through gpurun.
set -e
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
CASES="${CASES:-c5s:mf c5s:mf@3,4096 c5:mf c5:mf@3,1024 c5:mf@4,16384}"
python3 profiles/pmc_target.py $CASES > $O/timing_tiles.log 2>&1
grep -E "CASE" $O/timing_tiles.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o pf -- python3 $R/profiles/pmc_target.py --calib --no-time $CASES > $O/pf_tiles.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o pw -- python3 $R/profiles/pmc_target.py --calib --no-time $CASES > $O/pw_tiles.log 2>&1
cd $R
cp $(find /tmp/prof_f -name "pf_counter_collection.csv" | head -1) $O/pmc_fetch_tiles.csv
cp $(find /tmp/prof_w -name "pw_counter_collection.csv" | head -1) $O/pmc_write_tiles.csv
python3 profiles/pmc_reduce_r03.py $O/pmc_fetch_tiles.csv $O/pmc_write_tiles.csv $O/timing_tiles.log $CASES > $O/pmc_summary_tiles.txt
cat $O/pmc_summary_tiles.txt
