#!/bin/bash
# PMC traffic of the banded product on c3 after the 16-byte x gathers (round 2, late): the two
# separate --pmc passes of profiles/pmc_calib.py, reduced into profiles/pmc_traffic.json.
set -e
R=$PWD
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_c -o pmc_fetch_c3 -- python3 $R/profiles/pmc_calib.py c3 > $O/pmc_fetch_c3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_c -o pmc_write_c3 -- python3 $R/profiles/pmc_calib.py c3 > $O/pmc_write_c3.log 2>&1
cd $R
cp $(find /tmp/prof_c -name "pmc_fetch_c3_counter_collection.csv" | head -1) $O/pmc_fetch_c3.csv
cp $(find /tmp/prof_c -name "pmc_write_c3_counter_collection.csv" | head -1) $O/pmc_write_c3.csv
python3 profiles/pmc_reduce.py c3 $O/pmc_fetch_c3.csv $O/pmc_write_c3.csv
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python3 -c "
import json; print(json.load(open('profiles/pmc_traffic.json'))['c3'])"
