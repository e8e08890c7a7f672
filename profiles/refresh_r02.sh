#!/bin/bash
# Round-2 measured artefacts (run from the repo root through gpurun); outputs land in
# gpurun_out/r02/ and the summaries are copied into profiles/ by hand.
set -e
R=$PWD
O=$R/gpurun_out/r02
mkdir -p $O/prof
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload c3x --no-expv --no-cpu > $O/bench_c3x.json 2> $O/bench_c3x.err
python bench.py --workload c3 --variant 2 --no-expv --no-cpu > $O/bench_c3_sell.json 2> $O/bench_c3_sell.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c3x -- python3 $R/bench.py --workload c3x --no-expv --no-cpu > $O/bench_c3x_prof.json 2> $O/bench_c3x_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c3 -- python3 $R/bench.py --no-cpu > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch_c3x -- python3 $R/profiles/pmc_calib.py c3x > $O/pmc_fetch_c3x.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write_c3x -- python3 $R/profiles/pmc_calib.py c3x > $O/pmc_write_c3x.log 2>&1
cd $R
for n in c3x c3; do find $O/prof -name "${n}_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${n}_kernel_stats.csv; done
find $O/prof -name "pmc_fetch_c3x_counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_fetch_c3x.csv
find $O/prof -name "pmc_write_c3x_counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_write_c3x.csv
rm -rf $O/prof
cat $O/bench_c3.json
