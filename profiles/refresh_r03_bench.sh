#!/bin/bash
# Round 3: the driver's bench command (python bench.py, defaults) and the rocprofv3 kernel statistics of the SAME command;
# the c5 line; the c3x line.  Run from the repo root through gpurun.
set -e
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 $R/bench.py > $O/bench_c3_prof.json 2> $O/bench_c3_prof.err
cd $R
cp $(find /tmp/prof_b -name "b_kernel_stats.csv" | head -1) $O/bench_c3_default_kernel_stats.csv
# the default run launches the banded kernel at two sizes (c3 and the spmv_1e7 block), which rocprofv3 --stats averages together;
# the same command without that block gives the per-kernel average that belongs to roofline.avg_launch_ms
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b2 -o b2 -- python3 $R/bench.py --no-1e7 > $O/bench_c3_no1e7_prof.json 2> $O/bench_c3_no1e7_prof.err
cd $R
cp $(find /tmp/prof_b2 -name "b2_kernel_stats.csv" | head -1) $O/bench_c3_kernel_stats.csv
python3 bench.py --workload c5 --steps 50 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --workload c3x --no-1e7 > $O/bench_c3x.json 2> $O/bench_c3x.err
python3 - <<'PY'
import json, csv, sys, os
O = os.environ.get("O", "gpurun_out/r03")
j = json.loads(open(f"{O}/bench_c3.json").read().strip().splitlines()[-1])
print("bench c3:", j["value"], "GB/s alg;", j["roofline"]["avg_launch_ms"] * 1e3, "us/launch; frac", j["roofline"]["frac"], "; 1e7 stored frac",
      j["spmv_1e7"]["stored"]["frac"], "at", j["spmv_1e7"]["stored"]["avg_launch_ms"] * 1e3, "us")
jp = json.loads(open(f"{O}/bench_c3_no1e7_prof.json").read().strip().splitlines()[-1])
print("profiled run (bench.py --no-1e7): HIP events", jp["roofline"]["avg_launch_ms"] * 1e3, "us/launch")
for r in csv.DictReader(open(f"{O}/bench_c3_kernel_stats.csv")):
    if "k_spmv<0" in r["Name"]:
        print("rocprofv3 (--no-1e7):", r["Name"][:48], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
