#!/bin/bash
# Round 4: the driver's bench command (python bench.py, defaults), the rocprofv3 kernel statistics of the SAME command, and the
# lines of the large workloads with their exp(tA)v blocks.  Run from the repo root through gpurun.
set -e
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b2 -o b2 -- python3 $R/bench.py --no-1e7 > $O/bench_c3_no1e7_prof.json 2> $O/bench_c3_no1e7_prof.err
cd $R
cp $(find /tmp/prof_b2 -name "b2_kernel_stats.csv" | head -1) $O/bench_c3_kernel_stats.csv
python3 bench.py --workload c3x --no-1e7 --no-cpu > $O/bench_c3x.json 2> $O/bench_c3x.err
python3 bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err
python3 bench.py --workload c5s --no-cpu > $O/bench_c5s.json 2> $O/bench_c5s.err
python3 bench.py --workload c5 --steps 50 --warmup 5 --no-cpu > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --workload c5 --matrix-free --steps 50 --warmup 5 --no-cpu > $O/bench_c5_mf.json 2> $O/bench_c5_mf.err
python3 - <<'PY'
import json, csv, os
O = "gpurun_out/r04"
for name in ("c3", "c3x", "c4", "c5s", "c5", "c5_mf"):
    try:
        j = json.loads(open(f"{O}/bench_{name}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "FAILED", e); continue
    r = j["roofline"]
    line = f"{name}: value {j['value']} GB/s (8d model), moved {j.get('value_moved_GBps')} GB/s = {j.get('frac_of_peak')} of peak, {r['avg_launch_ms']*1e3:.2f} us/launch"
    if "matrix_free" in j and "avg_launch_ms" in j["matrix_free"]:
        line += f"; matrix-free {j['matrix_free']['avg_launch_ms']*1e3:.2f} us"
    if "expv" in j:
        line += f"; expv c2 {j['expv']['ms_per_step']} ms/step"
    if "expv_workload" in j:
        w = j["expv_workload"]
        line += f"; expv on the workload {w['ms_per_step']} ms/step = {w['moved_GBps']} GB/s moved ({w['frac_of_peak']})"
    print(line)
for r in csv.DictReader(open(f"{O}/bench_c3_kernel_stats.csv")):
    if "k_spmv<0" in r["Name"]:
        print("rocprofv3 (--no-1e7):", r["Name"][:48], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
