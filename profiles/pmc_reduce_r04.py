"""Reduce the FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_target.py (run with --calib --no-time, so every case is
exactly 5 launches): per case the median HBM bytes per launch (FETCH_SIZE x the factor calibrated on the 2 GiB
k_stream_read launches of the same run, guides/MI355X_MICROARCH.md HBM section; WRITE_SIZE exact) beside the
layout's own byte count and the HIP-event time of the timing run.

    python3 profiles/pmc_reduce_r03.py <fetch_csv> <write_csv> <timing_log> case [case ...] > summary"""
import csv
import re
import sys


def launches(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


fcsv, wcsv, tlog = sys.argv[1:4]
cases = sys.argv[4:]
f, w = launches(fcsv, "FETCH_SIZE"), launches(wcsv, "WRITE_SIZE")
med = lambda a: sorted(a)[len(a) // 2]
cal = {}
for r in f:
    k = r["Kernel_Name"]
    if "k_stream_read" in k:
        width = {"<int>": 4, "<double>": 8}.get(k[k.index("<"):k.index(">") + 1], 16)
        cal.setdefault(width, []).append(float(r["Counter_Value"]))
factor = sum((2 << 30) / (med(v) * 1024.0) for v in cal.values()) / len(cal)
print(f"FETCH_SIZE calibration factor {factor:.4f} (2 GiB reads at {sorted(cal)} B per lane)")
fs = [float(r["Counter_Value"]) for r in f if ("k_spmv<0" in r["Kernel_Name"] or "k_spmv_pencil<0" in r["Kernel_Name"] or "k_spmv_slab<0" in r["Kernel_Name"])]
ws = [float(r["Counter_Value"]) for r in w if ("k_spmv<0" in r["Kernel_Name"] or "k_spmv_pencil<0" in r["Kernel_Name"] or "k_spmv_slab<0" in r["Kernel_Name"])]
names = [r["Kernel_Name"][:60] for r in f if ("k_spmv<0" in r["Kernel_Name"] or "k_spmv_pencil<0" in r["Kernel_Name"] or "k_spmv_slab<0" in r["Kernel_Name"])]
timing = {}
for line in open(tlog):
    m = re.match(r"CASE (\S+) n=(\d+) ms=(\S+) real_bytes=(\d+)", line)
    if m:
        timing[m.group(1)] = (int(m.group(2)), float(m.group(3)), int(m.group(4)))
assert len(fs) == 5 * len(cases) == len(ws), (len(fs), len(ws), len(cases))
print(f"{'case':18s} {'states':>10s} {'us/launch':>10s} {'layout MB':>10s} {'PMC read MB':>12s} {'PMC write MB':>12s} {'PMC total MB':>12s} "
      f"{'PMC/layout':>10s} {'layout GB/s':>11s} {'PMC GB/s':>9s} {'frac(max)':>9s}  kernel")
for i, case in enumerate(cases):
    rd = med(fs[5 * i:5 * i + 5]) * 1024.0 * factor
    wr = med(ws[5 * i:5 * i + 5]) * 1024.0
    n, ms, real = timing.get(case, (0, float("nan"), 0))
    tot = rd + wr
    sec = ms * 1e-3
    print(f"{case:18s} {n:10d} {ms * 1e3:10.2f} {real / 1e6:10.1f} {rd / 1e6:12.1f} {wr / 1e6:12.1f} {tot / 1e6:12.1f} "
          f"{tot / max(real, 1):10.3f} {real / sec / 1e9:11.0f} {tot / sec / 1e9:9.0f} {max(real, tot) / sec / 8e12:9.3f}  {names[5 * i]}")
