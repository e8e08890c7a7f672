"""How a run of dependent launches divides into kernel time and the gaps between them.
Reads a rocprofv3 --kernel-trace CSV, keeps the last `tail` fraction of the events on the busiest queue (the steady
state), and prints per kernel: calls, average duration, average idle gap in FRONT of it (start - previous end), and the
totals - the share of the wall span the device spent between kernels.
usage: python profiles/arnoldi_gaps.py <kernel_trace.csv> [tail=0.5]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
ev.sort()
ev = ev[int(len(ev) * (1.0 - tail)):]
span = ev[-1][1] - ev[0][0]
busy = sum(e[1] - e[0] for e in ev)
stat = defaultdict(lambda: [0, 0, 0])
for prev, cur in zip(ev, ev[1:]):
    s = stat[cur[2].split("(")[0][:70]]
    s[0] += 1
    s[1] += cur[1] - cur[0]
    s[2] += max(0, cur[0] - prev[1])
print(f"{len(ev)} launches, span {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us ({busy / span:.3f}), gaps {(span - busy) / 1e3:.1f} us "
      f"= {(span - busy) / max(1, len(ev) - 1) / 1e3:.2f} us per boundary")
for name, (n, d, g) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:72s} calls {n:6d}  avg dur {d / n / 1e3:8.2f} us  avg gap before {g / n / 1e3:6.2f} us")
