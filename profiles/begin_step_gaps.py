"""What the GPU does around every k_copy_nrm2 (kfsp_begin_step) of a traced adaptive run:
reads rocprofv3 kernel + memory-copy traces, prints for the slowest begin_steps the GPU-side
timeline of the preceding 40 ms."""
import csv
import sys

kern, mem = sys.argv[1], sys.argv[2]
ev = []
for r in csv.DictReader(open(kern)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for r in csv.DictReader(open(mem)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))))
ev.sort()
idx = [i for i, e in enumerate(ev) if "k_copy_nrm2" in e[2]]
gaps = []
for i in idx:
    prev_end = max(e[1] for e in ev[max(0, i - 50):i]) if i else ev[i][0]
    gaps.append((ev[i][0] - prev_end, i))
gaps.sort(reverse=True)
print("begin_steps:", len(idx), " idle gap before k_copy_nrm2 (us): top 10", [round(g[0] / 1e3) for g in gaps[:10]])
for g, i in gaps[:3]:
    print(f"--- gap {g/1e3:.0f} us before event {i}")
    t0 = ev[i][0]
    for e in ev[max(0, i - 14):i + 3]:
        print(f"  start {(e[0]-t0)/1e3:10.1f} us  dur {(e[1]-e[0])/1e3:9.1f} us  {e[2]}")
