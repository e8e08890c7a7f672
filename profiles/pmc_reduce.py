"""Reduce the rocprofv3 --pmc passes of profiles/pmc_calib.py to
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    python profiles/pmc_reduce.py <workload> <fetch_csv> <write_csv> [<hit_csv>]

Method (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes (they do not fit one pass) and are in KiB.  On
gfx950 FETCH_SIZE under-reports wide coalesced reads; the factor is calibrated
in the same run on k_stream_read launches that read exactly 2 GiB with 4, 8 and
16 bytes per lane (the SpMV's own access widths).  WRITE_SIZE is exact.
"""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def by_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    wl, fcsv, wcsv = sys.argv[1:4]
    fetch = by_kernel(fcsv, "FETCH_SIZE")
    write = by_kernel(wcsv, "WRITE_SIZE")
    calib = {}
    for k, v in fetch.items():
        if "k_stream_read" in k:
            width = {"<int>": 4, "<double>": 8}.get(k[k.index("<"):k.index(">") + 1], 16)
            calib[width] = (2 << 30) / (sorted(v)[len(v) // 2] * 1024.0)
    spmv_f = [v for k, v in fetch.items() if "k_spmv<0" in k][0]
    spmv_w = [v for k, v in write.items() if "k_spmv<0" in k][0]
    factor = sum(calib.values()) / len(calib)
    med = lambda a: sorted(a)[len(a) // 2]
    rd = med(spmv_f) * 1024.0 * factor
    wr = med(spmv_w) * 1024.0
    entry = {
        "hbm_bytes_per_launch": round(rd + wr),
        "read_bytes": round(rd), "write_bytes": round(wr),
        "fetch_size_kib_raw": med(spmv_f), "write_size_kib_raw": med(spmv_w),
        "fetch_calibration_factor": {str(k): round(v, 4) for k, v in sorted(calib.items())},
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x factor "
                  "calibrated on 2 GiB k_stream_read launches at 4/8/16 B per lane in the same run",
    }
    if len(sys.argv) > 4:
        hit = by_kernel(sys.argv[4], "TCC_HIT_sum")
        miss = by_kernel(sys.argv[4], "TCC_MISS_sum")
        h = med([v for k, v in hit.items() if "k_spmv<0" in k][0])
        m = med([v for k, v in miss.items() if "k_spmv<0" in k][0])
        entry["l2_hit_rate"] = round(h / (h + m), 4)
        entry["l2_miss_x_128B"] = round(m * 128)
    path = os.path.join(HERE, "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[wl] = entry
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
