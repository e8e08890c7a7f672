"""Counter calibration + SpMV traffic run for rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE  -d <dir> --output-format csv -- python3 profiles/pmc_calib.py
    rocprofv3 --pmc WRITE_SIZE  -d <dir> --output-format csv -- python3 profiles/pmc_calib.py

Launches, in order: 3 x k_stream_read for each lane width 4/8/16 B over a
2 GiB buffer (known bytes, far larger than the 256 MiB Infinity Cache), then 5 x
the generator SpMV of the workload given as argument (c3, c3x, c4, c4_nomask).  profiles/pmc_reduce.py turns the two CSVs into
profiles/pmc_traffic.json.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
ctx = KfspContext(0)
for w in (4, 8, 16):
    ctx.selftest_stream(2 << 30, w, 3)
if workload.startswith("c4"):                     # c4 / c4_nomask: BASELINE config 4, with / without group masks
    mdl = synth.GoutsiasConserved(150, 150, 150)
    if workload == "c4_nomask":
        ctx.set_option("dia_mask", 0)
else:
    mdl = synth.repressilator(171) if workload == "c3" else synth.repressilator(216)
ctx.set_matrix_csr(mdl.n, *mdl.csr_rows())
ctx.set_vector(np.random.default_rng(12345).random(mdl.n))
ctx.begin_step()
ctx.spmv_bench(5)
print(workload, mdl.n, mdl.nnz(), ctx.matrix_info())
ctx.close()
