"""Dispatch latency after short idle periods (the stall kfsp_begin_step absorbs in adaptive runs,
profiles/r02_begin_step_gaps.txt): burst of products; host sleeps d ms (GPU idle); a small upload
(kfsp_set_vector, synchronous); then kfsp_begin_step (two tiny kernels + an 8-byte copy) is timed."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

mdl = synth.toggle(1000, 700)
w = np.random.default_rng(0).random(mdl.n)
with KfspContext(0) as c:
    c.set_matrix_csr(mdl.n, *mdl.csr_rows())
    c.set_vector(w)
    c.begin_step()
    for d in (0, 1, 2, 4, 6, 8, 10, 12, 15, 20, 30, 50, 100, 200):
        ts, us = [], []
        for _ in range(8):
            c.spmv_bench(50)
            time.sleep(d * 1e-3)
            t0 = time.perf_counter()
            c.set_vector(w)
            t1 = time.perf_counter()
            c.begin_step()
            t2 = time.perf_counter()
            us.append(t1 - t0)
            ts.append(t2 - t1)
        print(f"idle {d:4d} ms -> set_vector {1e3*np.median(us):7.3f} ms (max {1e3*max(us):7.3f})   "
              f"begin_step {1e3*np.median(ts):7.3f} ms (max {1e3*max(ts):7.3f})", flush=True)
