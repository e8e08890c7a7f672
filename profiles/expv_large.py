"""exp(tau A)v with fixed m = 30 on HBM-resident sizes (c3: 5.0e6 states, c3x: 1.0e7 states).
usage: python profiles/expv_large.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

for name, mdl in (("c3 171^3", synth.repressilator(171)), ("c3x 216^3", synth.repressilator(216)),
                  ("c3 171^3 again", synth.repressilator(171))):
    rp, cc, vv = mdl.csr_rows()
    p0 = synth.poisson_p0(mdl, 20.0)
    m, tau, steps = 30, 0.002, 5
    with KfspContext(0) as c:
        c.set_matrix_csr(mdl.n, rp, cc, vv)
        c.set_vector(p0)
        c.expv_fixed(m, tau, 1)
        c.set_vector(p0)
        c.timers(reset=True)
        t0 = time.perf_counter()
        ws = c.expv_fixed(m, tau, steps)
        dt = (time.perf_counter() - t0) / steps
        print("   timers (ms per step):", {k: round(v / steps, 3) for k, v in c.timers().items()})
    nnz, n = mdl.nnz(), mdl.n
    b_ref = (m + 1) * synth.spmv_alg_bytes(nnz, n) + m * 104 * n + 8 * n * (m + 1) + 24 * n     # SURVEY 8(d), unfused
    b_ours = (m + 1) * (8 * (nnz - n) + 24 * n) + m * (16 * n + 32 * n) + 8 * n * (m + 1) + 8 * n + 16 * n
    print(f"{name}: N={n} m={m}: {dt * 1e3:8.3f} ms/step  reference-count bytes {b_ref / 1e9:6.2f} GB -> {b_ref / dt / 1e9:7.1f} GB/s "
          f"algorithmic; bytes our kernels move {b_ours / 1e9:6.2f} GB -> {b_ours / dt / 1e9:7.1f} GB/s; mass {ws[-1]:.15f}", flush=True)
