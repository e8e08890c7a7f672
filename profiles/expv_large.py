"""exp(tau A)v with fixed m = 30 on HBM-resident sizes (c3: 5.0e6 states, c3x: 1.0e7 states), stored
(banded) and matrix-free generator; every step timed on its own, and the first case repeated, to
separate the steady rate from what the first steps of a process pay (round 1 saw 4.97 vs 3.53 ms).
usage: python profiles/expv_large.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

m, tau, steps = 30, 0.002, 8
for name, mdl, box in (("c3 171^3", synth.repressilator(171), False), ("c3 171^3 again", synth.repressilator(171), False),
                       ("c3x 216^3", synth.repressilator(216), False), ("c3 171^3 matrix-free", synth.repressilator(171), True),
                       ("c3x 216^3 matrix-free", synth.repressilator(216), True)):
    p0 = synth.poisson_p0(mdl, 20.0)
    with KfspContext(0) as c:
        if box:
            c.set_matrix_box(mdl)
        else:
            c.set_matrix_csr(mdl.n, *mdl.csr_rows())
        c.set_vector(p0)
        per = []
        for s in range(steps):
            t0 = time.perf_counter()
            ws = c.expv_fixed(m, tau, 1)
            per.append((time.perf_counter() - t0) * 1e3)
        c.timers(reset=True)
        t0 = time.perf_counter()
        ws = c.expv_fixed(m, tau, steps)
        dt = (time.perf_counter() - t0) / steps
        tm = {k: round(v / steps, 3) for k, v in c.timers().items()}
    nnz, n = mdl.nnz(), mdl.n
    b_ref = (m + 1) * synth.spmv_alg_bytes(nnz, n) + m * 104 * n + 8 * n * (m + 1) + 24 * n     # SURVEY 8(d), unfused
    gen = 16 * n if box else 8 * (nnz - n) + 24 * n
    b_ours = (m + 1) * gen + m * (16 * n + 32 * n) + 8 * n * (m + 1) + 8 * n + 16 * n
    print(f"{name}: N={n} m={m}: first steps (ms) {[round(x, 2) for x in per]}; steady {dt * 1e3:7.3f} ms/step = "
          f"{b_ref / dt / 1e9:7.1f} GB/s by the reference's byte count, {b_ours / dt / 1e9:7.1f} GB/s over the bytes our kernels move; "
          f"per step {tm}; mass {ws[-1]:.15f}", flush=True)
