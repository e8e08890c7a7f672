"""Option sweep for the c2 exp(tA)v recipe and the c3 SpMV (tuning aid).
Configurations are interleaved over several rounds in ONE process (guide rule
24) and the per-configuration minimum and median are printed.
usage: python profiles/sweep_expv.py"""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

tg = synth.toggle(1000, 1000)
rp, cc, vv = tg.csr_rows()
p0 = synth.poisson_p0(tg, 30.0)
rep = synth.repressilator(171)
rrp, rcc, rvv = rep.csr_rows()
x = np.random.default_rng(1).random(rep.n)
CONFIGS = [dict(), dict(fused_ortho=0), dict(grid_blocks=1024), dict(vec_grid_blocks=512), dict(vec_grid_blocks=2048),
           dict(grid_blocks=1024, vec_grid_blocks=512), dict(format=1)]
res = {i: ([], []) for i in range(len(CONFIGS))}
for rnd in range(3):
    for ci, opts in enumerate(CONFIGS):
        with KfspContext(0) as c:
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_matrix_csr(tg.n, rp, cc, vv)
            c.set_vector(p0)
            c.expv_fixed(30, 0.01, 2)
            for _ in range(3):
                c.set_vector(p0)
                t0 = time.perf_counter()
                c.expv_fixed(30, 0.01, 10)
                res[ci][0].append((time.perf_counter() - t0) / 10 * 1e3)
            c.set_matrix_csr(rep.n, rrp, rcc, rvv)
            c.set_vector(x)
            c.begin_step()
            c.spmv_bench(20)
            for _ in range(3):
                res[ci][1].append(c.spmv_bench(100) / 100 * 1e3)
for ci, opts in enumerate(CONFIGS):
    e, s = res[ci]
    print(f"{str(opts):52s} expv_c2 min {min(e):6.3f} med {statistics.median(e):6.3f} ms/step | "
          f"spmv_c3 min {min(s):6.2f} med {statistics.median(s):6.2f} us", flush=True)
