"""Cost of one FSP change on the device side: set_matrix_ell + set_vector +
begin_step for a growing, arbitrarily ordered (non-banded) generator.
usage: python profiles/fsp_change_timing.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

rng = np.random.default_rng(0)
mdl = synth.goutsias_box((30, 30, 30, 3, 3, 5))
adj, off, diag = mdl.ell()
perm = rng.permutation(mdl.n)
inv = np.empty_like(perm)
inv[perm] = np.arange(mdl.n)
adj = np.where(adj[perm] > 0, inv[np.maximum(adj[perm], 1) - 1] + 1, adj[perm]).astype(np.int32)
off, diag = off[perm], diag[perm]
w = rng.random(mdl.n)
with KfspContext(0) as c:
    for n in [200_000, 400_000, 600_000, 800_000, 900_000, 1_000_000, 1_100_000, 1_200_000, mdl.n]:
        a = np.where(adj[:n] > n, 0, adj[:n])          # links outside the first n states -> "not in FSP"
        t0 = time.perf_counter()
        c.set_matrix_ell(a, off[:n], diag[:n])
        t1 = time.perf_counter()
        c.set_vector(w[:n])
        t2 = time.perf_counter()
        c.begin_step()
        t3 = time.perf_counter()
        c.begin_step()
        t4 = time.perf_counter()
        c.arnoldi(30)
        t5 = time.perf_counter()
        print(f"n={n:8d} set_matrix {1e3 * (t1 - t0):7.2f} ms  set_vector {1e3 * (t2 - t1):6.2f}  begin_step {1e3 * (t3 - t2):6.2f} "
              f"(again {1e3 * (t4 - t3):5.2f})  arnoldi(30) {1e3 * (t5 - t4):6.2f}", flush=True)
