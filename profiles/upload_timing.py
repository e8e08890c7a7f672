"""Time kfsp_set_matrix_ell (reference layout -> device gather form): host
counting-sort transpose vs the device build.  usage: python profiles/upload_timing.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

rng = np.random.default_rng(0)
cases = [("toggle box 1000x1000 (banded)", synth.toggle(1000, 1000), False),
         ("repressilator 120^3 (banded)", synth.repressilator(120), False),
         ("goutsias box 1.1e6, rows shuffled (SELL)", synth.goutsias_box((30, 30, 30, 3, 3, 5)), True)]
for name, mdl, shuffle in cases:
    adj, off, diag = mdl.ell()
    if shuffle:           # an arbitrary state order, like the reference's hash/BFS order
        perm = rng.permutation(mdl.n)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(mdl.n)
        adj = adj[perm]
        off = off[perm]
        diag = diag[perm]
        adj = np.where(adj > 0, inv[np.maximum(adj, 1) - 1] + 1, adj).astype(np.int32)
    for hb in (1, 0):
        with KfspContext(0) as c:
            c.set_option("host_build", hb)
            c.set_matrix_ell(adj, off, diag)
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                c.set_matrix_ell(adj, off, diag)
                t.append(time.perf_counter() - t0)
            info = c.matrix_info()
        print(f"{name:44s} N={mdl.n:8d} bw={adj.shape[1]:2d} {'host' if hb else 'device'} build: "
              f"{min(t) * 1e3:8.2f} ms  slots={info['slots']}", flush=True)
