"""SELL product on the SSA-grown Goutsias FSP (profiles/fsp_spmv_r02.sh makes /tmp/fsp.bin), internal
state order: launch duration against the workgroup count (option grid_blocks).
    python3 profiles/fsp_grid_sweep.py /tmp/fsp.bin"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from krylovfspssa_amd import KfspContext  # noqa: E402

with open(sys.argv[1], "rb") as f:
    ns, nr, n = (int(v) for v in np.fromfile(f, dtype=np.int32, count=3))
    adj = np.fromfile(f, dtype=np.int32, count=nr * n).reshape(n, nr)
    off = np.fromfile(f, dtype=np.float64, count=nr * n).reshape(n, nr)
    diag = np.fromfile(f, dtype=np.float64, count=n)
    state = np.fromfile(f, dtype=np.int32, count=ns * n).reshape(n, ns)
x = np.random.default_rng(12345).random(n)
for order, sigma in ((1, 0), (1, 128), (1, 256), (1, 512), (1, 1024), (0, 0)):
    row = []
    for grid in (1024,):
        with KfspContext(0) as c:
            c.set_option("grid_blocks", grid)
            c.set_option("sell_sigma", sigma)
            c.set_option("state_order", order)
            c.set_option("state_order_products", 0)
            if order:
                c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            c.set_vector(x)
            c.begin_step()
            slots = c.matrix_info()["slots"]
            c.spmv_bench(20)
            row.append((grid, round(1e3 * min(c.spmv_bench(200) for _ in range(3)) / 200, 2)))
    print(("internal order, sigma %d" % sigma) if order else "discovery order", n, "us per launch:", row, "slots", slots, flush=True)
