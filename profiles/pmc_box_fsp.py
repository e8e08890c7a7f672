"""rocprofv3 --pmc target: 5 launches each of (a) the matrix-free product on c3x (10^7 states) and
(b) the SELL-64 product on the SSA-grown Goutsias FSP of profiles/statespace_bench.f90 (/tmp/fsp.bin)
in discovery order and in the internal state order.
    rocprofv3 --pmc FETCH_SIZE -d <dir> --output-format csv -- python3 profiles/pmc_box_fsp.py
    rocprofv3 --pmc WRITE_SIZE -d <dir> --output-format csv -- python3 profiles/pmc_box_fsp.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

mdl = synth.repressilator(216)
with KfspContext(0) as c:
    c.set_matrix_box(mdl)
    c.set_vector(np.random.default_rng(1).random(mdl.n))
    c.begin_step()
    c.spmv_bench(5)
    print("box", mdl.n, c.matrix_bytes())
if os.path.exists("/tmp/fsp.bin"):
    with open("/tmp/fsp.bin", "rb") as f:
        ns, nr, n = (int(v) for v in np.fromfile(f, dtype=np.int32, count=3))
        adj = np.fromfile(f, dtype=np.int32, count=nr * n).reshape(n, nr)
        off = np.fromfile(f, dtype=np.float64, count=nr * n).reshape(n, nr)
        diag = np.fromfile(f, dtype=np.float64, count=n)
        state = np.fromfile(f, dtype=np.int32, count=ns * n).reshape(n, ns)
    x = np.random.default_rng(2).random(n)
    for order in (0, 1):
        with KfspContext(0) as c:
            c.set_option("state_order", order)
            c.set_option("state_order_products", 0)
            if order:
                c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            c.set_vector(x)
            c.begin_step()
            c.spmv_bench(5)
            print("fsp order", order, n, c.matrix_info(), c.matrix_bytes())
