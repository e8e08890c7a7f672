"""rocprofv3 --pmc target: 5 launches of the matrix-free product on c3x (10^7 states), nothing else.
    rocprofv3 --pmc <counters> -d <dir> --output-format csv -- python3 profiles/pmc_box_sq.py"""
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
