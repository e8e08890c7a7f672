"""Matrix-free product: launch duration against the workgroup count (option "grid_blocks"), per box.
    python3 profiles/box_grid_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

boxes = [("c3x", synth.repressilator(216)), ("c3", synth.repressilator(171)), ("c2", synth.toggle(1000, 1000)),
         ("c5s", synth.birth_death((22, 22, 22, 22, 22, 3)))]
for name, mdl in boxes:
    row = []
    for grid in (256, 512, 768, 1024, 1536, 2048):
        with KfspContext(0) as c:
            c.set_option("grid_blocks", grid)
            c.set_matrix_box(mdl)
            c.set_vector(np.random.default_rng(1).random(mdl.n))
            c.begin_step()
            c.spmv_bench(20)
            row.append((grid, round(1e3 * min(c.spmv_bench(200) for _ in range(3)) / 200, 2)))
    print(name, mdl.n, "us per launch:", row, flush=True)
