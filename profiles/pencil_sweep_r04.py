"""Round 4: the pencil product (format 7) on the 6-species boxes against launch geometry and base-trip order.
    python3 profiles/pencil_sweep_r04.py c5s|c5
grid = workgroups of 256 (4 wavefronts each; 1024 = 4 per CU), box_tile = order of the base trips (0 ascending, -1 small tiles)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5s"
mdl = {"c5s": synth.birth_death((22, 22, 22, 22, 22, 3)), "c5": synth.birth_death((22,) * 6)}[wl]
x = np.random.default_rng(1).random(mdl.n)
y0 = None
cases = [(0, 0, 1024), (-1, 0, 1024), (-1, -1, 1024), (-1, -1, 512), (-1, -1, 768), (-1, -1, 1280), (-1, -1, 1536),
         (-1, -1, 2048), (-1, 0, 1280), (-1, 0, 2048)]
if "slab" in sys.argv:
    cases = [(0, 0, 1024), (1, 0, 768), (2, 0, 0), (2, 0, 256), (2, 0, 512), (-1, -1, 0)]
if "fine" in sys.argv:
    cases = [(0, 0, 1024)] + [(-1, 0, g) for g in (576, 640, 704, 768, 832, 896, 960)] + [(-1, -1, 768)]
if "waves" in sys.argv:
    cases = [(1, 0, 768)] + [(2, w, 0) for w in (2, 3, 4, 6, 8, 11)]          # (second field: box_slab_waves)
for pencil, tile, grid in cases:
    with KfspContext(0) as c:
        c.set_option("m_max", 8)
        if "waves" in sys.argv and pencil == 2:
            c.set_option("box_slab_waves", tile)
            tile = 0
        c.set_option("box_pencil", pencil)
        c.set_option("box_tile", tile)
        if grid:
            c.set_option("grid_blocks", grid)
        c.set_matrix_box(mdl, store=False)
        c.set_vector(x)
        c.begin_step()
        y = c.spmv_w()
        if y0 is None:
            y0 = y
        c.spmv_bench(10)
        ms = min(c.spmv_bench(50) for _ in range(3)) / 50
        print(f"{wl}: format {c.layout_info()['format']} box_tile {tile:2d} grid {grid:5d}: {ms * 1e3:9.2f} us  same bits: {np.array_equal(y, y0)}", flush=True)
