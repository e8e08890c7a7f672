"""Round 4: SMALL tiles for the 6-species boxes.  Round 3's tiled orders used blocks of 3-12
planes of 22^3 rows (32k-128k rows): with 22 lines of the fifth species per block that is 5-22 MB of x between a row and its
+-22^5 neighbour - no L2 holds it.  Here the block is B rows of the index below the stride of species 5 (22^4), and for each block
ALL (species 5, species 6) lines are swept back to back: the +-22^4 neighbours are one line away (B rows), the +-22^5
neighbours 22 lines (22 B rows = 0.7 MB at B = 4096): three planes of a block fit an XCD's 4 MiB L2.
    python3 profiles/trip_order_sweep_r04.py c5s|c5 mf|stored [B ...]
Same bits in every order (rows are independent); prints us per product."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5s"
form = sys.argv[2] if len(sys.argv) > 2 else "mf"
Bs = [int(v) for v in sys.argv[3:]] or [1024, 2048, 4096, 8192, 16384, 32768]
mdl = {"c5s": synth.birth_death((22, 22, 22, 22, 22, 3)), "c5": synth.birth_death((22,) * 6)}[wl]
n = mdl.n
trips = (n + 127) // 128
strides = [int(s) for s in mdl.strides]


def order(cut, B):
    """blocks of B rows of the index below strides[cut]; per block every slower line back to back"""
    Sc = strides[cut]
    r = np.arange(trips, dtype=np.int64) * 128
    lo, hi = r % Sc, r // Sc
    key = ((lo // B) * (hi.max() + 1) + hi) * Sc + lo
    return np.argsort(key, kind="stable").astype(np.int32)


with KfspContext(0) as c:
    c.set_option("m_max", 8)
    c.set_matrix_box(mdl, store=(form == "stored"))
    x = np.random.default_rng(1).random(n)
    c.set_vector(x)
    c.begin_step()
    y0 = c.spmv_w()

    def run(label, o):
        c.set_trip_order(o)
        c.spmv_bench(10)
        ms = min(c.spmv_bench(50) for _ in range(3)) / 50
        same = np.array_equal(c.spmv_w(), y0)
        print(f"{wl}:{form} {label:44s} {ms * 1e3:9.2f} us  same bits: {same}", flush=True)

    run("ascending", None)
    for cut in (4, 3):
        for B in Bs:
            if B < strides[cut]:
                run(f"cut below species {cut + 1}, B = {B} rows", order(cut, B))
    c.set_trip_order(None)
