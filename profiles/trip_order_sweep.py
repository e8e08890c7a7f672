"""Tiled trip orders for boxes whose far strides do not fit an XCD's L2: time and (under rocprofv3)
traffic of the product for a few tilings, stored and matrix-free, on the config-5 slab / the full 22^6 box / config 4.

    python3 profiles/trip_order_sweep.py c5s|c5|c4 [mf|stored]
order key: rows r = 128 c of trip c; lo = r mod S_p (index below the pivot stride), x_p = (r div S_p) mod d_p, hi = r div
(S_p d_p); blocks of B rows of lo.  'b,hi,xp' sorts by (block, hi, x_p, lo): all pivot planes of a block back to back (the
+-S_p neighbours stay in L2), the slower species around them (their neighbours one sweep of the planes away: Infinity Cache)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5s"
form = sys.argv[2] if len(sys.argv) > 2 else "mf"
mdl = {"c5s": synth.birth_death((22, 22, 22, 22, 22, 3)), "c5": synth.birth_death((22,) * 6),
       "c3x": synth.repressilator(216)}[wl]
n = mdl.n
trips = (n + 127) // 128
strides = [int(s) for s in mdl.strides]


def order(sp_index, B, mode):
    Sp = strides[sp_index]
    dp = mdl.dims[sp_index]
    r = np.arange(trips, dtype=np.int64) * 128
    lo = r % Sp
    xp = (r // Sp) % dp
    hi = r // (Sp * dp)
    b = lo // B
    if mode == "b,hi,xp":
        key = ((b * (hi.max() + 1) + hi) * dp + xp) * Sp + lo
    elif mode == "hi,b,xp":
        key = ((hi * (b.max() + 1) + b) * dp + xp) * Sp + lo
    else:
        raise ValueError(mode)
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
        c.spmv_bench(20)
        ms = min(c.spmv_bench(100) for _ in range(3)) / 100
        same = np.array_equal(c.spmv_w(), y0)
        print(f"{wl}:{form} {label:34s} {ms * 1e3:9.2f} us  same bits: {same}", flush=True)

    run("ascending", None)
    plane = strides[3] if len(strides) > 3 else strides[-1]
    for sp in ([4, 5] if len(strides) >= 6 else [len(strides) - 1]):
        for k in (3, 6, 12):
            for mode in ("b,hi,xp", "hi,b,xp"):
                run(f"pivot species {sp + 1}, B = {k} planes, {mode}", order(sp, k * plane, mode))
    c.set_trip_order(None)
