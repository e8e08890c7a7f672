"""Host Pade timing: exp(tH) of the order-102 Hessenberg fixture, per call.
python profiles/padm_timing.py   (box host, EPYC 9575F: 170 us at m = 62, 466 us at m = 102)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from krylovfspssa_amd import host  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "padm.npz"))
for c in range(int(g["ncase"])):
    H, t = g[f"H{c}"], float(g[f"t{c}"])
    host.padm(H, t)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        host.padm(H, t)
    dt = (time.perf_counter() - t0) / n
    print(f"m={H.shape[0]:4d} {dt * 1e6:9.1f} us per exponential")
