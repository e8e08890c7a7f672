"""Where kfsp_padm spends its time at order 102 (kfsp_padm_profile), for KFSP_PADE_THREADS threads: python profiles/padm_profile.py"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from krylovfspssa_amd import host
lib = host.load_library()
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "padm.npz"))
H, t = g["H4"], float(g["t4"])
host.padm(H, t)
out = (C.c_double * 4)()
lib.kfsp_padm_profile(out, 1)
n = 200
t0 = time.perf_counter()
for _ in range(n): host.padm(H, t)
dt = (time.perf_counter() - t0) / n
lib.kfsp_padm_profile(out, 0)
print(f"threads={os.environ.get('KFSP_PADE_THREADS')} m={H.shape[0]} {dt*1e6:.1f} us/call: dense {out[0]/n*1e6:.1f} banded {out[1]/n*1e6:.1f} solve {out[2]/n*1e6:.1f} whole {out[3]/n*1e6:.1f}")
