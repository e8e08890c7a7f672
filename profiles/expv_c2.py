"""BASELINE config 2 alone: toggle box 1000 x 1000 (10^6 states, banded generator, everything Infinity-Cache resident),
exp(tau A)v with fixed m = 30, tau = 0.01 - the launch-sensitive regime VERDICT r03 #9 asks about.  Prints ms/step with
the per-phase timers; run under `rocprofv3 --kernel-trace` and feed the trace to profiles/arnoldi_gaps.py to see how the
step divides into kernel time and the gaps between dependent launches.
usage: python profiles/expv_c2.py [steps] [option=value ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
m, tau = 30, 0.01
mdl = synth.toggle(1000, 1000)
p0 = synth.poisson_p0(mdl, 30.0)
with KfspContext(0) as c:
    for k, v in opts.items():
        c.set_option(k, int(v))
    c.set_matrix_csr(mdl.n, *mdl.csr_rows())
    c.set_vector(p0)
    c.expv_fixed(m, tau, 5)
    c.timers(reset=True)
    t0 = time.perf_counter()
    ws = c.expv_fixed(m, tau, steps)
    dt = (time.perf_counter() - t0) / steps
    tm = {k: round(v / steps, 4) for k, v in c.timers().items()}
    w = c.get_vector()
print(f"c2 toggle 1000^2 N={mdl.n} m={m} tau={tau} options={opts}: {dt * 1e3:.4f} ms/step over {steps} steps; per step {tm}; "
      f"mass {ws[-1]:.15f} checksum {float(np.dot(w, np.arange(mdl.n) % 97)):.15e}", flush=True)
