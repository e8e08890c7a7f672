import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from krylovfspssa_amd import KfspContext, synth
mdl = synth.toggle(1000, 1000)
adj, off, diag = mdl.ell()
w = np.random.default_rng(0).random(mdl.n)
with KfspContext(0) as c:
    c.set_matrix_ell(adj, off, diag); c.set_vector(w); c.begin_step()
    for idle in (0.0, 0.05, 0.2, 0.5, 1.0):
        ts = []
        for _ in range(3):
            c.set_matrix_ell(adj, off, diag); c.set_vector(w)
            time.sleep(idle)
            t0 = time.perf_counter(); c.begin_step(); ts.append(time.perf_counter() - t0)
        print(f"idle {idle:4.2f}s -> begin_step {1e3*min(ts):6.2f} .. {1e3*max(ts):6.2f} ms", flush=True)
