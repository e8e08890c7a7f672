"""Generator SpMV on an FSP in the reference's OWN state order (SSA / one-step
discovery order, not a lexicographic box): the matrix dumped by
profiles/statespace_bench.f90 goes through kfsp_set_matrix_ell (device build ->
SELL-64) and is timed like bench.py times c3.
    /tmp/ssb 2.0 38 /tmp/fsp.bin && python profiles/fsp_spmv_timing.py /tmp/fsp.bin"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

with open(sys.argv[1], "rb") as f:
    ns, nr, n = (int(v) for v in np.fromfile(f, dtype=np.int32, count=3))
    adj = np.fromfile(f, dtype=np.int32, count=nr * n).reshape(n, nr)
    off = np.fromfile(f, dtype=np.float64, count=nr * n).reshape(n, nr)
    diag = np.fromfile(f, dtype=np.float64, count=n)
    state = np.fromfile(f, dtype=np.int32, count=ns * n).reshape(n, ns)
nnz = int((adj > 0).sum()) + n
ctx = KfspContext(0)
ctx.set_option("state_order", 0)               # first: the caller's (discovery) order as it is
ctx.set_matrix_ell(adj, off, diag)
info = ctx.matrix_info()
x = np.random.default_rng(12345).random(n)
ctx.set_vector(x)
ctx.begin_step()
# spot check against numpy (scatter form on a sample of columns is awkward: use the gather rows)
y = ctx.spmv_w()
ref = -diag * x
src = np.repeat(np.arange(n), nr)[(adj > 0).ravel()]
dst = adj[adj > 0] - 1
np.add.at(ref, dst, off[adj > 0] * x[src])
print(f"N={n} nnz={nnz} stored slots={info['slots']} max |err| = {np.abs(y - ref).max():.3e}")
ctx.spmv_bench(20)
reps = 200
ms = ctx.spmv_bench(reps) / reps
b_alg = synth.spmv_alg_bytes(nnz, n)
b_real = 12 * info["slots"] + 24 * n
span = np.abs(dst - src)
print(f"SpMV {ms * 1e3:.1f} us/launch: algorithmic {b_alg / ms / 1e6:.0f} GB/s, stored bytes (12/slot + 24/row) "
      f"{b_real / ms / 1e6:.0f} GB/s; |col-row| median {int(np.median(span))}, 90% {int(np.quantile(span, 0.9))}, max {int(span.max())}")

# the same call sequence with the coordinates handed over first: the library keeps
# generator and vectors in lexicographic state order internally
ctx.set_option("state_order", 1)               # the default
ctx.set_option("state_order_products", 0)      # (by default only once the previous generator saw 48 products)
ctx.set_state_coords(state)
ctx.set_matrix_ell(adj, off, diag)
assert ctx.state_order_active()
info2 = ctx.matrix_info()
ctx.set_vector(x)
ctx.begin_step()
y2 = ctx.spmv_w()
print(f"with kfsp_set_state_coords: stored slots={info2['slots']} max |err| vs numpy = {np.abs(y2 - ref).max():.3e}; "
      f"bit-identical to the plain product: {bool(np.array_equal(y, y2))}")
ctx.spmv_bench(20)
ms2 = ctx.spmv_bench(reps) / reps
print(f"SpMV {ms2 * 1e3:.1f} us/launch: algorithmic {b_alg / ms2 / 1e6:.0f} GB/s, stored bytes "
      f"{(12 * info2['slots'] + 24 * n) / ms2 / 1e6:.0f} GB/s")
t = ctx.timers(reset=True)
import time
t0 = time.perf_counter()
for _ in range(10):
    ctx.set_state_coords(state)
    ctx.set_matrix_ell(adj, off, diag)
t1 = time.perf_counter()
for _ in range(10):
    ctx.set_matrix_ell(adj, off, diag)
t2 = time.perf_counter()
print(f"upload + device build: {(t1 - t0) * 100:.2f} ms with the state order, {(t2 - t1) * 100:.2f} ms without")
