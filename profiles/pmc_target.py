"""rocprofv3 target of round 3: a list of (workload, form) cases, each timed with HIP events (200 launches) and
then launched EXACTLY 5 more times, in the order given - so that the k-th group of 5 k_spmv launches after the
warm-ups in a counter CSV belongs to the k-th case.  With --calib first 3 x k_stream_read for each lane width
(known 2 GiB reads, the FETCH_SIZE calibration of pmc_reduce.py).

    python3 profiles/pmc_target.py [--calib] [--no-time] case [case ...]
    case = workload:form   workload c3 | c3x | c4 | c5s | c5 | fsp (Goutsias ellipsoid, 1.0e7 states, search order)
                           form     stored | mf (matrix-free: format 7 pencils where eligible, else format 4) | mf4 (format 4, ascending trips) | mf7 / mf8 (format 7 / format 8 forced) | mf6 / mf6r2 (matrix-free with x staged in LDS, reach 512 / 2: format 6) | sell (plain SELL-64) | coded (SELL-64, coded columns)
                                    | sell_search (plain SELL in the caller's search order; fsp only)
Every case prints one line `CASE <case> n=<states> ms=<per launch> real_bytes=<kfsp_matrix_bytes> info=<layout>`;
launch markers `MARK <case> <first launch index> 5` count the k_spmv<0,...> launches of this process."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext, synth  # noqa: E402

args = sys.argv[1:]
calib = "--calib" in args
timed = "--no-time" not in args
cases = [a for a in args if not a.startswith("--")]
ctx = KfspContext(0)
if calib:
    for w in (4, 8, 16):
        ctx.selftest_stream(2 << 30, w, 3)
launches = 0
fsp = None


def box(workload):
    if workload == "c3":
        return synth.repressilator(171)
    if workload == "c3x":
        return synth.repressilator(216)
    if workload == "c5s":
        return synth.birth_death((22, 22, 22, 22, 22, 3))
    if workload == "c5":
        return synth.birth_death((22,) * 6)
    raise ValueError(workload)


def tiled_order(mdl, cut, B):
    """profiles/trip_order_sweep_r04.py: blocks of B rows of the index below strides[cut], per block every slower line back to back"""
    trips = (mdl.n + 127) // 128
    Sc = int(mdl.strides[cut])
    r = np.arange(trips, dtype=np.int64) * 128
    lo, hi = r % Sc, r // Sc
    return np.argsort(((lo // B) * (hi.max() + 1) + hi) * Sc + lo, kind="stable").astype(np.int32)


for case in cases:
    workload, form = case.split(":")
    tile = None
    if "@" in form:                                 # form@cut,B : a tiled trip order (round 4), e.g. c5:mf@3,1024
        form, t = form.split("@")
        tile = tuple(int(v) for v in t.split(","))
    ctx.set_option("m_max", 8)                      # the product needs two basis columns
    ctx.set_option("format", 0)
    ctx.set_option("sell_code", -1)
    ctx.set_option("state_order", 1)
    variant = 0
    if workload == "fsp":
        if fsp is None:
            t0 = time.time()
            fsp = synth.GoutsiasEllipsoid()
            fsp_arrays = fsp.ell()
            print(f"fsp generated: n={fsp.n} in {time.time() - t0:.1f} s", flush=True)
        adj, off, diag = fsp_arrays
        n = fsp.n
        ctx.set_option("state_order_min", 1)
        ctx.set_option("state_order_products", 0)
        ctx.set_option("state_order", 0 if form == "sell_search" else 1)
        ctx.set_option("sell_code", 1 if form == "coded" else 0)
        ctx.set_state_coords(fsp.state)
        ctx.set_matrix_ell(adj, off, diag)
    elif workload == "c4":
        mdl = synth.GoutsiasConserved(150, 150, 150)
        n = mdl.n
        if form in ("sell", "coded"):
            ctx.set_option("format", 1)
            ctx.set_option("sell_code", 1 if form == "coded" else 0)
        ctx.set_matrix_csr(mdl.n, *mdl.csr_rows())
    else:
        mdl = box(workload)
        n = mdl.n
        if form in ("mf", "mf4", "mf7", "mf8", "mf6", "mf6r2"):
            ctx.set_option("box_pencil", {"mf4": 0, "mf7": 1, "mf8": 2}.get(form, -1))   # mf7 / mf8: pencils (format 7) / pencils in slabs (format 8)   # mf4: format 4 also where round 4's pencils (format 7) apply
            ctx.set_option("box_tile", 0 if form == "mf4" and tile is None else -1)
            ctx.set_option("box_lds", 0 if form in ("mf", "mf4", "mf7", "mf8") else 1)   # mf6: the near part of x staged in LDS (format 6)
            ctx.set_option("box_reach", 2 if form == "mf6r2" else 512)
            ctx.set_matrix_box(mdl, store=False)
        elif form == "stored":
            ctx.set_matrix_box(mdl, store=True)
        else:
            ctx.set_option("format", 1)
            ctx.set_option("sell_code", 1 if form == "coded" else 0)
            ctx.set_matrix_csr(mdl.n, *mdl.csr_rows())
    ctx.set_vector(np.random.default_rng(12345).random(n))
    ctx.begin_step()
    if tile is not None:
        ctx.set_trip_order(tiled_order(mdl, *tile))
    ms = float("nan")
    if timed:
        ctx.spmv_bench(20, variant)
        ms = min(ctx.spmv_bench(200, variant) for _ in range(3)) / 200
        launches += 620
    print(f"MARK {case} {launches} 5", flush=True)
    ctx.spmv_bench(5, variant)
    launches += 5
    print(f"CASE {case} n={n} ms={ms:.5f} real_bytes={ctx.matrix_bytes()} info={ctx.layout_info()} matrix={ctx.matrix_info()}", flush=True)
ctx.close()
