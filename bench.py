#!/usr/bin/env python3
"""Benchmark of the exp(tA)v hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A timed "step" is one generator SpMV launch y = A x (FMATVEC, the dominant
kernel of the path; with N > 1 ranks each launch is preceded by the RCCL
all-gather of the source slab, exactly as in the solver).  The workload is
BASELINE.json configs[2] - the configuration the SpMV-GB/s half of the metric is
quoted on: repressilator_model.input propensities on a 171^3 box, N = 5 000 211
states per GPU (weak scaling: the slowest box dimension grows with the number
of ranks, rows are partitioned contiguously, no data-path collective except the
all-gather the path itself has).  value = algorithmic GB/s of the whole job,
B_alg = 12 nnz + 20 N per launch.

With --gpus N > 1 and no launcher (WORLD_SIZE unset) the script starts its N ranks itself:
child processes of this same file, one per GPU, created BEFORE anything here touches the
GPU; rank 0's JSON line is relayed.  Under torch.distributed.run it joins the ranks it is given.

The same JSON line carries
  roofline      the SpMV kernel against the HBM roofline (HIP events on the
                library's own stream, measured over the timed region): `achieved` / `frac`
                count the bytes that really cross the HBM interface per launch - the
                kernel's own layout (kfsp_matrix_bytes) or the rocprofv3 PMC traffic
                where that is larger - so frac <= 1 by construction; `alg_GBps` keeps
                the SURVEY 8(d) CSR model (what `value` reports), which the banded form
                does not move (it stores no column indices)
  spmv_1e7      (default run) the same product at 10^7 states per GPU - the size of the
                60 % target: c3x = repressilator box 216^3, generator written out ON THE
                DEVICE from the propensity tables, stored and matrix-free, same fields
  self_check    the product that is timed, checked in this process against numpy on sampled
                rows of every rank (always, also on one GPU)
  expv          the exp(tA)v half of the metric: BASELINE configs[1] (toggle box
                10^6 states per GPU, Krylov m = 30, tau = 0.01, 10 steps) wall
                time per step and its l1 error against the CPU path
  cpu_baseline  the oracle (plain-C port of the reference loops) on this host,
                1 core, on a bounded sample of the same workload
--workload c5 is BASELINE configs[4] at its stated size (22^6 = 1.13e8 states, 1.41e9 nnz):
the whole generator on however many ranks there are, each rank's rows written out by its
own device (no host arrays of that size exist); "scaling": "strong".
Inputs are synthetic and resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c3x", "c4", "c5s", "c5", "tiny"])
    ap.add_argument("--device-build", action="store_true",
                    help="box workloads: the stored generator is written out by the device from the propensity tables "
                         "(kfsp_set_matrix_box, option box_store) instead of uploading gather rows made by numpy; always on for c5")
    ap.add_argument("--no-1e7", action="store_true", help="skip the spmv_1e7 block (c3x, 10^7 states per GPU) of the default run")
    ap.add_argument("--fsp", action="store_true",
                    help="add the spmv_fsp block: the generator product on a NON-BOX FSP of 1.0e7 states (Goutsias on an ellipsoid x 6 DNA "
                         "configurations, listed in search order; SELL-64 in the caller's order, in the internal order, with coded columns); "
                         "single GPU, ~40 s of input generation and upload")
    ap.add_argument("--variant", type=int, default=0, choices=[0, 2], help="SpMV kernel variant (0 auto: DIA/SELL, 2 SELL-64)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-expv", action="store_true")
    ap.add_argument("--expv-steps", type=int, default=10)
    ap.add_argument("--opt", action="append", default=[], help="library option name=value")
    ap.add_argument("--matrix-free", action="store_true",
                    help="box workloads: no stored generator, propensities from factor tables in the kernel (kfsp_set_matrix_box)")
    ap.add_argument("--force-comm", action="store_true",
                    help="create the RCCL communicator even with one rank (exercises the collective path on one GPU)")
    return ap.parse_args()


def spmv_model(workload, nranks):
    from krylovfspssa_amd import synth
    if workload == "c3":
        return synth.repressilator(dims=(171, 171, 171 * nranks)), "repressilator_model.input propensities, box 171x171x(171*ranks)"
    if workload == "c3x":
        return synth.repressilator(dims=(216, 216, 216 * nranks)), "repressilator_model.input propensities, box 216x216x(216*ranks) (10^7 states/GPU)"
    if workload == "c4":
        # the conserved-DNA state set of BASELINE config 4 (2.025e7 states per GPU); ranks stack along RNA
        return (synth.GoutsiasConserved(150, 150, 150) if nranks == 1 else None,
                "goutsias_model.input propensities, M,D,RNA in [0,150)^3 x 6 DNA configurations (single GPU only)")
    if workload == "c5s":
        # per-GPU slab of BASELINE config 5 (6-species birth-death network on 22^6, 12 reactions): three planes
        # of the slowest species per rank, i.e. 22^5 x 24 = 1.24e8 states on 8 GPUs
        return (synth.birth_death((22, 22, 22, 22, 22, 3 * nranks)),
                "synthetic 6-species birth-death network, box 22^5 x (3*ranks) (1.55e7 states per GPU)")
    if workload == "c5":
        # BASELINE config 5 as stated: the whole 22^6 box (1.13e8 states, 1.41e9 nonzeros), row-partitioned over
        # however many ranks there are (strong scaling; on one GPU the whole generator is resident: 10.9 GB stored)
        return (synth.birth_death((22,) * 6),
                "synthetic 6-species birth-death network, box 22^6 = 113 379 904 states, 12 reactions, rows partitioned over the ranks")
    if workload == "c2":
        return synth.toggle(1000, 1000 * nranks), "toggle_model.input propensities, box 1000x(1000*ranks)"
    return synth.repressilator(dims=(40, 40, 40 * nranks)), "tiny repressilator box 40x40x(40*ranks)"


def spawn_ranks(nranks, argv, script=None, env=None, popen=None, deadline_s=None, poll_s=0.2):
    """Start `nranks` copies of this script as child processes, one per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, as torch.distributed.run would
    set them), relay rank 0's stdout (the JSON line).  The parent never touches the GPU and never
    exec()s.  All children are polled together: the first one that exits non-zero - or the overall
    deadline (KFSP_BENCH_DEADLINE_S, default 1500 s) - ends the job at once; the others, which would
    otherwise sit in the rendezvous or a collective waiting for it for ever, are terminated (then
    killed), and the failing rank is named.  -> exit code (0 only if every rank succeeded)."""
    import socket
    import subprocess
    import tempfile
    popen = popen or subprocess.Popen
    script = script or os.path.abspath(__file__)
    base = dict(os.environ if env is None else env)
    if deadline_s is None:
        deadline_s = float(base.get("KFSP_BENCH_DEADLINE_S", "1500"))
    if "MASTER_PORT" not in base:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            base["MASTER_PORT"] = str(s.getsockname()[1])
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    # rank 0's stdout goes to a file, not a pipe: nothing has to drain it while all ranks are polled
    with tempfile.TemporaryFile() as out0:
        for r in range(nranks):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks))
            procs.append(popen([sys.executable, script] + list(argv), env=e,
                               stdout=out0 if r == 0 else subprocess.DEVNULL))
        t_end = time.monotonic() + deadline_s
        codes = [None] * nranks
        why = None
        while True:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                why = f"ranks failed: {bad}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > t_end:
                why = f"deadline of {deadline_s:.0f} s passed; still running: {[r for r, c in enumerate(codes) if c is None]}"
                break
            time.sleep(poll_s)
        if why is not None:
            # fresh children of this process, by their own handles (never by pattern)
            alive = [p for r, p in enumerate(procs) if codes[r] is None]
            for p in alive:
                p.terminate()
            t_kill = time.monotonic() + 5.0
            for p in alive:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            print(f"[bench] {why}" + (f"; terminated {len(alive)} other rank(s)" if alive else ""), file=sys.stderr)
            return 1
        out0.seek(0)
        text = out0.read()
    if text:
        sys.stdout.write(text.decode())
        sys.stdout.flush()
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: be our own launcher (before torch / HIP are even imported)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    # stdout must carry exactly one JSON line, but RCCL writes its version banner
    # and topology warnings there: keep the real stdout aside and point fd 1 at
    # stderr for everything else
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # the pool's host driver only supports dmabuf IPC: without this RCCL cannot share buffers across ranks
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from krylovfspssa_amd import KfspContext, synth

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx = KfspContext(local_rank)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    def comm_setup():
        """(re)create the library's communicator: a fresh unique id from rank 0, broadcast by the launcher"""
        if world == 1 and args.force_comm:
            ctx.comm_init(1, 0, KfspContext.unique_id())
        if world > 1:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.from_numpy(KfspContext.unique_id()))
            dist.broadcast(idt, 0)
            ctx.comm_init(world, rank, idt.cpu().numpy())

    comm_setup()

    # ---------------------------------------------------------------- SpMV
    mdl, desc = spmv_model(args.workload, world)
    if mdl is None:
        sys.exit(f"workload {args.workload} is defined for one GPU")
    row0, nrows = ctx.row_block(mdl.n)
    device_build = (args.device_build or args.workload == "c5") and hasattr(mdl, "rows_at") and not args.matrix_free
    t0 = time.time()
    if device_build or args.matrix_free and hasattr(mdl, "rows_at"):
        rowptr = col = val = None                     # no host arrays of the size of the FSP at all
        nnz_local = mdl.nnz() if world == 1 else mdl.nnz_rows(row0, nrows)
    else:
        rowptr, col, val = mdl.csr_rows(row0, nrows)
        nnz_local = int(rowptr[-1])
    t_gen = time.time() - t0
    nnz_global = mdl.nnz()
    x = np.random.default_rng(12345 + rank).random(nrows)

    def sampled_rows(model, r0, nr, csr, xg, k=2048):
        """(local rows, A x there, |A||x| there) in numpy: from the uploaded gather rows, or - when there are none -
        from the model itself at those states"""
        rows = np.unique(np.concatenate([np.arange(min(nr, 256)), np.arange(max(nr - 256, 0), nr),
                                         np.random.default_rng(7).integers(0, nr, k)]))
        if csr is not None:
            rp, cc, vv = csr
            ref = np.array([vv[rp[r]:rp[r + 1]] @ xg[cc[rp[r]:rp[r + 1]]] for r in rows])
            mag = np.array([np.abs(vv[rp[r]:rp[r + 1]]) @ np.abs(xg[cc[rp[r]:rp[r + 1]]]) for r in rows])
            return rows, ref, mag
        cols, vals = model.rows_at(r0 + rows)
        xs = xg[np.where(cols != np.iinfo(np.int64).max, cols, 0)]
        return rows, (vals * xs).sum(axis=1), (np.abs(vals) * np.abs(xs)).sum(axis=1)

    def load_generator():
        if args.matrix_free:
            ctx.set_matrix_box(mdl, store=False)
        elif device_build:
            ctx.set_matrix_box(mdl, store=True)
        else:
            ctx.set_matrix_csr(mdl.n, rowptr, col, val)
        ctx.set_vector(x)
        ctx.begin_step()                  # source column of the SpMV = x

    if args.workload == "c5":
        ctx.set_option("m_max", 30)       # 33 basis columns instead of 105 (0.9 GB each at 1.13e8 states)
    load_generator()
    info = ctx.matrix_info()
    b_alg_global = synth.spmv_alg_bytes(nnz_global, mdl.n)
    b_alg_local = synth.spmv_alg_bytes(nnz_local, nrows)

    inject = [int(v) for v in os.environ.get("KFSP_BENCH_INJECT_RAISE", "").split(",") if v]   # (tests: attempts that "raise")
    attempts = [0]

    def gathered(xloc, Lblk, nloc):
        """the whole source vector as every rank's kernel sees it (global index of row k of rank p = p L + k)"""
        if world == 1:
            return xloc
        xs = torch.zeros(Lblk, dtype=torch.float64, device="cuda")
        xs[:nloc] = torch.from_numpy(xloc).cuda()
        xall = torch.zeros(world * Lblk, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(xall, xs)
        return xall.cpu().numpy()

    def product_check(model=None, r0=None, nr=None, csr="main", xloc=None):
        """y = A x through the solver's own path (plain launch on one rank; halo strips or
        all-gather with more) against numpy on a sample of this rank's rows.
        -> (worst relative error over all ranks, some rank's product raised)"""
        model = mdl if model is None else model
        r0, nr = (row0, nrows) if r0 is None else (r0, nr)
        if csr == "main":
            csr = None if rowptr is None else (rowptr, col, val)
        xloc = x if xloc is None else xloc
        failed = None
        attempts[0] += 1
        try:
            y = ctx.spmv_w()
            if attempts[0] in inject:
                raise RuntimeError("injected failure (KFSP_BENCH_INJECT_RAISE)")
        except RuntimeError as e:
            failed = e
            y = None
        if world > 1 or args.force_comm:
            if max_over_ranks(1.0 if failed is not None else 0.0) > 0.0:
                print(f"[bench] rank {rank}: product failed on some rank ({failed})", file=sys.stderr)
                return float("inf"), True
        elif failed is not None:
            raise failed
        xg = gathered(xloc, _host.partition(model.n, world, rank)[2], nr)
        bad = 0.0
        if nr > 0:
            rows, ref, mag = sampled_rows(model, r0, nr, csr, xg)
            bad = float(np.max(np.abs(y[rows] - ref) / (mag + 1e-300)))
        return max_over_ranks(bad), False

    from krylovfspssa_amd import host as _host
    exchange = "none (single rank)"
    check_err, raised = product_check()
    if world > 1 or args.force_comm:
        exchange = "halo strips (banded generator), overlapped with the interior rows when the block is large"
        # never report a number from a wrong product: step down to the simpler exchanges.  A rank whose
        # product RAISED has left the library's communicator out of step with the others: every rank
        # learns of it (the launcher's own process group carries the verdict) and the communicator is
        # thrown away and made anew before the next mode is tried.
        for opt, label in (("overlap", "halo strips, not overlapped (overlap self-check failed)"),
                           ("halo", "all-gather of the whole vector (halo self-check failed)")):
            if check_err < 1e-12:
                break
            ctx.set_option(opt, 0)
            if raised:
                comm_setup()
            load_generator()
            exchange = label
            check_err, raised = product_check()
        if raised:
            print(f"[bench] rank {rank}: the product raised in every exchange mode; aborting all ranks", file=sys.stderr)
            sys.exit(3)
    check_ok = check_err < 1e-12
    ctx.set_vector(x)
    ctx.begin_step()

    ctx.spmv_bench(max(args.warmup, 1), args.variant)
    barrier()
    t0 = time.perf_counter()
    ms_events = ctx.spmv_bench(args.steps, args.variant)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ms_events = max_over_ranks(ms_events)
    value = args.steps * b_alg_global / elapsed / 1e9
    kern_ms = ms_events / args.steps
    # the exchange of the source vector alone, per rank (what every product of the partition pays before its rows)
    exch = None
    if world > 1 or args.force_comm:
        ex_ms, ex_bytes = ctx.exchange_bench(args.steps)
        ex_ms /= args.steps
        t = torch.tensor([ex_ms, float(ex_bytes)], dtype=torch.float64, device="cuda")
        if world > 1:
            allv = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allv, t)
        else:
            allv = [t]
        exch = {"per_rank_exchange_ms": [round(float(v[0]), 5) for v in allv],
                "per_rank_bytes_in": [int(v[1]) for v in allv],
                "per_rank_link_GBps": [round(float(v[1]) / max(float(v[0]), 1e-9) / 1e6, 2) for v in allv],
                "what": "the exchange that precedes every product (halo strips or all-gather), timed alone with HIP events on "
                        "each rank; link GB/s = bytes a rank receives per exchange / that time"}

    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    plain_run = world == 1 and not args.opt and not args.force_comm and args.variant in (0, 2) and not args.matrix_free

    def pmc_traffic(key):
        """HBM bytes per launch rocprofv3 counted for this workload (profiles/, NOT this run), or None"""
        if not (plain_run and os.path.exists(tpath)):
            return None, None
        try:
            t = json.load(open(tpath)).get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            t = None
        return t, (None if t is None else
                   "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not this run)")

    def roofline(ms, alg_bytes, real_bytes, traffic, traffic_src, kernel="k_spmv"):
        """The kernel against the HBM roofline.  `achieved` / `frac` count the bytes that really cross the HBM
        interface per launch - max(the layout's own byte count, the PMC-measured traffic) - so the fraction can
        not exceed 1; the SURVEY 8(d) CSR-model rate is kept beside it as alg_GBps (the banded and matrix-free forms
        store no column indices, so that model overstates what they move; no fraction of the peak is formed from it)."""
        moved = max(real_bytes, traffic or 0)
        sec = ms * 1e-3
        return {
            "bound": "hbm", "achieved": round(moved / sec / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(moved / sec / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kernel, "avg_launch_ms": round(ms, 5),
            "bytes_counted": "max(real_bytes_per_launch, traffic)",
            "real_bytes_per_launch": int(real_bytes),
            "real_GBps": round(real_bytes / sec / 1e9, 2),
            "frac_traffic": round(real_bytes / sec / 1e9 / HBM_PEAK_GBS, 4),
            "alg_bytes_per_launch": int(alg_bytes),
            "alg_GBps": round(alg_bytes / sec / 1e9, 2),
        }

    # bytes the kernel's own layout moves per launch (kfsp_matrix_bytes) and, when this run is the
    # profiled configuration, the HBM bytes rocprofv3 counted for it
    real_bytes = ctx.matrix_bytes(force_sell=(args.variant == 2))
    traffic, traffic_src = pmc_traffic(args.workload + ("_sell64" if args.variant == 2 else ""))
    roof = roofline(kern_ms, b_alg_local, real_bytes, traffic, traffic_src)
    roof["note"] = ("achieved/frac: bytes that cross the HBM interface per launch (the layout's own count: generator as stored + "
                    "24 B/row, or the rocprofv3 PMC traffic where it is larger) / HIP-event time of the timed launches / 8 TB/s. "
                    "alg_GBps: the SURVEY 8(d) CSR model (12 nnz + 20 N) over the same time - the figure `value` reports; it is NOT "
                    "a rate of bytes moved (the banded kernel stores no column indices) and may exceed the peak"
                    + ("; the time includes the exchange of the source vector" if world > 1 else ""))

    out = {
        "metric": "cme_generator_spmv_GBps",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong" if args.workload == "c5" else "weak",
        "vs_baseline": None,
        #  the physically bounded figure beside `value`: bytes that really cross the HBM interface per launch
        # over the same time, and its share of the 8 TB/s peak (= roofline.achieved / roofline.frac)
        "value_moved_GBps": roof["achieved"],
        "frac_of_peak": roof["frac"],
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {desc}; generator SpMV y=A x (FMATVEC) per step",
            "states_per_gpu": int(nrows), "states_total": int(mdl.n),
            "nnz_total": int(nnz_global), "alg_bytes_per_launch_total": int(b_alg_global),
            "value_is": "SURVEY 8(d) algorithmic bytes (12 nnz + 20 N) of the whole job / wall time of the timed launches",
            "partition": f"rows x{world}" if world > 1 else "single GPU",
            "exchange": exchange,
            "kernel_variant": "matrix-free box (no stored generator; factor tables in LDS)" if args.matrix_free else
                              {0: "auto (banded DIA when the rows allow it, else SELL-64)", 2: "sell64"}[args.variant],
            "generator_built": "on the device from the propensity tables (kfsp_set_matrix_box, box_store)" if device_build else
                               ("nothing stored" if args.matrix_free else "gather rows made by numpy, uploaded (kfsp_set_matrix_csr)"),
            "stored_slots_local": info["slots"],
        },
        "self_check": {"ok": bool(check_ok), "max_rel_err": check_err,
                       "what": "the timed product (same context, same exchange) vs numpy on 2.5k sampled rows per rank"},
        "roofline": roof,
        "exchange": exch,
        "input_generation_s": round(t_gen, 2),
    }

    # ------------------------------------------- the same product, matrix-free
    # (box workloads with separable propensities: no stored generator, kfsp_set_matrix_box; reported
    # next to the headline number, which stays the stored generator every FSP can use)
    if not args.matrix_free and getattr(mdl, "deps", None) is not None and args.variant == 0:
        ctx.set_matrix_box(mdl, store=False)
        ctx.set_vector(x)
        ctx.begin_step()
        mf_err, mf_raised = product_check()
        if mf_raised:
            # (an extra, not the headline: note it, make the communicator anew, carry on)
            out["matrix_free"] = {"failed": "the matrix-free product raised on some rank"}
            comm_setup()
        else:
            ctx.set_vector(x)
            ctx.begin_step()
            ctx.spmv_bench(max(args.warmup, 1), 0)
            barrier()
            mf_ms = max_over_ranks(ctx.spmv_bench(args.steps, 0)) / args.steps
            barrier()
            mf_bytes = ctx.matrix_bytes()
            mf_traffic, _ = pmc_traffic(args.workload + "_matrix_free")
            mf_moved = max(mf_bytes, mf_traffic or 0)
            out["matrix_free"] = {
                "what": "y = A x of the same workload with NO stored generator: propensity factor tables in LDS, rows rebuilt "
                        "from the row index (kfsp_set_matrix_box)",
                "avg_launch_ms": round(mf_ms, 5),
                "alg_GBps": round(b_alg_local / (mf_ms * 1e-3) / 1e9, 2),
                "real_bytes_per_launch": mf_bytes,
                "real_GBps": round(mf_bytes / (mf_ms * 1e-3) / 1e9, 2),
                "speedup_vs_stored": round(kern_ms / mf_ms, 3),
                "self_check": {"ok": bool(mf_err < 1e-12), "max_rel_err": mf_err},
                "traffic": mf_traffic,
                "frac": round(mf_moved / (mf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_16B_per_state": round(mf_bytes / (mf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bound": "3-species boxes: within 10 % of the fabric rate for its traffic; 6-species boxes: re-fetching x through L2s "
                         "that cannot hold its far strides (DESIGN 4.1b, 10.5; profiles/r03_box_lds_and_tiling.txt)",
            }

    # ------------------------------------------- the 10^7-state product (the size of the 60 % target)
    # c3x: repressilator box 216 x 216 x (216 * ranks), 1.0e7 states per GPU, built on the device from the
    # propensity tables (no host arrays), stored (banded) and matrix-free, both byte counts
    if args.workload == "c3" and not args.no_1e7 and not args.matrix_free and args.variant == 0:
        big, _ = spmv_model("c3x", world)
        br0, bnr = ctx.row_block(big.n)
        xb = np.random.default_rng(54321 + rank).random(bnr)
        blk = {"workload": f"c3x: repressilator_model.input propensities, box 216x216x(216*ranks), {bnr} states per GPU, "
                           "stored generator written out on the device (box_store)"}
        for label, store in (("stored", True), ("matrix_free", False)):
            ctx.set_matrix_box(big, store=store)
            ctx.set_vector(xb)
            ctx.begin_step()
            e7, r7 = product_check(big, br0, bnr, None, xb)
            if r7:
                blk[label] = {"failed": "the product raised on some rank"}
                comm_setup()
                continue
            ctx.set_vector(xb)
            ctx.begin_step()
            ctx.spmv_bench(max(args.warmup, 1), 0)
            barrier()
            ms7 = max_over_ranks(ctx.spmv_bench(args.steps, 0)) / args.steps
            barrier()
            nnz7 = big.nnz() if world == 1 else big.nnz_rows(br0, bnr)
            t7, t7src = pmc_traffic("c3x" if store else "c3x_matrix_free")
            blk[label] = roofline(ms7, synth.spmv_alg_bytes(nnz7, bnr), ctx.matrix_bytes(), t7, t7src)
            blk[label]["self_check"] = {"ok": bool(e7 < 1e-12), "max_rel_err": e7}
        out["spmv_1e7"] = blk

    # ------------------------------------------- a real (non-box) FSP at 10^7 states: the SELL kernel, HBM-resident
    if args.fsp and world == 1:
        t0 = time.time()
        fsp = synth.GoutsiasEllipsoid()
        fadj, foff, fdiag = fsp.ell()
        nnz_f = int((fadj > 0).sum()) + fsp.n
        xf = np.random.default_rng(777).random(fsp.n)
        # numpy reference on sampled rows (gather form of the reference arrays: row r collects OFFDIAG(k, i) x_i over ADJ(k, i) = r)
        rows = np.random.default_rng(8).integers(0, fsp.n, 1500)
        want = np.isin(fadj, rows + 1)
        src, slot = np.nonzero(want)
        tgt = fadj[src, slot] - 1
        ref = -fdiag[rows] * xf[rows]
        acc = {int(r): 0.0 for r in rows}
        mag = {int(r): abs(fdiag[r] * xf[r]) for r in rows}
        for i, k, r in zip(src.tolist(), slot.tolist(), tgt.tolist()):
            acc[r] += foff[i, k] * xf[i]
            mag[r] += abs(foff[i, k] * xf[i])
        ref = np.array([acc[int(r)] - fdiag[r] * xf[r] for r in rows])
        magv = np.array([mag[int(r)] for r in rows])
        blk = {"workload": f"Goutsias model on an ellipsoid of (M, D, RNA) x the 6 conserved DNA configurations, {fsp.n} states, {nnz_f} nonzeros, "
                           "listed in the order of a reachability search (synth.GoutsiasEllipsoid); reference-layout upload (kfsp_set_matrix_ell)",
               "input_generation_s": round(time.time() - t0, 1)}
        for label, order, code in (("sell_caller_order", 0, 0), ("sell_internal_order", 1, 0), ("sell_internal_order_coded_columns", 1, 1)):
            ctx.set_option("state_order", order)
            ctx.set_option("state_order_min", 1)
            ctx.set_option("state_order_products", 0)
            ctx.set_option("sell_code", code)
            ctx.set_option("m_max", 8)
            ctx.set_state_coords(fsp.state)
            ctx.set_matrix_ell(fadj, foff, fdiag)
            yf = ctx.spmv(xf)
            err = float(np.max(np.abs(yf[rows] - ref) / (magv + 1e-300)))
            ctx.set_vector(xf)
            ctx.begin_step()
            ctx.spmv_bench(max(args.warmup, 1), 0)
            msf = ctx.spmv_bench(args.steps, 0) / args.steps
            key = {"sell_caller_order": "fsp_sell_search_order", "sell_internal_order": "fsp_sell", "sell_internal_order_coded_columns": "fsp_sell_coded"}[label]
            tr, trs = pmc_traffic(key)
            blk[label] = roofline(msf, synth.spmv_alg_bytes(nnz_f, fsp.n), ctx.matrix_bytes(), tr, trs)
            blk[label]["self_check"] = {"ok": bool(err < 1e-12), "max_rel_err": err}
            blk[label]["layout"] = ctx.layout_info()
        out["spmv_fsp"] = blk
        ctx.set_option("state_order", 1)
        ctx.set_option("state_order_min", 32768)
        ctx.set_option("state_order_products", 48)
        ctx.set_option("sell_code", -1)
        ctx.set_option("m_max", 100)
        del fadj, foff, fdiag

    # ------------------------------------------- exp(tA)v on THIS workload's own generator (the large boxes)
    # fixed Krylov dimension m = 30 as in the c2 recipe; tau small enough that the FSP keeps its mass (the bench times the
    # kernels, not the physics); p0 = normalised Poisson product.  One GPU; c5 keeps its 33-column basis (m_max = 30).
    if args.workload in ("c3x", "c4", "c5s", "c5") and not args.no_expv and world == 1 and args.variant == 0:
        m_big, tau_big, n_big = 30, 1.0e-3, 4
        load_generator()
        lam = 8.0 if args.workload in ("c5", "c5s") else 20.0
        vs = []
        for dim in mdl.dims:                     # (normalised product of Poisson pmfs over the box's own axes, fastest first)
            xk = np.arange(dim, dtype=np.float64)
            lg = np.concatenate(([0.0], np.cumsum(np.log(np.arange(1, dim, dtype=np.float64)))))
            vs.append(np.exp(xk * np.log(lam) - lam - lg))
        pb = synth._outer_fastest_first(vs)
        pb /= pb.sum()
        ctx.set_vector(pb)
        ctx.expv_fixed(m_big, tau_big, 1)
        ctx.timers(reset=True)
        barrier()
        t0 = time.perf_counter()
        wsb = ctx.expv_fixed(m_big, tau_big, n_big)
        barrier()
        tb = time.perf_counter() - t0
        gen_b = ctx.matrix_bytes()
        moved = (m_big + 1) * gen_b + m_big * (16 + 32) * mdl.n + 8 * mdl.n * (m_big + 1) + 24 * mdl.n
        b_ref = (m_big + 1) * b_alg_global + m_big * 104 * mdl.n + 8 * mdl.n * (m_big + 1) + 24 * mdl.n
        out["expv_workload"] = {
            "workload": f"{args.workload}: exp(tau A)v on the SpMV workload's own generator ({'matrix-free' if args.matrix_free else 'stored'}), "
                        f"N={mdl.n}, Krylov m={m_big}, tau={tau_big}, {n_big} steps",
            "ms_per_step": round(tb / n_big * 1e3, 3),
            "moved_GBps": round(n_big * moved / tb / 1e9, 1),
            "frac_of_peak": round(n_big * moved / tb / 1e9 / HBM_PEAK_GBS, 4),
            "alg_GBps": round(n_big * b_ref / tb / 1e9, 1),
            "bytes_moved_per_step": int(moved),
            "bytes_counted": "(m+1) x (generator as stored + 24 B/row) + m x 48 B/row (two dots read with the product, one fused update "
                             "pass) + the combine's (m+1) column reads + w",
            "mass_final": float(wsb[-1]),
            "timers_ms_per_step": {k: round(v / n_big, 3) for k, v in ctx.timers().items()},
        }
        del pb

    # ---------------------------------------------------------------- expv
    if not args.no_expv:
        ctx.set_option("m_max", 100)
        tg = synth.toggle(1000, 1000 * world) if args.workload != "tiny" else synth.toggle(100, 80 * world)
        r0, nr = ctx.row_block(tg.n)
        rp, cc, vv = tg.csr_rows(r0, nr)
        ctx.set_matrix_csr(tg.n, rp, cc, vv)
        p0 = synth.poisson_p0(tg, 30.0)
        m, tau = 30, 0.01
        ctx.set_vector(p0[r0:r0 + nr])
        ctx.expv_fixed(m, tau, 1)                 # warm-up
        ctx.set_vector(p0[r0:r0 + nr])
        barrier()
        t0 = time.perf_counter()
        ws = ctx.expv_fixed(m, tau, args.expv_steps)
        barrier()
        t_expv = max_over_ranks(time.perf_counter() - t0)
        w_gpu = ctx.get_vector()
        nnz_t = tg.nnz()
        # reference (unfused) byte count of one fixed-m step
        b_step = (m + 1) * synth.spmv_alg_bytes(nnz_t, tg.n) + m * 104 * tg.n + 8 * tg.n * (m + 1) + 24 * tg.n
        out["expv"] = {
            "workload": f"c2: toggle_model.input propensities, box 1000x(1000*ranks), N={tg.n}, Krylov m={m}, tau={tau}, {args.expv_steps} steps",
            "ms_per_step": round(t_expv / args.expv_steps * 1e3, 4),
            "wall_s": round(t_expv, 5),
            "alg_GBps": round(args.expv_steps * b_step / t_expv / 1e9, 1),
            "mass_final": float(ws[-1]),
            "timers_ms": ctx.timers(),
        }
        if not args.matrix_free:
            # the same recipe on the matrix-free form of the same box
            ctx.set_matrix_box(tg, store=False)
            ctx.set_vector(p0[r0:r0 + nr])
            ctx.expv_fixed(m, tau, 1)
            ctx.set_vector(p0[r0:r0 + nr])
            barrier()
            t0 = time.perf_counter()
            ws2 = ctx.expv_fixed(m, tau, args.expv_steps)
            barrier()
            t2 = max_over_ranks(time.perf_counter() - t0)
            out["expv"]["matrix_free_ms_per_step"] = round(t2 / args.expv_steps * 1e3, 4)
            out["expv"]["matrix_free_l1_vs_stored"] = float(np.abs(ctx.get_vector() - w_gpu).sum())
            ctx.set_matrix_csr(tg.n, rp, cc, vv)      # back to the stored form for the parity leg below
    else:
        tg = None

    # ---------------------------------------------------------- CPU baseline
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as O
        O.lib()
        cpu_mdl, cpu_what = mdl, f"the same {args.workload} matrix"
        if args.workload == "c5":
            # bounded sample: the per-GPU slab of the same network (22^5 x 3 = 1.55e7 states, 1.9e8 nonzeros)
            cpu_mdl, cpu_what = synth.birth_death((22, 22, 22, 22, 22, 3)), "the 22^5 x 3 slab of the same network (c5s)"
        adj, off, diag = cpu_mdl.ell()
        A = O.EllMatrix(adj, off, diag)
        xc = np.random.default_rng(12345).random(cpu_mdl.n)
        b_cpu = synth.spmv_alg_bytes(cpu_mdl.nnz(), cpu_mdl.n)
        O.spmv_ell(A, xc)
        reps = 0
        t0 = time.perf_counter()
        while True:
            O.spmv_ell(A, xc)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 12.0 or reps >= 400:
                break
        cpu_gbs = reps * b_cpu / dt / 1e9
        out["cpu_baseline"] = {
            "value": round(cpu_gbs, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} scatter-form SpMVs (KrylovSolver.f90:593-606 loop order) on {cpu_what}, "
                      f"{dt:.1f} s, 1 of {os.cpu_count()} host cores",
            "ms_per_spmv": round(dt / reps * 1e3, 3),
        }
        del A, adj, off, diag
        if tg is not None:
            adj, off, diag = tg.ell()
            A = O.EllMatrix(adj, off, diag)
            ncpu = 2
            t0 = time.perf_counter()
            wc, wsc = O.expv_fixed(A, synth.poisson_p0(tg, 30.0), 30, 0.01, ncpu)
            dtc = time.perf_counter() - t0
            out["cpu_baseline"]["expv_ms_per_step"] = round(dtc / ncpu * 1e3, 2)
            out["cpu_baseline"]["expv_sample"] = f"{ncpu} steps of the c2 recipe"
            # parity of the GPU result after the same number of steps
            ctx.set_vector(synth.poisson_p0(tg, 30.0))
            ctx.expv_fixed(30, 0.01, ncpu)
            out["expv"]["l1_err_vs_cpu"] = float(np.abs(ctx.get_vector() - wc).sum())
            out["expv"]["l1_err_steps"] = ncpu

    ctx.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()
    if not check_ok:
        print(f"[bench] SELF-CHECK FAILED: max relative error {check_err:.3e}", file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
