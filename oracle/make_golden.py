#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Generates tests/golden/*.npz from the UNMODIFIED reference, compiled out of
/root/reference by oracle/Makefile into oracle/_ref/ref_dump (our driver
oracle/ref_dump.f90 linked with the reference's objects).

Run in the build container only (needs /root/reference and flang):

    make -C oracle && python oracle/make_golden.py

The fixtures are data only: inputs and the outputs the reference computed.
Nothing here is needed at test time; tests read the committed .npz files.
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def run_dump(args, log=None):
    """Run ref_dump in _ref/models (the .input files are opened relative to cwd,
    README.md:34) with an unlimited stack: DGEXPV_FSP keeps a 5 GB local
    workspace (KrylovSolver.f90:50-52)."""
    cmd = "ulimit -s unlimited && exec ../ref_dump " + " ".join(args)
    env = dict(os.environ, MKL_NUM_THREADS="1")
    out = subprocess.run(["bash", "-c", cmd], cwd=os.path.join(REF_DIR, "models"),
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         check=True, text=True).stdout
    if log is not None:
        with open(log, "w") as f:
            f.write(out)
    return out


def read_fsp(path):
    with open(path, "rb") as f:
        ns, nr, n = np.fromfile(f, dtype=np.int32, count=3)
        state = np.fromfile(f, dtype=np.int32, count=ns * n).reshape(n, ns)
        adj = np.fromfile(f, dtype=np.int32, count=nr * n).reshape(n, nr)
        offdiag = np.fromfile(f, dtype=np.float64, count=nr * n).reshape(n, nr)
        diag = np.fromfile(f, dtype=np.float64, count=n)
        vector = np.fromfile(f, dtype=np.float64, count=n)
    # arrays are stored [state][slot] = the Fortran (slot, state) column-major
    # layout of StateSpace.f90:13-17 read row-major
    return dict(ns=int(ns), nr=int(nr), n=int(n), state=state, adj=adj,
                offdiag=offdiag, diag=diag, vector=vector)


def parse_log(text):
    """Per-step records printed by PRINT_STATS (KrylovSolver.f90:641-651) and
    every WSUM evaluation (:452)."""
    fl = r"([-+0-9.Ee]+)"
    steps = []
    for m in re.finditer(r"TIMESTEP\s+(\d+)\s+-+\s*\n\s*FSP SIZE\s+=\s*(\d+)\s*\n\s*STEP_SIZE\s+=\s*" + fl +
                         r"\s*\n\s*NEXT_STEP\s+=\s*" + fl + r"\s*\n\s*T_NOW\s+=\s*" + fl +
                         r"\s*\n\s*KRYLOV DIMENSION\s+=\s*(\d+)", text):
        steps.append((int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)),
                      float(m.group(5)), int(m.group(6))))
    wsum = [float(x) for x in re.findall(r"WSUM=\s*" + fl, text)]
    s = np.array(steps, dtype=np.float64).reshape(-1, 6)
    return dict(step_no=s[:, 0].astype(np.int32), step_n=s[:, 1].astype(np.int32), step_tau=s[:, 2],
                step_next=s[:, 3], step_tnow=s[:, 4], step_m=s[:, 5].astype(np.int32),
                wsum=np.array(wsum), n_ssa=np.int32(len(re.findall("CALLING SSA", text))),
                n_reject=np.int32(len(re.findall("STEPSIZE REJECTED", text))),
                n_dimchange=np.int32(len(re.findall("DIMENSION CHANGED", text))))


SOLVE_CASES = {
    # fixture name: (ref_dump case, T, FSPTOL, KRYTOL); tolerances must match
    # oracle/ref_dump.f90 DO_SOLVE, T is passed as the override argument.
    "toggle_input": ("toggle_input", 1000.0, 1e-4, 1e-10),
    "toggle_example": ("toggle_example", 100.0, 1e-4, 1e-8),
    # short horizons of the same workloads: the FSP is grown by SSA + one-step
    # reachability several times, but the run ends before rounding-level
    # differences in the error estimate (a ~1e-10 entry of exp(tau*H)) can flip
    # a step-size / dimension decision, so state lists can be compared bit for bit
    "toggle_input_T02": ("toggle_input", 0.2, 1e-4, 1e-10),
    "toggle_input_T05": ("toggle_input", 0.5, 1e-4, 1e-10),
    "toggle_input_T2": ("toggle_input", 2.0, 1e-4, 1e-10),
    "toggle_example_T05": ("toggle_example", 0.5, 1e-4, 1e-8),
    "toggle_example_T2": ("toggle_example", 2.0, 1e-4, 1e-8),
    # closed systems, FSP never changes.  The short horizons stay away from the
    # stationary regime, where the reference's accept/reject decisions hinge on
    # 1e-10-level rounding of the scaled-and-squared Pade and on the absolute
    # happy-breakdown threshold (KrylovSolver.f90:249), so that an independent
    # implementation follows the identical (tau, m) sequence.
    "ring6": ("ring6", 20.0, 1e-4, 1e-10),
    "ring6_T40": ("ring6", 40.0, 1e-4, 1e-10),
    "ring6_long": ("ring6", 60.0, 1e-4, 1e-10),   # trajectory may fork late
    "ring4": ("ring4", 6.2, 1e-6, 1e-8),
    # the other two shipped models through the whole adaptive loop (3 and 6
    # species; parsed propensities, SSA + one-step growth, drops with compaction)
    "repressilator_input_T03": ("repressilator_input", 0.3, 1e-4, 1e-10),
    "repressilator_input_T1": ("repressilator_input", 1.0, 1e-4, 1e-10),
    "goutsias_input_T4": ("goutsias_input", 4.0, 1e-6, 1e-8),
    "goutsias_input_T15": ("goutsias_input", 15.0, 1e-6, 1e-8),
    "goutsias_input_T40": ("goutsias_input", 40.0, 1e-6, 1e-8),
    # compiled-in propensity functions (MODEL%CUSTOMPROP, oracle/ref_cases.f90): the reference's repressilator example over
    # a short horizon, and three variants of OURS whose production law depends on two species (not a product), on three, and
    # on a plane no probe visits - how the Fortran host's probing of such functions (KFSP_CUSTOMPROP) is tested
    "repressilator_example_T1": ("repressilator_example", 1.0, 1e-4, 1e-14),
    "goutsias_example_T15": ("goutsias_example", 15.0, 1e-6, 1e-8),
    "repressilator_pair_T2": ("repressilator_pair", 2.0, 1e-4, 1e-10),
    "repressilator_triple_T1": ("repressilator_triple", 1.0, 1e-4, 1e-10),
    "repressilator_trap_T2": ("repressilator_trap", 2.0, 1e-4, 1e-10),
}

ASSEMBLY_CASES = [("toggle", 5), ("toggle", 10), ("toggle", 20),
                  ("repressilator", 5), ("repressilator", 10),
                  ("goutsias", 5), ("goutsias", 10), ("goutsias", 16)]


# (model, rounds of SSA_EXTENDER + ONESTEP_EXTENDER, path duration)
SSA_CASES = [("toggle", 10, 0.1), ("repressilator", 8, 0.05), ("goutsias", 12, 2.0)]


def ssa_fixture_name(name, k, dt):
    return f"ssa_{name}_k{k}_dt{dt:g}.npz"


def make_ssa(tmp):
    # G1b: SSA paths + one-step reachability on the default random stream
    # (StateSpace.f90:347-396, :550-630); integer arrays bit-exact
    for name, k, dt in SSA_CASES:
        p = os.path.join(tmp, f"ssa_{name}_{k}.bin")
        run_dump(["ssa", name, str(k), repr(dt), p])
        d = read_fsp(p)
        np.savez_compressed(os.path.join(GOLDEN, ssa_fixture_name(name, k, dt)), k=np.int32(k), dt=dt,
                            ns=d["ns"], nr=d["nr"], n=d["n"], state=d["state"], adj=d["adj"], diag=d["diag"],
                            next_uniform=d["vector"][0])
        print(f"ssa {name} k={k} dt={dt}: N={d['n']}")


# (model, one-step sweeps before the drop, mass bound DSUM)
DROP_CASES = [("toggle", 20, 1e-6), ("goutsias", 12, 1e-7), ("goutsias", 16, 1e-12), ("repressilator", 10, 1e-4)]


def drop_fixture_name(name, k, dsum):
    return f"drop_{name}_k{k}_dsum{dsum:g}.npz"


def make_drop(tmp):
    # G1c: DROP_STATES + the sweep that follows it (StateSpace.f90:398-548), and
    # FIND_DROPTOL alone over ten mass bounds
    for name, k, dsum in DROP_CASES:
        p = os.path.join(tmp, f"drop_{name}_{k}.bin")
        run_dump(["drop", name, str(k), repr(dsum), p])
        d = read_fsp(p)
        np.savez_compressed(os.path.join(GOLDEN, drop_fixture_name(name, k, dsum)), k=np.int32(k), dsum=dsum, **d)
        print(f"drop {name} k={k} dsum={dsum}: N={d['n']}")
    p = os.path.join(tmp, "droptol.bin")
    run_dump(["droptol", p])
    a = np.fromfile(p).reshape(2, -1)
    np.savez_compressed(os.path.join(GOLDEN, "droptol.npz"), dsum=a[0], droptol=a[1])
    print("droptol", a[1])


def make_exprtable(tmp):
    # G4b: the reference's expression type on a model file of our own that exercises operator
    # classes, associativity, unary minus, '**', all functions, D/E exponents, dotted names
    # and the error rules (x/0, log of a non-positive number)
    p = os.path.join(tmp, "exprtable.bin")
    run_dump(["exprtable", p])
    P = np.fromfile(p).reshape(13, 13, 3, 16)
    st = np.fromfile(p + ".stoich", dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN, "exprtable.npz"), P=P, stoich=st[2:].reshape(int(st[1]), int(st[0])))
    print("exprtable", P.shape, float(np.abs(P).max()))


def make_api(tmp):
    # G1d: ADD / INDEX / PROBABILITY as a driver may call them (StateSpace.f90:19-45)
    p = os.path.join(tmp, "api.bin")
    run_dump(["api", p])
    d = read_fsp(p)
    with open(p + ".q", "rb") as f:
        idx = np.fromfile(f, dtype=np.int32, count=8)
        prob = np.fromfile(f, dtype=np.float64, count=8)
    np.savez_compressed(os.path.join(GOLDEN, "api_toggle.npz"), idx=idx, prob=prob, **d)
    print("api", d["n"], idx)


# (fixture, ref_cases workload, number of leading steps whose solution vector is kept)
# (fixture name = a SOLVE_CASES entry, ref_cases workload, number of leading steps whose solution vector is kept)
LOCKSTEP_CASES = [("toggle_input", "toggle_input", 72), ("toggle_example", "toggle_example", 72),
                  ("repressilator_input_T1", "repressilator_input", 40), ("goutsias_input_T40", "goutsias_input", 60)]


def run_trace(case, trace, out, T=None):
    cmd = "ulimit -s unlimited && exec ../ref_trace " + " ".join([case, trace, out] + ([repr(T)] if T else []))
    env = dict(os.environ, MKL_NUM_THREADS="1")
    return subprocess.run(["bash", "-c", cmd], cwd=os.path.join(REF_DIR, "models"), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True, text=True).stdout


def make_lockstep(tmp, only=()):
    """G7: the reference's own decisions, state lists and solution vectors step by step
    (oracle/ref_trace_main.f90 + the BLAS observers of oracle/ref_trace.c), for the lock-step
    tests.  The observed run must be bit-identical to the plain one (solve_<name>.npz)."""
    from oracle import lockstep as L
    for name, case, keep in LOCKSTEP_CASES:
        if only and name not in only:
            continue
        _, T, fsptol, krytol = SOLVE_CASES[name]
        trace, out = os.path.join(tmp, f"{name}.trace"), os.path.join(tmp, f"{name}.tr.bin")
        text = run_trace(case, trace, out, T)
        plain = np.load(os.path.join(GOLDEN, f"solve_{name}.npz"))
        dout = read_fsp(out)
        assert np.array_equal(dout["state"], plain["state"]) and np.array_equal(dout["vector"], plain["vector"]), \
            "the observers changed the run"
        log = parse_log(text)
        r = L.build_script(L.read_trace(trace), log, T, fsptol)
        ev = r["events"]
        arrays = dict(T=T, fsptol=fsptol, krytol=krytol, script=r["script"], n_after=r["n_after"],
                      beta=np.array([e["beta"] for e in ev if e["tag"] == "B"]),
                      fsp_at=np.array([k for k, _ in r["fsps"]], dtype=np.int64), keep=np.int64(keep),
                      final_state=dout["state"], final_vector=dout["vector"])
        for k, f in r["fsps"]:
            arrays[f"state_{k}"] = f["state"]
        for k in range(min(keep, len(r["w_after"]))):
            arrays[f"w_{k}"] = r["w_after"][k]
        # time at every B event (accepted steps advance it, give-ups do not)
        t_now, times = 0.0, [0.0]
        for row in r["script"]:
            if row[0] == L.FSP and row[1] == 0.0:
                t_now += row[2]
            if row[0] == L.END:
                times.append(t_now)
        arrays["t_at"] = np.array(times[:len(r["n_after"])])
        np.savez_compressed(os.path.join(GOLDEN, f"lockstep_{name}.npz"), **arrays)
        print(f"lockstep {name}: {len(r['script'])} script rows, {len(r['n_after'])} steps, {len(r['fsps'])} state lists")
        if name == "toggle_input":
            make_sensitivity_sample(r, os.path.join(GOLDEN, "lockstep_sample_toggle.npz"))


from oracle.lockstep import list_hash, state_weights  # noqa: E402  (shared with the tests)


# longer runs whose state lists and vectors are too big to keep: per step a checksum of the state list
# (oracle.lockstep.list_hash) and eight weighted sums of the solution vector (|difference of a sum| <= l1
# difference of the vectors)
LOCKSTEP_DIGEST_CASES = [("goutsias_input_T100", "goutsias_input", 100.0, 1e-6, 1e-8),
                         ("repressilator_input_T10", "repressilator_input", 10.0, 1e-4, 1e-10),
                         # the horizon of the reference's own example (N -> 1.03e6; the reference needs ~40 min and
                         # writes a ~50 GB trace, which is streamed): KFSP_CASE_CAPACITY as for digest_goutsias_input_T300
                         ("goutsias_input_T300", "goutsias_input", 300.0, 1e-6, 1e-8)]


def make_lockstep_digest(tmp, only=()):
    from oracle import lockstep as L
    for name, case, T, fsptol, krytol in LOCKSTEP_DIGEST_CASES:
        if (only and name not in only) or (not only and T > 100.0):
            continue
        trace, out = os.path.join(tmp, f"{name}.trace"), os.path.join(tmp, f"{name}.tr.bin")
        os.environ["KFSP_CASE_CAPACITY"] = "4194319"
        text = run_trace(case, trace, out, T)
        dout = read_fsp(out)
        r = L.build_script(L.read_trace_digest(trace), parse_log(text), T, fsptol)
        os.remove(trace)
        t_now, times = 0.0, [0.0]
        for row in r["script"]:
            if row[0] == L.FSP and row[1] == 0.0:
                t_now += row[2]
            if row[0] == L.END:
                times.append(t_now)
        np.savez_compressed(os.path.join(GOLDEN, f"lockstep_digest_{name}.npz"), T=T, fsptol=fsptol, krytol=krytol,
                            script=r["script"], n_after=r["n_after"], t_at=np.array(times[:len(r["n_after"])]),
                            list_hash=np.array([w["list_hash"] for w in r["w_after"]]),
                            proj=np.array([w["proj"] for w in r["w_after"]]), final_hash=list_hash(dout["state"]),
                            final_proj=dout["vector"] @ state_weights(dout["state"]), final_n=np.int64(dout["n"]))
        print(f"lockstep digest {name}: {len(r['script'])} script rows, {len(r['n_after'])} steps, N -> {dout['n']}")


def make_sensitivity_sample(r, path):
    """One Krylov pass of the toggle_input run in full (the first pass whose error estimate an
    independent implementation does not reproduce, step 4: N = 438, m = 75): start vector,
    generator, the reference's Hessenberg matrix and AVNORM, its step sizes, the coefficient vector
    and the solution it produced; plus a few (H, t, exp(tH) e1) triples of DGPADM(norm) calls."""
    from oracle import oracle as O
    ev = r["events"]

    def err_loc(E, m, beta, avnorm):
        p1, p2 = abs(E[m, 0]) * beta, abs(E[m + 1, 0]) * beta * avnorm
        return p2 if p1 > 10.0 * p2 else (p1 * p2 / (p1 - p2) if p1 > p2 else p1)

    # the first time step (single Krylov pass: no dimension change inside it) whose error estimate
    # the C restatement misses by more than 1 %
    bs = [i for i, e in enumerate(ev) if e["tag"] == "B"]
    F, chosen = None, None
    for a, b in zip(bs, bs[1:]):
        seg = ev[a:b]
        F = next((e for e in seg if e["tag"] == "F"), F)
        ps = [e for e in seg if e["tag"] == "P"]
        cs = [e for e in seg if e["tag"] == "C"]
        if len({p["m"] for p in ps}) != 1 or ps[0]["mx"] != ps[0]["m"] + 2 or not cs or cs[0]["mx"] != ps[0]["m"] + 1:
            continue
        m, beta = ps[0]["m"], ev[a]["beta"]
        A = O.EllMatrix(F["adj"], F["offdiag"], F["diag"])
        V, H, mb, k1, av = O.arnoldi(A, ev[a]["w"] / beta, m)
        e_o = err_loc(O.padm(H, ps[0]["t"])[0], m, beta, av)
        e_r = err_loc(O.padm(ps[0]["H"], ps[0]["t"])[0], m, beta, ps[0]["avnorm"])
        if abs(e_o - e_r) > 1e-2 * e_r:
            chosen = (a, seg, ps, cs)
            break
    assert chosen is not None
    a, seg, ps, cs = chosen
    ib = a
    # krylov evaluations before the first solution update, and that update
    kp = [e for e in seg[:seg.index(cs[0])] if e["tag"] == "P"]
    c = cs[0]
    s = seg[seg.index(c) + 1]
    assert s["tag"] == "S"
    out = dict(adj=F["adj"], offdiag=F["offdiag"], diag=F["diag"], w0=ev[ib]["w"], beta=ev[ib]["beta"],
               H=kp[0]["H"], avnorm=kp[0]["avnorm"], t_first=kp[0]["t"], t_second=kp[-1]["t"],
               y=c["y"], w1=s["w"], wsum=s["wsum"])
    ps = kp
    # DGPADM samples: (H, t) of an accepted evaluation and the coefficient vector DGEMV received
    k = 0
    last_p = None
    for e in ev:
        if e["tag"] == "P":
            last_p = e
        if e["tag"] == "C" and last_p is not None and last_p["mx"] >= e["mx"] and k < 8 and e["mx"] in (38, 79, 77, 102, 93, 68, 58, 45, 90):
            out[f"pH{k}"] = last_p["H"]
            out[f"pt{k}"] = last_p["t"]
            out[f"py{k}"] = e["y"]
            k += 1
    out["npade"] = np.int32(k)
    np.savez_compressed(path, **out)
    print(f"sensitivity sample: step {bs.index(ib)} N={F['n']} m={ps[0]['m']} t={[p['t'] for p in ps]} pade samples={k}")


def fsp_digest(d, top=4000):
    """What a 10^6-state result is compared by (the full list is 40 MB): size, mass, the marginal
    distribution of every species, and the `top` most probable states with their probabilities."""
    st, p = d["state"], d["vector"]
    out = dict(n=np.int64(d["n"]), mass=float(p.sum()), ns=np.int32(d["ns"]))
    for s in range(d["ns"]):
        out[f"marginal_{s}"] = np.bincount(st[:, s], weights=p)
    order = np.argsort(-p, kind="stable")[:top]
    out["top_state"] = st[order]
    out["top_prob"] = p[order]
    return out


def make_goutsias_T300():
    """G8: models/goutsias_model.input over the horizon of examples/transcr6d.f90:16 (T = 300, FSPTOL 1e-6,
    KRYTOL 1e-8; N -> 1.0e6 states, ~40 min on one core): digest of the reference's result."""
    tmp = tempfile.mkdtemp(prefix="kfsp_golden_")
    p = os.path.join(tmp, "g300.bin")
    cmd = "ulimit -s unlimited && exec ../ref_dump solve goutsias_input " + p + " 300"
    env = dict(os.environ, MKL_NUM_THREADS="1", KFSP_CASE_CAPACITY="2097169")
    text = subprocess.run(["bash", "-c", cmd], cwd=os.path.join(REF_DIR, "models"), env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, check=True, text=True).stdout
    save_goutsias_T300(p, text)


def save_goutsias_T300(binfile, logtext):
    d = read_fsp(binfile)
    log = parse_log(logtext)
    np.savez_compressed(os.path.join(GOLDEN, "digest_goutsias_input_T300.npz"), T=300.0, fsptol=1e-6, krytol=1e-8,
                        steps=np.int32(len(log["step_no"])), n_ssa=log["n_ssa"], peak_n=np.int64(log["step_n"].max()),
                        **fsp_digest(d))
    print(f"goutsias T=300: N={d['n']} steps={len(log['step_no'])} mass={d['vector'].sum()!r}")


# G9: digests of whole reference runs at the horizons the end-to-end times are quoted on (name -> case, T, FSPTOL,
# KRYTOL, FSP capacity); `make_golden.py digest [name ...]` runs the compiled reference, `make_golden.py digest
# name file.bin file.log` digests an existing run.  goutsias_example = examples/transcr6d.f90 (compiled-in propensities,
# ~40 min on one core), repressilator_example = examples/repressilator.f90, toggle_example = examples/toggle.f90.
DIGEST_CASES = {
    "toggle_input_T1000": ("toggle_input", 1000.0, 1e-4, 1e-10, None),
    "repressilator_input_T10": ("repressilator_input", 10.0, 1e-4, 1e-10, 2097169),
    "toggle_example_T100": ("toggle_example", 100.0, 1e-4, 1e-8, None),
    "repressilator_example_T10": ("repressilator_example", 10.0, 1e-4, 1e-14, 2097169),
    "goutsias_example_T300": ("goutsias_example", 300.0, 1e-6, 1e-8, 2097169),
}


def save_digest(name, binfile, logtext):
    case, T, fsptol, krytol, cap = DIGEST_CASES[name]
    d = read_fsp(binfile)
    log = parse_log(logtext)
    np.savez_compressed(os.path.join(GOLDEN, f"digest_{name}.npz"), T=T, fsptol=fsptol, krytol=krytol,
                        steps=np.int32(len(log["step_no"])), n_ssa=log["n_ssa"], peak_n=np.int64(log["step_n"].max()),
                        **fsp_digest(d))
    print(f"digest {name}: N={d['n']} steps={len(log['step_no'])} mass={d['vector'].sum()!r}")


def make_digests(only=()):
    tmp = tempfile.mkdtemp(prefix="kfsp_golden_")
    for name, (case, T, fsptol, krytol, cap) in DIGEST_CASES.items():
        if only and name not in only:
            continue
        p = os.path.join(tmp, name + ".bin")
        env = dict(os.environ, MKL_NUM_THREADS="1", **({"KFSP_CASE_CAPACITY": str(cap)} if cap else {}))
        cmd = f"ulimit -s unlimited && exec ../ref_dump solve {case} {p} {T!r}"
        text = subprocess.run(["bash", "-c", cmd], cwd=os.path.join(REF_DIR, "models"), env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, check=True, text=True).stdout
        save_digest(name, p, text)


def main():
    if sys.argv[1:2] == ["digest"]:
        if len(sys.argv) == 5 and sys.argv[2] in DIGEST_CASES and os.path.exists(sys.argv[3]):
            save_digest(sys.argv[2], sys.argv[3], open(sys.argv[4]).read())      # from an existing run
        else:
            make_digests(set(sys.argv[2:]))
        return
    if sys.argv[1:2] == ["goutsias300"]:
        if len(sys.argv) > 3:
            save_goutsias_T300(sys.argv[2], open(sys.argv[3]).read())      # from an existing run
        else:
            make_goutsias_T300()
        return
    if sys.argv[1:2] == ["lockstep"]:
        make_lockstep(tempfile.mkdtemp(prefix="kfsp_golden_"), set(sys.argv[2:]))
        return
    if sys.argv[1:2] == ["lockstep_digest"]:
        make_lockstep_digest(tempfile.mkdtemp(prefix="kfsp_golden_"), set(sys.argv[2:]))
        return
    if not os.path.exists(os.path.join(REF_DIR, "ref_dump")):
        sys.exit("oracle/_ref/ref_dump missing: run `make -C oracle` where /root/reference exists")
    os.makedirs(GOLDEN, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="kfsp_golden_")
    # `make_golden.py statespace` / `make_golden.py solve <fixture> ...` refresh a subset
    only = set(sys.argv[2:]) if sys.argv[1:2] == ["solve"] else None
    if not only:
        make_ssa(tmp)
        make_drop(tmp)
        make_exprtable(tmp)
        make_api(tmp)
    if sys.argv[1:] == ["statespace"]:
        return

    # G1: assembly (integer arrays bit-exact, StateSpace.f90:248-396)
    for name, k in ([] if only else ASSEMBLY_CASES):
        p = os.path.join(tmp, f"asm_{name}_{k}.bin")
        run_dump(["assembly", name, str(k), p])
        d = read_fsp(p)
        d.pop("vector")
        np.savez_compressed(os.path.join(GOLDEN, f"assembly_{name}_k{k}.npz"), k=np.int32(k), **d)
        print(f"assembly {name} k={k}: N={d['n']}")

    # G3/G6: CME_SOLVE end to end
    for name, (case, T, fsptol, krytol) in SOLVE_CASES.items():
        if only and name not in only:
            continue
        p = os.path.join(tmp, f"solve_{name}.bin")
        text = run_dump(["solve", case, p, repr(T)])
        din = read_fsp(p + ".in")
        dout = read_fsp(p)
        log = parse_log(text)
        np.savez_compressed(
            os.path.join(GOLDEN, f"solve_{name}.npz"),
            T=T, fsptol=fsptol, krytol=krytol, ns=dout["ns"], nr=dout["nr"],
            in_n=din["n"], in_state=din["state"], in_vector=din["vector"],
            n=dout["n"], state=dout["state"], adj=dout["adj"], offdiag=dout["offdiag"],
            diag=dout["diag"], vector=dout["vector"], **log)
        print(f"solve {name}: N={dout['n']} steps={len(log['step_no'])} sum={dout['vector'].sum()!r}")
    if only:
        return

    # G5: DGPADM
    p = os.path.join(tmp, "padm.bin")
    run_dump(["padm", p])
    out = {}
    with open(p, "rb") as f:
        ncase = int(np.fromfile(f, dtype=np.int32, count=1)[0])
        for c in range(ncase):
            m = int(np.fromfile(f, dtype=np.int32, count=1)[0])
            t = float(np.fromfile(f, dtype=np.float64, count=1)[0])
            H = np.fromfile(f, dtype=np.float64, count=m * m).reshape(m, m).T.copy()
            ns = int(np.fromfile(f, dtype=np.int32, count=1)[0])
            E = np.fromfile(f, dtype=np.float64, count=m * m).reshape(m, m).T.copy()
            out[f"m{c}"] = np.int32(m)
            out[f"t{c}"] = t
            out[f"H{c}"] = H
            out[f"ns{c}"] = np.int32(ns)
            out[f"E{c}"] = E
    np.savez_compressed(os.path.join(GOLDEN, "padm.npz"), ncase=np.int32(ncase), **out)
    print(f"padm: {ncase} cases")

    # G4: parser propensity table (TestModelParser.f90:33-43)
    p = os.path.join(tmp, "prop.bin")
    run_dump(["proptable", p])
    P = np.fromfile(p, dtype=np.float64).reshape(50, 50, 4)   # [i-1][j-1][r-1]
    np.savez_compressed(os.path.join(GOLDEN, "proptable_toggle_test.npz"), P=P,
                        params=np.array([5000.0, 1600.0, 1.0, 1.0]))
    print("proptable: ok")


if __name__ == "__main__":
    main()
