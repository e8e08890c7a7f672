"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Reader for the traces of oracle/_ref/ref_trace (record layout: oracle/ref_trace.c)
and the translation of a trace + the reference's stdout log into a DECISION SCRIPT:
the sequence of choices DGEXPV_FSP made (KrylovSolver.f90:206-550) - step sizes,
Krylov dimensions, Krylov accept/reject, FSP accept/reject/give-up - that
kfsp_dgexpv_replay (include/kfsp.h) follows in lock step.

Script records are rows of 4 doubles (kind, a, b, c):
    BEGIN   (1, t_step, m, 0)          a time step starts: T_STEP (:208; 0 = not on record), M (:211)
    KRYLOV  (2, code, t_step, m)       outcome of the error test (:375): code 0 accept,
                                       1 new step size t_step (same basis), 2 new dimension m
                                       (and step size t_step)
    FSP     (3, code, t_step, wsum)    outcome of the mass test (:458) for the reference's WSUM:
                                       0 accept, 1 retry with t_step, 2 give up and expand (:466)
    END     (4, n_after, t_new, 0)     the step is over: FSP size the next step runs on, and the
                                       reference's T_NEW as printed (:647; 0 when it printed none)
"""
import struct

import numpy as np

BEGIN, KRYLOV, FSP, END = 1.0, 2.0, 3.0, 4.0


def read_trace(path):
    ev = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0

    def i32():
        nonlocal pos
        v = struct.unpack_from("<i", data, pos)[0]
        pos += 4
        return v

    def f64():
        nonlocal pos
        v = struct.unpack_from("<d", data, pos)[0]
        pos += 8
        return v

    def arr(dtype, count):
        nonlocal pos
        a = np.frombuffer(data, dtype=dtype, count=count, offset=pos).copy()
        pos += a.nbytes
        return a

    while pos < len(data):
        tag = chr(data[pos])
        pos += 1
        if tag == "B":
            n = i32()
            beta = f64()
            ev.append(dict(tag="B", n=n, beta=beta, w=arr(np.float64, n)))
        elif tag == "F":
            ns, nr, n = i32(), i32(), i32()
            ev.append(dict(tag="F", ns=ns, nr=nr, n=n, state=arr(np.int32, ns * n).reshape(n, ns),
                           adj=arr(np.int32, nr * n).reshape(n, nr), offdiag=arr(np.float64, nr * n).reshape(n, nr),
                           diag=arr(np.float64, n)))
        elif tag == "P":
            m, lda = i32(), i32()
            alpha = f64()
            ev.append(dict(tag="P", m=m, lda=lda, alpha=alpha, H=arr(np.float64, m * m).reshape(m, m).T.copy()))
        elif tag == "G":
            m = i32()
            alpha = f64()
            ev.append(dict(tag="G", m=m, alpha=alpha, a_is_b=i32()))
        elif tag == "V":
            ev.append(dict(tag="V", n=i32()))
        elif tag == "N":
            n = i32()
            ev.append(dict(tag="N", n=n, value=f64()))
        elif tag == "C":
            n, mx = i32(), i32()
            beta = f64()
            ev.append(dict(tag="C", n=n, mx=mx, beta=beta, y=arr(np.float64, mx)))
        elif tag == "S":
            n = i32()
            wsum = f64()
            ev.append(dict(tag="S", n=n, wsum=wsum, w=arr(np.float64, n)))
        else:
            raise ValueError(f"bad tag {tag!r} at {pos - 1}")
    return ev


HASH_MULT = np.array([1000003, 998244353, 19260817, 1000000007, 74207281, 433494437, 2971215073, 32452843], dtype=np.int64)
HASH_M = 2147483647


def row_hash(state):
    """one number < 2^31 per state, from its coordinates only"""
    return (state.astype(np.int64) * HASH_MULT[:state.shape[1]]).sum(axis=1) % HASH_M


def list_hash(state):
    """order-sensitive checksum of a state list: two sums mod 2^31 - 1 (a Fortran observer can form
    them too: oracle/replay_main.f90 OBSERVE, digest mode)"""
    h = row_hash(state)
    i = np.arange(1, len(h) + 1, dtype=np.int64)
    a = ((i % 1000003 + 1) * h) % HASH_M
    b = ((i % 999983 + 7) * ((h * 48271 + 11) % HASH_M)) % HASH_M
    return np.array([int(a.sum() % HASH_M), int(b.sum() % HASH_M)], dtype=np.int64)


def state_weights(state, nproj=8):
    """nproj pseudo-random weights in [-1, 1] per state, a function of its coordinates only (two runs
    that list the same states in the same order weigh them alike)"""
    c = 0.0001 * np.arange(1, nproj + 1) + 1e-9 * np.arange(1, nproj + 1) ** 2
    return np.cos(np.outer(row_hash(state).astype(np.float64), c))


def read_trace_digest(path):
    """read_trace for traces too big to hold (a run to 10^6 states writes ~50 GB): the file is
    streamed; a 'B' record keeps, instead of the vector, its eight weighted sums (state_weights of
    the list in force) and the list's checksum; 'F' keeps nothing but its size; 'S' keeps wsum."""
    import os
    ev = []
    size = os.path.getsize(path)
    cur, cur_w8, cur_hash = None, None, None
    pending = None                       # the last 'B' record, its vector not yet digested

    def settle():
        nonlocal pending
        if pending is not None:
            w = pending.pop("w")
            pending["proj"] = w @ cur_w8
            pending["list_hash"] = cur_hash
            pending = None

    with open(path, "rb") as f:
        def i32():
            return struct.unpack("<i", f.read(4))[0]

        def f64():
            return struct.unpack("<d", f.read(8))[0]

        while f.tell() < size:
            tag = f.read(1).decode()
            if tag != "F":
                settle()
            if tag == "B":
                n = i32()
                beta = f64()
                pending = dict(tag="B", n=n, beta=beta, w=np.fromfile(f, dtype=np.float64, count=n))
                ev.append(pending)
            elif tag == "F":
                ns, nr, n = i32(), i32(), i32()
                cur = np.fromfile(f, dtype=np.int32, count=ns * n).reshape(n, ns)
                f.seek(nr * n * 4 + nr * n * 8 + n * 8, 1)
                cur_w8, cur_hash = state_weights(cur), list_hash(cur)
                ev.append(dict(tag="F", ns=ns, nr=nr, n=n))
                settle()
            elif tag == "P":
                m, lda = i32(), i32()
                alpha = f64()
                ev.append(dict(tag="P", m=m, lda=lda, alpha=alpha,
                               H=np.fromfile(f, dtype=np.float64, count=m * m).reshape(m, m).T.copy()))
            elif tag == "G":
                m = i32()
                alpha = f64()
                ev.append(dict(tag="G", m=m, alpha=alpha, a_is_b=i32()))
            elif tag == "V":
                ev.append(dict(tag="V", n=i32()))
            elif tag == "N":
                n = i32()
                ev.append(dict(tag="N", n=n, value=f64()))
            elif tag == "C":
                n, mx = i32(), i32()
                beta = f64()
                ev.append(dict(tag="C", n=n, mx=mx, beta=beta, y=np.fromfile(f, dtype=np.float64, count=mx)))
            elif tag == "S":
                n = i32()
                wsum = f64()
                f.seek(n * 8, 1)
                ev.append(dict(tag="S", n=n, wsum=wsum))
            elif tag == "D":
                n = i32()
                beta = f64()
                h = np.fromfile(f, dtype=np.int64, count=2)
                ev.append(dict(tag="B", n=n, beta=beta, list_hash=h, proj=np.fromfile(f, dtype=np.float64, count=8)))
            else:
                raise ValueError(f"bad tag {tag!r} at {f.tell() - 1}")
        settle()
    return ev


def pade_calls(ev):
    """Fold the DGEMM/DGESV records of each DGPADM(norm) call (dgpadm.f:100-163: H*H, five Horner
    products, the odd part, DGESV, ns squarings) into one record with the step size recovered
    exactly: t = scale * 2**ns, scale = alpha of the 7th product."""
    out = []
    i = 0
    last_norm = None
    while i < len(ev):
        e = ev[i]
        if e["tag"] == "N":
            last_norm = e["value"]      # the last DNRM2 before a DGPADMNORM call is AVNORM (:263)
            i += 1
            continue
        if e["tag"] != "P":
            out.append(e)
            i += 1
            continue
        g = ev[i + 1:i + 7]
        assert all(x["tag"] == "G" and x["m"] == e["m"] for x in g), "DGPADM product sequence"
        assert all(x["alpha"] == 1.0 and not x["a_is_b"] for x in g[:5])
        scale = g[5]["alpha"]
        assert scale * scale == e["alpha"]
        assert ev[i + 7]["tag"] == "V"
        j = i + 8
        ns = 0
        while j < len(ev) and ev[j]["tag"] == "G" and ev[j]["a_is_b"] and ev[j]["alpha"] == 1.0:
            ns += 1
            j += 1
        t = scale * 2.0 ** ns
        # the same ns the routine derived from hnorm (dgpadm.f:86)
        hnorm = abs(t * np.abs(e["H"]).sum(axis=1).max())
        assert ns == max(0, int(np.log(hnorm) / np.log(2.0)) + 2), (ns, hnorm)
        # lda = MH = M + 2 (:215, :274); mx < lda means a happy breakdown (:249-256, MX = MBRKDWN)
        out.append(dict(tag="P", mx=e["m"], m=e["lda"] - 2, t=t, ns=ns, H=e["H"], avnorm=last_norm))
        i = j
    return out


def build_script(ev, log, t_out, fsptol):
    """Decision script (see module docstring) + the reference's FSP after every change and its
    solution vector after every step.  log: oracle.make_golden.parse_log of the same run."""
    ev = pade_calls(ev)
    rows = []
    fsps = []          # (index into steps of the B event it belongs to, F event)
    w_after = []       # solution vector at every B event (start vector first)
    n_after = []
    combos = []        # (mx, beta, y) of every solution update
    wsums = []
    t_now = 0.0
    acc = 0            # accepted steps so far (index into the log's TIMESTEP blocks)
    i = 0
    assert ev[0]["tag"] == "B"
    while i < len(ev):
        e = ev[i]
        assert e["tag"] == "B", e["tag"]
        w_after.append(e["w"] if "w" in e else dict(proj=e["proj"], list_hash=e["list_hash"]))
        n_after.append(e["n"])
        if rows:
            # END of the previous step: size it left behind; T_NEW is filled in below
            rows[-1][1] = float(e["n"])
        i += 1
        if i < len(ev) and ev[i]["tag"] == "F":
            fsps.append((len(w_after) - 1, ev[i]))
            i += 1
        if i >= len(ev):
            break
        # ---- one time step: Krylov phase
        p = ev[i]
        assert p["tag"] == "P"
        m = p["m"]
        # after a happy breakdown the step size is T_OUT - T_NOW (:254), what :208 chose is not on record
        rows.append([BEGIN, p["t"] if p["mx"] == m + 2 else 0.0, float(m), 0.0])
        t_step = p["t"]
        i += 1
        while ev[i]["tag"] == "P":
            q = ev[i]
            if q["m"] == m:
                rows.append([KRYLOV, 1.0, q["t"], float(m)])
            else:
                m = q["m"]
                rows.append([KRYLOV, 2.0, q["t"], float(m)])
            p = q
            t_step = q["t"]
            i += 1
        rows.append([KRYLOV, 0.0, t_step, float(m)])
        # ---- FSP phase: (C S P)* C S
        rejects = 0
        gave_up = False
        while True:
            c, s = ev[i], ev[i + 1]
            assert c["tag"] == "C" and s["tag"] == "S", (c["tag"], s["tag"])
            combos.append((c["mx"], c["beta"], c["y"]))
            wsums.append(s["wsum"])
            i += 2
            # the reference's own test on its own numbers (:458, :615)
            ok = s["wsum"] >= 1.0 - (t_now + t_step) * fsptol / t_out
            if ok:
                rows.append([FSP, 0.0, t_step, s["wsum"]])
                assert ev[i]["tag"] == "B" if i < len(ev) else True
                break
            rejects += 1
            if rejects >= 5:
                rows.append([FSP, 2.0, t_step, s["wsum"]])
                gave_up = True
                assert ev[i]["tag"] == "B"
                break
            q = ev[i]
            assert q["tag"] == "P" and q["mx"] == c["mx"], (q["tag"], q.get("mx"), c["mx"])
            t_step = q["t"]
            rows.append([FSP, 1.0, t_step, s["wsum"]])
            i += 1
        t_new = 0.0
        if not gave_up:
            assert abs(log["step_tau"][acc] - t_step) == 0.0, (acc, log["step_tau"][acc], t_step)
            assert int(log["step_m"][acc]) == m
            t_now = t_now + t_step
            assert t_now == log["step_tnow"][acc], (t_now, log["step_tnow"][acc])
            t_new = float(log["step_next"][acc])
            acc += 1
        rows.append([END, 0.0, t_new, 0.0])
    if rows and rows[-1][0] == END and rows[-1][1] == 0.0:
        rows[-1][1] = float(n_after[-1])
    assert acc == len(log["step_tau"]), (acc, len(log["step_tau"]))
    assert np.array_equal(np.array(wsums), log["wsum"])
    return dict(script=np.array(rows, dtype=np.float64), fsps=fsps, w_after=w_after,
                n_after=np.array(n_after, dtype=np.int64), combos=combos, wsums=np.array(wsums), events=ev)


def write_script(path, script):
    """script.bin for oracle/replay_main.f90: int64 rows; f64 script(4, rows)."""
    s = np.ascontiguousarray(script, dtype=np.float64)
    with open(path, "wb") as f:
        f.write(struct.pack("<q", s.shape[0]))
        f.write(s.tobytes())


FORK_KINDS = {1: "BEGIN_TAU", 2: "BEGIN_M", 3: "KRYLOV_TEST", 4: "KRYLOV_CHOICE", 5: "KRYLOV_VALUE",
              6: "FSP_TEST", 7: "FSP_TAU", 8: "T_NEW", 9: "FSP_SIZE", 10: "UNSAFE_ACCEPT", 11: "BREAKDOWN"}


def read_forks(path):
    """-> (rc, n_forks, max_wsum_diff, n_safe_extensions, [dict(step, kind, own, forced, lhs, rhs)])."""
    with open(path, "rb") as f:
        data = f.read()
    rc, nf, wd, nx, _ = struct.unpack_from("<iidii", data, 0)
    pos = 24
    out = []
    while pos < len(data):
        step, kind = struct.unpack_from("<ii", data, pos)
        v = struct.unpack_from("<6d", data, pos + 8)
        pos += 56
        out.append(dict(step=step, kind=FORK_KINDS.get(kind, kind), own=v[0:2], forced=v[2:4], lhs=v[4], rhs=v[5]))
    return rc, nf, wd, nx, out
