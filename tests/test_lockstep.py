"""Lock step with the reference: the decisions DGEXPV_FSP took on
BASELINE config 1 (toggle_input, T = 1000) and on examples/toggle (T = 100), recorded from the
UNMODIFIED reference by BLAS observers (oracle/ref_trace.c), steer kfsp_dgexpv_replay; the device
arithmetic and our host state-space code run as in a free solve and are compared with the
reference after EVERY time step: state lists bit for bit, probabilities in l1.

Why lock step: the reference's accept/reject decisions hinge on the tail of exp(tau*H) e1, and the
IOP(2) Hessenberg matrix is chaotic in the rounding of its BLAS (the CPU tests below measure it:
two restatements that differ only in summation order share 16 digits of H in column 1 and 2 in
column 26).  A free-running solve of ANY other implementation therefore forks from the
reference's trajectory within a few steps; what can and must agree is every step's arithmetic."""
import os
import subprocess

import numpy as np
import pytest

from oracle import lockstep as L
from oracle import oracle as O
from tests.conftest import GOLDEN, ROOT

FDIR = os.path.join(ROOT, "krylovfspssa_amd", "fortran")
REPLAY = os.path.join(FDIR, "_build", "kfsp_replay")
MODELS = os.path.join(GOLDEN, "models")
DELTA = 1.2        # KrylovSolver.f90:85


def _err_loc(E, m, beta, avnorm):
    """KrylovSolver.f90:293-304"""
    p1 = abs(E[m, 0]) * beta
    p2 = abs(E[m + 1, 0]) * beta * avnorm
    if p1 > 10.0 * p2:
        return p2
    if p1 > p2:
        return p1 * p2 / (p1 - p2)
    return p1


# ------------------------------------------------------------------ CPU: why trajectories fork

def test_script_fixtures_are_well_formed():
    for name in ("toggle_input", "toggle_example", "repressilator_input_T1", "goutsias_input_T40"):
        g = np.load(os.path.join(GOLDEN, f"lockstep_{name}.npz"))
        s = g["script"]
        assert s.shape[1] == 4 and s[0, 0] == L.BEGIN and s[-1, 0] == L.END
        assert set(np.unique(s[:, 0])) == {L.BEGIN, L.KRYLOV, L.FSP, L.END}
        n_steps = int((s[:, 0] == L.END).sum())
        assert n_steps + 1 == len(g["n_after"]) + (0 if name else 0) or n_steps == len(g["n_after"])
        # every step: BEGIN, >= 1 KRYLOV ending in an accept, >= 1 FSP ending in accept / give-up, END
        kinds = "".join("BKFE"[int(k) - 1] for k in s[:, 0])
        import re
        assert re.fullmatch(r"(BK+F+E)+", kinds)
        acc = s[(s[:, 0] == L.FSP) & (s[:, 1] == 0.0)]
        assert abs(acc[:, 2].sum() - float(g["T"])) < 1e-9 * float(g["T"])      # accepted steps add up to T
        # the final state list of the observed run is the plain run's (fixture solve_<name>.npz)
        plain = np.load(os.path.join(GOLDEN, f"solve_{name}.npz"))
        assert np.array_equal(g["final_state"], plain["state"]) and np.array_equal(g["final_vector"], plain["vector"])


def test_hessenberg_tail_is_rounding_noise_but_the_solution_is_not():
    """One Krylov pass of the reference's toggle_input run (step 14: N = 2570, m = 80, tau = 0.46;
    the first single-pass step whose error estimate the C restatement misses by > 1 %).  The oracle
    (plain C loops, the reference's own operation order except inside BLAS) starts from the same
    vector and generator.  Its H agrees with the reference's to rounding in the first two
    columns and to three digits from column ~22 on (IOP(2) does not keep the basis orthogonal,
    every column amplifies the difference ~4x), so the local error estimate - the LAST entries of
    exp(tau H) e1 - is implementation specific at the per-cent level; the solution update
    beta V exp(tau H) e1 built from the same pass still agrees to 1e-13."""
    g = np.load(os.path.join(GOLDEN, "lockstep_sample_toggle.npz"))
    A = O.EllMatrix(g["adj"], g["offdiag"], g["diag"])
    beta, Href = float(g["beta"]), g["H"]
    m = Href.shape[0] - 2
    V, H, mb, k1, av = O.arnoldi(A, g["w0"] / beta, m)
    assert (mb, k1) == (m, 2)
    sub_o = np.array([H[j + 1, j] for j in range(m)])
    sub_r = np.array([Href[j + 1, j] for j in range(m)])
    rel = np.abs(sub_o - sub_r) / np.abs(sub_r)
    assert rel[:2].max() < 1e-13            # same arithmetic ...
    assert rel[25:].max() > 1e-4            # ... whose rounding differences grow by ~12 orders of magnitude
    growth = rel[25:40].max() / max(rel[:2].max(), 1e-16)
    assert growth > 1e9
    # loss of orthogonality, the mechanism: v_40 is far from orthogonal to v_1..v_37
    G = V[:, :m + 1].T @ V[:, :m + 1]
    assert np.abs(G[40, :38]).max() > 0.1
    # the error estimates differ visibly, for both step sizes the reference tried
    for t in {float(g["t_first"]), float(g["t_second"])}:
        e_o = _err_loc(O.padm(H, t)[0], m, beta, av)
        e_r = _err_loc(O.padm(Href, t)[0], m, beta, float(g["avnorm"]))
        assert abs(e_o - e_r) > 1e-2 * e_r, (t, e_o, e_r)
    # ... while the solution of the pass (coefficients from each side's own H) is the same
    t = float(g["t_second"])
    mx = m + 1
    y_o = O.padm(H, t)[0][:mx, 0]
    w_o = beta * (V[:, :mx] @ y_o)
    w_o[w_o < 0] = 0.0
    assert np.abs(y_o - g["y"]).max() > 1e-10                # different coefficients on different bases
    assert np.abs(w_o - g["w1"]).sum() < 1e-13               # same vector
    assert abs(w_o.sum() - float(g["wsum"])) < 1e-12


def test_library_pade_reproduces_the_reference_on_its_own_hessenberg_matrices():
    """kfsp_padm (host code of the product) on (H, tau) pairs the reference handed to
    DGPADM(norm) during the toggle_input run, against the coefficient vectors exp(tau H) e1 its
    DGEMV then received: the exponential is NOT where trajectories fork (it agrees to ~1e-13 of the
    largest coefficient, and the error estimate formed from it to 1e-9 relative)."""
    from krylovfspssa_amd import build, host
    build.build_lib()
    g = np.load(os.path.join(GOLDEN, "lockstep_sample_toggle.npz"))
    assert int(g["npade"]) >= 6
    for k in range(int(g["npade"])):
        H, t, y = g[f"pH{k}"], float(g[f"pt{k}"]), g[f"py{k}"]
        E, _, _ = host.padm(H, t)
        assert np.abs(E[:len(y), 0] - y).max() <= 1e-12 * np.abs(y).max(), k
        tail = np.abs(y[-2:])
        if tail.min() > 1e-300:
            assert np.abs(E[len(y) - 2:len(y), 0] - y[-2:]).max() <= 1e-8 * tail.max(), k


# ------------------------------------------------------------------ GPU: the lock step itself

@pytest.fixture(scope="module")
def replay():
    if not os.path.exists(REPLAY):
        from krylovfspssa_amd import build
        build.build_lib()
        subprocess.run(["make", "-s", "-C", FDIR, "_build/kfsp_replay"], check=True)
    return REPLAY


def _run_replay(replay, tmp_path, name, case, safe, horizon=None, prefix="lockstep_"):
    g = np.load(os.path.join(GOLDEN, f"{prefix}{name}.npz"))
    script, steps, out = str(tmp_path / "script.bin"), str(tmp_path / "steps.bin"), str(tmp_path / "out.bin")
    L.write_script(script, g["script"])
    r = subprocess.run([replay, case, script, steps, out, repr(horizon) if horizon else "-"] + (["safe"] if safe else []), cwd=MODELS,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    ev = [e for e in L.read_trace(steps) if e["tag"] in "BF"]
    rc, nforks, wdiff, next_, forks = L.read_forks(steps + ".forks")
    # per B event: (n, w, state list in force)
    ours, cur = [], None
    for e in ev:
        if e["tag"] == "B":
            ours.append([e, cur])
        else:
            cur = e
            ours[-1][1] = e
    return g, ours, rc, forks, next_


def _compare(g, ours, upto):
    """-> (first step whose state list differs or None, l1 per step)"""
    fsp_at = set(int(k) for k in g["fsp_at"])
    cur = None
    l1 = []
    for k in range(min(upto, len(ours), int(g["keep"]))):
        if k in fsp_at:
            cur = g[f"state_{k}"]
        b, f = ours[k]
        if f["n"] != cur.shape[0] or not np.array_equal(f["state"], cur):
            return k, np.array(l1)
        l1.append(np.abs(b["w"] - g[f"w_{k}"]).sum())
    return None, np.array(l1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,case", [("toggle_input", "toggle_input"), ("toggle_example", "toggle_example")])
def test_lock_step_with_the_reference(replay, tmp_path, name, case):
    """Recorded step sizes / dimensions / FSP decisions, our arithmetic (safe mode: where the
    recorded acceptance fails OUR error test the basis is enlarged instead of injecting an error
    the reference does not have).  After every one of the first steps - SSA expansions, one-step
    sweeps and drops with compaction included - the state list is the reference's bit for bit and
    the probability vector agrees to 1e-10; later, while both runs stay on the recorded time grid,
    to the solver's own tolerance 2 * DELTA * KRYTOL * t (each side's Krylov error is only
    controlled to DELTA * KRYTOL per unit time, KrylovSolver.f90:314,375)."""
    g, ours, rc, forks, next_ = _run_replay(replay, tmp_path, name, case, safe=True)
    krytol = float(g["krytol"])
    hard = [f for f in forks if f["kind"] in ("UNSAFE_ACCEPT", "BREAKDOWN", "FSP_SIZE", "FSP_TEST")]
    first_hard = min([f["step"] for f in hard], default=10 ** 9)
    bad, l1 = _compare(g, ours, upto=first_hard)
    nstep = len(l1)
    t_at = g["t_at"][:nstep]
    tight = t_at <= 5.0
    print(f"{name}: rc={rc} forks={len(forks)} basis extensions={next_} first hard fork at step {first_hard}; "
          f"compared {nstep} steps (t <= {t_at[-1]:.3g}), state lists equal through all of them={bad is None}; "
          f"max l1 for t<=5: {l1[tight].max():.3e} ({int(tight.sum())} steps), overall {l1.max():.3e}")
    assert bad is None, f"state list differs at step {bad}"
    assert nstep >= 40                                   # well past every expansion of the transient
    assert int(tight.sum()) >= 20
    assert l1[tight].max() < 1e-10
    assert np.all(l1 <= 1e-10 + 2.0 * DELTA * krytol * t_at)


@pytest.mark.gpu
@pytest.mark.parametrize("name,case", [("repressilator_input_T1", "repressilator_input"), ("goutsias_input_T40", "goutsias_input")])
def test_lock_step_on_three_and_six_species(replay, tmp_path, name, case):
    """The same on the 3-species repressilator (T = 1: ten steps, FSP 91 -> 27 816 -> 21 559 states)
    and the 6-species Goutsias model (T = 40: twenty steps, twelve expansions, seven compacting
    drops, FSP -> 12 214 states): after EVERY step of the whole run the state list is the
    reference's bit for bit - each list is the product of an SSA expansion on the shared random
    stream, a one-step sweep and a drop decided on our vector - and the solution agrees to the
    solver's tolerance."""
    g, ours, rc, forks, next_ = _run_replay(replay, tmp_path, name, case, safe=True, horizon=float(g_T(name)))
    krytol = float(g["krytol"])
    hard = [f for f in forks if f["kind"] in ("UNSAFE_ACCEPT", "BREAKDOWN", "FSP_SIZE", "FSP_TEST")]
    first_hard = min([f["step"] for f in hard], default=10 ** 9)
    bad, l1 = _compare(g, ours, upto=first_hard)
    t_at = g["t_at"][:len(l1)]
    print(f"{name}: rc={rc} forks={len(forks)} basis extensions={next_} first hard fork at step {first_hard}; "
          f"compared {len(l1)} of {len(g['n_after'])} steps, state lists equal={bad is None}; max l1 {l1.max():.3e}")
    assert bad is None, f"state list differs at step {bad}"
    assert len(l1) == len(g["n_after"])                  # the whole run
    assert np.all(l1 <= 1e-10 + 2.0 * DELTA * krytol * t_at)
    from oracle.make_golden import read_fsp
    final = read_fsp(str(tmp_path / "out.bin"))
    assert np.array_equal(final["state"], g["final_state"])
    assert np.abs(final["vector"] - g["final_vector"]).sum() <= 1e-10 + 2.0 * DELTA * krytol * float(g["T"])


def _run_replay_digest(replay, tmp_path, name, case, horizon, env=None):
    """kfsp_replay in digest mode (its observer writes checksums, not lists) -> fixture, per-step digests"""
    g = np.load(os.path.join(GOLDEN, f"lockstep_digest_{name}.npz"))
    script, steps, out = str(tmp_path / "script.bin"), str(tmp_path / "steps.bin"), str(tmp_path / "out.bin")
    L.write_script(script, g["script"])
    r = subprocess.run([replay, case, script, steps, out, repr(horizon), "safestop", "digest"], cwd=MODELS,
                       env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    ours = [e for e in L.read_trace_digest(steps) if e["tag"] == "B"]
    rc, nforks, wdiff, next_, forks = L.read_forks(steps + ".forks")
    return g, ours, rc, forks, next_


def _compare_digests(name, g, ours, rc, forks, next_, min_steps):
    krytol = float(g["krytol"])
    # (the run stops at the first step that leaves a list of another size - "safestop" -, so every record
    # it wrote precedes that fork; the other hard forks are numbered by accepted steps <= record index)
    hard = [f for f in forks if f["kind"] in ("UNSAFE_ACCEPT", "BREAKDOWN", "FSP_SIZE", "FSP_TEST")]
    first_hard = min([f["step"] for f in hard if f["kind"] != "FSP_SIZE"], default=10 ** 9)
    nstep = min(len(ours), len(g["n_after"]), first_hard)
    worst = 0.0
    for k in range(nstep):
        assert ours[k]["n"] == int(g["n_after"][k]), f"FSP size differs at step {k}"
        assert np.array_equal(ours[k]["list_hash"], g["list_hash"][k]), f"state list differs at step {k}"
        d = np.abs(ours[k]["proj"] - g["proj"][k]).max()
        worst = max(worst, d)
        assert d <= 1e-10 + 2.0 * DELTA * krytol * float(g["t_at"][k]), (k, d)
    parted = [f for f in hard if f["kind"] == "FSP_SIZE"]
    print(f"{name}: rc={rc} forks={len(forks)} basis extensions={next_}; {nstep} of {len(g['n_after'])} records compared "
          f"(t <= {float(g['t_at'][nstep - 1]):.4g}, N <= {int(g['n_after'][:nstep].max())}), state lists equal in all of them, "
          f"max difference of a weighted sum {worst:.3e}"
          + (f"; the lists part at accepted step {parted[0]['step']}: {int(parted[0]['own'][0])} states here, "
             f"{int(parted[0]['forced'][0])} in the record" if parted else ""))
    assert nstep >= min_steps
    return nstep == len(g["n_after"]) and not hard


@pytest.mark.gpu
def test_lock_step_on_a_longer_repressilator_run_by_digests(replay, tmp_path):
    """repressilator `.input` model to T = 10 (the horizon of the workload in oracle/ref_cases.f90: 41 steps,
    FSP -> 105 229 states), by digests as below"""
    name = "repressilator_input_T10"
    g, ours, rc, forks, next_ = _run_replay_digest(replay, tmp_path, name, "repressilator_input", 10.0,
                                                   env={"KFSP_CASE_CAPACITY": "4194319"})
    _compare_digests(name, g, ours, rc, forks, next_, min_steps=19)


@pytest.mark.gpu
def test_lock_step_on_a_longer_goutsias_run_by_digests(replay, tmp_path):
    """Goutsias `.input` model to T = 100: 37 steps, FSP -> 90 961 states - lists and vectors too big
    to keep, so the fixture holds per step a checksum of the reference's state list (two order-sensitive
    sums mod 2^31 - 1, oracle.lockstep.list_hash) and eight weighted sums of its solution vector
    (weights a function of the state's coordinates; the difference of a sum is at most the l1
    difference of the vectors); our side's observer forms the same numbers.  Same protocol as above:
    recorded step sizes and dimensions, our arithmetic and state-space code."""
    from oracle.make_golden import read_fsp
    name = "goutsias_input_T100"
    g, ours, rc, forks, next_ = _run_replay_digest(replay, tmp_path, name, "goutsias_input", 100.0)
    whole = _compare_digests(name, g, ours, rc, forks, next_, min_steps=30)
    if whole:
        final = read_fsp(str(tmp_path / "out.bin"))
        assert final["n"] == int(g["final_n"]) and np.array_equal(L.list_hash(final["state"]), g["final_hash"])
        assert np.abs(final["vector"] @ L.state_weights(final["state"]) - g["final_proj"]).max() <= 1e-10 + 2.0 * DELTA * float(g["krytol"]) * 100.0


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "lockstep_digest_goutsias_input_T300.npz")),
                    reason="fixture not generated (python -m oracle.make_golden lockstep_digest goutsias_input_T300: ~1 h, 50 GB of scratch)")
def test_lock_step_over_the_full_horizon_of_the_goutsias_example(replay, tmp_path):
    """The horizon of the reference's own example (examples/transcr6d.f90: T = 300, FSPTOL 1e-6, KRYTOL
    1e-8; 457 records, FSP -> 9.6e5 states) in lock step, by digests, for as long as the state lists
    can stay equal: DROP_STATES compares entries of the solution with a threshold (StateSpace.f90:470-495),
    and the two solutions are only equal to the solver's own tolerance (KRYTOL 1e-8 per unit time: the
    weighted sums differ by 1e-10 at t = 99), so sooner or later an entry next to the threshold falls on
    the other side.  Here that is the drop at t = 109.3: 51 692 states kept instead of 51 695.  Up to
    there - 32 records, eleven expansions, ten compacting drops, FSP up to 76 317 states - every list
    carries the reference's checksum."""
    name = "goutsias_input_T300"
    g, ours, rc, forks, next_ = _run_replay_digest(replay, tmp_path, name, "goutsias_input", 300.0,
                                                   env={"KFSP_CASE_CAPACITY": "4194319"})
    whole = _compare_digests(name, g, ours, rc, forks, next_, min_steps=30)
    if whole:
        from oracle.make_golden import read_fsp
        final = read_fsp(str(tmp_path / "out.bin"))
        assert final["n"] == int(g["final_n"]) and np.array_equal(L.list_hash(final["state"]), g["final_hash"])
        assert np.abs(final["vector"] @ L.state_weights(final["state"]) - g["final_proj"]).max() <= 1e-10 + 2.0 * DELTA * float(g["krytol"]) * 300.0


def g_T(name):
    return np.load(os.path.join(GOLDEN, f"lockstep_{name}.npz"))["T"]


@pytest.mark.gpu
def test_strict_lock_step_on_the_transient(replay, tmp_path):
    """Every recorded choice carried out literally (no safety net) over the transient of config 1
    (the first 8 steps: five expansions by SSA + one-step reachability, FSP 21 -> 1588 states, all
    step-size and dimension rejections of the record): state lists bit-exact, l1 < 1e-13 per step,
    and the mass sums WSUM of every solution update agree to 1e-11."""
    g, ours, rc, forks, _ = _run_replay(replay, tmp_path, "toggle_input", "toggle_input", safe=False)
    bad, l1 = _compare(g, ours, upto=9)
    assert bad is None and len(l1) == 9
    assert l1.max() < 1e-13
