"""Parity of the HIP hot path (through the C ABI, krylovfspssa_amd/host.py) with
the CPU oracle and the reference's golden fixtures.  Needs a real MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from krylovfspssa_amd import KfspContext
    c = KfspContext(0)
    yield c
    c.close()


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _abs_gen(oracle, adj, off, diag, x):
    """|A| |x| row-wise: the scale rounding errors of one product live on"""
    A = oracle.EllMatrix(adj, np.abs(off), -np.abs(diag))
    return oracle.spmv_ell(A, np.abs(x))


SPMV_FIXTURES = ["assembly_toggle_k5.npz", "assembly_toggle_k20.npz", "assembly_repressilator_k10.npz",
                 "assembly_goutsias_k10.npz", "assembly_goutsias_k16.npz", "solve_toggle_input.npz",
                 "solve_toggle_example.npz", "solve_ring6.npz"]


@pytest.mark.parametrize("fixture", SPMV_FIXTURES)
def test_spmv_on_reference_matrices(ctx, oracle, golden_dir, fixture):
    """kfsp_spmv (FMATVEC seam, KrylovSolver.f90:577-607) on matrices assembled by
    the reference itself.  Tolerance: 1e-13 of |A||x| per entry (different
    summation order than the scatter loop, fused multiply-add on the GPU)."""
    g = _golden(golden_dir, fixture)
    adj, off, diag = g["adj"], g["offdiag"], g["diag"]
    A = oracle.EllMatrix(adj, off, diag)
    ctx.set_matrix_ell(adj, off, diag)
    info = ctx.matrix_info()
    assert info["rows"] == A.n and info["nnz"] == A.nnz()
    rng = np.random.default_rng(12345)
    for x in (np.eye(1, A.n, 0).ravel(), np.ones(A.n), rng.random(A.n)):
        y = ctx.spmv(x)
        ref = oracle.spmv_ell(A, x)
        scale = _abs_gen(oracle, adj, off, diag, x)
        assert np.all(np.abs(y - ref) <= 1e-13 * scale + 1e-300)


def test_spmv_synthetic_boxes_ell_and_csr_agree(ctx, oracle):
    from krylovfspssa_amd import synth
    rng = np.random.default_rng(7)
    for mdl in (synth.toggle(130, 77), synth.repressilator(dims=(23, 19, 17)),
                synth.goutsias_box((7, 6, 5, 3, 3, 3)), synth.birth_death((9, 8, 7, 6)),
                synth.GoutsiasConserved(11, 9, 8)):
        adj, off, diag = mdl.ell()
        A = oracle.EllMatrix(adj, off, diag)
        x = rng.random(mdl.n)
        ref = oracle.spmv_ell(A, x)
        scale = _abs_gen(oracle, adj, off, diag, x)
        ctx.set_matrix_ell(adj, off, diag)
        y1 = ctx.spmv(x)
        ctx.set_matrix_csr(mdl.n, *mdl.csr_rows())
        y2 = ctx.spmv(x)
        assert np.all(np.abs(y1 - ref) <= 1e-13 * scale)
        assert np.array_equal(y1, y2)       # same device layout either way
        assert ctx.matrix_info()["nnz"] == mdl.nnz()


def test_spmv_is_deterministic_and_handles_ragged_sizes(ctx, oracle):
    from krylovfspssa_amd import synth
    rng = np.random.default_rng(3)
    for dims in ((1, 2), (5, 1), (63, 1), (64, 1), (65, 3), (257, 5)):
        mdl = synth.toggle(*dims)
        adj, off, diag = mdl.ell()
        ctx.set_matrix_ell(adj, off, diag)
        x = rng.random(mdl.n)
        y = ctx.spmv(x)
        assert np.array_equal(y, ctx.spmv(x))
        ref = oracle.spmv_ell(oracle.EllMatrix(adj, off, diag), x)
        assert np.abs(y - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def _setup(ctx, oracle, adj, off, diag, w):
    ctx.set_matrix_ell(adj, off, diag)
    ctx.set_vector(w)
    return oracle.EllMatrix(adj, off, diag)


def test_begin_step_norms_and_restore(ctx, oracle, golden_dir):
    g = _golden(golden_dir, "solve_toggle_input.npz")
    w = g["vector"]
    _setup(ctx, oracle, g["adj"], g["offdiag"], g["diag"], w)
    beta = ctx.begin_step()
    assert beta == pytest.approx(np.sqrt((w * w).sum()), rel=1e-14)
    assert ctx.nrm2_w() == pytest.approx(beta, rel=1e-14)
    assert ctx.asum_w() == pytest.approx(np.abs(w).sum(), rel=1e-14)
    v1 = ctx.get_basis(1)
    assert np.abs(v1 - w / beta).max() < 1e-15
    ctx.restore_w(beta)                                  # KrylovSolver.f90:467
    assert np.abs(ctx.get_vector() - w).max() <= 1e-16
    assert np.abs(ctx.spmv_w() - oracle.spmv_ell(oracle.EllMatrix(g["adj"], g["offdiag"], g["diag"]), w)).max() < 1e-13


@pytest.mark.parametrize("fixture,m", [("solve_toggle_input.npz", 30), ("solve_ring6.npz", 29),
                                       ("assembly_goutsias_k16.npz", 60), ("assembly_toggle_k5.npz", 10)])
def test_arnoldi_matches_oracle(ctx, oracle, golden_dir, fixture, m):
    """IOP Arnoldi (KrylovSolver.f90:236-266): Hessenberg entries, basis, AVNORM."""
    g = _golden(golden_dir, fixture)
    n = int(g["n"])
    rng = np.random.default_rng(11)
    # a generic start vector: the fixtures' final vectors are (near) stationary,
    # where A v1 ~ 0 and every later column is rounding noise in both codes
    w = g["in_vector"] if fixture == "solve_ring6.npz" else rng.random(n)
    A = _setup(ctx, oracle, g["adj"], g["offdiag"], g["diag"], w)
    beta = ctx.begin_step()
    H, mb, k1, av = ctx.arnoldi(m)
    V, Href, mbr, k1r, avr = oracle.arnoldi(A, w / np.sqrt((w * w).sum()), m)
    assert (mb, k1) == (mbr, k1r) == (m, 2)
    hs = np.abs(Href).max()
    assert np.abs(H - Href).max() <= 1e-11 * hs
    assert av == pytest.approx(avr, rel=1e-10)
    assert H[m + 1, m] == 1.0
    for j in (1, 2, m // 2, m + 1):
        assert np.abs(ctx.get_basis(j) - V[:, j - 1]).max() <= 1e-10
    assert beta == pytest.approx(np.sqrt((w * w).sum()), rel=1e-14)


def test_arnoldi_restart_extends_basis_like_reference(ctx, oracle, golden_dir):
    """Dimension change (KrylovSolver.f90:400-432): a second call with jold = m_old
    recomputes column m_old and continues; shrinking takes the AVNORM product
    from column jold."""
    g = _golden(golden_dir, "solve_ring6.npz")
    w = g["in_vector"]
    A = _setup(ctx, oracle, g["adj"], g["offdiag"], g["diag"], w)
    ctx.begin_step()
    m0, m1 = 12, 20
    H0, *_ = ctx.arnoldi(m0)
    H1 = np.zeros((m1 + 2, m1 + 2), order="F")
    H1[:m0 + 1, :m0] = H0[:m0 + 1, :m0]
    H1, mb, k1, av = ctx.arnoldi(m1, jold=m0, H=H1)
    _, Href, _, _, avr = oracle.arnoldi(A, w / np.sqrt((w * w).sum()), m1)
    assert np.abs(H1 - Href).max() <= 1e-11 * np.abs(Href).max()
    assert av == pytest.approx(avr, rel=1e-10)
    # shrink below jold: loop body skipped, product A v_jold
    H2 = np.zeros((10 + 2, 10 + 2), order="F")
    _, mb, k1, av2 = ctx.arnoldi(10, jold=m1, H=H2)
    vj = ctx.get_basis(m1)
    assert av2 == pytest.approx(np.sqrt((oracle.spmv_ell(A, vj) ** 2).sum()), rel=1e-12)
    assert (mb, k1) == (10, 2) and H2[11, 10] == 1.0


def test_happy_breakdown_is_detected(ctx, oracle):
    """A stationary start vector: A v1 = 0, so column 1 breaks down (:249-256)."""
    adj = np.array([[2, -1], [1, -1], [0, 0]], dtype=np.int32)[:2]
    off = np.array([[1.0, 0.0], [1.0, 0.0]])
    diag = np.array([1.0, 1.0])
    # pad to a 70-state reducible chain so that m < n
    n = 70
    ADJ = np.full((n, 2), -1, dtype=np.int32)
    OFF = np.zeros((n, 2))
    DIAG = np.zeros(n)
    ADJ[:2], OFF[:2], DIAG[:2] = adj, off, diag
    for i in range(2, n):
        ADJ[i, 0] = i if i + 1 > n - 1 else i + 2
        OFF[i, 0] = 0.5
        DIAG[i] = 0.5
    w = np.zeros(n)
    w[:2] = 0.5
    A = _setup(ctx, oracle, ADJ, OFF, DIAG, w)
    ctx.begin_step()
    H, mb, k1, av = ctx.arnoldi(10)
    _, Href, mbr, k1r, _ = oracle.arnoldi(A, w / np.sqrt(0.5), 10)
    assert (mb, k1) == (mbr, k1r) == (1, 0)
    assert abs(H[0, 0] - Href[0, 0]) < 1e-15 and H[1, 0] == 0.0 and H[11, 10] == 1.0


@pytest.mark.parametrize("mdl_name,m,tau,steps", [("toggle", 30, 0.01, 4), ("repressilator", 30, 0.002, 3),
                                                  ("ring6", 20, 0.05, 5)])
def test_expv_fixed_matches_oracle(ctx, oracle, golden_dir, mdl_name, m, tau, steps):
    """Benchmark-mode exp(tau A)^steps p0 (BASELINE config 2 recipe at a size the
    oracle finishes in seconds): l1 error < 1e-10, mass after every step."""
    from krylovfspssa_amd import synth
    if mdl_name == "ring6":
        g = _golden(golden_dir, "solve_ring6.npz")
        adj, off, diag, p0 = g["adj"], g["offdiag"], g["diag"], g["in_vector"]
    else:
        mdl = synth.toggle(160, 140) if mdl_name == "toggle" else synth.repressilator(dims=(40, 36, 30))
        adj, off, diag = mdl.ell()
        p0 = synth.poisson_p0(mdl, 30.0 if mdl_name == "toggle" else 12.0)
    A = _setup(ctx, oracle, adj, off, diag, p0)
    ws = ctx.expv_fixed(m, tau, steps)
    w = ctx.get_vector()
    wref, wsref = oracle.expv_fixed(A, p0, m, tau, steps)
    assert np.abs(w - wref).sum() < 1e-10
    assert np.abs(ws - wsref).max() < 1e-12
    assert np.all(w >= 0.0)


def test_combine_clamps_and_sums(ctx, oracle, golden_dir):
    g = _golden(golden_dir, "solve_ring4.npz")
    w = g["in_vector"]
    A = _setup(ctx, oracle, g["adj"], g["offdiag"], g["diag"], w)
    beta = ctx.begin_step()
    m = 12
    ctx.arnoldi(m)
    V, *_ = oracle.arnoldi(A, w / beta, m)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(m + 1)          # forces negative entries
    wsum = ctx.combine(m + 1, beta, y)
    ref = np.maximum(beta * V[:, :m + 1] @ y, 0.0)
    got = ctx.get_vector()
    assert np.abs(got - ref).max() < 1e-12
    assert (ref == 0).sum() > 0 and np.all(got >= 0)
    assert wsum == pytest.approx(ref.sum(), rel=1e-13)


def test_size_independent_properties_at_benchmark_size(ctx):
    """BASELINE config 2 matrix (toggle box 1000 x 1000, N = 10^6): linearity,
    the mass-balance identity 1^T A x = -leak . x, determinism."""
    from krylovfspssa_amd import synth
    mdl = synth.toggle(1000, 1000)
    rowptr, col, val = mdl.csr_rows()
    assert rowptr[-1] == 4_996_000 == mdl.nnz()
    ctx.set_matrix_csr(mdl.n, rowptr, col, val)
    rng = np.random.default_rng(12345)
    x, z = rng.random(mdl.n), rng.random(mdl.n)
    ax, az = ctx.spmv(x), ctx.spmv(z)
    lin = ctx.spmv(2.0 * x - 3.0 * z)
    assert np.abs(lin - (2.0 * ax - 3.0 * az)).max() <= 1e-9 * np.abs(ax).max()
    # column sums of the generator = -(propensity leaving the box)
    colsum = np.zeros(mdl.n)
    np.add.at(colsum, col, val)
    assert (ax.sum() - colsum @ x) == pytest.approx(0.0, abs=1e-7 * np.abs(ax).sum())
    assert np.array_equal(ax, ctx.spmv(x))
    # expv on it conserves mass up to the leak and stays non-negative
    p0 = synth.poisson_p0(mdl, 30.0)
    ctx.set_vector(p0)
    ws = ctx.expv_fixed(30, 0.01, 2)
    w = ctx.get_vector()
    assert np.all(w >= 0) and 0.999 < ws[-1] <= 1.0 + 1e-12
    assert ws[-1] == pytest.approx(w.sum(), rel=1e-13)


def test_size_independent_properties_at_config3_size(oracle):
    """BASELINE config 3 (repressilator box 171^3, N = 5 000 211, the bench.py
    workload) at full size: banded and SELL-64 forms agree, linearity, the
    mass-balance identity, determinism; a sample of rows against the oracle's
    scatter product."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.repressilator(171)
    assert mdl.n == 5_000_211
    rowptr, col, val = mdl.csr_rows()
    assert rowptr[-1] == mdl.nnz() == 34_826_031
    rng = np.random.default_rng(12345)
    x, z = rng.random(mdl.n), rng.random(mdl.n)
    c = KfspContext(0)
    try:
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        ax, az = c.spmv(x), c.spmv(z)
        mag = np.abs(ax).max()
        assert np.abs(c.spmv(2.0 * x - 3.0 * z) - (2.0 * ax - 3.0 * az)).max() <= 1e-9 * mag
        colsum = np.zeros(mdl.n)
        np.add.at(colsum, col, val)
        assert (ax.sum() - colsum @ x) == pytest.approx(0.0, abs=1e-7 * np.abs(ax).sum())
        assert np.array_equal(ax, c.spmv(x))
        c.set_option("format", 1)                                   # SELL-64 on the same rows
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        assert np.abs(c.spmv(x) - ax).max() <= 1e-12 * mag
    finally:
        c.close()
    rows = np.unique(np.concatenate([np.arange(512), np.arange(mdl.n - 512, mdl.n), rng.integers(0, mdl.n, 4096)]))
    ref = np.array([val[rowptr[r]:rowptr[r + 1]] @ x[col[rowptr[r]:rowptr[r + 1]]] for r in rows])
    scale = np.array([np.abs(val[rowptr[r]:rowptr[r + 1]]) @ x[col[rowptr[r]:rowptr[r + 1]]] for r in rows])
    assert np.all(np.abs(ax[rows] - ref) <= 1e-13 * scale)


def test_size_independent_properties_on_the_config4_state_set(ctx):
    """BASELINE config 4's generator (Goutsias, conserved DNA: M, D, RNA boxes x 6
    DNA configurations; here 60^3 x 6 = 1.3e6 states, the full 150^3 x 6 is
    bench.py --workload c4): banded upload (gather rows) and SELL upload (reference
    layout) agree, linearity, the mass-balance identity, determinism."""
    from krylovfspssa_amd import synth
    mdl = synth.GoutsiasConserved(60, 60, 60)
    rowptr, col, val = mdl.csr_rows()
    assert rowptr[-1] == mdl.nnz()
    rng = np.random.default_rng(5)
    x, z = rng.random(mdl.n), rng.random(mdl.n)
    ctx.set_matrix_ell(*mdl.ell())
    y_sell = ctx.spmv(x)
    ctx.set_matrix_csr(mdl.n, rowptr, col, val)
    ax, az = ctx.spmv(x), ctx.spmv(z)
    mag = np.abs(ax).max()
    assert np.abs(ax - y_sell).max() <= 1e-12 * mag
    assert np.abs(ctx.spmv(2.0 * x - 3.0 * z) - (2.0 * ax - 3.0 * az)).max() <= 1e-9 * mag
    colsum = np.zeros(mdl.n)
    np.add.at(colsum, col, val)
    assert (ax.sum() - colsum @ x) == pytest.approx(0.0, abs=1e-7 * np.abs(ax).sum())
    assert np.array_equal(ax, ctx.spmv(x))
    # a few fixed steps from a point mass keep the mass (up to the leak) and the sign
    p0 = np.zeros(mdl.n)
    p0[2 + 60 * (6 + 60 * 0)] = 1.0                     # (M, D, RNA) = (2, 6, 0), two free DNA copies
    ctx.set_vector(p0)
    ws = ctx.expv_fixed(20, 0.05, 2)
    w = ctx.get_vector()
    assert np.all(w >= 0) and 0.999 < ws[-1] <= 1.0 + 1e-12
    assert ws[-1] == pytest.approx(w.sum(), rel=1e-13)


# ---------------------------------------------------------------- adaptive solver

def _would_drop(ctx, dsum):
    """DROP_STATES (StateSpace.f90:431-548) up to its compaction decision, in numpy
    on vectors fetched through the C ABI: FIND_DROPTOL :398-427, marking :475-495,
    the 10 % rule :497.  Returns True when the reference would compact the FSP."""
    w = ctx.get_vector()
    droptol = 1e-8
    while True:
        s = w[(w < droptol) & (w > 0)].sum()
        if s < dsum:
            break
        droptol /= 10.0
    cnt = int((w < droptol).sum()) - int((ctx.spmv_w() > 1e-8).sum())
    return cnt / len(w) > 0.1


@pytest.mark.parametrize("name", ["ring6", "ring6_T40", "ring4"])
def test_adaptive_solver_follows_reference_trajectory(ctx, golden_dir, name):
    """kfsp_dgexpv against CME_SOLVE of the unmodified reference on closed systems:
    identical step sizes and Krylov dimensions, WSUM sequence and final vector
    within l1 < 1e-10 (BASELINE north_star tolerance)."""
    from krylovfspssa_amd import host
    g = _golden(golden_dir, f"solve_{name}.npz")
    ctx.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
    ctx.set_vector(g["in_vector"])
    dropped = []

    def drop(dsum):
        dropped.append(_would_drop(ctx, dsum))
        return None

    rc, st, log = ctx.dgexpv(float(g["T"]), float(g["fsptol"]), float(g["krytol"]), int(g["nr"]), drop=drop)
    assert rc == 0 and not any(dropped)
    steps = [v for ev, v in log if ev == host.EV_STEP]
    wsums = np.array([v[0] for ev, v in log if ev == host.EV_WSUM])
    assert np.array_equal([s[2] for s in steps], g["step_tau"])
    assert np.array_equal([int(s[5]) for s in steps], g["step_m"])
    assert np.array_equal([s[4] for s in steps], g["step_tnow"])
    assert len(wsums) == len(g["wsum"]) and np.abs(wsums - g["wsum"]).max() < 1e-10
    assert sum(ev == host.EV_REJECT_STEP for ev, _ in log) == int(g["n_reject"])
    assert sum(ev == host.EV_DIM_CHANGE for ev, _ in log) == int(g["n_dimchange"])
    w = ctx.get_vector()
    assert np.abs(w - g["vector"]).sum() < 1e-10
    assert st.nstep == len(g["step_tau"]) and st.t_now == float(g["T"])


def test_adaptive_solver_reports_needed_expansion(ctx, golden_dir):
    """An open system from a point mass leaks probability at once: without an
    expand callback the solver stops with code 10 after the FSP test fails
    (KrylovSolver.f90:458-470, 518-534) and leaves w = beta*v1 restored."""
    g = _golden(golden_dir, "assembly_toggle_k5.npz")
    ctx.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
    p0 = np.zeros(int(g["n"]))
    p0[0] = 1.0
    ctx.set_vector(p0)
    rc, st, log = ctx.dgexpv(1000.0, 1e-4, 1e-10, int(g["nr"]))
    assert rc == 10
    assert any(ev == 6 for ev, _ in log)


# ------------------------------------------------------- formats and fused variants

@pytest.mark.parametrize("opts", [dict(format=0, fused_ortho=1), dict(format=1, fused_ortho=1),
                                  dict(format=0, fused_ortho=0), dict(format=1, fused_ortho=0)])
def test_generator_formats_and_ortho_variants_agree_with_oracle(oracle, opts):
    """Banded (DIA) vs SELL-64 generator kernels and the one-pass vs two-pass
    IOP(2) orthogonalisation: every combination within the same tolerances."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.repressilator(dims=(37, 29, 23))
    adj, off, diag = mdl.ell()
    A = oracle.EllMatrix(adj, off, diag)
    p0 = synth.poisson_p0(mdl, 9.0)
    with KfspContext(0) as c:
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_matrix_ell(adj, off, diag)
        info = c.matrix_info()
        nact = -(-mdl.n // 128) * 128
        assert info["slots"] == (6 * nact if opts["format"] == 0 else info["slots"])
        x = np.random.default_rng(1).random(mdl.n)
        y = c.spmv(x)
        ref = oracle.spmv_ell(A, x)
        assert np.abs(y - ref).max() <= 1e-13 * np.abs(oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), x)).max()
        c.set_vector(p0)
        beta = c.begin_step()
        H, mb, k1, av = c.arnoldi(25)
        V, Href, _, _, avr = oracle.arnoldi(A, p0 / beta, 25)
        assert np.abs(H - Href).max() <= 1e-11 * np.abs(Href).max()
        assert av == pytest.approx(avr, rel=1e-10)
        assert np.abs(c.get_basis(26) - V[:, 25]).max() < 1e-10
        # dimension-change restart through the fused path
        H2 = np.zeros((32, 32), order="F")
        H2[:26, :25] = H[:26, :25]
        H2, *_ = c.arnoldi(30, jold=25, H=H2)
        _, Href2, *_ = oracle.arnoldi(A, p0 / beta, 30)
        assert np.abs(H2 - Href2).max() <= 1e-11 * np.abs(Href2).max()
        c.set_vector(p0)
        ws = c.expv_fixed(30, 0.004, 3)
        w = c.get_vector()
    wref, wsref = oracle.expv_fixed(A, p0, 30, 0.004, 3)
    assert np.abs(w - wref).sum() < 1e-10 and np.abs(ws - wsref).max() < 1e-12


# ------------------------------------------------ device-side generator build

@pytest.mark.parametrize("fixture", ["assembly_goutsias_k16.npz", "solve_toggle_input.npz", "assembly_toggle_k5.npz",
                                     "solve_ring6.npz"])
def test_device_transpose_equals_host_transpose(oracle, golden_dir, fixture):
    """kfsp_set_matrix_ell builds the gather form on the device (histogram, scan,
    ticketed scatter, per-row sort); the result must be the layout the host
    counting sort produces: identical products, bit for bit, run after run."""
    from krylovfspssa_amd import KfspContext
    g = _golden(golden_dir, fixture)
    adj, off, diag = g["adj"], g["offdiag"], g["diag"]
    x = np.random.default_rng(2).random(int(g["n"]))
    ys, infos = [], []
    for host_build in (0, 1, 0):
        with KfspContext(0) as c:
            c.set_option("host_build", host_build)
            c.set_matrix_ell(adj, off, diag)
            ys.append(c.spmv(x))
            infos.append(c.matrix_info())
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    assert infos[0]["nnz"] == infos[1]["nnz"] == oracle.EllMatrix(adj, off, diag).nnz()


def test_device_build_detects_banded_generators(oracle):
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.birth_death((13, 11, 10, 9))
    adj, off, diag = mdl.ell()
    x = np.random.default_rng(4).random(mdl.n)
    ref = oracle.spmv_ell(oracle.EllMatrix(adj, off, diag), x)
    out = {}
    for host_build in (0, 1):
        with KfspContext(0) as c:
            c.set_option("host_build", host_build)
            c.set_matrix_ell(adj, off, diag)
            out[host_build] = (c.spmv(x), c.matrix_info())
    nact = -(-mdl.n // 128) * 128
    assert out[0][1]["slots"] == out[1][1]["slots"] == 8 * nact           # 8 diagonals, no index array
    assert out[0][1]["nnz"] == out[1][1]["nnz"] == mdl.nnz()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.abs(out[0][0] - ref).max() <= 1e-13 * np.abs(ref).max() + 1e-300


# ------------------------------------------------------------- collective path

def test_one_rank_communicator_runs_the_collective_path(oracle):
    """kfsp_comm_init with a unique id and nranks = 1 creates a real RCCL
    communicator: every product is preceded by ncclAllGather into the padded
    global vector, every scalar goes through finish + ncclAllReduce.  Results
    must be bit-identical to the collective-free path."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.repressilator(dims=(31, 23, 19))
    rp, cc, vv = mdl.csr_rows()
    p0 = synth.poisson_p0(mdl, 8.0)
    out = []
    for use_comm in (False, True):
        with KfspContext(0) as c:
            c.set_option("small_kernel", 0)
            if use_comm:
                c.comm_init(1, 0, KfspContext.unique_id())
            assert c.row_block(mdl.n) == (0, mdl.n)
            c.set_matrix_csr(mdl.n, rp, cc, vv)
            c.set_vector(p0)
            beta = c.begin_step()
            H, mb, k1, av = c.arnoldi(20)
            c.set_vector(p0)
            ws = c.expv_fixed(20, 0.01, 3)
            ms = c.spmv_bench(3)
            out.append((beta, H.copy(), av, ws.copy(), c.get_vector()))
    assert out[0][0] == out[1][0] and out[0][2] == out[1][2]
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][3], out[1][3]) and np.array_equal(out[0][4], out[1][4])
    # (13 547 rows: small enough for the one-launch pass, which is switched off above so that
    # both runs use the same multi-launch kernels and differ only in the collectives)
    adj, off, diag = mdl.ell()
    wref, _ = oracle.expv_fixed(oracle.EllMatrix(adj, off, diag), p0, 20, 0.01, 3)
    assert np.abs(out[1][4] - wref).sum() < 1e-10


def test_one_rank_communicator_halo_layout(oracle):
    """Banded generator + communicator: basis columns get halo margins, every
    product packs / all-gathers the boundary strips.  With one rank there is no
    neighbour, so the result must equal the plain path bit for bit; the
    all-gather fallback (option halo=0) likewise."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.toggle(300, 211)          # 63 300 rows: interior + two boundary launches when overlapped
    rp, cc, vv = mdl.csr_rows()
    p0 = synth.poisson_p0(mdl, 25.0)
    res = []
    for mode in ("plain", "halo+overlap", "halo", "allgather"):
        with KfspContext(0) as c:
            if mode != "plain":
                c.comm_init(1, 0, KfspContext.unique_id())
            if mode == "halo+overlap":
                c.set_option("overlap", 2)      # force the split even at this small size
            if mode == "halo":
                c.set_option("overlap", 0)      # exchange, then one launch
            if mode == "allgather":
                c.set_option("halo", 0)
            c.set_matrix_csr(mdl.n, rp, cc, vv)
            c.set_vector(p0)
            y = c.spmv_w()
            ws = c.expv_fixed(25, 0.01, 2)
            res.append((y, ws.copy(), c.get_vector()))
    for mode, r in zip(("halo+overlap", "halo", "allgather"), res[1:]):
        assert np.array_equal(r[0], res[0][0])                     # the product itself: same rows, same order
        if mode == "halo+overlap":
            # three launches group the dot-product partials differently: rounding-level only
            assert np.abs(r[1] - res[0][1]).max() < 1e-14 and np.abs(r[2] - res[0][2]).sum() < 1e-14
        else:
            assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])


def test_empty_diagonal_segments_are_skipped_without_changing_results(oracle):
    """Banded generators whose diagonals are empty over whole 128-row groups (the
    config-4 state set: four reactions exist for three of the six DNA
    configurations only) take the masked kernel variant; option dia_mask = 0 keeps
    the plain one.  Same bits either way, and the oracle agrees."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.GoutsiasConserved(20, 16, 12)
    rowptr, col, val = mdl.csr_rows()
    adj, off, diag = mdl.ell()
    rng = np.random.default_rng(9)
    x = rng.random(mdl.n)
    ref = oracle.spmv_ell(oracle.EllMatrix(adj, off, diag), x)
    scale = _abs_gen(oracle, adj, off, diag, x)
    out = []
    for mask in (1, 0):
        c = KfspContext(0)
        try:
            c.set_option("dia_mask", mask)
            c.set_matrix_csr(mdl.n, rowptr, col, val)
            assert c.matrix_info()["slots"] == 10 * ((mdl.n + 127) // 128 * 128)     # ten stored diagonals
            y = c.spmv(x)
            c.set_vector(x)
            c.begin_step()
            H = c.arnoldi(12)[0]
            out.append((y, H))
        finally:
            c.close()
    assert np.all(np.abs(out[0][0] - ref) <= 1e-13 * scale)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
