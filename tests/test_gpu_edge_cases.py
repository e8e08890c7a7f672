"""Edge cases and error behaviour of the C ABI on the device (status codes
instead of exits, wide / irregular rows, other orthogonalisation windows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from krylovfspssa_amd import KfspContext
    c = KfspContext(0)
    yield c
    c.close()


def _random_generator(n, bw, rng, fill=0.7):
    """a random, unstructured 'reaction network' in the reference layout: every slot of
    every state links to a random state, is outside the FSP (0) or illegal (-1)"""
    adj = rng.integers(1, n + 1, size=(n, bw)).astype(np.int32)
    u = rng.random((n, bw))
    adj[u > fill] = 0
    adj[u > 0.5 * (1 + fill)] = -1
    off = rng.random((n, bw)) * 10.0
    return adj, off, off.sum(axis=1)


def test_bad_arguments_return_status_codes(ctx):
    from krylovfspssa_amd import KfspError
    rng = np.random.default_rng(0)
    adj, off, diag = _random_generator(100, 3, rng)
    bad = adj.copy()
    bad[5, 1] = 101                                  # link beyond n
    with pytest.raises(KfspError, match="-5"):
        ctx.set_matrix_ell(bad, off, diag)
    ctx.set_matrix_ell(adj, off, diag)
    with pytest.raises(KfspError, match="-2"):
        ctx.set_vector(np.ones(99))                  # not this rank's block length
    ctx.set_vector(np.ones(100) / 100)
    with pytest.raises(KfspError, match="-2"):
        ctx.arnoldi(101)                             # m > M_MAX
    with pytest.raises(KfspError, match="-2"):
        ctx.get_basis(200)
    with pytest.raises(KfspError, match="-2"):
        ctx.combine(0, 1.0, np.ones(4))
    with pytest.raises(KfspError):
        ctx.dgexpv(0.0, 1e-4, 1e-8, 3)               # T = 0
    with pytest.raises(KfspError):
        ctx.dgexpv(1.0, -1.0, 1e-8, 3)               # FSPTOL <= 0
    # the context is still usable afterwards
    assert ctx.begin_step() == pytest.approx(0.1)


def test_raised_m_max_applies_from_the_next_generator(oracle):
    """Option m_max sizes the basis when a generator is set.  Raising it afterwards must not let kfsp_arnoldi (or the
    scratch column of kfsp_spmv_w) run past the columns that ARE allocated: the larger m is refused (-2) until the next
    generator re-lays the basis, then it works and agrees with the oracle."""
    from krylovfspssa_amd import KfspContext, KfspError
    rng = np.random.default_rng(3)
    adj, off, diag = _random_generator(500, 4, rng)
    A = oracle.EllMatrix(adj, off, diag)
    p0 = rng.random(500)
    p0 /= p0.sum()
    with KfspContext(0) as c:
        c.set_option("m_max", 8)
        c.set_matrix_ell(adj, off, diag)
        c.set_vector(p0)
        c.begin_step()
        c.arnoldi(8)
        c.set_option("m_max", 100)                    # the basis still has 8 + 3 columns
        with pytest.raises(KfspError, match="-2"):
            c.arnoldi(20)
        with pytest.raises(KfspError, match="-2"):
            c.get_basis(20)
        y = c.spmv_w()                               # scratch column = the last ALLOCATED one
        assert np.abs(y - oracle.spmv_ell(A, p0)).max() <= 1e-13 * np.abs(off).max()
        c.arnoldi(8)                                 # ... and what fits still runs
        c.set_matrix_ell(adj, off, diag)              # re-lays the basis for 100 + 3 columns
        c.set_vector(p0)
        c.begin_step()
        H, mb, k1, av = c.arnoldi(20)
        V, Href, mbr, k1r, avr = oracle.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), 20)
        assert mb == mbr and np.abs(H[:22, :21] - Href[:22, :21]).max() <= 1e-9 * np.abs(Href).max()


@pytest.mark.parametrize("n,bw", [(2, 1), (3, 2), (65, 20), (1000, 33), (4097, 64)])
def test_unstructured_generators_any_width(ctx, oracle, n, bw):
    """rows with up to 64 incoming links, nothing banded: SELL path, device build"""
    rng = np.random.default_rng(n * 131 + bw)
    adj, off, diag = _random_generator(n, bw, rng)
    A = oracle.EllMatrix(adj, off, diag)
    ctx.set_matrix_ell(adj, off, diag)
    assert ctx.matrix_info()["nnz"] == A.nnz()
    x = rng.standard_normal(n)
    y = ctx.spmv(x)
    ref = oracle.spmv_ell(A, x)
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    assert np.all(np.abs(y - ref) <= 1e-13 * scale + 1e-300)
    assert np.array_equal(y, ctx.spmv(x))


@pytest.mark.parametrize("qiop", [0, 1, 3, 5])
def test_other_orthogonalisation_windows(ctx, oracle, golden_dir, qiop):
    """QIOP is a constant 2 in the reference (KrylovSolver.f90:137) but the loop
    (:241-246) is general: 0 = against every previous vector"""
    import os
    g = np.load(os.path.join(golden_dir, "solve_ring4.npz"))
    A = oracle.EllMatrix(g["adj"], g["offdiag"], g["diag"])
    w = g["in_vector"]
    ctx.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
    ctx.set_vector(w)
    beta = ctx.begin_step()
    m = 14
    H, mb, k1, av = ctx.arnoldi(m, qiop=qiop)
    V, Href, mbr, k1r, avr = oracle.arnoldi(A, w / beta, m, qiop=qiop)
    assert (mb, k1) == (mbr, k1r)
    assert np.abs(H - Href).max() <= 1e-11 * np.abs(Href).max()
    assert av == pytest.approx(avr, rel=1e-10)


def test_tiny_state_spaces(ctx, oracle):
    """n = 3: Krylov dimension capped at n-1 by the solver (KrylovSolver.f90:211)"""
    adj = np.array([[2, -1], [3, 1], [0, 2]], dtype=np.int32)
    off = np.array([[2.0, 0.0], [1.0, 3.0], [0.5, 1.5]])
    diag = off.sum(axis=1)
    A = oracle.EllMatrix(adj, off, diag)
    p0 = np.array([1.0, 0.0, 0.0])
    ctx.set_matrix_ell(adj, off, diag)
    ctx.set_vector(p0)
    ws = ctx.expv_fixed(2, 0.05, 3)
    wref, wsref = oracle.expv_fixed(A, p0, 2, 0.05, 3)
    assert np.abs(ctx.get_vector() - wref).sum() < 1e-13 and np.abs(ws - wsref).max() < 1e-14


def test_growing_and_shrinking_fsp_reuses_the_context(ctx, oracle, golden_dir):
    """generator re-uploads of changing size (what every FSP change does): results never
    depend on what an earlier, larger FSP left in device memory"""
    import os
    rng = np.random.default_rng(9)
    names = ["assembly_goutsias_k16.npz", "assembly_toggle_k5.npz", "assembly_goutsias_k10.npz",
             "assembly_repressilator_k10.npz", "assembly_goutsias_k16.npz"]
    for name in names:
        g = np.load(os.path.join(golden_dir, name))
        n = int(g["n"])
        A = oracle.EllMatrix(g["adj"], g["offdiag"], g["diag"])
        ctx.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
        w = rng.random(n)
        w /= w.sum()
        ctx.set_vector(w)
        m = min(20, n - 1)
        ws = ctx.expv_fixed(m, 1e-3, 2)
        wref, wsref = oracle.expv_fixed(A, w, m, 1e-3, 2)
        assert np.abs(ctx.get_vector() - wref).sum() < 1e-11
        assert np.abs(ws - wsref).max() < 1e-12


@pytest.mark.parametrize("fixture", ["solve_toggle_input.npz", "solve_ring6.npz", "assembly_goutsias_k10.npz"])
def test_one_launch_arnoldi_equals_multi_launch(oracle, golden_dir, fixture):
    """State spaces of <= 4096 rows run a whole IOP(2) pass in one launch of one
    workgroup (k_arnoldi_small); it must reproduce the multi-launch pass -
    Hessenberg, basis, AVNORM, restart and breakdown behaviour - to rounding."""
    import os
    from krylovfspssa_amd import KfspContext
    g = np.load(os.path.join(golden_dir, fixture))
    n = int(g["n"])
    w = np.random.default_rng(21).random(n)
    A = oracle.EllMatrix(g["adj"], g["offdiag"], g["diag"])
    out = {}
    for small in (1, 0):
        with KfspContext(0) as c:
            c.set_option("small_kernel", small)
            c.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
            c.set_vector(w)
            beta = c.begin_step()
            H, mb, k1, av = c.arnoldi(40)
            H2 = np.zeros((52, 52), order="F")
            H2[:41, :40] = H[:41, :40]
            H2, mb2, k12, av2 = c.arnoldi(50, jold=40, H=H2)              # dimension change restart
            v51 = c.get_basis(51)
            H3 = np.zeros((22, 22), order="F")
            _, mb3, k13, av3 = c.arnoldi(20, jold=50, H=H3)               # shrink below jold: column 51 := A v_50
            out[small] = (H.copy(), av, H2.copy(), av2, av3, v51, (mb, k1, mb2, k12, mb3, k13))
    V, Href, _, _, avr = oracle.arnoldi(A, w / np.sqrt(w @ w), 50)
    for small in (1, 0):
        H, av, H2, av2, av3, v51, flags = out[small]
        assert flags == (40, 2, 50, 2, 20, 2)
        assert np.abs(H2 - Href).max() <= 1e-11 * np.abs(Href).max()
        assert av2 == pytest.approx(avr, rel=1e-10)
        assert np.abs(v51 - V[:, 50]).max() < 1e-10
    assert np.abs(out[1][0] - out[0][0]).max() <= 1e-12 * np.abs(Href).max()
    assert out[1][4] == pytest.approx(out[0][4], rel=1e-12)


@pytest.mark.parametrize("fixture", ["solve_toggle_input.npz", "solve_ring6.npz"])
def test_one_launch_arnoldi_with_and_without_the_generator_in_lds(golden_dir, fixture):
    """When the SELL slots fit beside the source column in the workgroup's LDS the
    one-launch kernel copies them there (values + 16-bit columns); option
    small_lds = 0 keeps them in global memory.  Same operations in the same order:
    identical bits."""
    import os
    from krylovfspssa_amd import KfspContext
    g = np.load(os.path.join(golden_dir, fixture))
    w = np.random.default_rng(5).random(int(g["n"]))
    out = []
    for lds in (1, 0):
        with KfspContext(0) as c:
            c.set_option("small_lds", lds)
            c.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
            c.set_vector(w)
            c.begin_step()
            H, mb, k1, av = c.arnoldi(30)
            out.append((H.copy(), av, c.get_basis(31), (mb, k1)))
    assert out[0][3] == out[1][3]
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])


def test_growing_fsp_uploads_only_the_new_propensity_columns(oracle, golden_dir):
    """kfsp_update_matrix_ell: after an expansion the OFFDIAG / DIAG columns of the states that were
    already listed are taken from the device's copy.  The reference's assembly after 5 sweeps is a
    prefix of the one after 10 (states are only appended, StateSpace.f90:136-246): upload k = 5, then
    k = 10 with the first n5 propensity columns of the HOST arrays poisoned - the product must still
    be the true one; a stale claim (another generator in between) is ignored."""
    import os
    from krylovfspssa_amd import KfspContext
    a = np.load(os.path.join(golden_dir, "assembly_goutsias_k5.npz"))
    b = np.load(os.path.join(golden_dir, "assembly_goutsias_k10.npz"))
    n5, n10 = a["adj"].shape[0], b["adj"].shape[0]
    assert np.array_equal(a["offdiag"], b["offdiag"][:n5]) and np.array_equal(a["diag"], b["diag"][:n5])
    A = oracle.EllMatrix(b["adj"], b["offdiag"], b["diag"])
    x = np.random.default_rng(8).random(n10)
    yref = oracle.spmv_ell(A, x)
    scale = oracle.spmv_ell(oracle.EllMatrix(b["adj"], np.abs(b["offdiag"]), -np.abs(b["diag"])), x)
    off_p, diag_p = b["offdiag"].copy(), b["diag"].copy()
    off_p[:n5] = 1.0e30
    diag_p[:n5] = -7.0
    with KfspContext(0) as c:
        c.set_matrix_ell(a["adj"], a["offdiag"], a["diag"])
        c.update_matrix_ell(b["adj"], off_p, diag_p, n5)
        assert np.all(np.abs(c.spmv(x) - yref) <= 1e-13 * scale)
        # in between another kind of generator: the claim no longer holds and is ignored (so the poison shows)
        c.set_matrix_csr(n10, *A.to_csr())
        c.update_matrix_ell(b["adj"], off_p, diag_p, n5)
        assert not np.all(np.abs(c.spmv(x) - yref) <= 1e-13 * scale)
        c.update_matrix_ell(b["adj"], b["offdiag"], b["diag"], 0)        # everything travels again
        assert np.all(np.abs(c.spmv(x) - yref) <= 1e-13 * scale)


@pytest.mark.parametrize("seed", range(6))
def test_random_banded_generators(ctx, seed):
    """Banded rows with random offsets (odd and even, +-1, some longer than the matrix) and ragged
    sizes: the banded kernel's paired x gathers at both ends of x (first and last rows, offsets that
    leave the vector), against numpy; the same rows forced through SELL-64 give the same bits."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(130, 5000))
    nd = int(rng.integers(1, 13))
    deltas = np.unique(np.concatenate([rng.integers(-n - 5, n + 6, nd), [(-1) ** seed]]))
    deltas = deltas[deltas != 0]
    rows = np.arange(n)
    cols_l, vals_l = [rows], [-(rng.random(n) + 0.5)]
    for d in deltas:
        c = rows + d
        ok = (c >= 0) & (c < n) & (rng.random(n) < 0.9)
        cols_l.append(np.where(ok, c, -1))
        vals_l.append(np.where(ok, rng.random(n), 0.0))
    C, V = np.stack(cols_l, 1), np.stack(vals_l, 1)
    order = np.argsort(np.where(C < 0, 1 << 40, C), axis=1, kind="stable")
    C, V = np.take_along_axis(C, order, 1), np.take_along_axis(V, order, 1)
    valid = C >= 0
    rowptr = np.concatenate(([0], np.cumsum(valid.sum(1)))).astype(np.int64)
    col, val = C[valid].astype(np.int32), V[valid]
    x = rng.standard_normal(n)
    ref = np.array([val[rowptr[r]:rowptr[r + 1]] @ x[col[rowptr[r]:rowptr[r + 1]]] for r in range(n)])
    mag = np.array([np.abs(val[rowptr[r]:rowptr[r + 1]]) @ np.abs(x[col[rowptr[r]:rowptr[r + 1]]]) for r in range(n)])
    out = {}
    for fmt in (0, 1):
        ctx.set_option("format", fmt)
        ctx.set_matrix_csr(n, rowptr, col, val)
        out[fmt] = ctx.spmv(x)
        assert np.all(np.abs(out[fmt] - ref) <= 1e-13 * mag + 1e-300)
    ctx.set_option("format", 0)
    assert np.array_equal(out[0], out[1])
