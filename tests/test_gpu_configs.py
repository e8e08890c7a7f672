"""BASELINE.json configs 4 and 5 at their stated per-GPU sizes through the C ABI, and a
6-species network small enough for the oracle.

  config 4  goutsias_model.input propensities on M, D, RNA in [0,150)^3 x the 6 conserved DNA
            configurations: N = 2.025e7 states, nnz = 1.81e8, on one context and row-partitioned over
            2 and 4 loop-back ranks at the full size
  config 5  synthetic 6-species birth-death network, the per-GPU slab 22^5 x 3 of the 22^6 box:
            N = 1.55e7 states, 12 reactions

At these sizes the oracle's scatter loop is no longer a seconds-scale check, so the generator
product is pinned by size-independent properties (linearity, the mass-balance identity
1^T A x = colsum . x, determinism, banded == SELL, masked == unmasked) and by sampled rows
against numpy; the same model classes are checked against the oracle at small sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _properties(ctx, mdl, rowptr, col, val, seed):
    rng = np.random.default_rng(seed)
    n = mdl.n
    x, z = rng.random(n), rng.random(n)
    ax, az = ctx.spmv(x), ctx.spmv(z)
    mag = np.abs(ax).max()
    assert np.abs(ctx.spmv(2.0 * x - 3.0 * z) - (2.0 * ax - 3.0 * az)).max() <= 1e-9 * mag
    colsum = np.bincount(col, weights=val, minlength=n)
    assert abs(ax.sum() - colsum @ x) <= 1e-7 * np.abs(ax).sum()
    assert np.array_equal(ax, ctx.spmv(x))
    rows = np.unique(np.concatenate([np.arange(512), np.arange(n - 512, n), rng.integers(0, n, 8192)]))
    ref = np.array([val[rowptr[r]:rowptr[r + 1]] @ x[col[rowptr[r]:rowptr[r + 1]]] for r in rows])
    scale = np.array([np.abs(val[rowptr[r]:rowptr[r + 1]]) @ x[col[rowptr[r]:rowptr[r + 1]]] for r in rows])
    assert np.all(np.abs(ax[rows] - ref) <= 1e-13 * scale)
    return x, ax, mag


def test_config4_at_full_size():
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.GoutsiasConserved(150, 150, 150)
    assert mdl.n == 20_250_000
    rowptr, col, val = mdl.csr_rows()
    assert rowptr[-1] == mdl.nnz() and rowptr[-1] > 1.8e8
    with KfspContext(0) as c:
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        info = c.matrix_info()
        assert info["slots"] % mdl.n == 0 or info["slots"] >= 10 * mdl.n      # banded: ten stored diagonals
        x, ax, mag = _properties(c, mdl, rowptr, col, val, 41)
        masked_bytes = c.matrix_bytes()
        c.set_option("dia_mask", 0)                       # same diagonals without the empty-segment masks
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        assert np.array_equal(c.spmv(x), ax)
        assert c.matrix_bytes() > masked_bytes * 1.1      # the masks skip > 10 % of the stored bytes here
        c.set_option("format", 1)                         # SELL-64 on the same rows
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        assert np.abs(c.spmv(x) - ax).max() <= 1e-12 * mag
    # a few fixed steps from the reference's initial state keep mass and sign
    with KfspContext(0) as c:
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        p0 = np.zeros(mdl.n)
        p0[2 + 150 * (6 + 150 * 0)] = 1.0                # (M, D, RNA) = (2, 6, 0), two free DNA copies
        c.set_vector(p0)
        ws = c.expv_fixed(20, 0.05, 2)
        w = c.get_vector()
        assert np.all(w >= 0) and 0.999 < ws[-1] <= 1.0 + 1e-12
        assert ws[-1] == pytest.approx(w.sum(), rel=1e-13)


@pytest.mark.parametrize("ranks", [2, 4])
def test_config4_at_full_size_row_partitioned(ranks):
    """BASELINE config 4 as it is stated: "row-partitioned across 2 and 4 GPUs" - here the 2 / 4 contexts of a loop-back
    group on the one GPU, at the FULL size (2.025e7 states, 1.81e8 nonzeros).  Every rank builds only its block of gather
    rows (numpy) and uploads it; the partitioned product (banded rows with group masks, halo strips: the configuration
    shift is 3.4e6 rows, within one block) against the single-context product, an Arnoldi pass with bit-identical scalars
    on all ranks, and two fixed-(m, tau) steps from the reference's initial state against the single-context steps."""
    from krylovfspssa_amd import KfspContext, host, synth
    mdl = synth.GoutsiasConserved(150, 150, 150)
    rng = np.random.default_rng(44)
    x = rng.random(mdl.n)
    p0 = np.zeros(mdl.n)
    p0[2 + 150 * (6 + 150 * 0)] = 1.0
    m = 8
    rowptr, col, val = mdl.csr_rows()
    with KfspContext(0) as c:
        c.set_option("m_max", 12)
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        c.set_vector(x)
        y1 = c.spmv_w()
        beta1 = c.begin_step()
        H1, mb1, k11, av1 = c.arnoldi(m)
        c.set_vector(p0)
        ws1 = c.expv_fixed(m, 0.05, 2)
        w1 = c.get_vector()
    del rowptr, col, val
    mag = np.abs(y1).max()

    def body(ctx, rank):
        ctx.set_option("m_max", 12)
        r0, nr = ctx.row_block(mdl.n)
        rp, cc, vv = mdl.csr_rows(r0, nr)
        nnz = int(rp[-1])
        ctx.set_matrix_csr(mdl.n, rp, cc, vv)
        del rp, cc, vv
        info = ctx.layout_info()
        ctx.set_vector(x[r0:r0 + nr])
        y = ctx.spmv_w()
        err = float(np.abs(y - y1[r0:r0 + nr]).max()) if nr else 0.0
        beta = ctx.begin_step()
        H, mb, k1, av = ctx.arnoldi(m)
        ctx.set_vector(p0[r0:r0 + nr])
        ws = ctx.expv_fixed(m, 0.05, 2)
        l1 = float(np.abs(ctx.get_vector() - w1[r0:r0 + nr]).sum())
        return err, beta, H.copy(), mb, k1, av, nnz, nr, ws, l1, info["exchange"], info["format"]

    res = host.run_loopback_ranks(ranks, body)
    assert sum(r[7] for r in res) == mdl.n and sum(r[6] for r in res) == mdl.nnz()
    assert max(r[0] for r in res) <= 1e-12 * mag
    for r in res[1:]:
        assert r[1] == res[0][1] and np.array_equal(r[2], res[0][2]) and r[3:6] == res[0][3:6] and np.array_equal(r[8], res[0][8])
    # banded rows on every rank; halo strips while the largest shift (two DNA configurations = 6.75e6 rows) fits one block
    # (2 ranks: 1.01e7 rows), the all-gather of the whole vector beyond that (4 ranks: 5.06e6 rows)
    assert all(r[11] in (1, 2) for r in res) and all(r[10] == (1 if ranks == 2 else 2) for r in res), [(r[10], r[11]) for r in res]
    assert abs(res[0][1] - beta1) <= 1e-13 * beta1 and (res[0][3], res[0][4]) == (mb1, k11)
    assert np.abs(res[0][2] - H1).max() <= 1e-10 * np.abs(H1).max() and abs(res[0][5] - av1) <= 1e-10 * av1
    assert np.abs(res[0][8] - ws1).max() < 1e-12 and sum(r[9] for r in res) < 1e-10


def test_config5_slab_at_full_size():
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.birth_death((22, 22, 22, 22, 22, 3))
    assert mdl.n == 15_460_896 and mdl.R == 12
    rowptr, col, val = mdl.csr_rows()
    assert rowptr[-1] == mdl.nnz()
    with KfspContext(0) as c:
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        x, ax, mag = _properties(c, mdl, rowptr, col, val, 42)
        c.set_option("format", 1)
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        assert np.abs(c.spmv(x) - ax).max() <= 1e-12 * mag
    with KfspContext(0) as c:
        c.set_matrix_csr(mdl.n, rowptr, col, val)
        p0 = synth.poisson_p0(mdl, 4.0)
        c.set_vector(p0)
        ws = c.expv_fixed(30, 0.01, 2)
        w = c.get_vector()
        # the slab is three planes thick: births of the sixth species leave it, so mass only decreases
        assert np.all(w >= 0) and 0.5 < ws[1] < ws[0] < 1.0
        assert ws[-1] == pytest.approx(w.sum(), rel=1e-13)


def _c5_samples(mdl, x, rng, k=8192):
    """rows of A x recomputed in numpy from the model at sampled states (first / last 512 and k random ones)"""
    n = mdl.n
    rows = np.unique(np.concatenate([np.arange(512), np.arange(n - 512, n), rng.integers(0, n, k)]))
    cols, vals = mdl.rows_at(rows)
    ok = cols != np.iinfo(np.int64).max
    xs = x[np.where(ok, cols, 0)]
    return rows, (vals * xs).sum(axis=1), (np.abs(vals) * np.abs(xs)).sum(axis=1)


def test_config5_at_full_size_on_one_context():
    """BASELINE config 5 at its STATED size: the 6-species birth-death network on 22^6 = 113 379 904 states,
    12 reactions, nnz = 1.41e9.  Matrix-free form (no generator arrays anywhere) and the stored form written
    out on the device (12 diagonals, 10.9 GB), both on ONE context: linearity, the mass-balance identity
    1^T A x = -(flux into the sink), determinism, >= 8k sampled rows recomputed in numpy from the model, the
    two forms against each other, and two fixed-(m, tau) steps."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.birth_death((22,) * 6)
    assert mdl.n == 113_379_904 and mdl.R == 12 and mdl.nnz() > 1.4e9
    rng = np.random.default_rng(55)
    x, z = rng.random(mdl.n), rng.random(mdl.n)
    rows, ref, scale = _c5_samples(mdl, x, rng)
    assert len(rows) >= 8192
    flux = mdl.colsum_dot(x)
    with KfspContext(0) as c:
        c.set_option("m_max", 24)                        # 27 basis columns of 0.9 GB instead of 105
        c.set_matrix_box(mdl)
        assert c.matrix_info() == dict(rows=mdl.n, slots=0, nnz=mdl.nnz())
        ax = c.spmv(x)
        mag = np.abs(ax).max()
        assert np.all(np.abs(ax[rows] - ref) <= 1e-13 * scale)
        assert abs(ax.sum() - flux) <= 1e-7 * np.abs(ax).sum()
        assert np.array_equal(ax, c.spmv(x))
        az = c.spmv(z)
        lin = c.spmv(2.0 * x - 3.0 * z)
        assert np.abs(lin - (2.0 * ax - 3.0 * az)).max() <= 1e-9 * mag
        del az, lin
        mf_bytes = c.matrix_bytes()
        # the stored form of the same generator, on the same context
        c.set_matrix_box(mdl, store=True)
        info = c.matrix_info()
        assert info["slots"] >= 12 * mdl.n and info["nnz"] == mdl.nnz()
        assert c.matrix_bytes() > 5 * mf_bytes
        ay = c.spmv(x)
        assert np.all(np.abs(ay[rows] - ref) <= 1e-13 * scale)
        assert np.abs(ay - ax).max() <= 1e-12 * mag          # (the fast matrix-free path sums a row species by species)
        del ay
        # two steps of exp(tau A) from the Poisson product: mass and sign, stored against matrix-free
        p0 = synth.poisson_p0(mdl, 4.0)
        c.set_vector(p0)
        ws = c.expv_fixed(20, 0.01, 2)
        w = c.get_vector()
        assert np.all(w >= 0) and 0.99 < ws[1] <= ws[0] <= 1.0 + 1e-12
        assert ws[-1] == pytest.approx(w.sum(), rel=1e-12)
        c.set_matrix_box(mdl, store=False)
        c.set_vector(p0)
        ws2 = c.expv_fixed(20, 0.01, 2)
        assert np.abs(ws2 - ws).max() < 1e-12 and np.abs(c.get_vector() - w).sum() < 1e-10


def test_config5_at_full_size_row_partitioned_over_8_ranks():
    """The same generator as BASELINE states it: 1.13e8 states row-partitioned over 8 ranks - here the 8 contexts
    of a loop-back group on the one GPU.  Every rank builds ONLY its 22^5 x ~2.75 slab of gather rows
    (synth csr_rows(row0, nrows), numpy) and uploads it; the partitioned product (halo strips) is compared with
    the matrix-free product of the whole box, and one Arnoldi pass must leave bit-identical scalars on all ranks
    that agree with the single-context pass."""
    from krylovfspssa_amd import KfspContext, host, synth
    mdl = synth.birth_death((22,) * 6)
    rng = np.random.default_rng(56)
    x = rng.random(mdl.n)
    m = 6
    with KfspContext(0) as c:
        c.set_option("m_max", 8)
        c.set_matrix_box(mdl)
        c.set_vector(x)
        y1 = c.spmv_w()
        beta1 = c.begin_step()
        H1, mb1, k11, av1 = c.arnoldi(m)
    mag = np.abs(y1).max()

    def body(ctx, rank):
        ctx.set_option("m_max", 8)
        r0, nr = ctx.row_block(mdl.n)
        rowptr, col, val = mdl.csr_rows(r0, nr)
        nnz = int(rowptr[-1])
        ctx.set_matrix_csr(mdl.n, rowptr, col, val)
        del rowptr, col, val
        ctx.set_vector(x[r0:r0 + nr])
        y = ctx.spmv_w()
        err = float(np.abs(y - y1[r0:r0 + nr]).max()) if nr else 0.0
        beta = ctx.begin_step()
        H, mb, k1, av = ctx.arnoldi(m)
        return err, beta, H.copy(), mb, k1, av, nnz, nr

    res = host.run_loopback_ranks(8, body)
    assert sum(r[7] for r in res) == mdl.n and sum(r[6] for r in res) == mdl.nnz()
    assert max(r[0] for r in res) <= 1e-12 * mag
    for r in res[1:]:                                             # every rank holds the same scalars, bit for bit
        assert r[1] == res[0][1] and np.array_equal(r[2], res[0][2]) and r[3:6] == res[0][3:6]
    assert abs(res[0][1] - beta1) <= 1e-13 * beta1
    assert (res[0][3], res[0][4]) == (mb1, k11)
    assert np.abs(res[0][2] - H1).max() <= 1e-10 * np.abs(H1).max()
    assert abs(res[0][5] - av1) <= 1e-10 * av1


@pytest.mark.parametrize("dims", [(5, 5, 5, 5, 5, 5), (7, 6, 5, 4, 3, 3)])
def test_six_species_network_matches_the_oracle(oracle, dims):
    """config 5's model (6 species, 12 reactions, 12 diagonals) at 5^6 = 15 625 and 7 560 states:
    product, Arnoldi pass and fixed-(m, tau) expv against the oracle, banded and SELL forms, through
    both upload routes (gather rows and the reference's column layout)."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.birth_death(dims)
    adj, off, diag = mdl.ell()
    A = oracle.EllMatrix(adj, off, diag)
    rng = np.random.default_rng(6)
    x = rng.random(mdl.n)
    yref = oracle.spmv_ell(A, x)
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    p0 = synth.poisson_p0(mdl, 1.5)
    m, tau, nsteps = 25, 0.02, 3
    wref, wsref = oracle.expv_fixed(A, p0, m, tau, nsteps)
    V, Href, mb, k1, av = oracle.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), m)
    for fmt in (0, 1):
        for route in ("csr", "ell"):
            with KfspContext(0) as c:
                c.set_option("format", fmt)
                c.set_option("small_kernel", 0)
                if route == "csr":
                    c.set_matrix_csr(mdl.n, *mdl.csr_rows())
                else:
                    c.set_matrix_ell(adj, off, diag)
                y = c.spmv(x)
                assert np.all(np.abs(y - yref) <= 1e-13 * np.abs(scale) + 1e-300), (fmt, route)
                c.set_vector(p0)
                c.begin_step()
                H, mb2, k12, av2 = c.arnoldi(m)
                assert (mb2, k12) == (mb, k1)
                # IOP(2) amplifies rounding differences from column to column (tests/test_lockstep.py):
                # the leading columns agree to rounding, the whole pass to 1e-6, the solution to 1e-10
                assert np.abs(H[:11, :10] - Href[:11, :10]).max() <= 1e-11 * np.abs(Href).max()
                assert np.abs(H[:m + 1, :m] - Href[:m + 1, :m]).max() <= 1e-6 * np.abs(Href).max()
                assert abs(av2 - av) <= 1e-6 * av
                c.set_vector(p0)
                ws = c.expv_fixed(m, tau, nsteps)
                assert np.abs(c.get_vector() - wref).sum() < 1e-10
                assert np.abs(ws - wsref).max() < 1e-12
