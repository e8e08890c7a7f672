"""Matrix-free generator of a lexicographic box (kfsp_set_matrix_box): the
kernel stores no generator entries and rebuilds FMATVEC's rows (KrylovSolver.f90:577-607) from the row
index and one-species factor tables of the propensities (ModelModule.f90:163-199 evaluated on the host
at every population count)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _models():
    from krylovfspssa_amd import synth
    return {
        "toggle": synth.toggle(97, 61),                          # 2 species, Hill factors
        "repressilator": synth.repressilator(dims=(31, 23, 19)),  # 3 species
        "birth_death6": synth.birth_death((5, 6, 4, 5, 3, 4)),    # 6 species, 12 reactions (config 5's model)
        "goutsias": synth.goutsias_box((9, 8, 7, 3, 3, 3)),       # two-factor propensities, a reaction that moves 3 species
        "line": synth.birth_death((1000,)),                       # one species
    }


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("name", list(_models()))
def test_matrix_free_product_equals_the_stored_generator(oracle, name, generic):
    """generic = 1 forces the run-time interpreted kernel also where the single-factor fast path
    applies (all models here except goutsias, whose propensities have two factors)."""
    from krylovfspssa_amd import KfspContext
    mdl = _models()[name]
    adj, off, diag = mdl.ell()
    A = oracle.EllMatrix(adj, off, diag)
    rng = np.random.default_rng(2)
    x = rng.random(mdl.n)
    yref = oracle.spmv_ell(A, x)
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    with KfspContext(0) as c, KfspContext(0) as b:
        c.set_option("box_generic", generic)
        c.set_matrix_box(mdl)
        b.set_matrix_csr(mdl.n, *mdl.csr_rows())                  # the stored (banded) form of the same rows
        info = c.matrix_info()
        assert info["slots"] == 0 and info["nnz"] == mdl.nnz()
        if mdl.n > 10000:
            assert c.matrix_bytes() < 0.6 * b.matrix_bytes()
        y = c.spmv(x)
        assert np.all(np.abs(y - yref) <= 1e-13 * np.abs(scale) + 1e-300)
        yb = b.spmv(x)
        if generic and max(len(d) for d in mdl.deps) == 1:
            # one factor per propensity: the table entries ARE the stored entries, and the interpreted
            # kernel sums a row in the banded kernel's order -> same bits (the fast path groups the
            # entries by species: same values, other order)
            assert np.array_equal(y, yb)
        assert np.all(np.abs(y - yb) <= 1e-13 * np.abs(scale) + 1e-300)
        # the whole path on it: Arnoldi pass and fixed-(m, tau) steps against the oracle
        p0 = rng.random(mdl.n)
        p0 /= p0.sum()
        m, tau, nsteps = 16, 0.003, 2
        for ctx in (c, b):
            ctx.set_option("small_kernel", 0)
            ctx.set_vector(p0)
        wref, wsref = oracle.expv_fixed(A, p0, m, tau, nsteps)
        ws = c.expv_fixed(m, tau, nsteps)
        assert np.abs(c.get_vector() - wref).sum() < 1e-10 and np.abs(ws - wsref).max() < 1e-12
        V, Href, mb, k1, av = oracle.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), m)
        c.set_vector(p0)
        c.begin_step()
        H, mb2, k12, av2 = c.arnoldi(m)
        assert (mb2, k12) == (mb, k1)
        assert np.abs(H[:9, :8] - Href[:9, :8]).max() <= 1e-11 * np.abs(Href).max()


@pytest.mark.parametrize("P", [2, 3])
def test_matrix_free_rows_are_partitioned_like_banded_ones(oracle, P):
    from krylovfspssa_amd import host, synth
    mdl = synth.repressilator(dims=(31, 23, 19))
    adj, off, diag = mdl.ell()
    A = oracle.EllMatrix(adj, off, diag)
    x = np.random.default_rng(4).random(mdl.n)

    def body(ctx, rank):
        ctx.set_matrix_box(mdl)
        r0, nr = ctx.row_block(mdl.n)
        ctx.set_vector(x[r0:r0 + nr])
        y = ctx.spmv_w()
        ctx.begin_step()
        H, mb, k1, av = ctx.arnoldi(10)
        return y, H.copy()

    res = host.run_loopback_ranks(P, body)
    y = np.concatenate([r[0] for r in res])
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    assert np.all(np.abs(y - oracle.spmv_ell(A, x)) <= 1e-13 * np.abs(scale) + 1e-300)
    V, Href, mb, k1, av = oracle.arnoldi(A, x / np.sqrt((x * x).sum()), 10)
    assert np.abs(res[0][1][:11, :10] - Href[:11, :10]).max() <= 1e-11 * np.abs(Href).max()


@pytest.mark.parametrize("name", list(_models()))
def test_box_written_out_on_the_device_equals_the_uploaded_rows(oracle, name):
    """option box_store: kfsp_set_matrix_box writes the generator out as stored diagonals on the device.  With
    one factor per propensity the entries are the table entries = the very numbers kfsp_set_matrix_csr gets for
    the same box, and both are banded generators summed in the same order: products bit-identical.  Goutsias
    (two-factor propensities: the product of the factor tables instead of the model's own evaluation): to
    1e-13 |A||x|.  Also under a 2-rank partition (row0 != 0, a short last block)."""
    from krylovfspssa_amd import KfspContext, host
    mdl = _models()[name]
    rng = np.random.default_rng(8)
    x = rng.random(mdl.n)
    adj, off, diag = mdl.ell()
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    one_factor = max(len(d) for d in mdl.deps) == 1
    with KfspContext(0) as c, KfspContext(0) as b:
        c.set_matrix_box(mdl, store=True)
        b.set_matrix_csr(mdl.n, *mdl.csr_rows())
        ic, ib = c.matrix_info(), b.matrix_info()
        assert ic["slots"] > 0 and ic["nnz"] == mdl.nnz() == ib["nnz"]
        y, yb = c.spmv(x), b.spmv(x)
        if one_factor:
            assert np.array_equal(y, yb)
        assert np.all(np.abs(y - yb) <= 1e-13 * np.abs(scale) + 1e-300)
        p0 = rng.random(mdl.n)
        p0 /= p0.sum()
        for ctx in (c, b):
            ctx.set_option("small_kernel", 0)
            ctx.set_vector(p0)
        wc, wb = c.expv_fixed(12, 0.004, 2), b.expv_fixed(12, 0.004, 2)
        if one_factor:
            assert np.array_equal(wc, wb) and np.array_equal(c.get_vector(), b.get_vector())
        assert np.abs(c.get_vector() - b.get_vector()).sum() < 1e-12

    def body(ctx, rank):
        ctx.set_matrix_box(mdl, store=True)
        r0, nr = ctx.row_block(mdl.n)
        ctx.set_vector(x[r0:r0 + nr])
        return ctx.spmv_w()

    y2 = np.concatenate(host.run_loopback_ranks(2, body))
    assert np.array_equal(y2, y)


@pytest.mark.parametrize("name", ["toggle", "repressilator", "birth_death6", "line"])
def test_lds_window_gives_the_bits_of_the_gather_kernel(name):
    """format 6 (option box_lds = 1: the near part of x staged in LDS per workgroup; built in round 3, measured SLOWER than
    format 4 and left off by default) against format 4 (every entry its own global gather): same table
    look-ups, same products, same order of additions per row - bit-identical products; also with a reach smaller than
    the box's strides (only +-1 from LDS) and on boxes whose last trips are partial."""
    from krylovfspssa_amd import KfspContext
    mdl = _models()[name]
    rng = np.random.default_rng(12)
    x = rng.random(mdl.n)
    p0 = rng.random(mdl.n)
    p0 /= p0.sum()
    out = {}
    for key, opts in {"gather": {}, "lds": {"box_lds": 1}, "lds, reach 2": {"box_lds": 1, "box_reach": 2}}.items():
        with KfspContext(0) as c:
            c.set_option("small_kernel", 0)
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_matrix_box(mdl)
            fmt = c.layout_info()["format"]
            assert fmt == (4 if key == "gather" else 6), (key, fmt)
            y = c.spmv(x)
            c.set_vector(p0)
            c.begin_step()
            H, mb, k1, av = c.arnoldi(9)
            c.set_vector(p0)
            ws = c.expv_fixed(9, 0.004, 2)
            out[key] = (y, H.copy(), av, ws, c.get_vector())
    for key in ("lds", "lds, reach 2"):
        g, l = out["gather"], out[key]
        assert np.array_equal(g[0], l[0]), key                    # the product itself: same bits
        # (the two kernels deal the trips to the workgroups differently, so block partial sums are added in another order)
        assert np.abs(g[1][:6, :5] - l[1][:6, :5]).max() <= 1e-12 * np.abs(g[1]).max() and abs(g[2] - l[2]) <= 1e-9 * abs(g[2])
        assert np.abs(g[3] - l[3]).max() < 1e-13 and np.abs(g[4] - l[4]).sum() < 1e-12


def test_bad_boxes_are_rejected():
    from krylovfspssa_amd import KfspContext, KfspError, synth
    with KfspContext(0) as c:
        mdl = synth.birth_death((3,) * 9)                        # nine species
        with pytest.raises(KfspError):
            c.set_matrix_box(mdl)
        big = synth.toggle(4000, 3000)                           # 7000 table entries
        with pytest.raises(KfspError):
            c.set_matrix_box(big)


def _random_box(rng, k):
    """a random reaction network on a random box: 1-6 species, dimensions 1..12 (odd and even, a few
    ones), 1-4 reactions per factor species, stoichiometry in {-2..2} on up to three species, one
    propensity factor each (-> the fast path) or, every third model, two factors (-> interpreted)"""
    from krylovfspssa_amd import synth
    d = int(rng.integers(1, 7))
    dims = [int(rng.integers(1, 13)) for _ in range(d)]
    if np.prod(dims) < 2:
        dims[0] = 5
    two = k % 3 == 2 and d >= 2
    R, stoich, deps, coef = 0, [], [], []
    for s in range(d):
        for _ in range(int(rng.integers(0 if d > 1 else 1, 5 if not two else 3))):
            col = np.zeros(d, dtype=np.int64)
            for t in rng.choice(d, size=min(d, int(rng.integers(1, 4))), replace=False):
                col[t] = int(rng.integers(-2, 3))
            if not col.any():
                col[s] = 1
            stoich.append(col)
            dep = (s, int((s + 1) % d)) if two else (s,)
            deps.append(dep)
            coef.append((float(rng.uniform(0.1, 3.0)), int(rng.integers(0, 3))))
            R += 1
    if R == 0:
        stoich, deps, coef, R = [np.eye(d, dtype=np.int64)[0]], [(0,)], [(1.0, 1)], 1
    if R > 16:
        stoich, deps, coef, R = stoich[:16], deps[:16], coef[:16], 16

    def prop(r, X):
        c, kind = coef[r]
        a = c * np.ones_like(X[deps[r][0]])
        for s in deps[r]:
            x = X[s]
            a = a * (x if kind == 0 else (1.0 + x * x if kind == 1 else 1.0 / (1.0 + 0.1 * x)))
        return a

    return synth.BoxModel(f"random{k}", dims, np.array(stoich).T, prop, deps=deps)


def test_matrix_free_on_random_boxes(oracle):
    """60 random networks on random boxes (see _random_box): the matrix-free product - fast path
    where the model allows it, and the interpreted kernel on the same model - against the stored
    generator and the oracle; one context throughout, so every model also runs on whatever the
    previous one left in device memory."""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(2024)
    fast_seen = 0
    with KfspContext(0) as c, KfspContext(0) as g, KfspContext(0) as b:
        g.set_option("box_generic", 1)
        for k in range(60):
            mdl = _random_box(rng, k)
            adj, off, diag = mdl.ell()
            A = oracle.EllMatrix(adj, off, diag)
            x = rng.standard_normal(mdl.n)
            yref = oracle.spmv_ell(A, x)
            scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
            c.set_matrix_box(mdl)
            g.set_matrix_box(mdl)
            b.set_matrix_csr(mdl.n, *mdl.csr_rows())
            ys = [ctx.spmv(x) for ctx in (c, g, b)]
            for y in ys:
                assert np.all(np.abs(y - yref) <= 2e-13 * np.abs(scale) + 1e-300), (k, mdl.dims, mdl.stoich.tolist())
            assert np.array_equal(ys[0], c.spmv(x))                   # deterministic
            fast_seen += max(len(dp) for dp in mdl.deps) == 1 and int(np.abs(mdl.stoich).max()) <= 2
    assert fast_seen >= 30


@pytest.mark.parametrize("P", [2, 3])
def test_matrix_free_random_boxes_with_a_row_partition(oracle, P):
    """the same random networks cut into P row blocks (loop-back ranks): blocks shorter than the
    generator's reach (all-gather instead of halo strips), ragged last blocks, empty ranks"""
    from krylovfspssa_amd import host
    rng = np.random.default_rng(77 + P)
    done = 0
    for k in range(40):
        mdl = _random_box(rng, k)
        if mdl.n < 200 or done == 10:
            continue
        done += 1
        adj, off, diag = mdl.ell()
        A = oracle.EllMatrix(adj, off, diag)
        x = rng.standard_normal(mdl.n)

        def body(ctx, rank):
            ctx.set_matrix_box(mdl)
            r0, nr = ctx.row_block(mdl.n)
            ctx.set_vector(x[r0:r0 + nr])
            return ctx.spmv_w()

        y = np.concatenate(host.run_loopback_ranks(P, body))
        scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
        assert np.all(np.abs(y - oracle.spmv_ell(A, x)) <= 2e-13 * np.abs(scale) + 1e-300), (k, mdl.dims, mdl.stoich.tolist())
    assert done == 10


@pytest.mark.parametrize("name,dims", [("repressilator", (31, 24, 19)), ("repressilator", (16, 64, 7)), ("birth_death6", (5, 6, 4, 5, 4, 7)),
                                       ("birth_death6", (8, 4, 4, 4, 4, 2)), ("birth_death4", (12, 9, 10, 6)),
                                       ("birth_death6", (6, 4, 3, 2, 22, 3)), ("birth_death6", (4, 4, 4, 2, 13, 2)),
                                       ("repressilator", (10, 29, 5))])
def test_pencil_product_equals_the_trip_product_bit_for_bit(oracle, name, dims):
    """Kernel formats 7 and 8 (option box_pencil = 1 / 2).  Format 8 (the default for large boxes): a WORKGROUP owns 128 rows in W
    consecutive lines of the second-slowest species, its wavefronts walk the planes in step and hand each other their own pairs
    through LDS (that species' +-1 entries); lines that are no multiple of the workgroup, workgroups at the box's edge and group
    boundaries (memory gathers there) are in the cases below.  Format 7: a wavefront owns 128 rows of one plane of the slowest species and walks the planes;
    the slowest species' own entries take their source elements from the lane's previous / next pair (registers) and everything
    that depends on the other coordinates is worked out once per pencil.  Every row is the same sequence of fused multiply-adds
    over the same operands as in format 4: y must be bit-identical (here box_pencil = 1 forces it on small boxes; by default it
    serves boxes with >= 4096 base trips - tests/test_gpu_configs.py runs config 5 through it).  Planes that are not a multiple
    of 128 rows (a last base trip with dead lanes), two planes only, a 4-species model (instantiated with 6: not eligible, the
    option must leave format 4), and the fused modes (Arnoldi pass, fixed-(m, tau) steps) against the oracle."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.repressilator(dims=dims) if name == "repressilator" else synth.birth_death(dims)
    adj, off, diag = mdl.ell()
    A = oracle.EllMatrix(adj, off, diag)
    rng = np.random.default_rng(12)
    x = rng.random(mdl.n) - 0.3                                  # (both signs: the 0.0 * x of an absent entry may be -0.0)
    p0 = rng.random(mdl.n)
    p0 /= p0.sum()
    m, tau, nsteps = 14, 0.004, 2
    out = {}
    for pencil in (0, 1, 2):                                      # format 4 / pencils (format 7) / pencils in slabs (format 8)
        with KfspContext(0) as c:
            c.set_option("small_kernel", 0)
            c.set_option("box_pencil", pencil)
            c.set_matrix_box(mdl)
            fmt = c.layout_info()["format"]
            line_rows = int(np.prod(dims[:-2]))                  # (slabs need lines of an even number of rows: else pencils)
            want = 4 if not pencil or len(dims) not in (3, 6) else (8 if pencil == 2 and line_rows % 2 == 0 else 7)
            assert fmt == want, (fmt, pencil)
            y = c.spmv(x)
            c.set_vector(p0)
            c.begin_step()
            H, mb, k1, av = c.arnoldi(m)
            c.set_vector(p0)
            ws = c.expv_fixed(m, tau, nsteps)
            out[pencil] = dict(y=y, H=H.copy(), mb=mb, k1=k1, ws=ws, w=c.get_vector())
    assert np.array_equal(out[0]["y"], out[1]["y"]) and np.array_equal(out[0]["y"], out[2]["y"])
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    assert np.all(np.abs(out[1]["y"] - oracle.spmv_ell(A, x)) <= 1e-13 * np.abs(scale) + 1e-300)
    wref, wsref = oracle.expv_fixed(A, p0, m, tau, nsteps)
    V, Href, mbr, k1r, avr = oracle.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), m)
    for o in out.values():
        assert np.abs(o["w"] - wref).sum() < 1e-10 and np.abs(o["ws"] - wsref).max() < 1e-12
        assert (o["mb"], o["k1"]) == (mbr, k1r) and np.abs(o["H"][:9, :8] - Href[:9, :8]).max() <= 1e-11 * np.abs(Href).max()
