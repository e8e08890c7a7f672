"""Synthetic CME generators on hyper-rectangular state sets.

Input construction for benchmarks and parity tests (numpy, host side, not part
of the timed path).  States are the lattice points of a box
[0,n_1) x ... x [0,n_d) in lexicographic order with species 1 fastest, i.e.
state index i = sum_k x_k * stride_k, stride_1 = 1.

Two output forms:
  * `ell(...)`      the reference's own FSP_MATRIX layout (StateSpace.f90:13-17):
                    ADJ/OFFDIAG [state][slot] (1-based, 0 = successor outside
                    the box, -1 = negative population), DIAG = sum of ALL
                    propensities (StateSpace.f90:207-212) -> kfsp_set_matrix_ell
  * `csr_rows(...)` gather rows of a row block with global columns and the
                    diagonal stored in place -> kfsp_set_matrix_csr (what each
                    rank of a row partition builds for itself)

The propensity formulas restate the reference's model files (data):
models/toggle_model.input:29-32, models/repressilator_model.input:32-37,
models/goutsias_model.input:43-52; parameter values are the ones its drivers
use (test/TestSolverFromFile.f90:31, examples/transcr6d.f90:23-32).
"""
import numpy as np


class BoxModel:
    """stoich: (d, R) integer array; prop(r, X) -> propensity of reaction r at the
    coordinate arrays X[0..d-1] (float64 arrays of equal shape)."""

    def __init__(self, name, dims, stoich, prop, deps=None):
        self.name = name
        # deps[r]: the species propensity r depends on (it must be a product of one-species factors);
        # None = unknown, no matrix-free form
        self.deps = deps
        self.dims = tuple(int(x) for x in dims)
        self.stoich = np.asarray(stoich, dtype=np.int64)
        self.prop = prop
        self.d, self.R = self.stoich.shape
        assert self.d == len(self.dims)
        self.strides = np.concatenate(([1], np.cumprod(self.dims[:-1]))).astype(np.int64)
        self.n = int(np.prod(self.dims))
        self.offsets = (self.stoich * self.strides[:, None]).sum(axis=0)   # index shift of each reaction

    def coords(self, idx):
        return [(idx // self.strides[k]) % self.dims[k] for k in range(self.d)]

    # ---- reference layout -------------------------------------------------
    def ell(self):
        idx = np.arange(self.n, dtype=np.int64)
        X = self.coords(idx)
        Xf = [x.astype(np.float64) for x in X]
        adj = np.empty((self.n, self.R), dtype=np.int32)
        off = np.empty((self.n, self.R), dtype=np.float64)
        diag = np.zeros(self.n, dtype=np.float64)
        for r in range(self.R):
            neg = np.zeros(self.n, dtype=bool)
            out = np.zeros(self.n, dtype=bool)
            for k in range(self.d):
                s = self.stoich[k, r]
                if s:
                    y = X[k] + s
                    neg |= y < 0
                    out |= y >= self.dims[k]
            a = np.where(out, 0, idx + self.offsets[r] + 1)
            adj[:, r] = np.where(neg, -1, a)
            off[:, r] = self.prop(r, Xf)
            diag += off[:, r]            # sequential over reactions, as StateSpace.f90:207-212
        return adj, off, diag

    # ---- gather rows ----------------------------------------------------------
    def rows_at(self, rows):
        """gather rows of A at the global state indices `rows` (any order): (cols, vals), both
        [len(rows)][R + 1], entries sorted by column, absent entries have column int64 max and value 0;
        the diagonal -(sum of all propensities) is stored in place."""
        rows = np.asarray(rows, dtype=np.int64)
        nrows = len(rows)
        X = self.coords(rows)
        Xf = [x.astype(np.float64) for x in X]
        big = np.iinfo(np.int64).max
        cols = np.full((nrows, self.R + 1), big, dtype=np.int64)
        vals = np.zeros((nrows, self.R + 1), dtype=np.float64)
        dsum = np.zeros(nrows, dtype=np.float64)
        for r in range(self.R):
            ok = np.ones(nrows, dtype=bool)
            P = []
            for k in range(self.d):
                s = self.stoich[k, r]
                y = X[k] - s if s else X[k]
                if s:
                    ok &= (y >= 0) & (y < self.dims[k])
                P.append(y.astype(np.float64))
            a = self.prop(r, P)                       # propensity at the predecessor
            cols[:, r] = np.where(ok, rows - self.offsets[r], big)
            vals[:, r] = np.where(ok, a, 0.0)
            dsum += self.prop(r, Xf)                  # every reaction leaves the state
        cols[:, self.R] = rows
        vals[:, self.R] = -dsum
        order = np.argsort(cols, axis=1, kind="stable")
        cols = np.take_along_axis(cols, order, axis=1)
        vals = np.take_along_axis(vals, order, axis=1)
        return cols, vals

    def csr_rows(self, row0=0, nrows=None):
        if nrows is None:
            nrows = self.n - row0
        cols, vals = self.rows_at(np.arange(row0, row0 + nrows, dtype=np.int64))
        valid = cols != np.iinfo(np.int64).max
        rowptr = np.concatenate(([0], np.cumsum(valid.sum(axis=1)))).astype(np.int64)
        return rowptr, cols[valid].astype(np.int32), vals[valid]

    def colsum_dot(self, x, chunk=1 << 22):
        """1^T A x without A: column j of A sums to -(propensities of the reactions that LEAVE the box from
        state j) - the probability flux into the absorbing sink (KrylovSolver.f90:446-458) - so
        1^T A x = -sum_j x_j * sum_{r: x_j + nu_r outside the box} a_r(x_j).  Evaluated in chunks."""
        total = 0.0
        for b in range(0, self.n, chunk):
            idx = np.arange(b, min(self.n, b + chunk), dtype=np.int64)
            X = self.coords(idx)
            Xf = [c.astype(np.float64) for c in X]
            leak = np.zeros(len(idx))
            for r in range(self.R):
                out = np.zeros(len(idx), dtype=bool)
                for k in range(self.d):
                    s = self.stoich[k, r]
                    if s:
                        y = X[k] + s
                        out |= (y < 0) | (y >= self.dims[k])
                leak += np.where(out, self.prop(r, Xf) * np.ones(len(idx)), 0.0)
            total -= float(leak @ x[b:b + len(idx)])
        return total

    # ---- matrix-free form ---------------------------------------------------
    def factors(self):
        """(ndep[R], dep_species[R][3], tables) for kfsp_set_matrix_box: propensity r as a product of
        one-species factor tables, made with self.prop itself.  A propensity of one species IS its
        table (same bits as the stored entries); for several species the first factor carries the
        constant, the others are normalised by the value at population 1.  Separability is verified."""
        assert self.deps is not None, f"{self.name}: no dependency list, no matrix-free form"
        ndep = np.zeros(self.R, dtype=np.int32)
        dep = np.zeros((self.R, 3), dtype=np.int32)
        tabs = []
        rng = np.random.default_rng(0)
        for r in range(self.R):
            sp = tuple(self.deps[r]) or (0,)
            assert 1 <= len(sp) <= 3
            ndep[r] = len(sp)
            dep[r, :len(sp)] = sp
            ref = [np.ones(1) for _ in range(self.d)]                 # all other populations at 1
            base = float((self.prop(r, ref) * np.ones(1))[0])
            mine = []
            for i, s in enumerate(sp):
                X = [np.ones(self.dims[s]) for _ in range(self.d)]
                X[s] = np.arange(self.dims[s], dtype=np.float64)
                t = np.asarray(self.prop(r, X), dtype=np.float64) * np.ones(self.dims[s])
                if i > 0:
                    t = t / base
                mine.append(t)
            # check on random states of the box
            idx = rng.integers(0, self.n, 2000)
            C = self.coords(idx)
            want = self.prop(r, [x.astype(np.float64) for x in C]) * np.ones(len(idx))
            got = np.ones(len(idx))
            for i, s in enumerate(sp):
                got = got * mine[i][C[s]]
            assert np.all(np.abs(got - want) <= 4e-16 * np.abs(want) * len(sp)), f"{self.name}: propensity {r} is not separable"
            tabs += mine
        return ndep, dep, np.concatenate(tabs)

    def nnz_rows(self, row0, nrows, chunk=1 << 23):
        """true nonzeros incl. the diagonal of the gather rows [row0, row0 + nrows) (counted, no values)"""
        total = int(nrows)
        for b in range(row0, row0 + nrows, chunk):
            idx = np.arange(b, min(row0 + nrows, b + chunk), dtype=np.int64)
            X = self.coords(idx)
            for r in range(self.R):
                ok = np.ones(len(idx), dtype=bool)
                for k in range(self.d):
                    s = self.stoich[k, r]
                    if s:
                        y = X[k] - s
                        ok &= (y >= 0) & (y < self.dims[k])
                total += int(ok.sum())
        return total

    def nnz(self):
        """true nonzeros incl. the diagonal"""
        nnz = self.n
        for r in range(self.R):
            c = 1
            for k in range(self.d):
                c *= max(self.dims[k] - abs(int(self.stoich[k, r])), 0)
            nnz += c
        return nnz


def _outer_fastest_first(vectors):
    p = vectors[0]
    for v in vectors[1:]:
        p = (v[:, None] * p[None, :]).reshape(-1)
    return p


def poisson_p0(model, lam=30.0):
    vs = []
    for k in range(model.d):
        x = np.arange(model.dims[k], dtype=np.float64)
        lg = np.concatenate(([0.0], np.cumsum(np.log(np.arange(1, model.dims[k], dtype=np.float64)))))
        vs.append(np.exp(x * np.log(lam) - lam - lg))
    p = _outer_fastest_first(vs)
    return p / p.sum()


def toggle(n1=1000, n2=1000, params=(1.0, 100.0, 1.0, 1.0, 100.0, 1.0)):
    bx, kx, dx, by, ky, dy = params
    st = [[1, -1, 0, 0], [0, 0, 1, -1]]

    def prop(r, X):
        x, y = X
        if r == 0:
            return bx + kx / (2.0 + 0.2 * y ** 2)
        if r == 1:
            return dx * x
        if r == 2:
            return by + ky / (1.0 + 0.5 * x ** 1.5)
        return dy * y
    return BoxModel("toggle", (n1, n2), st, prop, deps=[(1,), (0,), (0,), (1,)])


def repressilator(n=171, params=(100.0, 100.0, 100.0, 1.0, 1.0, 1.0), dims=None):
    a1, a2, a3, b1, b2, b3 = params
    st = [[1, 0, 0, -1, 0, 0], [0, 1, 0, 0, -1, 0], [0, 0, 1, 0, 0, -1]]

    def prop(r, X):
        s1, s2, s3 = X
        if r == 0:
            return a1 / (1.0 + s3 ** 2.5)
        if r == 1:
            return a2 / (1.0 + s1 ** 1.5)
        if r == 2:
            return a3 / (1.0 + s2 ** 1.5)
        return (b1 * s1, b2 * s2, b3 * s3)[r - 3]
    return BoxModel("repressilator", dims if dims is not None else (n, n, n), st, prop,
                    deps=[(2,), (0,), (1,), (0,), (1,), (2,)])


def birth_death(dims, k=None, g=None):
    """d-species network of 2d reactions +-e_i with propensities k_i and g_i*x_i
    (SURVEY.md 8(d), config C5)."""
    d = len(dims)
    k = np.linspace(5.0, 9.0, d) if k is None else np.asarray(k, dtype=np.float64)
    g = np.linspace(0.6, 1.4, d) if g is None else np.asarray(g, dtype=np.float64)
    st = np.zeros((d, 2 * d), dtype=np.int64)
    for i in range(d):
        st[i, 2 * i] = 1
        st[i, 2 * i + 1] = -1

    def prop(r, X):
        i = r // 2
        return np.full_like(X[i], k[i]) if r % 2 == 0 else g[i] * X[i]
    return BoxModel("birth_death", dims, st, prop, deps=[(r // 2,) for r in range(2 * d)])


GOUTSIAS_PARAMS = (0.043, 0.0007, 0.0715, 0.0039, 0.0199264663575241, 0.4791,
                   0.000199264663575241, 0.8765e-11, 0.0830269431563506104, 0.5)


def goutsias_box(dims, params=GOUTSIAS_PARAMS):
    """Goutsias transcription model (species M, D, RNA, DNA, DNA.D, DNA.2D) on a
    plain box; used at small sizes for parity tests."""
    c = params
    st = np.zeros((6, 10), dtype=np.int64)
    M, D, RNA, DNA, DNAD, DNA2D = range(6)
    st[M, 0] = 1
    st[M, 1] = -1
    st[RNA, 2] = 1
    st[RNA, 3] = -1
    st[DNA, 4] = -1; st[D, 4] = -1; st[DNAD, 4] = 1
    st[DNA, 5] = 1; st[D, 5] = 1; st[DNAD, 5] = -1
    st[DNAD, 6] = -1; st[D, 6] = -1; st[DNA2D, 6] = 1
    st[DNAD, 7] = 1; st[D, 7] = 1; st[DNA2D, 7] = -1
    st[M, 8] = -2; st[D, 8] = 1
    st[M, 9] = 2; st[D, 9] = -1

    def prop(r, X):
        m, d_, rna, dna, dnad, dna2d = X
        return (c[0] * rna, c[1] * m, c[2] * dnad, c[3] * rna, c[4] * dna * d_, c[5] * dnad,
                c[6] * dnad * d_, c[7] * dna2d, c[8] * m * (m - 1) / 2.0, c[9] * d_)[r]
    M_, D_, RNA_, DNA_, DNAD_, DNA2D_ = range(6)
    return BoxModel("goutsias", dims, st, prop,
                    deps=[(RNA_,), (M_,), (DNAD_,), (RNA_,), (DNA_, D_), (DNAD_,), (DNAD_, D_), (DNA2D_,), (M_,), (D_,)])


class GoutsiasConserved:
    """BASELINE config 4: the Goutsias model on M, D, RNA in [0, nM) x [0, nD) x [0, nRNA)
    times the six ways two DNA copies split into (DNA, DNA.D, DNA.2D) - the state
    set a run from (2, 6, 0, 2, 0, 0) lives on (examples/transcr6d.f90:50), 150^3 x 6 =
    2.025e7 states at full size.  Index = M + nM (D + nD (RNA + nRNA cfg)).  Same
    interface as BoxModel (ell / csr_rows / nnz); species order of the reference's
    model file: M, D, RNA, DNA, DNA.D, DNA.2D."""

    CFG = np.array([(2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2)], dtype=np.int64)
    # (dM, dD, dRNA) and the successor configuration of every reaction (-1: a DNA form would go negative)
    _SAME = (0, 1, 2, 3, 4, 5)
    REACTIONS = [((1, 0, 0), _SAME), ((-1, 0, 0), _SAME), ((0, 0, 1), _SAME), ((0, 0, -1), _SAME),
                 ((0, -1, 0), (1, 3, 4, -1, -1, -1)), ((0, 1, 0), (-1, 0, -1, 1, 2, -1)),
                 ((0, -1, 0), (-1, 2, -1, 4, 5, -1)), ((0, 1, 0), (-1, -1, 1, -1, 3, 4)),
                 ((-2, 1, 0), _SAME), ((2, -1, 0), _SAME)]

    def __init__(self, nM=150, nD=150, nRNA=150, params=None):
        self.name = "goutsias_conserved"
        self.box = (int(nM), int(nD), int(nRNA))
        self.dims = self.box + (6,)
        self.c = GOUTSIAS_PARAMS if params is None else params
        self.strides = np.array([1, nM, nM * nD, nM * nD * nRNA], dtype=np.int64)
        self.n = int(nM * nD * nRNA * 6)
        self.d, self.R = 6, 10

    def coords(self, idx):
        """species counts (M, D, RNA, DNA, DNA.D, DNA.2D) of the states idx"""
        box = [(idx // self.strides[k]) % self.dims[k] for k in range(3)]
        cfg = idx // self.strides[3]
        return box + [self.CFG[cfg, k] for k in range(3)]

    def _prop(self, r, X):
        m, d_, rna, dna, dnad, dna2d = X
        c = self.c
        return (c[0] * rna, c[1] * m, c[2] * dnad, c[3] * rna, c[4] * dna * d_, c[5] * dnad,
                c[6] * dnad * d_, c[7] * dna2d, c[8] * m * (m - 1) / 2.0, c[9] * d_)[r]

    def _step(self, idx, r, sign):
        """index of the state reached from idx by reaction r (sign +1) or of the state
        that reaches idx by it (sign -1); -1 = negative population, -2 = outside the box"""
        delta, succ = self.REACTIONS[r]
        cmap = np.asarray(succ, dtype=np.int64)
        if sign < 0:
            inv = np.full(6, -1, dtype=np.int64)
            inv[cmap[cmap >= 0]] = np.nonzero(cmap >= 0)[0]
            cmap = inv
        cfg = idx // self.strides[3]
        cfg2 = cmap[cfg]
        tgt = idx + (cfg2 - cfg) * self.strides[3]
        neg = cfg2 < 0
        out = np.zeros(len(idx), dtype=bool)
        for k in range(3):
            s = sign * delta[k]
            if s:
                y = (idx // self.strides[k]) % self.dims[k] + s
                neg |= y < 0
                out |= y >= self.dims[k]
                tgt = tgt + s * self.strides[k]
        return np.where(neg, -1, np.where(out, -2, tgt))

    def ell(self):
        idx = np.arange(self.n, dtype=np.int64)
        Xf = [x.astype(np.float64) for x in self.coords(idx)]
        adj = np.empty((self.n, self.R), dtype=np.int32)
        off = np.empty((self.n, self.R), dtype=np.float64)
        diag = np.zeros(self.n, dtype=np.float64)
        for r in range(self.R):
            t = self._step(idx, r, +1)
            adj[:, r] = np.where(t == -1, -1, np.where(t == -2, 0, t + 1))
            off[:, r] = self._prop(r, Xf)
            diag += off[:, r]
        return adj, off, diag

    def csr_rows(self, row0=0, nrows=None):
        if nrows is None:
            nrows = self.n - row0
        rows = np.arange(row0, row0 + nrows, dtype=np.int64)
        Xf = [x.astype(np.float64) for x in self.coords(rows)]
        big = np.iinfo(np.int64).max
        cols = np.full((nrows, self.R + 1), big, dtype=np.int64)
        vals = np.zeros((nrows, self.R + 1), dtype=np.float64)
        dsum = np.zeros(nrows, dtype=np.float64)
        for r in range(self.R):
            src = self._step(rows, r, -1)
            ok = src >= 0
            P = [x.astype(np.float64) for x in self.coords(np.where(ok, src, 0))]
            cols[:, r] = np.where(ok, src, big)
            vals[:, r] = np.where(ok, self._prop(r, P), 0.0)
            dsum += self._prop(r, Xf)
        cols[:, self.R] = rows
        vals[:, self.R] = -dsum
        order = np.argsort(cols, axis=1, kind="stable")
        cols = np.take_along_axis(cols, order, axis=1)
        vals = np.take_along_axis(vals, order, axis=1)
        valid = cols != big
        rowptr = np.concatenate(([0], np.cumsum(valid.sum(axis=1)))).astype(np.int64)
        return rowptr, cols[valid].astype(np.int32), vals[valid]

    def nnz(self):
        """true nonzeros incl. the diagonal (links whose propensity happens to be 0 count, as in BoxModel)"""
        nnz = self.n
        for delta, succ in self.REACTIONS:
            c = sum(1 for s in succ if s >= 0)
            for k in range(3):
                c *= max(self.box[k] - abs(delta[k]), 0)
            nnz += c
        return nnz


class GoutsiasEllipsoid:
    """A NON-BOX FSP of the Goutsias model for bandwidth measurements at sizes the reference's own solver never
    reaches: the states (M, D, RNA, DNA, DNA.D, DNA.2D) with DNA + DNA.D + DNA.2D = 2 (the six configurations a run
    from (2, 6, 0, 2, 0, 0) lives on, examples/transcr6d.f90:50) and (M, D, RNA) inside an ellipsoid clipped at 0 -
    the shape a probability-mass truncation gives - listed NOT in lexicographic order but in the order of a
    reachability search from a seed state: by graph distance |dM| + |dD| + |dRNA| from the seed, ties shuffled
    (fixed seed) - which scatters the neighbours of a state over the vector as SSA_EXTENDER / ONESTEP_EXTENDER
    (StateSpace.f90:347-396, :550-630) do.  `ell()` gives the reference's FSP_MATRIX arrays (1-based ADJ, 0 =
    target not listed, -1 = negative population), `state` the species counts [n][6] (FSP%STATE)."""

    def __init__(self, center=(60, 50, 20), axes=(110, 90, 80), seed_state=(2, 6, 0), params=None, order="search", rng_seed=7):
        self.c = GOUTSIAS_PARAMS if params is None else params
        cM, cD, cR = center
        aM, aD, aR = axes
        m = np.arange(0, int(cM + aM) + 1, dtype=np.int64)
        d = np.arange(0, int(cD + aD) + 1, dtype=np.int64)
        r = np.arange(0, int(cR + aR) + 1, dtype=np.int64)
        inside = (((m - cM) / aM) ** 2)[:, None, None] + (((d - cD) / aD) ** 2)[None, :, None] + \
                 (((r - cR) / aR) ** 2)[None, None, :] <= 1.0
        M, D, R = np.nonzero(inside)                               # lexicographic with RNA fastest here; reordered below
        nb = len(M)
        cfg = np.repeat(np.arange(6, dtype=np.int64), nb)
        M, D, R = np.tile(M, 6), np.tile(D, 6), np.tile(R, 6)
        n = len(M)
        if order == "search":
            dist = np.abs(M - seed_state[0]) + np.abs(D - seed_state[1]) + np.abs(R - seed_state[2]) + 3 * cfg
            tie = np.random.default_rng(rng_seed).permutation(n)
            o = np.lexsort((tie, dist))
        elif order == "lex":                                       # species 1 (M) fastest, the configuration slowest
            o = np.lexsort((M, D, R, cfg))
        else:
            raise ValueError(order)
        M, D, R, cfg = M[o], D[o], R[o], cfg[o]
        C = GoutsiasConserved.CFG
        self.state = np.stack([M, D, R, C[cfg, 0], C[cfg, 1], C[cfg, 2]], axis=1).astype(np.int32)
        self.n, self.d, self.R = n, 6, 10
        self._M, self._D, self._R, self._cfg = M, D, R, cfg
        self._shape = (len(m), len(d), len(r), 6)

    def ell(self):
        M, D, R, cfg = self._M, self._D, self._R, self._cfg
        nM, nD, nR, _ = self._shape
        n = self.n
        lut = np.zeros(nM * nD * nR * 6, dtype=np.int32)           # dense (M, D, RNA, cfg) -> 1-based index, 0 = not listed
        key = ((cfg * nR + R) * nD + D) * nM + M
        lut[key] = np.arange(1, n + 1, dtype=np.int32)
        X = [M.astype(np.float64), D.astype(np.float64), R.astype(np.float64)] + \
            [GoutsiasConserved.CFG[cfg, k].astype(np.float64) for k in range(3)]
        m, d_, rna, dna, dnad, dna2d = X
        c = self.c
        props = (c[0] * rna, c[1] * m, c[2] * dnad, c[3] * rna, c[4] * dna * d_, c[5] * dnad,
                 c[6] * dnad * d_, c[7] * dna2d, c[8] * m * (m - 1) / 2.0, c[9] * d_)
        adj = np.empty((n, 10), dtype=np.int32)
        off = np.empty((n, 10), dtype=np.float64)
        diag = np.zeros(n, dtype=np.float64)
        for k, (delta, succ) in enumerate(GoutsiasConserved.REACTIONS):
            cmap = np.asarray(succ, dtype=np.int64)
            c2 = cmap[cfg]
            m2, d2, r2 = M + delta[0], D + delta[1], R + delta[2]
            neg = (c2 < 0) | (m2 < 0) | (d2 < 0) | (r2 < 0)
            out = (m2 >= nM) | (d2 >= nD) | (r2 >= nR)
            ok = ~(neg | out)
            k2 = ((np.where(ok, c2, 0) * nR + np.where(ok, r2, 0)) * nD + np.where(ok, d2, 0)) * nM + np.where(ok, m2, 0)
            adj[:, k] = np.where(neg, -1, np.where(out, 0, lut[k2]))
            off[:, k] = props[k]
            diag += off[:, k]
        return adj, off, diag


def spmv_alg_bytes(nnz, n):
    """Algorithmic bytes of one generator SpMV: CSR with f64
    values + int32 columns (12 B per nonzero incl. the diagonal) and per row a
    4-B row pointer, x read once, y written once (20 B)."""
    return 12 * int(nnz) + 20 * int(n)
