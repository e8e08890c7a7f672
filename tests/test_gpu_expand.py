"""The expansion step on the resident lists (kfsp_expand_resident, csrc/kfsp_expand.hip): SSA_EXTENDER (independent
streams) + ONESTEP_EXTENDER of KrylovSolver.f90:518-534 without a trip to the host.  It must give exactly what the two
calls it fuses - kfsp_ssa_streams and kfsp_onestep_columns, each pinned on its own (tests/test_fortran_host.py against
the host walk, tests/test_gpu_onestep.py against the reference's assemblies) - give on host copies of the lists: the
same states in the same order, the same links, the same propensity columns; the resident vector padded with zeros; the
rebuilt generator's products bit-identical to those of a generator uploaded from the downloaded lists.  Then a whole
cycle drop -> rebuild -> expand, all on the device."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stoich(a):
    nr = a["adj"].shape[1]
    nu = [None] * nr
    for i, row in enumerate(a["adj"]):
        for r, j in enumerate(row):
            if j > 0 and nu[r] is None:
                nu[r] = a["state"][j - 1] - a["state"][i]
    assert all(v is not None for v in nu)
    return np.array(nu, dtype=np.int32)


def _mass_action(nu):
    """a_k = c_k * prod of the species reaction k consumes (postfix code of kfsp_set_propensity_program; + - * / only:
    the device's columns are the same bits wherever they are made)"""
    nr, ns = nu.shape
    MUL = 5
    progs, params = [], []
    for k in range(nr):
        params.append(0.05 + 0.01 * k)
        code = [100 + ns + 1 + k]
        for s in range(ns):
            if nu[k, s] < 0:
                code += [100 + s + 1, MUL]
        progs.append((code, []))
    return np.array(params), progs


def _grown(c, name, golden_dir, sweeps):
    a = np.load(os.path.join(golden_dir, f"assembly_{name}.npz"))
    nu = _stoich(a)
    state, adj = a["state"], a["adj"]
    for _ in range(sweeps):
        state, adj = c.onestep(nu, state, adj)
    return nu, state, adj


def _host_composition(c, t, seed, nu, state, adj, off, diag):
    s1, o1, d1 = c.ssa_streams(t, seed, nu, state, adj, off, diag)
    state1 = np.concatenate([state, s1])
    adj1 = np.concatenate([adj, np.zeros((len(s1), adj.shape[1]), dtype=np.int32)])       # appended, not linked
    state2, adj2, o2, d2 = c.onestep_columns(nu, state1, adj1)
    return len(s1), state2, adj2, np.concatenate([off, o1, o2]), np.concatenate([diag, d1, d2])


@pytest.mark.parametrize("name,sweeps,order", [("toggle_k20", 6, 0), ("goutsias_k16", 4, 0), ("goutsias_k16", 4, 1),
                                               ("repressilator_k10", 3, 1)])
def test_resident_expansion_equals_the_two_calls_on_host_lists(golden_dir, name, sweeps, order):
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(5)
    with KfspContext(0) as c, KfspContext(0) as ref:
        nu, state, adj = _grown(c, name, golden_dir, sweeps)
        nr, ns = nu.shape
        params, progs = _mass_action(nu)
        for x in (c, ref):
            x.set_propensity_program(ns, params, progs)
            x.set_option("state_order", order)
            x.set_option("state_order_min", 64)
            x.set_option("state_order_products", 0)
        ref.set_option("build_speculate", 0)              # (the reference side waits for every number and insertion-sorts its rows)
        ref.set_option("ssa_regs", 0)                     # (and walks unlisted states through the interpreter)
        off, diag = c.propensities(state)
        n = len(state)
        t = 2.0 / float(np.mean(diag[diag > 0]))
        seed = 123456789
        nssa_want, state2, adj2, off2, diag2 = _host_composition(ref, t, seed, nu, state, adj, off, diag)
        assert nssa_want > 0 and len(state2) > n + nssa_want          # both halves of the step appended something

        c.set_option("keep_coords", 1)
        c.set_state_coords(state)
        c.set_matrix_ell(adj, off, diag)
        assert bool(c.state_order_active()) == bool(order)
        w = rng.random(n)
        c.set_vector(w)
        n2, nssa = c.expand_resident(t, seed, nu)
        assert (n2, nssa) == (len(state2), nssa_want)
        s_r, a_r, o_r, d_r = c.download_fsp(ns, nr)
        assert np.array_equal(s_r, state2)
        assert np.array_equal(a_r, adj2)
        assert np.array_equal(o_r, off2)
        assert np.array_equal(d_r, diag2)
        assert np.array_equal(c.get_vector(), np.concatenate([w, np.zeros(n2 - n)]))
        assert bool(c.state_order_active()) == bool(order)

        # the rebuilt generator against one uploaded from the same lists
        ref.set_state_coords(state2)
        ref.set_matrix_ell(adj2, off2, diag2)
        x = rng.random(n2)
        assert np.array_equal(c.spmv(x), ref.spmv(x))


def test_drop_rebuild_expand_cycle_stays_on_the_device(golden_dir):
    """three rounds of: products, a drop decided and applied on the device, the resident expansion - against a host that
    carries its own copy of the lists through the same calls"""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(11)
    with KfspContext(0) as c, KfspContext(0) as ref:
        nu, state, adj = _grown(c, "goutsias_k16", golden_dir, 4)
        nr, ns = nu.shape
        params, progs = _mass_action(nu)
        for x in (c, ref):
            x.set_propensity_program(ns, params, progs)
            x.set_option("state_order_min", 64)
            x.set_option("state_order_products", 0)
        ref.set_option("build_speculate", 0)              # (the reference side waits for every number and insertion-sorts its rows)
        off, diag = c.propensities(state)
        c.set_option("keep_coords", 1)
        c.set_state_coords(state)
        c.set_matrix_ell(adj, off, diag)
        # a vector with a tail of tiny entries, so that the drop rule has something to drop
        # (the later half of the list - the outer states - carries next to nothing: the drop rule has something to drop)
        w = rng.random(len(state)) * np.where(np.arange(len(state)) >= len(state) // 2, 1e-14, 1.0)
        w /= w.sum()
        c.set_vector(w)
        for cycle in range(3):
            n = len(state)
            droptol, cnt, nflag = c.drop_plan(1e-7)
            assert nflag > n // 20, (cycle, nflag, n)            # (the 10 % rule is the caller's; the mechanics are tested here)
            flags = c.drop_flags()
            nk = c.drop_compact()
            c.drop_rebuild()
            keep = flags == 0
            assert nk == int(keep.sum())
            # the host's compaction of its copy (StateSpace.f90:500-546)
            newidx = np.cumsum(keep) * keep
            adj = np.where(adj > 0, newidx[np.maximum(adj, 1) - 1], adj)[keep].astype(np.int32)
            state, off, diag, w = state[keep], off[keep], diag[keep], w[keep]
            s_r, a_r, o_r, d_r = c.download_fsp(ns, nr)
            assert np.array_equal(s_r, state) and np.array_equal(a_r, adj) and np.array_equal(o_r, off) and np.array_equal(d_r, diag)
            t = 2.0 / float(np.mean(diag[diag > 0]))
            seed = 1000 + cycle
            _, state, adj, off, diag = _host_composition(ref, t, seed, nu, state, adj, off, diag)
            n2, _ = c.expand_resident(t, seed, nu)
            assert n2 == len(state)
            s_r, a_r, o_r, d_r = c.download_fsp(ns, nr)
            assert np.array_equal(s_r, state) and np.array_equal(a_r, adj) and np.array_equal(o_r, off) and np.array_equal(d_r, diag)
            w = np.concatenate([w, np.zeros(n2 - len(w))])
            assert np.array_equal(c.get_vector(), w)
            ref.set_state_coords(state)
            ref.set_matrix_ell(adj, off, diag)
            x = rng.random(n2)
            y = c.spmv(x)
            assert np.array_equal(y, ref.spmv(x))
            # the next cycle's vector: a few products' worth of spreading, tiny tail again
            w = np.abs(x) * np.where(np.arange(n2) >= n2 // 2, 1e-14, 1.0)
            w /= w.sum()
            c.set_vector(w)
        info = c.build_info()
        assert info["speculative"] + info["repeated"] >= 6 and info["sell"] == 1, info     # 3 drops + 3 expansions, none the slow way first
        assert info["orders_carried_over"] >= 5, info      # (compacted after a drop, the appended keys merged in after an expansion)
        assert ref.build_info()["speculative"] == 0


def test_a_speculation_that_does_not_hold_is_repeated_the_slow_way(golden_dir):
    """The resident FSP is re-ordered with the key layout of the LAST order (fields as wide as the bits they occupy) and
    rebuilt as the SELL generator the last one was, with one synchronisation at the end (option build_speculate).  Growing
    an FSP until a population crosses a power of two makes the cached layout too narrow: the check at the end must see it
    and repeat order and build - the generator is the one a fresh upload of the same lists gives, bit for bit, every time."""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(17)
    with KfspContext(0) as c, KfspContext(0) as ref:
        nu, state, adj = _grown(c, "goutsias_k16", golden_dir, 2)
        nr, ns = nu.shape
        params, progs = _mass_action(nu)
        for x in (c, ref):
            x.set_propensity_program(ns, params, progs)
            x.set_option("state_order_min", 64)
            x.set_option("state_order_products", 0)
        ref.set_option("build_speculate", 0)
        off, diag = c.propensities(state)
        c.set_option("keep_coords", 1)
        c.set_state_coords(state)
        c.set_matrix_ell(adj, off, diag)
        c.set_vector(np.full(len(state), 1.0 / len(state)))
        top = int(state.max())
        for step in range(12):
            n2, _ = c.expand_resident(4.0 / float(np.mean(diag[diag > 0])), 77 + step, nu, max_count=100000)
            state, adj, off, diag = c.download_fsp(ns, nr)
            assert len(state) == n2
            ref.set_state_coords(state)
            ref.set_matrix_ell(adj, off, diag)
            x = rng.random(n2)
            assert np.array_equal(c.spmv(x), ref.spmv(x)), step
            info = c.build_info()
            if info["repeated"] >= 1 and info["speculative"] >= 1:
                break
        assert info["repeated"] >= 1 and info["speculative"] >= 1, (info, top, int(state.max()), n2)


def test_random_sequences_of_drops_and_expansions_keep_the_carried_order_right(golden_dir):
    """24 FSP changes in random order - expansions of random horizons, drops of random depth, two drops in a row, the state
    order switched off for a while (the carried order must not survive that) and on again: after every change the resident
    generator multiplies exactly as one uploaded from the downloaded lists into a context that sorts every key and waits for
    every number (build_speculate = 0)."""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(2026)
    with KfspContext(0) as c, KfspContext(0) as ref:
        nu, state, adj = _grown(c, "goutsias_k16", golden_dir, 3)
        nr, ns = nu.shape
        params, progs = _mass_action(nu)
        for x in (c, ref):
            x.set_propensity_program(ns, params, progs)
            x.set_option("state_order_min", 64)
            x.set_option("state_order_products", 0)
        ref.set_option("build_speculate", 0)
        off, diag = c.propensities(state)
        c.set_option("keep_coords", 1)
        c.set_state_coords(state)
        c.set_matrix_ell(adj, off, diag)
        n = len(state)
        carried = 0
        for step in range(24):
            if step in (8, 16):                                   # the order off for four changes, then on again
                c.set_option("state_order", 0)
            if step in (12, 20):
                c.set_option("state_order", 1)
            w = rng.random(n) * np.where(rng.random(n) < rng.uniform(0.2, 0.6), 1e-14, 1.0)
            c.set_vector(w / w.sum())
            drop = n > 3000 and rng.random() < 0.5
            if drop:
                _, _, nflag = c.drop_plan(1e-7)
                if nflag == 0 or nflag == n:
                    drop = False
                else:
                    n = c.drop_compact()
                    c.drop_rebuild()
            if not drop:
                t = float(rng.uniform(0.5, 3.0)) / float(np.mean(diag[diag > 0]))
                n, _ = c.expand_resident(t, int(rng.integers(1, 2 ** 31 - 2)), nu, max_count=100000)
            state, adj, off, diag = c.download_fsp(ns, nr)
            assert len(state) == n
            ref.set_option("state_order", 1 if c.state_order_active() else 0)
            ref.set_state_coords(state)
            ref.set_matrix_ell(adj, off, diag)
            x = rng.random(n)
            assert np.array_equal(c.spmv(x), ref.spmv(x)), (step, drop, n)
            carried = c.build_info()["orders_carried_over"]
        info = c.build_info()
        assert info["speculative"] >= 12 and carried >= 8, info


@pytest.mark.parametrize("P", [2, 3])
@pytest.mark.parametrize("order", [0, 1])
def test_resident_expansion_under_a_row_partition(golden_dir, P, order):
    """a group context (P loop-back ranks): every rank holds the whole lists and expands them redundantly, the vector is
    assembled and dealt out again in the new partition - the head must give what one context gives"""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(3)
    res = []
    for group in (None, P):
        with KfspContext(0, group=group) as c:
            nu, state, adj = _grown(c, "goutsias_k16", golden_dir, 3)
            nr, ns = nu.shape
            params, progs = _mass_action(nu)
            c.set_propensity_program(ns, params, progs)
            c.set_option("state_order", order)
            c.set_option("state_order_min", 64)
            c.set_option("state_order_products", 0)
            c.set_option("small_kernel", 0)
            c.set_option("keep_coords", 1)
            off, diag = c.propensities(state)
            c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            n = len(state)
            w = np.random.default_rng(8).random(n)
            c.set_vector(w)
            t = 2.0 / float(np.mean(diag[diag > 0]))
            n2, nssa = c.expand_resident(t, 4242, nu)
            lists = c.download_fsp(ns, nr)
            x = np.random.default_rng(9).random(n2)
            res.append(dict(n2=n2, nssa=nssa, lists=lists, w=c.get_vector(), y=c.spmv(x)))
            # a second round on the grown FSP, after a drop decided on the device
            wv = np.random.default_rng(10).random(n2) * np.where(np.arange(n2) >= n2 // 2, 1e-14, 1.0)
            c.set_vector(wv / wv.sum())
            c.drop_plan(1e-7)
            nk = c.drop_compact()
            c.drop_rebuild()
            n3, _ = c.expand_resident(t, 4243, nu)
            res[-1].update(nk=nk, n3=n3, lists3=c.download_fsp(ns, nr), w3=c.get_vector(), info=c.build_info())
    a, b = res
    # (the ranks of the partition rebuild speculatively and carry their order over as one context does: 2 expansions + 1 drop)
    assert b["info"]["speculative"] + b["info"]["repeated"] >= 3, b["info"]
    assert b["info"]["orders_carried_over"] >= (2 if order else 0), b["info"]
    assert (a["n2"], a["nssa"], a["nk"], a["n3"]) == (b["n2"], b["nssa"], b["nk"], b["n3"]) and a["nssa"] > 0 and a["n2"] > a["nssa"]
    for u, v in zip(a["lists"] + a["lists3"], b["lists"] + b["lists3"]):
        assert np.array_equal(u, v)
    assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["w3"], b["w3"])
    assert np.array_equal(a["y"], b["y"])                  # every row is summed in FMATVEC's order whoever owns it


def test_walk_register_path_with_a_tabulated_reaction(golden_dir):
    """The walk's register path (ssa_regs) reads a reaction that is NOT a product chain from its one-species table, and meets
    the listed states through the tagged table behind its bit-map filter (ssa_filter): a Goutsias-like program whose
    dimerisation c X (X - 1) / 2 is tabulated - the other reactions are chains - must give the records the interpreter and
    the unfiltered table give, from every seed, for a short and a long horizon (paths that stay inside / leave the FSP)."""
    from krylovfspssa_amd import KfspContext
    MUL, SUB, DIV, IMM = 5, 4, 6, 1
    with KfspContext(0) as a, KfspContext(0) as b:
        nu, state, adj = _grown(a, "goutsias_k16", golden_dir, 3)
        nr, ns = nu.shape
        params, progs = _mass_action(nu)
        dimer = [k for k in range(nr) if (nu[k] == -2).any()]
        assert dimer, "the model has a dimerisation"
        tab_len = 4096
        ts = np.full(nr, -1, dtype=np.int32)
        tab = np.zeros((nr, tab_len))
        for k in dimer:
            s = int(np.where(nu[k] == -2)[0][0])
            c = 100 + ns + 1 + k                                     # parameter k
            progs[k] = ([c, 100 + s + 1, MUL, 100 + s + 1, IMM, SUB, MUL, IMM, DIV], [1.0, 2.0])
            v = np.arange(tab_len, dtype=np.float64)
            tab[k] = ((params[k] * v) * (v - 1.0)) / 2.0             # the program's operations in its order
            ts[k] = s
        for x in (a, b):
            x.set_propensity_program(ns, params, progs, tables=(ts, tab))
        b.set_option("ssa_regs", 0)
        b.set_option("ssa_filter", 0)
        off, diag = a.propensities(state)
        off_b, diag_b = b.propensities(state)
        assert np.array_equal(off, off_b) and np.array_equal(diag, diag_b)
        for scale, seed in ((0.5, 11), (8.0, 12), (40.0, 13)):
            t = scale / float(np.mean(diag[diag > 0]))
            ra = a.ssa_streams(t, seed, nu, state, adj, off, diag, max_count=tab_len - 1, capacity_new=1 << 20)
            rb = b.ssa_streams(t, seed, nu, state, adj, off, diag, max_count=tab_len - 1, capacity_new=1 << 20)
            assert len(ra[0]) > 0
            for u, v in zip(ra, rb):
                assert np.array_equal(u, v), (scale, len(ra[0]), len(rb[0]))


def test_refusals(golden_dir):
    from krylovfspssa_amd import KfspContext, KfspError
    with KfspContext(0) as c:
        nu, state, adj = _grown(c, "toggle_k20", golden_dir, 0)
        nr, ns = nu.shape
        params, progs = _mass_action(nu)
        c.set_propensity_program(ns, params, progs)
        off, diag = c.propensities(state)
        c.set_matrix_ell(adj, off, diag)                       # no coordinates on the device
        c.set_vector(np.ones(len(state)))
        with pytest.raises(KfspError):
            c.expand_resident(1.0, 1, nu)
        c.set_option("keep_coords", 1)
        c.set_state_coords(state)
        c.set_matrix_ell(adj, off, diag)
        with pytest.raises(KfspError):                          # capacity: the reference STOPs here
            c.expand_resident(1.0, 1, nu, capacity=len(state) + 1)
        # the refused expansion left the FSP as it was
        s_r, a_r, _, _ = c.download_fsp(ns, nr)
        assert np.array_equal(s_r, state) and np.array_equal(a_r, adj)


def test_random_networks_resident_equals_host_lists():
    """40 random networks (1-4 species, 1-6 reactions with entries in [-2, 2]; mass-action propensities, so the walk's
    product-chain path and the interpreter both occur - a reaction that consumes nothing is a constant) on random subsets
    of small boxes, random horizon: the resident expansion against the two calls on host copies of the lists"""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(99)
    for case in range(40):
        ns, nr = int(rng.integers(1, 5)), int(rng.integers(1, 7))
        nu = rng.integers(-2, 3, size=(nr, ns)).astype(np.int32)
        side = int(rng.integers(3, 7))
        box = np.array(np.meshgrid(*[np.arange(side)] * ns, indexing="ij")).reshape(ns, -1).T
        keep = rng.random(len(box)) < rng.uniform(0.3, 1.0)
        keep[int(rng.integers(0, len(box)))] = True
        state = box[keep][rng.permutation(int(keep.sum()))].astype(np.int32)
        params, progs = _mass_action(nu)
        with KfspContext(0) as c, KfspContext(0) as ref:
            for x in (c, ref):
                x.set_propensity_program(ns, params, progs)
                x.set_option("state_order", case % 2)
                x.set_option("state_order_min", 1)
                x.set_option("state_order_products", 0)
            ref.set_option("ssa_regs", 0)                      # (the reference side walks unlisted states through the interpreter)
            idx = {tuple(v): i + 1 for i, v in enumerate(state.tolist())}
            adj = np.zeros((len(state), nr), dtype=np.int32)                # complete links among the listed states
            for j, v in enumerate(state):
                for k in range(nr):
                    y = v + nu[k]
                    adj[j, k] = -1 if y.min() < 0 else idx.get(tuple(y.tolist()), 0)
            off, diag = c.propensities(state)
            n = len(state)
            t = float(rng.uniform(0.5, 6.0)) / max(float(diag.max()), 1e-3)
            seed = int(rng.integers(1, 2 ** 31 - 2))
            s1, o1, d1 = ref.ssa_streams(t, seed, nu, state, adj, off, diag, max_count=side + 3)
            st1 = np.concatenate([state, s1])
            ad1 = np.concatenate([adj, np.zeros((len(s1), nr), dtype=np.int32)])
            st2, ad2, o2, d2 = ref.onestep_columns(nu, st1, ad1, max_count=side + 3)
            c.set_option("keep_coords", 1)
            c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            w = rng.random(n)
            c.set_vector(w)
            n2, nssa = c.expand_resident(t, seed, nu, max_count=side + 3)
            assert (n2, nssa) == (len(st2), len(s1)), case
            s_r, a_r, o_r, d_r = c.download_fsp(ns, nr)
            assert np.array_equal(s_r, st2) and np.array_equal(a_r, ad2), case
            assert np.array_equal(o_r, np.concatenate([off, o1, o2])) and np.array_equal(d_r, np.concatenate([diag, d1, d2])), case
            assert np.array_equal(c.get_vector(), np.concatenate([w, np.zeros(n2 - n)])), case
