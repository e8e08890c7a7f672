"""ONESTEP_EXTENDER on the device (kfsp_onestep) against the reference's
own assemblies: starting from the FSP of k one-step sweeps (fixture assembly_*_k<k>), further
sweeps on the device must give the state list and the link array of the reference's k' > k sweeps,
bit for bit (StateSpace.f90:136-246, 347-396)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("toggle", 5, 10), ("toggle", 10, 20), ("repressilator", 5, 10), ("goutsias", 5, 10), ("goutsias", 10, 16)]


def _stoich(a):
    """reaction vectors from a reference assembly: successor - state of any linked pair"""
    nr = a["adj"].shape[1]
    nu = [None] * nr
    for i, row in enumerate(a["adj"]):
        for r, j in enumerate(row):
            if j > 0 and nu[r] is None:
                nu[r] = a["state"][j - 1] - a["state"][i]
    assert all(v is not None for v in nu)
    return np.array(nu, dtype=np.int32)


@pytest.mark.parametrize("name,k0,k1", CASES)
def test_device_sweeps_reproduce_the_reference_assembly(golden_dir, name, k0, k1):
    from krylovfspssa_amd import KfspContext
    a = np.load(os.path.join(golden_dir, f"assembly_{name}_k{k0}.npz"))
    b = np.load(os.path.join(golden_dir, f"assembly_{name}_k{k1}.npz"))
    nu = _stoich(b)
    state, adj = a["state"], a["adj"]
    with KfspContext(0) as c:
        for _ in range(k1 - k0):
            state, adj = c.onestep(nu, state, adj)
    assert state.shape == b["state"].shape
    assert np.array_equal(state, b["state"])
    assert np.array_equal(adj, b["adj"])


def test_population_cap_and_capacity(golden_dir):
    """targets above MAXNUMBERMOLECULES are not states (their links stay 0); a capacity that is too
    small is reported like the reference's STOP"""
    from krylovfspssa_amd import KfspContext, KfspError
    a = np.load(os.path.join(golden_dir, "assembly_toggle_k5.npz"))
    nu = _stoich(np.load(os.path.join(golden_dir, "assembly_toggle_k10.npz")))
    cap = int(a["state"].max())
    with KfspContext(0) as c:
        state, adj = c.onestep(nu, a["state"], a["adj"], max_count=cap)
        assert state.max() == cap                       # nothing beyond the cap was added
        assert len(state) > len(a["state"])
        top = np.where((state == cap).any(axis=1))[0]
        assert (adj[top] == 0).any()
        with pytest.raises(KfspError):
            c.onestep(nu, a["state"], a["adj"], capacity=len(a["state"]) + 1)


@pytest.mark.parametrize("name,k0,k1", [("toggle", 5, 10), ("goutsias", 10, 16), ("repressilator", 5, 10)])
def test_columns_that_arrive_unlinked_are_completed(golden_dir, name, k0, k1):
    """A caller may hand over states it appended without linking them: their columns arrive as zeros, and
    so do the entries of older states that point at them.  The sweep must then give what it gives from
    the fully linked array - successors present, -1 for a negative population (StateSpace.f90:213-244)."""
    from krylovfspssa_amd import KfspContext
    a = np.load(os.path.join(golden_dir, f"assembly_{name}_k{k0}.npz"))
    nu = _stoich(np.load(os.path.join(golden_dir, f"assembly_{name}_k{k1}.npz")))
    state, adj = a["state"], a["adj"]
    n = len(state)
    lo = n - n // 3                                     # the last third plays the appended states
    open_adj = adj.copy()
    open_adj[lo:] = 0
    open_adj[:lo][adj[:lo] > lo] = 0                    # links of older states into the tail: not made yet
    assert (adj[lo:] == -1).any() and (adj[:lo] > lo).any()      # both kinds of entries are exercised
    with KfspContext(0) as c:
        s_ref, a_ref = c.onestep(nu, state, adj)
        s_new, a_new = c.onestep(nu, state, open_adj)
    assert np.array_equal(s_new, s_ref)
    assert np.array_equal(a_new, a_ref)


def _onestep_py(nu, state, adj, max_count):
    """ONESTEP_EXTENDER restated in plain Python (StateSpace.f90:347-396 with ADD_STATE :136-246): the open links
    of the listed states in (state, reaction) order, a target appended the first time it is named, then the
    appended states' own columns"""
    nr, ns = nu.shape
    state = [tuple(int(v) for v in s) for s in state]
    adj = [[int(v) for v in r] for r in adj]
    idx = {s: i + 1 for i, s in enumerate(state)}
    n0 = len(state)
    for j in range(n0):
        for k in range(nr):
            if adj[j][k] != 0:
                continue
            y = tuple(state[j][s] + int(nu[k, s]) for s in range(ns))
            if min(y) < 0:
                adj[j][k] = -1
            elif max(y) > max_count:
                pass
            elif y in idx:
                adj[j][k] = idx[y]
            else:
                state.append(y)
                idx[y] = len(state)
                adj.append([0] * nr)
                adj[j][k] = len(state)
    for i in range(n0, len(state)):
        for k in range(nr):
            y = tuple(state[i][s] + int(nu[k, s]) for s in range(ns))
            adj[i][k] = -1 if min(y) < 0 else (0 if max(y) > max_count else idx.get(y, 0))
    return np.array(state, dtype=np.int32).reshape(-1, ns), np.array(adj, dtype=np.int32).reshape(-1, nr)


def test_random_networks_against_the_plain_restatement():
    """300 random cases: 1-4 species, 1-6 reactions with entries in [-2, 2] (duplicates and null reactions included),
    a random subset of a small box as the listed states in random order, a random part of the correct links already
    made and the rest open, a population cap that cuts some targets off"""
    from krylovfspssa_amd import KfspContext
    rng = np.random.default_rng(77)
    with KfspContext(0) as c:
        for case in range(300):
            ns, nr = int(rng.integers(1, 5)), int(rng.integers(1, 7))
            nu = rng.integers(-2, 3, size=(nr, ns)).astype(np.int32)
            side = int(rng.integers(2, 6))
            box = np.array(np.meshgrid(*[np.arange(side)] * ns, indexing="ij")).reshape(ns, -1).T
            keep = rng.random(len(box)) < rng.uniform(0.2, 1.0)
            keep[int(rng.integers(0, len(box)))] = True
            state = box[keep][rng.permutation(int(keep.sum()))].astype(np.int32)
            max_count = int(rng.integers(side - 1, side + 3))
            idx = {tuple(s): i + 1 for i, s in enumerate(state.tolist())}
            adj = np.zeros((len(state), nr), dtype=np.int32)
            for j, s in enumerate(state):
                for k in range(nr):
                    if rng.random() < 0.5:                # this link is made already (correctly), the others are open
                        y = s + nu[k]
                        adj[j, k] = -1 if y.min() < 0 else (0 if y.max() > max_count else idx.get(tuple(y.tolist()), 0))
            want_s, want_a = _onestep_py(nu, state, adj, max_count)
            got_s, got_a = c.onestep(nu, state, adj, max_count=max_count)
            assert np.array_equal(got_s, want_s), case
            assert np.array_equal(got_a, want_a), case
