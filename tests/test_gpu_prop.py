"""Propensities on the device (kfsp_set_propensity_program / kfsp_propensities / kfsp_onestep_columns, SURVEY.md 8(f)
rank 4): the postfix program of the model's .input expressions (ModelModule.f90:163-199 through the stack machine of
FortranParser.f90:187-302) evaluated for whole lists of states.

Pinned against tests/golden/exprtable.npz - the table the REFERENCE's parser gives for tests/golden/models/
expr_test_model.input (16 expressions covering every operator class, all 14 functions, x/0 and log(<= 0)) on a
13 x 13 x 3 grid.  Tolerances, as the header states them: expressions built from + - * / (and negation) are
bit-exact; an expression of ONE species is bit-exact whatever it contains (it travels as a table made by the host's
own evaluator); pow / exp / log / trigonometric functions of several species come from the device's math library
and agree with the host's to <= 4 ulp."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

FDIR = os.path.join(ROOT, "krylovfspssa_amd", "fortran")
REPLAY = os.path.join(FDIR, "_build", "kfsp_replay")
MODELS = os.path.join(GOLDEN, "models")
EPS = np.finfo(np.float64).eps

# reactions (0-based) of expr_test_model.input by what they are made of
EXACT_OPS = [0, 1, 2, 8, 10, 13, 14]             # + - * / only (incl. the x / 0 rule of reaction 15)
ONE_SPECIES = [2, 4, 8, 9, 12, 15]               # depend on one species: tabulated by the host when tables are on
CONSTANT = [0, 1]


def _table(tmp_path, mode):
    if not os.path.exists(REPLAY):
        from krylovfspssa_amd import build
        build.build_lib()
        subprocess.run(["make", "-s", "-C", FDIR, "_build/kfsp_replay"], check=True)
    p = str(tmp_path / f"prop_{mode}.bin")
    out = subprocess.run([REPLAY, "proptable", p, mode], cwd=MODELS, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    raw = np.fromfile(p)
    n = 13 * 13 * 3
    return raw[:16 * n].reshape(13, 13, 3, 16), raw[16 * n:].reshape(13, 13, 3)


def _ulps(a, b):
    return np.abs(a - b) / (EPS * np.maximum(np.abs(b), np.finfo(np.float64).tiny))


@pytest.mark.parametrize("mode", ["tab", "notab"])
def test_device_propensities_match_the_reference_parser(tmp_path, mode):
    P, D = _table(tmp_path, mode)
    G = np.load(os.path.join(GOLDEN, "exprtable.npz"))["P"]
    exact = set(EXACT_OPS) | set(CONSTANT) | (set(ONE_SPECIES) if mode == "tab" else set())
    worst = 0.0
    for r in range(16):
        if r in exact:
            assert np.array_equal(P[..., r], G[..., r]), f"reaction {r + 1} ({mode})"
        else:
            u = _ulps(P[..., r], G[..., r])
            # the rules that zero a whole expression (x / 0, log of x <= 0) must agree exactly
            assert np.array_equal(P[..., r] == 0.0, G[..., r] == 0.0), f"reaction {r + 1}"
            assert u.max() <= 4.0, f"reaction {r + 1} ({mode}): {u.max()} ulp"
            worst = max(worst, float(u.max()))
    print(f"{mode}: worst difference of the library-function expressions {worst:.2f} ulp")
    # DIAG = the propensities added in reaction order (ADD_STATE, StateSpace.f90:207-212)
    want = np.zeros_like(D)
    for r in range(16):
        want = want + P[..., r]
    assert np.array_equal(D, want)


def test_program_through_the_c_abi():
    """hand-assembled programs: mass action c X Y (exact), x / 0 -> 0, a table beside the code, and the
    checks kfsp_set_propensity_program makes before anything reaches the device"""
    from krylovfspssa_amd import KfspContext, KfspError
    IMM, NEG, ADD, SUB, MUL, DIV, POW = 1, 2, 3, 4, 5, 6, 7
    X, Y, C0 = 101, 102, 103                               # two species, one parameter
    progs = [([C0, X, MUL, Y, MUL], []),                   # c * X * Y
             ([X, Y, IMM, SUB, DIV], [3.0]),               # X / (Y - 3)
             ([X, X, IMM, SUB, MUL, IMM, DIV, C0, MUL], [1.0, 2.0]),     # X (X - 1) / 2 * c
             ([X, IMM, POW], [2.5])]                       # X ** 2.5 (device pow)
    st = np.array([[x, y] for x in range(0, 40, 3) for y in range(0, 7)], dtype=np.int32)
    c0 = 0.0199264663575241
    xf, yf = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
    with KfspContext(0) as c:
        c.set_propensity_program(2, [c0], progs)
        off, diag = c.propensities(st)
        assert np.array_equal(off[:, 0], c0 * xf * yf)
        with np.errstate(divide="ignore", invalid="ignore"):
            want = np.where(yf - 3.0 == 0.0, 0.0, xf / (yf - 3.0))
        assert np.array_equal(off[:, 1], want)
        assert np.array_equal(off[:, 2], xf * (xf - 1.0) / 2.0 * c0)
        assert _ulps(off[:, 3], xf ** 2.5).max() <= 4.0
        assert np.array_equal(diag, ((off[:, 0] + off[:, 1]) + off[:, 2]) + off[:, 3])
        # the fourth propensity as a host-made table: now the host's bits
        tab = np.zeros((4, 64))
        tab[3] = np.arange(64.0) ** 2.5
        c.set_propensity_program(2, [c0], progs, tables=([-1, -1, -1, 0], tab))
        off2, _ = c.propensities(st)
        assert np.array_equal(off2[:, 3], xf ** 2.5) and np.array_equal(off2[:, :3], off[:, :3])
        for bad in ([([ADD], [])], [([X, 99], [])], [([IMM], [])], [([X] * 40, [])]):
            with pytest.raises(KfspError):
                c.set_propensity_program(2, [c0], bad)
        with pytest.raises(KfspError):
            c.propensities(st)                            # a rejected program leaves none behind
