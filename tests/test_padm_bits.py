"""kfsp_padm's bits are pinned: the adaptive solver's step-size and dimension decisions hang on entries of exp(tau H)
(KrylovSolver.f90:290-305), and the lock-step / exact-trajectory tests were recorded with them.  The products inside come
in three builds (baseline, 256-bit and 512-bit fused multiply-add) that must give the same bits on a host with FMA;
this regression pin (digests made by the library itself, round 3) catches a change of the summation order or of what the
compiler fuses - which product of "a b + c d" that is, is the compiler's choice.  Parity with the reference's DGPADM is
tests/test_abi_symbols.py (to tolerance); this file is about reproducibility."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

PIN = os.path.join(GOLDEN, "padm_bits.json")


def _cases():
    rng = np.random.default_rng(20261005)
    for trial in range(24):
        m = int(rng.integers(3, 104))
        H = np.zeros((m, m), order="F")
        for i in range(m):
            for j in range(max(0, i - 1), min(m, i + 3)):
                H[i, j] = rng.standard_normal() * 10 ** rng.uniform(0, 4)
        if trial % 3 == 0:
            H = np.asfortranarray(rng.standard_normal((m, m)) * 100)
        yield m, 10 ** rng.uniform(-4, -1), H


def _digests(lib):
    out = []
    for m, t, H in _cases():
        E = np.zeros((m, m), order="F")
        ns, hn = C.c_int(0), C.c_double(0)
        assert lib.kfsp_padm(6, m, t, H.ctypes.data, m, E.ctypes.data, C.byref(ns), C.byref(hn)) == 0
        out.append(hashlib.sha256(E.tobytes(order="F")).hexdigest()[:16])
    return out


def test_padm_bits_are_reproducible():
    from krylovfspssa_amd import host
    lib = host.load_library()
    lib.kfsp_padm.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    with open("/proc/cpuinfo") as f:
        flags = f.read()
    if " fma" not in flags or " avx2" not in flags:
        pytest.skip("host without fused multiply-add: the baseline build rounds twice per product")
    got = _digests(lib)
    with open(PIN) as f:
        want = json.load(f)
    assert got == want
