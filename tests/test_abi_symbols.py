"""CPU-side checks of the drop-in boundary: the shared library loads and exports
every entry point include/kfsp.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "kfsp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kfsp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from krylovfspssa_amd import build, host
    build.build_lib()          # hipcc cross-compiles gfx950 without a GPU
    return host.load_library()


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 25
    raw = ctypes.CDLL(lib._name)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_abi_version_and_host_only_entry_points(lib):
    assert lib.kfsp_abi_version() >= 1
    # kfsp_padm is host code by design (the Hessenberg Pade stays on the host)
    from krylovfspssa_amd import host
    H = np.array([[-1.0, 0.5], [1.0, -0.5]])
    E, ns, hn = host.padm(H, 0.3)
    import scipy.linalg as sl
    assert np.abs(E - sl.expm(0.3 * H)).max() < 1e-14
    assert hn == pytest.approx(0.3 * 1.5)


def test_bad_arguments_are_reported_not_fatal(lib):
    assert lib.kfsp_create(0, None) == -2
    assert lib.kfsp_padm(6, 0, 1.0, None, 1, None, None, None) == -2
    assert lib.kfsp_comm_unique_id(None) == -1


def test_library_padm_matches_reference_dgpadm(lib, golden_dir):
    from krylovfspssa_amd import host
    g = np.load(os.path.join(golden_dir, "padm.npz"))
    for c in range(int(g["ncase"])):
        E, ns, _ = host.padm(g[f"H{c}"], float(g[f"t{c}"]))
        ref = g[f"E{c}"]
        assert ns == int(g[f"ns{c}"])
        assert np.abs(E - ref).max() <= 1e-13 * np.abs(ref).max()


def test_every_library_option_is_documented_in_the_header():
    """kfsp_set_option names (csrc/kfsp_api.cpp) all appear in include/kfsp.h."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "krylovfspssa_amd", "csrc", "kfsp_api.cpp")).read()
    hdr = open(os.path.join(root, "include", "kfsp.h")).read()
    names = re.findall(r'k == "([a-z_]+)"', src)
    assert len(names) >= 10
    assert not [n for n in names if f'"{n}"' not in hdr]
