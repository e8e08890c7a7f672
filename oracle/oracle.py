"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

ctypes front end of oracle/libkfsp_oracle.so (the plain-C restatement of the
reference hot path, oracle/kfsp_oracle.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkfsp_oracle.so")
_lib = None


class _Ell(C.Structure):
    _fields_ = [("n", C.c_int32), ("bw", C.c_int32), ("ld", C.c_int32),
                ("adj", C.c_void_p), ("offdiag", C.c_void_p), ("diag", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("nmult", C.c_int), ("nexph", C.c_int), ("nscale", C.c_int), ("nstep", C.c_int),
                ("nreject", C.c_int), ("ibrkflag", C.c_int), ("mbrkdwn", C.c_int),
                ("n_wsum", C.c_int), ("status", C.c_int), ("t_now", C.c_double)]


def build():
    """Compile the C restatement (gcc only; needs no reference and no GPU)."""
    subprocess.run(["make", "-s", "-C", _HERE, "libkfsp_oracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.kfo_dot.restype = C.c_double
        _lib.kfo_nrm2.restype = C.c_double
        _lib.kfo_asum.restype = C.c_double
        _lib.kfo_ell_count_nnz.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class EllMatrix:
    """The reference's FSP_MATRIX arrays (StateSpace.f90:13-17) as numpy arrays
    shaped [state][slot] (= Fortran (slot, state) column-major)."""

    def __init__(self, adj, offdiag, diag):
        self.adj = np.ascontiguousarray(adj, dtype=np.int32)
        self.offdiag = np.ascontiguousarray(offdiag, dtype=np.float64)
        self.diag = np.ascontiguousarray(diag, dtype=np.float64)
        self.n, self.bw = self.adj.shape
        assert self.offdiag.shape == (self.n, self.bw) and self.diag.shape == (self.n,)
        self._s = _Ell(self.n, self.bw, self.bw, _p(self.adj), _p(self.offdiag), _p(self.diag))

    @property
    def ref(self):
        return C.byref(self._s)

    def nnz(self):
        return int(lib().kfo_ell_count_nnz(self.ref))

    def to_csr(self):
        nnz = self.nnz()
        rowptr = np.empty(self.n + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=np.float64)
        lib().kfo_ell_to_csr(self.ref, _p(rowptr), _p(col), _p(val))
        return rowptr, col, val


def spmv_ell(A, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.n, dtype=np.float64)
    lib().kfo_spmv_ell(A.ref, _p(x), _p(y))
    return y


def spmv_csr(rowptr, col, val, x):
    n = len(rowptr) - 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(n, dtype=np.float64)
    lib().kfo_spmv_csr(C.c_int(n), _p(rowptr), _p(col), _p(val), _p(x), _p(y))
    return y


def padm(H, t, ideg=6):
    """exp(t*H) -> (E, ns, hnorm)."""
    Hf = np.asfortranarray(H, dtype=np.float64)
    m = Hf.shape[0]
    E = np.empty((m, m), dtype=np.float64, order="F")
    ns = C.c_int(0)
    hn = C.c_double(0.0)
    rc = lib().kfo_padm(C.c_int(ideg), C.c_int(m), C.c_double(t), _p(Hf), C.c_int(m), _p(E),
                        C.byref(ns), C.byref(hn))
    if rc:
        raise RuntimeError(f"kfo_padm failed: {rc}")
    return E, ns.value, hn.value


def arnoldi(A, v1, m, qiop=2, break_tol=1e-7):
    """One IOP pass from the unit start vector v1 -> (V[n,m+2] F-order, H[m+2,m+2],
    mbrkdwn, k1, avnorm)."""
    n = A.n
    V = np.zeros((n, m + 2), dtype=np.float64, order="F")
    V[:, 0] = v1
    mh = m + 2
    H = np.zeros((mh, mh), dtype=np.float64, order="F")
    av = C.c_double(0.0)
    k1 = C.c_int(0)
    nm = C.c_int(0)
    mb = lib().kfo_arnoldi(A.ref, C.c_int(m), C.c_int(1), C.c_int(qiop), C.c_double(break_tol),
                           _p(V), _p(H), C.c_int(mh), C.byref(av), C.byref(k1), C.byref(nm))
    return V, H, mb, k1.value, av.value


def expv_fixed(A, w0, m, tau, nsteps):
    w = np.array(w0, dtype=np.float64, copy=True)
    ws = np.zeros(nsteps, dtype=np.float64)
    rc = lib().kfo_expv_fixed(A.ref, C.c_int(m), C.c_double(tau), C.c_int(nsteps), _p(w), _p(ws))
    if rc:
        raise RuntimeError(f"kfo_expv_fixed failed: {rc}")
    return w, ws


def dgexpv_fixed_fsp(A, t, v, fsptol, krytol, max_log=4096):
    v = np.ascontiguousarray(v, dtype=np.float64)
    w = np.zeros(A.n, dtype=np.float64)
    st = Stats()
    ltau = np.zeros(max_log)
    lm = np.zeros(max_log, dtype=np.int32)
    lws = np.zeros(max_log)
    lib().kfo_dgexpv_fixed_fsp(A.ref, C.c_double(t), _p(v), _p(w), C.c_double(fsptol),
                               C.c_double(krytol), C.byref(st), C.c_int(max_log),
                               _p(ltau), _p(lm), _p(lws))
    ns = min(st.nstep, max_log)
    nw = min(st.n_wsum, max_log)
    return w, st, ltau[:ns].copy(), lm[:ns].copy(), lws[:nw].copy()
