"""ctypes mirror of the C ABI in include/kfsp.h (libkfsp_hip.so).

This is plumbing for tests, bench.py and Python drivers: numpy arrays in the
reference's own layouts go in, numpy arrays come out; all arithmetic of the hot
path happens in the HIP library.  There is no CPU fallback: importing works
anywhere (so the symbol table can be checked without a GPU), but creating a
context without a usable MI355X raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

M_MAX = 100
T_NAMES = ("arnoldi", "combine", "begin_step", "fsp_callbacks", "host_pade", "upload", "device_onestep")

_lib = None


class KfspError(RuntimeError):
    pass


def library_path():
    return _build.LIB


def load_library():
    """Load libkfsp_hip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise KfspError(f"{path} is missing: run `python -m krylovfspssa_amd.build` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    sig = {
        "kfsp_create": [C.c_int, C.POINTER(vp)],
        "kfsp_destroy": [vp],
        "kfsp_create_group": [C.c_int, vp, C.POINTER(vp)],
        "kfsp_group_size": [vp, C.POINTER(C.c_int)],
        "kfsp_abi_version": [],
        "kfsp_comm_unique_id": [vp],
        "kfsp_comm_init": [vp, C.c_int, C.c_int, vp],
        "kfsp_loopback_create": [C.c_int, C.POINTER(vp)],
        "kfsp_loopback_destroy": [vp],
        "kfsp_comm_init_loopback": [vp, vp, C.c_int],
        "kfsp_row_block": [vp, i64, C.POINTER(i64), C.POINTER(i64)],
        "kfsp_partition": [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
        "kfsp_set_matrix_ell": [vp, i32, i32, i32, vp, vp, vp],
        "kfsp_update_matrix_ell": [vp, i32, i32, i32, vp, vp, vp, i32],
        "kfsp_set_matrix_csr": [vp, i64, i64, i64, vp, vp, vp],
        "kfsp_set_state_coords": [vp, i32, i32, i32, vp],
        "kfsp_update_state_coords": [vp, i32, i32, i32, vp, i32],
        "kfsp_set_matrix_box": [vp, i32, vp, i32, vp, vp, vp, vp],
        "kfsp_state_order_active": [vp, C.POINTER(C.c_int)],
        "kfsp_matrix_info": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
        "kfsp_matrix_bytes": [vp, C.c_int, C.POINTER(i64)],
        "kfsp_num_states": [vp, C.POINTER(i64)],
        "kfsp_layout_info": [vp, vp],
        "kfsp_build_info": [vp, vp],
        "kfsp_set_trip_order": [vp, i64, vp],
        "kfsp_onestep": [vp, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, C.POINTER(i32), vp, vp],
        "kfsp_onestep_columns": [vp, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, C.POINTER(i32), vp, vp, vp, i32, vp],
        "kfsp_set_propensity_program": [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp],
        "kfsp_propensities": [vp, i32, vp, i32, vp, i32, vp],
        "kfsp_ssa_streams": [vp, dbl, i64, i32, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, C.POINTER(i32), vp, vp, i32, vp],
        "kfsp_drop_plan": [vp, dbl, C.POINTER(dbl), C.POINTER(i64), C.POINTER(i64)],
        "kfsp_drop_flags": [vp, i64, vp],
        "kfsp_drop_compact": [vp, C.POINTER(i64)],
        "kfsp_drop_rebuild": [vp],
        "kfsp_expand_resident": [vp, dbl, i64, i32, i32, vp, i32, i32, C.POINTER(i64), C.POINTER(i64)],
        "kfsp_download_fsp": [vp, i32, vp, i32, vp, vp, i32, vp],
        "kfsp_dgexpv": [vp, dbl, dbl, dbl, C.c_int, vp, vp],
        "kfsp_set_vector": [vp, i64, vp],
        "kfsp_get_vector": [vp, i64, vp],
        "kfsp_begin_step": [vp, C.POINTER(dbl)],
        "kfsp_arnoldi": [vp, C.c_int, C.c_int, C.c_int, dbl, vp, C.c_int, C.POINTER(C.c_int),
                         C.POINTER(C.c_int), C.POINTER(dbl)],
        "kfsp_combine": [vp, C.c_int, dbl, vp, C.POINTER(dbl)],
        "kfsp_restore_w": [vp, dbl],
        "kfsp_spmv": [vp, vp, vp],
        "kfsp_spmv_w": [vp, vp],
        "kfsp_nrm2_w": [vp, C.POINTER(dbl)],
        "kfsp_asum_w": [vp, C.POINTER(dbl)],
        "kfsp_get_basis": [vp, C.c_int, i64, vp],
        "kfsp_padm": [C.c_int, C.c_int, dbl, vp, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(dbl)],
        "kfsp_expv_fixed": [vp, C.c_int, dbl, C.c_int, vp],
        "kfsp_spmv_bench": [vp, C.c_int, C.c_int, C.POINTER(C.c_float)],
        "kfsp_selftest_stream": [vp, i64, C.c_int, C.c_int, C.POINTER(C.c_float)],
        "kfsp_exchange_bench": [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(i64)],
        "kfsp_add_timer": [vp, C.c_int, dbl],
        "kfsp_get_timers": [vp, vp, C.c_int],
        "kfsp_set_option": [vp, C.c_char_p, i64],
    }
    for name, args in sig.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    lib.kfsp_last_error.argtypes = [vp]
    lib.kfsp_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_DROP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.POINTER(C.c_int64))
_EXPAND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.POINTER(C.c_int64))
_LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int)

EV_BEGIN_IOP, EV_WSUM, EV_STEP, EV_REJECT_STEP, EV_DIM_CHANGE, EV_CALL_SSA = 1, 2, 3, 4, 5, 6


class FspOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("drop", _DROP_FN), ("expand", _EXPAND_FN), ("log", _LOG_FN)]


class SolveStats(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("nmult", "nexph", "nscale", "nstep", "nreject", "ibrkflag", "mbrkdwn",
                                         "n_wsum", "n_expand", "n_drop_calls")] + \
               [(k, C.c_double) for k in ("step_min", "step_max", "x_error", "s_error", "tbrkdwn", "t_now",
                                          "hump", "beta")]


def padm(H, t, ideg=6):
    """Host Pade exponential of the library (dgpadm.f:2-169 semantics)."""
    lib = load_library()
    Hf = np.asfortranarray(H, dtype=np.float64)
    m = Hf.shape[0]
    E = np.empty((m, m), dtype=np.float64, order="F")
    ns, hn = C.c_int(0), C.c_double(0.0)
    rc = lib.kfsp_padm(ideg, m, float(t), _p(Hf), m, _p(E), C.byref(ns), C.byref(hn))
    if rc:
        raise KfspError(f"kfsp_padm -> {rc}")
    return E, ns.value, hn.value


def partition(n, nranks, rank):
    """(row0, nrows, L): the contiguous row block of `rank` and the padded block
    length; global index of local row k of rank p is p*L + k (host arithmetic
    of the library, no GPU needed)."""
    r0, nr, L = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    rc = load_library().kfsp_partition(int(n), int(nranks), int(rank), C.byref(r0), C.byref(nr), C.byref(L))
    if rc:
        raise KfspError(f"kfsp_partition -> {rc}")
    return r0.value, nr.value, L.value


class LoopbackGroup:
    """kfsp_loopback_create: nranks contexts of this process exchanging through device copies
    (one-GPU rehearsal of the row partition; see include/kfsp.h)."""

    def __init__(self, nranks):
        self.handle = C.c_void_p()
        rc = load_library().kfsp_loopback_create(int(nranks), C.byref(self.handle))
        if rc:
            raise KfspError(f"kfsp_loopback_create -> {rc}")
        self.nranks = int(nranks)

    def close(self):
        if self.handle:
            load_library().kfsp_loopback_destroy(self.handle)
            self.handle = C.c_void_p()


def run_loopback_ranks(nranks, body, device=0):
    """body(ctx, rank) on nranks contexts of one loop-back group, one thread per rank (ctypes
    releases the GIL inside the library); returns the list of results, re-raises the first error."""
    import threading
    group = LoopbackGroup(nranks)
    results, errors = [None] * nranks, [None] * nranks

    def work(rank):
        try:
            with KfspContext(device) as ctx:
                ctx.comm_init_loopback(group, rank)
                results[rank] = body(ctx, rank)
        except BaseException as e:   # noqa: BLE001 - reported to the caller below
            errors[rank] = e

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    group.close()
    for e in errors:
        if e is not None:
            raise e
    return results


class KfspContext:
    """One device context = one rank's share of the solver workspace."""

    def __init__(self, device=0, group=None):
        """group = P (int: P ranks, all on `device`, loop-back) or a list of device ids (distinct: RCCL):
        a head handle over a row partition driven by this one thread (kfsp_create_group); it behaves like
        a one-rank context with whole vectors in and out."""
        self._lib = load_library()
        self._h = C.c_void_p()
        if group is not None:
            devs = [int(device)] * int(group) if isinstance(group, int) else [int(d) for d in group]
            arr = (C.c_int * len(devs))(*devs)
            rc = self._lib.kfsp_create_group(len(devs), arr, C.byref(self._h))
            self.group_size = len(devs)
        else:
            rc = self._lib.kfsp_create(int(device), C.byref(self._h))
            self.group_size = 1
        if rc:
            self._h = C.c_void_p()
            raise KfspError(f"kfsp_create(device={device}, group={group}) -> {rc}: no usable HIP device; "
                            "the exp(tA)v hot path has no CPU fallback")
        self.n = 0
        self.nranks, self.rank = 1, 0

    # -- lifetime
    def close(self):
        if self._h:
            self._lib.kfsp_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            msg = self._lib.kfsp_last_error(self._h)
            raise KfspError(f"{what} -> {rc}: {msg.decode() if msg else ''}")

    # -- partition
    @staticmethod
    def unique_id():
        buf = np.zeros(128, dtype=np.uint8)
        rc = load_library().kfsp_comm_unique_id(_p(buf))
        if rc:
            raise KfspError(f"kfsp_comm_unique_id -> {rc}")
        return buf

    def comm_init(self, nranks, rank, id_bytes=None):
        buf = None if id_bytes is None else np.ascontiguousarray(id_bytes, dtype=np.uint8)
        self._chk(self._lib.kfsp_comm_init(self._h, nranks, rank, None if buf is None else _p(buf)), "kfsp_comm_init")
        self.nranks, self.rank = nranks, rank

    def comm_init_loopback(self, group, rank):
        """Join a LoopbackGroup (ranks = contexts of this process, one host thread each)."""
        self._chk(self._lib.kfsp_comm_init_loopback(self._h, group.handle, int(rank)), "kfsp_comm_init_loopback")
        self.nranks, self.rank = group.nranks, int(rank)

    def row_block(self, n):
        r0, nr = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.kfsp_row_block(self._h, int(n), C.byref(r0), C.byref(nr)), "kfsp_row_block")
        return r0.value, nr.value

    # -- generator
    def set_matrix_ell(self, adj, offdiag, diag):
        """FSP_MATRIX arrays as [state][slot] numpy arrays (= Fortran (slot,state))."""
        adj = np.ascontiguousarray(adj, dtype=np.int32)
        offdiag = np.ascontiguousarray(offdiag, dtype=np.float64)
        diag = np.ascontiguousarray(diag, dtype=np.float64)
        n, bw = adj.shape
        assert offdiag.shape == (n, bw) and diag.shape == (n,)
        self._chk(self._lib.kfsp_set_matrix_ell(self._h, n, bw, bw, _p(adj), _p(offdiag), _p(diag)),
                  "kfsp_set_matrix_ell")
        self.n = n
        self.row0, self.nloc = self.row_block(n)

    def update_matrix_ell(self, adj, offdiag, diag, n_unchanged):
        """set_matrix_ell after the FSP grew: OFFDIAG / DIAG of the first n_unchanged states stay on the device."""
        adj = np.ascontiguousarray(adj, dtype=np.int32)
        offdiag = np.ascontiguousarray(offdiag, dtype=np.float64)
        diag = np.ascontiguousarray(diag, dtype=np.float64)
        n, bw = adj.shape
        self._chk(self._lib.kfsp_update_matrix_ell(self._h, n, bw, bw, _p(adj), _p(offdiag), _p(diag), int(n_unchanged)),
                  "kfsp_update_matrix_ell")
        self.n = n
        self.row0, self.nloc = self.row_block(n)

    def set_state_coords(self, state):
        """FSP%STATE as a [state][species] array, for the next set_matrix_ell of the
        same size (internal lexicographic state order, include/kfsp.h)."""
        state = np.ascontiguousarray(state, dtype=np.int32)
        n, ns = state.shape
        self._chk(self._lib.kfsp_set_state_coords(self._h, n, ns, ns, _p(state)), "kfsp_set_state_coords")

    def update_state_coords(self, state, n_unchanged):
        """set_state_coords after the FSP grew: only the coordinates behind the first n_unchanged states travel"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        n, ns = state.shape
        self._chk(self._lib.kfsp_update_state_coords(self._h, n, ns, ns, _p(state), int(n_unchanged)), "kfsp_update_state_coords")

    def state_order_active(self):
        a = C.c_int(0)
        self._chk(self._lib.kfsp_state_order_active(self._h, C.byref(a)), "kfsp_state_order_active")
        return bool(a.value)

    def set_matrix_csr(self, n, rowptr, col, val, row0=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        r0, nr = self.row_block(n)
        if row0 is None:
            row0 = r0
        assert len(rowptr) == nr + 1, (len(rowptr), nr)
        self._chk(self._lib.kfsp_set_matrix_csr(self._h, int(n), int(row0), int(nr), _p(rowptr), _p(col), _p(val)),
                  "kfsp_set_matrix_csr")
        self.n = int(n)
        self.row0, self.nloc = r0, nr

    def set_matrix_box(self, model, store=None):
        """Matrix-free generator of a synth.BoxModel with separable propensities (model.factors()).
        store=True: the device writes the same generator out as stored diagonals (option box_store)."""
        if store is not None:
            self.set_option("box_store", 1 if store else 0)
        dims = np.ascontiguousarray(model.dims, dtype=np.int32)
        stoich = np.ascontiguousarray(np.asarray(model.stoich).T, dtype=np.int32)      # [nr][ns]
        ndep, deps, tables = model.factors()
        ndep = np.ascontiguousarray(ndep, dtype=np.int32)
        deps = np.ascontiguousarray(deps, dtype=np.int32)
        tables = np.ascontiguousarray(tables, dtype=np.float64)
        self._chk(self._lib.kfsp_set_matrix_box(self._h, len(dims), _p(dims), stoich.shape[0], _p(stoich), _p(ndep),
                                                _p(deps), _p(tables)), "kfsp_set_matrix_box")
        self.n = int(model.n)
        self.row0, self.nloc = self.row_block(self.n)

    def matrix_info(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.kfsp_matrix_info(self._h, C.byref(a), C.byref(b), C.byref(c)), "kfsp_matrix_info")
        return dict(rows=a.value, slots=b.value, nnz=c.value)

    # -- vectors
    def onestep(self, stoich, state, adj, max_count=10000, capacity=None):
        """ONESTEP_EXTENDER's integer work on the device: (state [n][ns], adj [n][nr]) -> (state', adj')
        of the extended FSP (new states appended in the reference's order, all links completed)."""
        state = np.ascontiguousarray(state, dtype=np.int32)
        adj = np.ascontiguousarray(adj, dtype=np.int32)
        stoich = np.ascontiguousarray(stoich, dtype=np.int32)          # [nr][ns]
        n, ns = state.shape
        nr = adj.shape[1]
        assert stoich.shape == (nr, ns)
        cap = int(capacity) if capacity else n * (nr + 1) + 16
        n_new = C.c_int32(0)
        st_new = np.zeros((cap - n + 1, ns), dtype=np.int32)
        adj_out = np.zeros((cap, nr), dtype=np.int32)
        self._chk(self._lib.kfsp_onestep(self._h, ns, nr, _p(stoich), n, _p(state), ns, _p(adj), nr, int(max_count), cap,
                                         C.byref(n_new), _p(st_new), _p(adj_out)), "kfsp_onestep")
        m = n_new.value
        return np.concatenate([state, st_new[:m - n]]), adj_out[:m].copy()

    def onestep_columns(self, stoich, state, adj, max_count=10000, capacity=None):
        """onestep that also returns the propensity columns (offdiag [m][nr], diag [m]) of the appended states"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        adj = np.ascontiguousarray(adj, dtype=np.int32)
        stoich = np.ascontiguousarray(stoich, dtype=np.int32)
        n, ns = state.shape
        nr = adj.shape[1]
        cap = int(capacity) if capacity else n * (nr + 1) + 16
        n_new = C.c_int32(0)
        st_new = np.zeros((cap - n + 1, ns), dtype=np.int32)
        adj_out = np.zeros((cap, nr), dtype=np.int32)
        off_new = np.zeros((cap - n + 1, nr))
        diag_new = np.zeros(cap - n + 1)
        self._chk(self._lib.kfsp_onestep_columns(self._h, ns, nr, _p(stoich), n, _p(state), ns, _p(adj), nr, int(max_count), cap,
                                                 C.byref(n_new), _p(st_new), _p(adj_out), _p(off_new), nr, _p(diag_new)),
                  "kfsp_onestep_columns")
        m = n_new.value
        return np.concatenate([state, st_new[:m - n]]), adj_out[:m].copy(), off_new[:m - n].copy(), diag_new[:m - n].copy()

    def ssa_streams(self, timestep, seedmix, stoich, state, adj, offdiag, diag, max_count=10000, capacity_new=None):
        """the independent-stream SSA walk: (new states [m][ns], offdiag [m][nr], diag [m]) in order of first occurrence"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        adj = np.ascontiguousarray(adj, dtype=np.int32)
        offdiag = np.ascontiguousarray(offdiag, dtype=np.float64)
        diag = np.ascontiguousarray(diag, dtype=np.float64)
        stoich = np.ascontiguousarray(stoich, dtype=np.int32)
        n, ns = state.shape
        nr = adj.shape[1]
        cap = int(capacity_new) if capacity_new else 4 * n + 4096
        nf = C.c_int32(0)
        st_new = np.zeros((cap, ns), dtype=np.int32)
        off_new = np.zeros((cap, nr))
        diag_new = np.zeros(cap)
        self._chk(self._lib.kfsp_ssa_streams(self._h, float(timestep), int(seedmix), ns, nr, _p(stoich), n, _p(state), ns, _p(adj),
                                             _p(offdiag), nr, _p(diag), int(max_count), cap, C.byref(nf), _p(st_new), _p(off_new), nr,
                                             _p(diag_new)), "kfsp_ssa_streams")
        m = nf.value
        return st_new[:m].copy(), off_new[:m].copy(), diag_new[:m].copy()

    def expand_resident(self, t_ssa, seedmix, stoich, max_count=10000, capacity=None):
        """SSA walk (t_ssa > 0) + one-step sweep on the resident lists; returns (n_new, n_from_ssa)"""
        stoich = np.ascontiguousarray(stoich, dtype=np.int32)
        nr, ns = stoich.shape
        cap = int(capacity) if capacity else 2 ** 31 - 2
        n_new, n_ssa = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.kfsp_expand_resident(self._h, float(t_ssa), int(seedmix), ns, nr, _p(stoich), int(max_count), cap,
                                                 C.byref(n_new), C.byref(n_ssa)), "kfsp_expand_resident")
        self.n = n_new.value
        self.row0, self.nloc = self.row_block(self.n)
        return n_new.value, n_ssa.value

    def download_fsp(self, ns, nr):
        """the resident lists: (state [n][ns], adj [n][nr], offdiag [n][nr], diag [n])"""
        n = self.n
        state = np.zeros((n, ns), dtype=np.int32)
        adj = np.zeros((n, nr), dtype=np.int32)
        off = np.zeros((n, nr))
        diag = np.zeros(n)
        self._chk(self._lib.kfsp_download_fsp(self._h, n, _p(state), ns, _p(adj), _p(off), nr, _p(diag)), "kfsp_download_fsp")
        return state, adj, off, diag

    def set_propensity_program(self, ns, params, programs, tables=None):
        """programs: per reaction (code list, immediates list); tables: None or (tab_species [nr], tab [nr][tab_len])"""
        nr = len(programs)
        code_off = np.concatenate(([0], np.cumsum([len(c) for c, _ in programs]))).astype(np.int32)
        imm_off = np.concatenate(([0], np.cumsum([len(i) for _, i in programs]))).astype(np.int32)
        code = np.array([v for c, _ in programs for v in c] or [0], dtype=np.int32)
        imm = np.array([v for _, i in programs for v in i] or [0.0], dtype=np.float64)
        params = np.ascontiguousarray(params, dtype=np.float64)
        if tables is None:
            ts, tab, tl = np.full(nr, -1, dtype=np.int32), np.zeros(1), 0
        else:
            ts = np.ascontiguousarray(tables[0], dtype=np.int32)
            tab = np.ascontiguousarray(tables[1], dtype=np.float64)
            tl = tab.shape[1]
        self._chk(self._lib.kfsp_set_propensity_program(self._h, int(ns), nr, len(params), _p(params) if len(params) else None,
                                                        _p(code_off), _p(code), _p(imm_off), _p(imm), _p(ts), int(tl), _p(tab)),
                  "kfsp_set_propensity_program")
        self._prop_nr = nr

    def propensities(self, state):
        state = np.ascontiguousarray(state, dtype=np.int32)
        n, ns = state.shape
        off = np.zeros((n, self._prop_nr))
        diag = np.zeros(n)
        self._chk(self._lib.kfsp_propensities(self._h, n, _p(state), ns, _p(off), self._prop_nr, _p(diag)), "kfsp_propensities")
        return off, diag

    def drop_plan(self, dsum):
        """DROP_STATES decision on the device -> (droptol, drop_count, n_flagged)."""
        tol, cnt, nf = C.c_double(0.0), C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.kfsp_drop_plan(self._h, float(dsum), C.byref(tol), C.byref(cnt), C.byref(nf)), "kfsp_drop_plan")
        return tol.value, cnt.value, nf.value

    def drop_flags(self):
        f = np.zeros(self.n, dtype=np.uint8)
        self._chk(self._lib.kfsp_drop_flags(self._h, int(self.n), _p(f)), "kfsp_drop_flags")
        return f

    def drop_compact(self):
        n = C.c_int64(0)
        self._chk(self._lib.kfsp_drop_compact(self._h, C.byref(n)), "kfsp_drop_compact")
        return n.value

    def drop_rebuild(self):
        """the generator of the compacted FSP from the device's own arrays (after drop_compact)"""
        self._chk(self._lib.kfsp_drop_rebuild(self._h), "kfsp_drop_rebuild")
        n = C.c_int64(0)
        self._lib.kfsp_num_states(self._h, C.byref(n))
        self.n = n.value
        self.row0, self.nloc = self.row_block(self.n)

    def matrix_bytes(self, force_sell=False):
        """force_sell: False / True (the SELL image) / 3 (the SELL image read with plain columns)"""
        b = C.c_int64(0)
        self._chk(self._lib.kfsp_matrix_bytes(self._h, int(force_sell), C.byref(b)), "kfsp_matrix_bytes")
        return b.value

    def set_trip_order(self, order):
        """order of the product's wavefront trips (None: ascending)"""
        if order is None:
            self._chk(self._lib.kfsp_set_trip_order(self._h, 0, None), "kfsp_set_trip_order")
            return
        order = np.ascontiguousarray(order, dtype=np.int32)
        self._chk(self._lib.kfsp_set_trip_order(self._h, len(order), _p(order)), "kfsp_set_trip_order")

    def layout_info(self):
        v = np.zeros(8, dtype=np.int64)
        self._chk(self._lib.kfsp_layout_info(self._h, _p(v)), "kfsp_layout_info")
        keys = ("format", "exchange", "halo_rows", "sell_reach", "coded_chunks", "chunks", "code_words", "state_order")
        return dict(zip(keys, (int(x) for x in v)))

    def build_info(self):
        """how the rebuilds of a resident FSP went (kfsp_build_info)"""
        v = np.zeros(6, dtype=np.int64)
        self._chk(self._lib.kfsp_build_info(self._h, _p(v)), "kfsp_build_info")
        return dict(zip(("speculative", "repeated", "sell", "key_layout_cached", "orders_carried_over"), (int(x) for x in v)))

    def set_vector(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        self._chk(self._lib.kfsp_set_vector(self._h, len(w), _p(w)), "kfsp_set_vector")

    def get_vector(self):
        w = np.empty(self.nloc, dtype=np.float64)
        self._chk(self._lib.kfsp_get_vector(self._h, self.nloc, _p(w)), "kfsp_get_vector")
        return w

    def get_basis(self, j):
        v = np.empty(self.nloc, dtype=np.float64)
        self._chk(self._lib.kfsp_get_basis(self._h, int(j), self.nloc, _p(v)), "kfsp_get_basis")
        return v

    # -- hot path
    def begin_step(self):
        b = C.c_double(0.0)
        self._chk(self._lib.kfsp_begin_step(self._h, C.byref(b)), "kfsp_begin_step")
        return b.value

    def arnoldi(self, m, jold=1, qiop=2, break_tol=1e-7, H=None):
        """-> (H[(m+2),(m+2)] F-order, mbrkdwn, k1, avnorm)."""
        mh = m + 2
        if H is None:
            H = np.zeros((mh, mh), dtype=np.float64, order="F")
        assert H.flags.f_contiguous and H.shape[0] >= mh
        mb, k1, av = C.c_int(0), C.c_int(0), C.c_double(0.0)
        self._chk(self._lib.kfsp_arnoldi(self._h, m, jold, qiop, float(break_tol), _p(H), H.shape[0],
                                         C.byref(mb), C.byref(k1), C.byref(av)), "kfsp_arnoldi")
        return H, mb.value, k1.value, av.value

    def combine(self, mx, beta, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert len(y) >= mx
        ws = C.c_double(0.0)
        self._chk(self._lib.kfsp_combine(self._h, int(mx), float(beta), _p(y), C.byref(ws)), "kfsp_combine")
        return ws.value

    def restore_w(self, beta):
        self._chk(self._lib.kfsp_restore_w(self._h, float(beta)), "kfsp_restore_w")

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert len(x) == self.n
        y = np.empty(self.nloc, dtype=np.float64)
        self._chk(self._lib.kfsp_spmv(self._h, _p(x), _p(y)), "kfsp_spmv")
        return y

    def spmv_w(self):
        y = np.empty(self.nloc, dtype=np.float64)
        self._chk(self._lib.kfsp_spmv_w(self._h, _p(y)), "kfsp_spmv_w")
        return y

    def nrm2_w(self):
        o = C.c_double(0.0)
        self._chk(self._lib.kfsp_nrm2_w(self._h, C.byref(o)), "kfsp_nrm2_w")
        return o.value

    def asum_w(self):
        o = C.c_double(0.0)
        self._chk(self._lib.kfsp_asum_w(self._h, C.byref(o)), "kfsp_asum_w")
        return o.value

    def dgexpv(self, t, fsptol, krytol, n_reactions, drop=None, expand=None):
        """The adaptive solver (DGEXPV_FSP, KrylovSolver.f90:151-573) on the resident
        generator and vector.  drop(dsum) / expand(t_ssa) are Python callables that
        return the new number of states (they re-upload through this context).
        -> (return code, SolveStats, log) with log = list of (event, values)."""
        log = []

        def _log(_u, ev, vals, n):
            log.append((ev, [vals[i] for i in range(n)]))

        def _wrap(fn):
            def cb(_u, x, n_new):
                try:
                    r = fn(x)
                    if r is not None:
                        n_new[0] = int(r)
                    return 0
                except Exception as e:      # never let an exception cross the C boundary
                    log.append((-1, repr(e)))
                    return 4000
            return cb
        ops = FspOps(None, _DROP_FN(_wrap(drop)) if drop else _DROP_FN(), _EXPAND_FN(_wrap(expand)) if expand else _EXPAND_FN(),
                     _LOG_FN(_log))
        st = SolveStats()
        rc = self._lib.kfsp_dgexpv(self._h, float(t), float(fsptol), float(krytol), int(n_reactions),
                                   C.byref(ops), C.byref(st))
        if rc not in (0, 10):
            self._chk(rc, "kfsp_dgexpv")
        n = C.c_int64(0)
        self._lib.kfsp_num_states(self._h, C.byref(n))
        self.n = n.value
        self.row0, self.nloc = self.row_block(self.n)
        return rc, st, log

    def expv_fixed(self, m, tau, nsteps):
        ws = np.zeros(max(nsteps, 1), dtype=np.float64)
        self._chk(self._lib.kfsp_expv_fixed(self._h, int(m), float(tau), int(nsteps), _p(ws)), "kfsp_expv_fixed")
        return ws[:nsteps]

    def spmv_bench(self, reps, variant=0):
        ms = C.c_float(0.0)
        self._chk(self._lib.kfsp_spmv_bench(self._h, int(reps), int(variant), C.byref(ms)), "kfsp_spmv_bench")
        return ms.value

    def exchange_bench(self, reps):
        """-> (ms for reps exchanges of the source vector alone, bytes this rank receives per exchange)"""
        ms, b = C.c_float(0.0), C.c_int64(0)
        self._chk(self._lib.kfsp_exchange_bench(self._h, int(reps), C.byref(ms), C.byref(b)), "kfsp_exchange_bench")
        return ms.value, b.value

    def selftest_stream(self, nbytes, elem_bytes, reps=1):
        ms = C.c_float(0.0)
        self._chk(self._lib.kfsp_selftest_stream(self._h, int(nbytes), int(elem_bytes), int(reps), C.byref(ms)),
                  "kfsp_selftest_stream")
        return ms.value

    def timers(self, reset=False):
        t = np.zeros(len(T_NAMES))
        self._chk(self._lib.kfsp_get_timers(self._h, _p(t), int(reset)), "kfsp_get_timers")
        return dict(zip(T_NAMES, t.tolist()))

    def set_option(self, name, value):
        self._chk(self._lib.kfsp_set_option(self._h, name.encode(), int(value)), "kfsp_set_option")
