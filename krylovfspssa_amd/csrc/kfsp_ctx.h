// Private: the device context behind the opaque kfsp_ctx handle, shared by the
// host translation units of libkfsp_hip.
#pragma once

#include "../../include/kfsp.h"
#include "kfsp_internal.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace kfsp {

constexpr int kNumPartial = 8;   // rotating block-partial buffers: a kernel never writes one of the last 7 it may read
constexpr int kNumStage = 8;     // rotating all-reduce staging scalars

template <class T>
struct DevBuf {
    // kGuard elements in front of p and behind p + cap belong to the buffer and are zero: a kernel
    // that loads two neighbouring elements at once may touch the element before the first or behind
    // the last one it needs (the matrix-free product does) - readable, finite, never used.
    static constexpr size_t kGuard = 256 / sizeof(T);
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n, bool zero)
    {
        if (n <= cap) return hipSuccess;
        // The FSP grows a little at every expansion: take as much again (at most 2^27 elements more), so
        // that hipFree/hipMalloc (both synchronise the device, and fresh memory
        // costs the first kernel that touches it) happen O(log n) times - the resident Goutsias run went
        // through 527 hipFree / 574 hipMalloc / 1171 hipMemset calls with half as much again (rocprofv3 --hip-trace).
        if (cap > 0) n = std::max(n, cap + std::min(cap, (size_t)1 << 27));
        release();
        T *q = nullptr;
        hipError_t e = alloc(n, &q);
        if (e != hipSuccess) return e;
        p = q;
        cap = n;
        if (zero) e = hipMemset(p, 0, n * sizeof(T));
        return e;
    }
    // the same, but the first `keep` elements survive a reallocation
    hipError_t reserve_keep(size_t n, size_t keep, hipStream_t st)
    {
        if (n <= cap) return hipSuccess;
        if (keep == 0 || !p) return reserve(n, false);
        n = std::max(n, cap + std::min(cap, (size_t)1 << 27));
        T *q = nullptr;
        hipError_t e = alloc(n, &q);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(q, p, std::min(keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(p - kGuard);
        p = q;
        cap = n;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p - kGuard);
        p = nullptr;
        cap = 0;
    }

private:
    static hipError_t alloc(size_t n, T **out)
    {
        T *base = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), (n + 2 * kGuard) * sizeof(T));
        if (e != hipSuccess) return e;
        e = hipMemset(base, 0, kGuard * sizeof(T));
        if (e == hipSuccess) e = hipMemset(base + kGuard + n, 0, kGuard * sizeof(T));
        if (e != hipSuccess) {
            (void)hipFree(base);
            return e;
        }
        *out = base + kGuard;
        return hipSuccess;
    }
};

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// thresholds of one FIND_DROPTOL batch (StateSpace.f90:416-426: 1e-8, then /10 per sweep)
constexpr int kDropLevels = 16;
struct DropLevels {
    double tol[kDropLevels];
};

// Loop-back transport: the ranks of a row partition are contexts of ONE process, each
// driven by its own host thread, and "collectives" are host barriers around plain device
// copies.  Everything above the collective itself - packing the strips, dropping the
// neighbours' strips into the halo margins, the split interior / boundary launches, the
// staged scalars - is the code that runs over RCCL, so a one-GPU box exercises the
// rank > 0 branches.  Not a performance path.
struct LoopGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> slot;   // what each rank contributes to the collective in flight
    bool aborted = false;             // sticky: a peer failed or a deadline expired (abort()); every barrier returns false
    // false = a peer did not arrive within 120 s (its thread died) or the group was aborted: the caller reports an error
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&] { return aborted || generation != gen; });
        return ok && generation != gen;
    }
    // releases every rank waiting in a barrier (they report 2999); the transport is dead afterwards
    void abort()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            aborted = true;
        }
        cv.notify_all();
    }
};

}  // namespace kfsp

using kfsp::DevBuf;
using kfsp::kMaxDiag;

namespace kfsp {
struct Group;
}

struct kfsp_ctx {
    // a head handle over a row partition driven by ONE host thread (kfsp_create_group, kfsp_group.cpp): no device
    // resources of its own; every entry point fans out to the ranks' contexts
    kfsp::Group *group = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    // partition
    int nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    std::mutex comm_mu;                // held while a collective is ENQUEUED on comm, and by comm_abort (kfsp_api.cpp)
    bool comm_aborted = false;         // sticky (under comm_mu): the communicator was aborted, every collective returns 2999
    kfsp::LoopGroup *loop = nullptr;   // loop-back transport instead of RCCL (kfsp_comm_init_loopback)
    double *h_loop = nullptr;          // its pinned scratch
    bool use_comm = false;   // collectives on the data path (nranks > 1, or a 1-rank communicator for testing)
    // halo exchange instead of the full all-gather (banded generators only):
    // every basis column carries `margin` rows on either side that receive the
    // neighbours' boundary strips, so the product kernel reads x in place
    bool use_halo = false;
    int64_t halo = 0;        // rows needed from each neighbour = max |delta|, agreed by all ranks
    int64_t margin = 0;      // rows reserved on either side of every column (>= halo)
    DevBuf<double> d_strip;  // [2*halo] send + [nranks*2*halo] receive
    // the exchange runs on its own stream so that the interior rows of a product
    // (which read no halo) are computed while the strips travel
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_src = nullptr, ev_halo = nullptr;

    // sizes
    int64_t n = 0;        // global states
    int64_t L = 0;        // rows per rank (padded block length), multiple of 64
    int64_t row0 = 0;     // first global row of this rank
    int64_t nloc = 0;     // rows owned
    int64_t ldv = 0;      // column stride of the basis, multiple of 256; grows with head room, never shrinks
    bool relayout = false;   // the partition changed: re-lay (and clear) the basis with the next generator

    // generator
    DevBuf<int64_t> d_off;
    DevBuf<int32_t> d_col;
    DevBuf<double> d_val, d_diag;
    int64_t nchunks = 0, slots = 0, nnz = 0;
    // banded (DIA) form, used instead of SELL when the rows allow it
    DevBuf<double> d_dia;
    bool use_dia = false;
    int nd = 0;
    int32_t delta[kMaxDiag] = {0};
    int64_t dia_ld = 0;
    bool have_sell = false;
    // dictionary-coded columns of the SELL image (kernel format 5; kfsp_internal.h kSellCode*)
    DevBuf<int32_t> d_dtab, d_dtlen;
    DevBuf<unsigned long long> d_code;
    DevBuf<int64_t> d_codeoff;
    bool sell_coded = false;
    int64_t code_words = 0;          // 64-bit code words stored (all chunks)
    int64_t coded_chunks = 0;        // chunks that use them
    int64_t coded_slots = 0;         // entries (incl. padding) of those chunks
    int64_t coded_tab_bytes = 0;     // bytes of their offset tables, in 64-byte lines
    int64_t sell_reach = -1;         // max |col - row| over the local SELL rows (-1: unknown)
    // which 128-row groups of which diagonals hold entries at all (used when enough are empty)
    DevBuf<uint32_t> d_gmask;
    DevBuf<double> d_zero;   // 128 zeros, the stand-in for an empty segment
    bool dia_masked = false;
    // optional order of the product's wavefront trips (kfsp_set_trip_order / the box tiling): trips of the CURRENT generator
    DevBuf<int32_t> d_trip_order;
    int64_t trip_order_n = 0;          // 0: ascending order
    // matrix-free box generator (kfsp_set_matrix_box): descriptor (host copy, passed as a kernel
    // argument) and the factor tables in device memory
    kfsp::BoxDev box;
    DevBuf<double> d_box;
    bool use_box = false;
    bool box_fast = false;        // the single-factor form (BoxFast) sits behind the tables
    // format 7 (pencils along the slowest species, kfsp_kernels.hip): eligible box, rows per plane, planes, base trips
    // (= ceil(rows per plane / 128)) and their tiled order (0 entries: ascending)
    bool box_pencil = false, pencil_simple = false;
    // format 8 (slabs): rows per line of the second-slowest species, lines, wavefronts per workgroup, workgroups per line set
    bool box_slab = false;
    int64_t slab_line_rows = 0, slab_lo_trips = 0;
    int slab_lines = 0, slab_waves = 0, slab_groups = 0;
    int64_t pencil_plane_rows = 0, pencil_trips = 0, pencil_order_n = 0;
    int pencil_planes = 0;
    DevBuf<int32_t> d_pencil_order;
    size_t box_lds_bytes = 0;
    int box_reach = 0;                // rows of x staged in LDS on either side of a workgroup's 512 (format 6), 0: format 4
    int64_t dia_empty_segments = 0;   // (diagonal, 128-row group) pairs without entries
    // device-side build from the reference layout (kfsp_build.hip)
    DevBuf<int32_t> d_ell_adj, d_cnt, d_ticket;
    // columns of OFFDIAG / DIAG resident from the last reference-layout upload (0: none that may be reused)
    int64_t ell_cols = 0;
    int32_t ell_ld = 0, ell_bw = 0;
    DevBuf<double> d_ell_off, d_ell_diag;
    DevBuf<int> d_slot;
    DevBuf<char> d_scan;
    // Speculative rebuild of a resident FSP (option build_speculate; kfsp_build.hip): the numbers the host would stop for
    // (coordinate ranges, link statistics, SELL slots) land in this pinned block instead and are checked ONCE, when the
    // generator is built - a failed check repeats order and build the slow way.  kc_*: the key layout of the last state
    // order (each species' field widened to its full bit width), valid while every coordinate stays inside [kc_lo, kc_hi].
    char *h_build = nullptr;
    bool kc_ok = false, order_check = false, last_build_sell = false, h_build_ready = false;
    int kc_ns = 0, kc_bits = 0, kc_lo[16] = {0}, kc_hi[16] = {0}, kc_shift[16] = {0};
    int64_t opt_build_speculate = 1, spec_builds = 0, spec_redone = 0;
    int64_t opt_ssa_filter = 1;           // a one-bit-per-slot filter in front of the walk's table of listed states
    int64_t opt_ssa_regs = 1;             // the walk evaluates unlisted states from descriptors in registers when the program allows it
    // the sorted keys of the current order (d_perm) and for how many states both are valid: an expansion merges the appended
    // states' keys into them, a drop compacts them (state_order_from_resident, state_order_after_drop)
    DevBuf<unsigned long long> d_skeys, d_skeys2;
    DevBuf<int32_t> d_perm2;
    int64_t order_n = 0, skeys_n = 0, order_merges = 0;
    // Internal state order (kfsp_set_state_coords): the device keeps generator and
    // vectors in lexicographic order of the state coordinates, the host sees its
    // own order.  perm[new] = old, iperm[old] = new (0-based).
    DevBuf<int32_t> d_perm, d_iperm, d_coords, d_coords2, d_ell_adj2;
    DevBuf<double> d_ell_off2, d_ell_diag2, d_pstage;
    DevBuf<unsigned long long> d_keys;     // 2 n sort keys (in, out)
    DevBuf<int32_t> d_sortidx;             // n identity indices (sort values in)
    DevBuf<char> d_sorttmp;
    int64_t perm_pending_n = 0;            // coordinates received for a generator of this size
    bool perm_on = false;                  // the current generator and vectors are permuted
    int64_t prod_count = 0, prod_last = 0; // products on the current / the previous generator

    // DROP_STATES on the device (kfsp_drop.hip): one flag byte per state in the caller's order
    // (+ the inverted copy the compaction needs), three counters, and the plan they belong to
    DevBuf<uint8_t> d_dropflag;
    DevBuf<unsigned long long> d_dropcnt;
    bool drop_planned = false;
    int64_t drop_ell_cols = 0;         // columns of the reference arrays resident when kfsp_drop_compact ran (kfsp_drop_rebuild)
    int32_t drop_bw = 0;               // reaction slots of that generator
    int64_t drop_n = 0, drop_flagged = 0;
    // the compacted w (caller's order) waits in d_tmp for the generator of the compacted FSP
    bool w_pending = false;
    int64_t w_pending_n = 0;

    // ONESTEP_EXTENDER on the device (kfsp_onestep.hip): two scratch arenas (+ one for the columns of the new states)
    DevBuf<char> d_os1, d_os2, d_os3, d_os4, d_os5;
    // the model's propensity program (kfsp_prop.hip): [code_off | imm_off | tab_species | code] and [params | imm | tables]
    DevBuf<int32_t> d_prop_i;
    DevBuf<double> d_prop_d;
    bool prop_ready = false;
    // the program runs through prop_eval_light: whatever the populations / as long as they stay below prop_tab_len (its
    // library functions sit in tabulated reactions only)
    bool prop_light = false, prop_light_tab = false;
    // every reaction is a product chain of at most three operands with at most one constant among them, or sits behind a
    // one-species table: the register-resident walk evaluates an unlisted state from descriptors it keeps in registers
    // (d_prop_fast_i: one word per reaction, d_prop_fast_d: its constant; kfsp_ssa.hip)
    bool prop_fast = false;
    DevBuf<int32_t> d_prop_fast_i;
    DevBuf<double> d_prop_fast_d;
    int prop_ncode = 0, prop_nimm = 0;
    size_t prop_mono_off = 0, prop_monoc_off = 0;      // where the product-chain tables sit in d_prop_i / d_prop_d
    int prop_ns = 0, prop_nr = 0, prop_np = 0, prop_np_pad = 1, prop_nimm_pad = 1, prop_tab_len = 0;
    // tables over TWO species (kfsp_set_propensity_tables2: a compiled-in CUSTOMPROP tabulated by the host): per reaction
    // [s1, s2, n1, n2] in d_prop_t2i (s1 < 0: none) and the offset of its n1 x n2 block in d_prop_t2d (d_prop_t2o); a
    // population beyond a table raises d_prop_oob[0] and leaves the largest population per species in d_prop_oob[1 + s]
    bool prop_has_tab2 = false;
    DevBuf<int32_t> d_prop_t2i, d_prop_oob;
    DevBuf<long long> d_prop_t2o;
    DevBuf<double> d_prop_t2d;
    int32_t prop_missed[16] = {0};                      // what the last overflow left in d_prop_oob[1 ..] (kfsp_propensity_overflow)

    // vectors
    DevBuf<double> d_V;    // (kMMax+2) columns, stride ldv, unnormalised basis
    DevBuf<double> d_w;    // probability vector, ldv
    DevBuf<double> d_xg;   // nranks*L gathered source (nranks > 1) or scratch x (kfsp_spmv)
    DevBuf<double> d_tmp;  // ldv scratch (kfsp_spmv output)
    DevBuf<double> d_full;   // n: a whole vector in the caller's order (internal state order under a communicator)
    bool w_pending_full = false;   // the pending vector is the whole vector in the caller's order in d_wfull (kfsp_expand_resident)
    DevBuf<double> d_wfull;  // n: the whole compacted w between kfsp_drop_compact and the next generator (communicator)
    DevBuf<uint8_t> d_flagloc;   // L * nranks: flags of the blocks in the internal order (communicator)

    // scalars
    DevBuf<double> d_part;   // kNumPartial * kMaxGrid
    DevBuf<double> d_stage;  // kNumStage
    DevBuf<double> d_H;      // kMH * kMH image + 2 (avnorm^2, avnorm)
    DevBuf<double> d_sq;     // finished squared norms, index = column (1-based)
    DevBuf<double> d_g;      // finished u_j . u_{j-1}, index = j
    DevBuf<double> d_y;      // kMH coefficients
    DevBuf<int> d_flag;
    int part_rr = 0, stage_rr = 0;
    double *h_H = nullptr;     // pinned image of d_H (+2) for the one copy per Arnoldi pass
    double *h_pin = nullptr;   // pinned scratch for scalars and the combine coefficients (kMH + 8 doubles)
    double avnorm_last = 0.0;

    // options
    int64_t opt_grid = 0;   // cap on the product kernels' grid, 0 = auto (2048)
    int64_t opt_vgrid = 0;  // cap on the streaming kernels' grid, 0 = auto (1024)
    int64_t opt_nt = -1;    // -1 auto, 0 off, 1 on
    int64_t opt_format = 0; // 0 auto (DIA when banded), 1 always SELL
    int64_t opt_fused = 1;  // 1: one-pass IOP(2) orthogonalisation (k_ortho2)
    int64_t opt_halo = 1;         // 0: always all-gather the whole source vector
    int64_t opt_sell_sigma = 0;           // rows per window of the SELL-sigma sort under the internal state order (0: off, the default:
                                          // measured slower - what it saves in padding it loses in gather coalescing)
    int64_t opt_halo_sell = 1;    // 0: SELL generators always all-gather, whatever their reach
    int64_t opt_halo_p2p = 0;     // 1: a rank exchanges its strips with its two neighbours only (send/recv); 0: all-gather of all strips
    int64_t opt_small = 1;        // 1: one-launch Arnoldi pass for <= 16384 rows
    int64_t opt_overlap = 1;      // 0: exchange and product strictly one after the other
    int64_t opt_host_build = 0;   // 1: transpose reference-layout input on the host (A/B testing)
    int64_t opt_small_lds = 1;            // 0: the one-launch Arnoldi kernel reads the generator from global memory
    int64_t lds_per_block = 65536;        // device limit (hipDeviceAttributeMaxSharedMemoryPerBlock)
    int64_t opt_dia_mask = 1;             // 0: never skip empty diagonal segments
    int64_t opt_box_pencil = -1;          // matrix-free boxes in pencils along the slowest species: -1 pencils (format 7) where eligible and large
                                          // enough, 0 never, 1 pencils whatever the size, 2 pencils in slabs (format 8: measured slower)
    int64_t opt_box_slab_waves = 12;      // format 8: most wavefronts (= lines of the second-slowest species) a workgroup takes
    int64_t opt_box_tile = -1;            // tiled trip order of box generators: -1 when the box outgrows the caches, 0 never, 1 always
    int64_t opt_box_lds = 0;              // 1: the single-factor matrix-free product stages the near part of x in LDS (format 6;
                                          // measured slower than format 4 on every box: DESIGN.md 4.1b - off by default)
    int64_t opt_box_reach = 512;          // largest shift (rows) served from that window
    int64_t opt_box_generic = 0;          // 1: matrix-free boxes always take the run-time interpreted kernel
    int64_t opt_ssa_partition = 1;         // the resident expansion under a communicator: each rank walks its share of the seeds (0: all of them)
    int64_t opt_mmax = kfsp::kMMax;        // largest Krylov dimension the NEXT basis is allocated for (m_max + 3 columns)
    int64_t v_mmax = 0;                    // ... and the one the basis in d_V WAS allocated for (resize); 0: no basis yet
    int64_t opt_ssa_resident = 0;          // 1: the caller vouches that the arrays given to kfsp_ssa_streams are the ones last uploaded
    int64_t coords_n = 0;                  // states whose coordinates sit in d_coords (kfsp_set_state_coords), 0: none
    int32_t coords_ld = 0, coords_ns = 0;
    int64_t opt_sell_code = -1;            // dictionary-coded SELL columns: -1 auto (under the internal state order), 0 never, 1 always try
    int64_t opt_box_store = 0;            // 1: kfsp_set_matrix_box writes the generator out as stored diagonals on the device (banded form)
    int64_t opt_state_order = 1;          // 1: use kfsp_set_state_coords for large, long-lived generators (0: never)
    int64_t opt_ssa_general = 0;          // 1: the SSA walk always runs its general kernel (A/B of the register-resident one)
    int64_t opt_keep_coords = 0;          // 1: coordinates handed over stay resident even when no order is derived from them
    int64_t opt_state_order_min = 32768;  // smallest generator that is reordered
    int64_t opt_state_order_products = 48;   // ... and only if its predecessor saw this many products
    double t_ms[KFSP_T_COUNT] = {0, 0, 0, 0, 0, 0, 0};
};

namespace kfsp {
// group contexts (kfsp_group.cpp): what each entry point of include/kfsp.h does when it is handed a head
void comm_abort(kfsp_ctx *ctx);      // kfsp_api.cpp: release a rank blocked in a collective (called from the group's watchdog)
int group_selftest(int nranks, int failing_rank, int hanging_rank, int work_ms, int hang_ms, int timeout_ms, int grace_ms,
                   int settle_ms, int *rc_out, int *who_out, double *seconds, int *broken, int *stuck);
int group_create(int nranks, const int *devices, kfsp_ctx **out);
int group_destroy(kfsp_ctx *h);
int group_size(const kfsp_ctx *h);
int group_set_option(kfsp_ctx *h, const char *name, int64_t value);
int group_update_matrix_ell(kfsp_ctx *h, int32_t n, int32_t bw, int32_t ld, const int32_t *adj, const double *offdiag,
                            const double *diag, int32_t n_unchanged);
int group_set_matrix_csr(kfsp_ctx *h, int64_t n, int64_t row0, int64_t nrows, const int64_t *rowptr, const int32_t *col,
                         const double *val);
int group_set_matrix_box(kfsp_ctx *h, int32_t ns, const int32_t *dims, int32_t nr, const int32_t *stoich, const int32_t *ndep,
                         const int32_t *dep_species, const double *tables);
int group_set_state_coords(kfsp_ctx *h, int32_t n, int32_t ns, int32_t ld, const int32_t *state);
int group_state_order_active(const kfsp_ctx *h, int *active);
int group_matrix_info(const kfsp_ctx *h, int64_t *nrows, int64_t *slots, int64_t *nnz);
int group_matrix_bytes(const kfsp_ctx *h, int force_sell, int64_t *bytes);
int group_set_vector(kfsp_ctx *h, int64_t n, const double *w);
int group_get_vector(kfsp_ctx *h, int64_t n, double *w);
int group_begin_step(kfsp_ctx *h, double *beta);
int group_arnoldi(kfsp_ctx *h, int m, int jold, int qiop, double break_tol, double *H, int ldh, int *mbrkdwn, int *k1, double *avnorm);
int group_combine(kfsp_ctx *h, int mx, double beta, const double *y, double *wsum);
int group_restore_w(kfsp_ctx *h, double beta);
int group_spmv(kfsp_ctx *h, const double *x, double *y);
int group_spmv_w(kfsp_ctx *h, double *y);
int group_drop_plan(kfsp_ctx *h, double dsum, double *droptol, int64_t *drop_count, int64_t *n_flagged);
int group_drop_flags(kfsp_ctx *h, int64_t n, uint8_t *dropped);
int group_drop_compact(kfsp_ctx *h, int64_t *n_new);
int group_reduce_w(kfsp_ctx *h, int squared, double *out);
int group_get_basis(kfsp_ctx *h, int j, int64_t n, double *v);
int group_spmv_bench(kfsp_ctx *h, int reps, int variant, float *ms_total);
int group_get_timers(kfsp_ctx *h, double *ms, int reset);
int group_layout_info(const kfsp_ctx *h, int64_t *v);
int group_drop_rebuild(kfsp_ctx *h);
int group_expand_resident(kfsp_ctx *h, double t_ssa, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t max_count,
                          int32_t capacity, int64_t *n_new, int64_t *n_from_ssa);
int group_set_propensity_tables2(kfsp_ctx *h, int32_t nr, const int32_t *s1, const int32_t *s2, const int32_t *n1, const int32_t *n2,
                                 const int64_t *off, int64_t len, const double *tab2);
int group_update_state_coords(kfsp_ctx *h, int32_t n, int32_t ns, int32_t ld, const int32_t *state, int32_t n_unchanged);
int group_set_propensity_program(kfsp_ctx *h, int32_t ns, int32_t nr, int32_t np, const double *params, const int32_t *code_off,
                                 const int32_t *code, const int32_t *imm_off, const double *imm, const int32_t *tab_species,
                                 int32_t tab_len, const double *tab);
kfsp_ctx *group_rank0(const kfsp_ctx *h);
// generator build on the device from the reference layout (kfsp_build.hip)
// keep: leading columns whose OFFDIAG / DIAG are resident and unchanged (only the rest is uploaded)
int build_from_ell_device(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, const int32_t *adj,
                          const double *offdiag, const double *diag, int64_t keep = 0);
// after a banded generator was stored: find the empty (diagonal, 128-row group) segments and
// switch the masked kernel variant on if they are worth skipping
constexpr int kRedoBuild = 7777;      // a speculative order / build did not hold: repeat both with speculate = false
int build_from_resident_ell(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, bool speculate = false);
int state_order_from_resident(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, bool *ok, bool speculate = false, int32_t n_prev = 0);
bool state_order_after_drop(kfsp_ctx *ctx, int32_t n_old, int32_t n_new, const uint8_t *keep, const int32_t *scan, int *rc);
int compact_resident_ell(kfsp_ctx *ctx, int64_t n, int bw, int ld, const uint8_t *keep, int64_t n_keep, bool with_coords, int lds);
int build_dia_mask(kfsp_ctx *ctx);
// after a SELL image was stored (d_off, d_col, d_val): try the dictionary-coded column form
int build_sell_code(kfsp_ctx *ctx);
// kfsp_set_matrix_box with option box_store: the box generator written out as stored diagonals (d_dia, d_diag)
int box_materialize(kfsp_ctx *ctx);
// lexicographic order of n states given as ns coordinates each (host array, leading
// dimension ld): fills d_perm / d_iperm; *ok = false when the packed key needs > 64 bits
int state_order_from_coords(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state, bool *ok, int64_t keep = 0,
                            bool order = true);
// dst[i'] = src[perm[i']] (host order -> device order) and dst[i] = src[iperm[i]] (back)
void launch_gather_index(int64_t n, const int32_t *index, const double *src, double *dst, hipStream_t st);
// DROP_STATES pieces (kfsp_drop.hip)
void launch_drop_sums(int grid, int64_t npairs, const double *w, const DropLevels &L, double *partial, double *out, hipStream_t st);
void launch_drop_flags(int64_t n, const double *w, const double *aw, double droptol, const int32_t *iperm, uint8_t *flag,
                       unsigned long long *cnt, hipStream_t st);
void launch_flags_to_caller(int64_t n, const uint8_t *all, const int32_t *iperm, uint8_t *flag, hipStream_t st);
int drop_compact_vector(kfsp_ctx *ctx, int64_t n, const double *src, double *dst, int *n_keep_dev);
// ONESTEP_EXTENDER's integer work (kfsp_expand.hip); all arrays are host memory
// off_new / diag_new (may be null): the propensity columns of the appended states, made by the program of kfsp_prop.hip
int onestep_device(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state, int32_t lds,
                   const int32_t *adj, int32_t lda, int32_t max_count, int32_t cap, int32_t *n_out, int32_t *state_new,
                   int32_t *adj_out, double *off_new = nullptr, int32_t ldo = 0, double *diag_new = nullptr);
// propensities on the device (kfsp_prop.hip)
int prop_set_program(kfsp_ctx *ctx, int32_t ns, int32_t nr, int32_t np, const double *params, const int32_t *code_off,
                     const int32_t *code, const int32_t *imm_off, const double *imm, const int32_t *tab_species, int32_t tab_len,
                     const double *tab);
int prop_eval_device(kfsp_ctx *ctx, int64_t n, const int32_t *d_state, int lds, double *d_off, int ldo, double *d_diag);
int prop_set_tables2(kfsp_ctx *ctx, int32_t nr, const int32_t *s1, const int32_t *s2, const int32_t *n1, const int32_t *n2,
                     const int64_t *off, int64_t len, const double *tab2);
// after an operation that evaluated propensities (and a stream synchronisation): -16 if a population missed a two-species
// table (the flag is cleared, prop_missed filled), else 0.  No-op for programs without such tables.
int prop_check_overflow(kfsp_ctx *ctx);
// SSA paths on independent streams (kfsp_ssa.hip); all arrays are host memory
int ssa_streams_device(kfsp_ctx *ctx, double tstep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n,
                       const int32_t *state, int32_t lds, const int32_t *adj, const double *offdiag, int32_t lda, const double *diag,
                       int32_t max_count, int32_t cap_new, int32_t *n_found, int32_t *state_new, double *off_new, int32_t ldo,
                       double *diag_new);
int prop_eval_host(kfsp_ctx *ctx, int32_t n, const int32_t *state, int32_t lds, double *offdiag, int32_t ldo, double *diag);
// the same walk on lists that are on the device; the states met and their columns stay there (d_pstage).  partitioned: called
// by EVERY rank of a communicator on identical lists - each rank walks the paths of its share of the seeds only and the
// records are all-gathered (same result on every rank as the unpartitioned walk, bit for bit)
int ssa_streams_core(kfsp_ctx *ctx, double tstep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n,
                     const int32_t *d_state, int32_t lds, const int32_t *d_adj, const double *d_off, int32_t lda, const double *d_diag,
                     int32_t max_count, int32_t cap_new, int32_t ldo, int32_t *n_found, int32_t **sn, double **on, double **dn,
                     bool partitioned = false);
int comm_gather_doubles(kfsp_ctx *ctx, const double *send, double *recv, size_t count, hipStream_t st);
int comm_gather_bytes(kfsp_ctx *ctx, const void *send, void *recv, size_t bytes, hipStream_t st);
// SSA walk + one-step sweep on the resident lists (kfsp_expand.hip)
int expand_resident_lists(kfsp_ctx *ctx, double tstep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich,
                          int32_t max_count, int32_t cap, int64_t *n_out, int64_t *n_ssa);
void launch_zero_pad(int64_t n0, int64_t n1, double *w, hipStream_t st);
}  // namespace kfsp
