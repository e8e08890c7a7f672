// Host side of libkfsp_hip: the C ABI of include/kfsp.h, the device context
// (basis, vectors, generator, scalar staging), the ELL -> SELL transpose, the
// RCCL row-partition plumbing and the host Pade exponential.
#include "kfsp_ctx.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

using namespace kfsp;

namespace {

constexpr int kAbiVersion = 2;   // 2: kfsp_dgexpv_replay, KFSP_EV_READY

}  // namespace

namespace {

struct PhaseTimer {
    kfsp_ctx *c;
    int phase;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer(kfsp_ctx *c_, int p) : c(c_), phase(p), t0(std::chrono::steady_clock::now()) {}
    ~PhaseTimer() { c->t_ms[phase] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

int fail(kfsp_ctx *c, int code, const char *what)
{
    if (c) c->err = what;
    return code;
}

// C++ exceptions (host allocations) end at the C boundary as status codes
template <class F>
int no_throw(kfsp_ctx *c, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(c, 4001, "out of host memory");
    } catch (const std::exception &e) {
        if (c) c->err = std::string("exception: ") + e.what();
        return 4000;
    } catch (...) {
        return fail(c, 4000, "unknown exception");
    }
}

int hip_fail(kfsp_ctx *c, hipError_t e, const char *where)
{
    if (c) c->err = std::string(where) + ": " + hipGetErrorString(e);
    return 1000 + (int)e;
}

int nccl_fail(kfsp_ctx *c, ncclResult_t r, const char *where)
{
    if (c) c->err = std::string(where) + ": " + ncclGetErrorString(r);
    return 2000 + (int)r;
}

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #expr); \
    } while (0)

#define NCCL_TRY(expr)                                           \
    do {                                                         \
        ncclResult_t r_ = (expr);                                \
        if (r_ != ncclSuccess) return nccl_fail(ctx, r_, #expr); \
    } while (0)

int spmv_grid(const kfsp_ctx *c)
{
    // a wavefront trip covers 64 rows (SELL) or 128 rows (banded, two rows per lane)
    const int64_t trips = c->use_dia ? (c->nchunks + 1) / 2 : c->nchunks;
    int64_t g = round_up((trips + 3) / 4, 8);
    // 1024 workgroups (4 per CU, 16 waves per CU) already saturate HBM with the
    // 5-9 independent loads a lane keeps in flight, and halve the partial sums
    // every consumer has to re-add (measured: profiles/r01_sweep.log)
    const int64_t cap = c->opt_grid > 0 ? round_up(c->opt_grid, 8) : 1024;
    g = std::min<int64_t>(g, std::min<int64_t>(cap, kMaxGrid));
    return (int)std::max<int64_t>(g, 8);
}

// Rows the streaming kernels touch: the SELL-padded local block.  Every pass
// rewrites exactly these rows of every column it uses, so stale data beyond
// them (left by a larger, earlier FSP) is never read.
int64_t act_pairs(const kfsp_ctx *c) { return c->nchunks * (kChunk / 2); }

// Streaming kernels: every consumer re-sums the producer's partials, so the
// grid is kept at <= 1024 workgroups (4 per CU) with >= 4 pairs per lane; the
// loops are unrolled so that this still keeps > 16 MB of loads in flight.
int vec_grid(const kfsp_ctx *c)
{
    int64_t g = (act_pairs(c) + 4 * kBlock - 1) / (4 * kBlock);
    const int64_t cap = c->opt_vgrid > 0 ? c->opt_vgrid : 1024;
    g = std::min<int64_t>(g, std::min<int64_t>(cap, kMaxGrid));
    return (int)std::max<int64_t>(g, 1);
}

bool use_nt(const kfsp_ctx *c)
{
    if (c->opt_nt >= 0) return c->opt_nt != 0;
    // stream the generator around the caches only when it cannot stay in the
    // 256 MiB Infinity Cache between two products anyway
    const double bytes = c->use_dia ? (double)c->nd * c->dia_ld * 8.0 : (double)c->slots * 12.0;
    return bytes > 192.0 * 1024 * 1024;
}

double *next_partial(kfsp_ctx *c)
{
    double *p = c->d_part.p + (size_t)c->part_rr * kMaxGrid;
    c->part_rr = (c->part_rr + 1) % kNumPartial;
    return p;
}

SellDev sell_of(const kfsp_ctx *c)
{
    return SellDev{c->nloc, c->nchunks, c->d_off.p, c->d_col.p, c->d_val.p, c->d_diag.p,
                   c->d_dtab.p, c->d_dtlen.p, c->d_code.p, c->d_codeoff.p};
}

// generator part of the product kernel's arguments
void set_matrix_args(const kfsp_ctx *c, SpmvArgs &a)
{
    a.A = sell_of(c);
    a.D.nd = c->nd;
    for (int d = 0; d < kMaxDiag; ++d) a.D.delta[d] = c->delta[d];
    a.D.val = c->d_dia.p;
    a.D.ld = c->dia_ld;
    a.D.diag = c->d_diag.p;
    a.D.nchunks = c->nchunks;
    a.D.n = c->n;
    a.D.gmask = c->dia_masked ? c->d_gmask.p : nullptr;
    a.D.zero = c->d_zero.p;
    if (c->use_box) a.B = c->box;
    else a.B.ns = a.B.nr = a.B.ntab = 0;
    a.box_tab = c->d_box.p;
    a.box_fast = reinterpret_cast<const BoxFast *>(c->d_box.p + (c->box_lds_bytes / sizeof(double)));
    a.udot2 = nullptr;
    a.partial2 = nullptr;
    a.trip_order = nullptr;
}

// ---- the two collectives of the data path, over RCCL or the loop-back transport ----
constexpr int kLoopVals = 16;                              // values one all-reduce may carry (the 16 FIND_DROPTOL sums)
constexpr int kLoopScratch = 64 * kLoopVals + kLoopVals;   // doubles: up to 64 ranks x 16 scalars (+ result)

// buf[0..count) <- sum (or max) over ranks, in place, on stream st
int comm_allreduce(kfsp_ctx *ctx, double *buf, int count, bool take_max, hipStream_t st)
{
    if (!ctx->loop) {
        std::lock_guard<std::mutex> lk(ctx->comm_mu);
        if (ctx->comm_aborted) return fail(ctx, 2999, "the communicator was aborted");
        NCCL_TRY(ncclAllReduce(buf, buf, (size_t)count, ncclDouble, take_max ? ncclMax : ncclSum, ctx->comm, st));
        return 0;
    }
    kfsp::LoopGroup *g = ctx->loop;
    if (count > kLoopVals || g->n > 64) return fail(ctx, -1, "loop-back all-reduce: too many values");
    HIP_TRY(hipStreamSynchronize(st));                  // this rank's contribution is in memory
    g->slot[(size_t)ctx->rank] = buf;
    if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");
    double *h = ctx->h_loop;
    for (int p = 0; p < g->n; ++p)
        HIP_TRY(hipMemcpyAsync(h + kLoopVals * p, g->slot[(size_t)p], (size_t)count * sizeof(double), hipMemcpyDefault, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");   // everyone has read
    double *r = h + kLoopVals * g->n;
    for (int i = 0; i < count; ++i) {
        double a = h[i];
        for (int p = 1; p < g->n; ++p) a = take_max ? std::max(a, h[kLoopVals * p + i]) : a + h[kLoopVals * p + i];   // rank order: same bits on every rank
        r[i] = a;
    }
    HIP_TRY(hipMemcpyAsync(buf, r, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// recv[p*count .. (p+1)*count) <- send of rank p, on stream st
int comm_allgather(kfsp_ctx *ctx, const double *send, double *recv, size_t count, hipStream_t st)
{
    if (!ctx->loop) {
        std::lock_guard<std::mutex> lk(ctx->comm_mu);
        if (ctx->comm_aborted) return fail(ctx, 2999, "the communicator was aborted");
        NCCL_TRY(ncclAllGather(send, recv, count, ncclDouble, ctx->comm, st));
        return 0;
    }
    kfsp::LoopGroup *g = ctx->loop;
    HIP_TRY(hipStreamSynchronize(st));
    g->slot[(size_t)ctx->rank] = send;
    if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");
    for (int p = 0; p < g->n; ++p)
        HIP_TRY(hipMemcpyAsync(recv + (size_t)p * count, g->slot[(size_t)p], count * sizeof(double), hipMemcpyDefault, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");   // nobody reuses its send buffer before all have copied
    return 0;
}

// the same for raw bytes (flags)
int comm_allgather_bytes(kfsp_ctx *ctx, const void *send, void *recv, size_t bytes, hipStream_t st)
{
    if (!ctx->loop) {
        std::lock_guard<std::mutex> lk(ctx->comm_mu);
        if (ctx->comm_aborted) return fail(ctx, 2999, "the communicator was aborted");
        NCCL_TRY(ncclAllGather(send, recv, bytes, ncclUint8, ctx->comm, st));
        return 0;
    }
    kfsp::LoopGroup *g = ctx->loop;
    HIP_TRY(hipStreamSynchronize(st));
    g->slot[(size_t)ctx->rank] = send;
    if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");
    for (int p = 0; p < g->n; ++p)
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + (size_t)p * bytes, g->slot[(size_t)p], bytes, hipMemcpyDefault, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");
    return 0;
}

}  // namespace

namespace kfsp {
// the collectives for the translation units of the expansion step (the walk's records of a partitioned expansion)
int comm_gather_doubles(kfsp_ctx *ctx, const double *send, double *recv, size_t count, hipStream_t st) { return comm_allgather(ctx, send, recv, count, st); }
int comm_gather_bytes(kfsp_ctx *ctx, const void *send, void *recv, size_t bytes, hipStream_t st) { return comm_allgather_bytes(ctx, send, recv, bytes, st); }

// Called from ANOTHER thread than the one that drives ctx (the watchdog of a group context, kfsp_group.cpp) when a peer
// failed or a deadline expired: the rank may sit in hipStreamSynchronize behind a collective its peers never entered.
// ncclCommAbort makes the collective's kernel give up; the loop-back transport releases its barriers.  The rank's
// pending call then returns an error, later collectives return 2999.  comm_mu keeps the abort from running while
// the driving thread is in the middle of enqueuing on the same communicator.
void comm_abort(kfsp_ctx *ctx)
{
    if (ctx->loop) {
        ctx->loop->abort();
        return;
    }
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    if (ctx->comm && !ctx->comm_aborted) (void)ncclCommAbort(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_aborted = true;
}
}  // namespace kfsp

namespace {

// Make block partials a scalar every rank agrees on.
int publish(kfsp_ctx *ctx, Pending local, Pending *out)
{
    if (!ctx->use_comm) {
        *out = local;
        return 0;
    }
    double *st = ctx->d_stage.p + ctx->stage_rr;
    ctx->stage_rr = (ctx->stage_rr + 1) % kNumStage;
    launch_finalize(local, st, nullptr, ctx->stream);
    if (int rc = comm_allreduce(ctx, st, 1, false, ctx->stream)) return rc;
    *out = Pending{st, 1};
    return 0;
}

int gather_source(kfsp_ctx *ctx, const double *src_local, const double **xg);
int exchange_strips(kfsp_ctx *ctx, const double *src_local, hipStream_t st);

int trips_grid(int64_t trips, int64_t cap)
{
    int64_t g = round_up((trips + 3) / 4, 8);
    g = std::min<int64_t>(g, cap);
    return (int)std::max<int64_t>(g, 8);
}

// One generator product y = (s) A x from the local source column `src` (or from
// a full global vector when src_is_global).  `a` carries everything except the
// source pointer, the trip range and the partial buffers.  With a banded
// generator and a communicator the product is split: the interior trips, which
// read no halo row, start at once on the compute stream while the strips are
// exchanged on the communication stream; the few boundary trips follow once
// the halo has landed.  p1/p2 receive the block partials (modes 1-3).
int run_product(kfsp_ctx *ctx, int mode, SpmvArgs a, const double *src, bool src_is_global, Pending *p1, Pending *p2,
                bool force_sell = false, bool force_plain_sell = false)
{
    hipStream_t st = ctx->stream;
    ++ctx->prod_count;
    const bool dia = ctx->use_dia && !force_sell;
    const bool nt = use_nt(ctx);
    const int64_t trips = dia ? (ctx->nchunks + 1) / 2 : ctx->nchunks;
    int64_t cap = ctx->opt_grid > 0 ? round_up(ctx->opt_grid, 8) : 1024;
    if (ctx->use_box && dia && ctx->opt_grid <= 0) {
        // every workgroup of a matrix-free product copies the table image into its LDS first: no more
        // workgroups than are resident at once (160 KB of LDS per CU), or the later ones pay that copy
        // again for fewer rows each (toggle 1000 x 1000, 64 KB image: 8.3 us with 512, 12.1 us with 1024)
        const int64_t per_cu = std::max<int64_t>(1, (int64_t)(160 * 1024) / (int64_t)(ctx->box_lds_bytes + 512));
        cap = std::min<int64_t>(cap, 256 * per_cu);
    }
    double *P1 = mode != 0 ? next_partial(ctx) : nullptr;
    double *P2 = mode == 3 ? next_partial(ctx) : nullptr;
    a.row0 = ctx->row0;

    const int fmt = (ctx->use_box && dia) ? (ctx->box_fast && !ctx->opt_box_generic ? 4 : 3)
                                          : (dia ? (ctx->dia_masked ? 2 : 1) : (ctx->sell_coded && !force_plain_sell ? 5 : 0));
    const int64_t H = ctx->halo, L = ctx->L;
    const int64_t trip_rows = dia ? 128 : 64;
    int64_t lo = 0, hi = 0;
    // (a SELL generator takes part when its reach is bounded - the internal state order - and the halo mode was agreed)
    bool split = !src_is_global && !force_sell && ctx->use_halo && ctx->opt_overlap != 0 && ctx->comm_stream != nullptr;
    if (split) {
        lo = (H + trip_rows - 1) / trip_rows;                        // first trip whose rows all lie >= H
        hi = std::min<int64_t>((L - H) / trip_rows, trips);          // trips [lo, hi) end below L - H
        // Three launches and two cross-stream waits cost ~10-15 us; that only pays once
        // the product itself is several times longer (>= ~2M rows per rank), or when
        // the caller insists (overlap = 2, used by the tests)
        const int64_t min_trips = ctx->opt_overlap >= 2 ? 64 : 16384;
        if (hi - lo < min_trips) split = false;
    }
    if (!split) {
        const double *xg = src;
        if (!src_is_global)
            if (int rc = gather_source(ctx, src, &xg)) return rc;
        a.xg = xg;
        a.partial = P1;
        a.partial2 = P2;
        a.trip_begin = 0;
        a.trip_end = trips;
        a.trip_split = INT64_MAX;
        a.trip_jump = 0;
        a.trip_order = (ctx->trip_order_n == trips && !force_sell) ? ctx->d_trip_order.p : nullptr;
        const int g = trips_grid(trips, cap);
        if (fmt == 4 && ctx->box_slab && ctx->opt_box_pencil == 2 && !ctx->use_comm && ctx->opt_box_lds == 0) {
            // format 8: one workgroup per 128 rows x W lines of the second-slowest species, walking the planes in step
            const int64_t total = ctx->slab_lo_trips * ctx->slab_groups;
            const size_t lds = ((ctx->box_lds_bytes + 15) & ~(size_t)15) + 2 * (size_t)ctx->slab_waves * 1024;
            // (workgroups of up to 1024 threads: as many as are resident at once, they loop over the slabs)
            const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(12 / ctx->slab_waves, (int64_t)(160 * 1024) / (int64_t)(lds + 1024)));
            const int64_t want = ctx->opt_grid > 0 ? ctx->opt_grid : 256 * per_cu;
            const int gs = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(total, want), kMaxGrid));
            launch_spmv_slab(mode, gs, ctx->slab_waves, a, st, lds, ctx->pencil_plane_rows, ctx->pencil_planes, ctx->slab_line_rows,
                             ctx->slab_lines, ctx->slab_groups, ctx->slab_lo_trips, ctx->pencil_simple);
            if (p1) *p1 = Pending{P1, gs};
            if (p2) *p2 = Pending{P2, gs};
            return 0;
        }
        if (fmt == 4 && ctx->box_pencil && ctx->opt_box_pencil != 0 && !ctx->use_comm && ctx->opt_box_lds == 0) {
            // format 7: one wavefront per pencil of 128 rows x all planes of the slowest species
            // (3 workgroups per CU: measured best on the 22^6 box and its slab - 768: 798 / 77.7 us, 512: 866 / 84, 1024: 936 / 86,
            // 2048: 816 / 81; anything that is not a multiple of 256 leaves some CUs with one more - profiles/r04_pencil_grid_sweep.txt)
            const int gp = trips_grid(ctx->pencil_trips, ctx->opt_grid > 0 ? cap : 768);
            launch_spmv_pencil(mode, gp, a, st, ctx->box_lds_bytes, ctx->pencil_plane_rows, ctx->pencil_planes, ctx->pencil_trips,
                               ctx->pencil_order_n == ctx->pencil_trips ? ctx->d_pencil_order.p : nullptr, ctx->pencil_simple);
            if (p1) *p1 = Pending{P1, gp};
            if (p2) *p2 = Pending{P2, gp};
            return 0;
        }
        if (fmt == 4 && ctx->box_reach > 0 && ctx->opt_box_lds != 0 && !ctx->use_comm && a.trip_order == nullptr) {
            // format 6: near entries from an LDS window of x (single rank: x is readable exactly on [0, n))
            const size_t img = (ctx->box_lds_bytes + 15) & ~(size_t)15;
            launch_spmv_boxlds(mode, g, a, st, img + 2 * (size_t)(512 + 2 * ctx->box_reach) * sizeof(double), ctx->box_reach);
        } else {
            launch_spmv(mode, g, a, nt, fmt, st, ctx->box_lds_bytes);
        }
        if (p1) *p1 = Pending{P1, g};
        if (p2) *p2 = Pending{P2, g};
        return 0;
    }
    // exchange on the communication stream, behind everything that produced src
    HIP_TRY(hipEventRecord(ctx->ev_src, st));
    HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_src, 0));
    if (int rc = exchange_strips(ctx, src, ctx->comm_stream)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_halo, ctx->comm_stream));
    a.xg = src - ctx->row0;
    int used = 0;
    auto launch_range = [&](int64_t b, int64_t e, int64_t split, int64_t jump, int64_t gcap) {
        if (e <= b) return;
        const int g = trips_grid(e - b, gcap);
        a.partial = P1 ? P1 + used : nullptr;
        a.partial2 = P2 ? P2 + used : nullptr;
        a.trip_begin = b;
        a.trip_end = e;
        a.trip_split = split;
        a.trip_jump = jump;
        launch_spmv(mode, g, a, nt, fmt, st, ctx->box_lds_bytes);
        used += g;
    };
    launch_range(lo, hi, INT64_MAX, 0, std::min<int64_t>(cap, kMaxGrid - 512));   // interior: no halo row is read
    HIP_TRY(hipStreamWaitEvent(st, ctx->ev_halo, 0));
    // one launch for both boundary ranges: linear trips [0, lo) are themselves,
    // [lo, lo + trips - hi) stand for [hi, trips)
    launch_range(0, lo + (trips - hi), lo, hi - lo, 512);
    if (p1) *p1 = Pending{P1, used};
    if (p2) *p2 = Pending{P2, used};
    return 0;
}

// Several scalars at once: one all-reduce for all of them.
int publish_n(kfsp_ctx *ctx, const Pending *local, int k, Pending *out)
{
    if (!ctx->use_comm) {
        for (int i = 0; i < k; ++i) out[i] = local[i];
        return 0;
    }
    if (ctx->stage_rr + k > kNumStage) ctx->stage_rr = 0;
    double *st = ctx->d_stage.p + ctx->stage_rr;
    ctx->stage_rr = (ctx->stage_rr + k) % kNumStage;
    for (int i = 0; i < k; ++i) launch_finalize(local[i], st + i, nullptr, ctx->stream);
    if (int rc = comm_allreduce(ctx, st, k, false, ctx->stream)) return rc;
    for (int i = 0; i < k; ++i) out[i] = Pending{st + i, 1};
    return 0;
}

int resize(kfsp_ctx *ctx, int64_t n);

// Banded generator: only the `halo` boundary rows of the two neighbours are ever
// read.  Every rank contributes [its first halo rows | its last halo rows]; one
// all-gather of these strips, then the two strips this rank needs are dropped
// into the margins of the source column itself.  All on stream st.
int exchange_strips(kfsp_ctx *ctx, const double *src_local, hipStream_t st)
{
    const int64_t H = ctx->halo, L = ctx->L;
    double *col = const_cast<double *>(src_local);
    if (ctx->opt_halo_p2p != 0) {
        // Neighbours only, and straight between the columns: a rank's first H rows go into the margin
        // behind the previous rank's block, its last H rows into the margin in front of the next rank's.
        // No staging copies, no strips of ranks that are not neighbours (an all-gather moves
        // nranks * 2H doubles to every rank for the 2H it needs).
        const bool up = ctx->rank > 0, down = ctx->rank + 1 < ctx->nranks;
        if (!ctx->loop) {
            std::lock_guard<std::mutex> lk(ctx->comm_mu);
            if (ctx->comm_aborted) return fail(ctx, 2999, "the communicator was aborted");
            NCCL_TRY(ncclGroupStart());
            if (up) {
                NCCL_TRY(ncclSend(src_local, (size_t)H, ncclDouble, ctx->rank - 1, ctx->comm, st));
                NCCL_TRY(ncclRecv(col - H, (size_t)H, ncclDouble, ctx->rank - 1, ctx->comm, st));
            }
            if (down) {
                NCCL_TRY(ncclSend(src_local + (L - H), (size_t)H, ncclDouble, ctx->rank + 1, ctx->comm, st));
                NCCL_TRY(ncclRecv(col + L, (size_t)H, ncclDouble, ctx->rank + 1, ctx->comm, st));
            }
            NCCL_TRY(ncclGroupEnd());
            return 0;
        }
        kfsp::LoopGroup *g = ctx->loop;
        HIP_TRY(hipStreamSynchronize(st));
        g->slot[(size_t)ctx->rank] = src_local;
        if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");
        if (up)      // the previous rank's LAST rows sit just below row 0
            HIP_TRY(hipMemcpyAsync(col - H, static_cast<const double *>(g->slot[(size_t)ctx->rank - 1]) + (L - H),
                                   (size_t)H * sizeof(double), hipMemcpyDefault, st));
        if (down)    // the next rank's FIRST rows follow row L-1
            HIP_TRY(hipMemcpyAsync(col + L, static_cast<const double *>(g->slot[(size_t)ctx->rank + 1]),
                                   (size_t)H * sizeof(double), hipMemcpyDefault, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!g->barrier()) return fail(ctx, 2999, "loop-back barrier timed out");   // nobody moves on before all have copied
        return 0;
    }
    double *send = ctx->d_strip.p, *recv = ctx->d_strip.p + 2 * H;
    HIP_TRY(hipMemcpyAsync(send, src_local, (size_t)H * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(send + H, src_local + (L - H), (size_t)H * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (int rc = comm_allgather(ctx, send, recv, (size_t)(2 * H), st)) return rc;
    if (ctx->rank > 0)                 // the previous rank's LAST rows sit just below row 0
        HIP_TRY(hipMemcpyAsync(col - H, recv + (size_t)(ctx->rank - 1) * 2 * H + H, (size_t)H * sizeof(double),
                               hipMemcpyDeviceToDevice, st));
    if (ctx->rank + 1 < ctx->nranks)   // the next rank's FIRST rows follow row L-1
        HIP_TRY(hipMemcpyAsync(col + L, recv + (size_t)(ctx->rank + 1) * 2 * H, (size_t)H * sizeof(double),
                               hipMemcpyDeviceToDevice, st));
    return 0;
}

// The source column must be visible in full on every rank before a product.
int gather_source(kfsp_ctx *ctx, const double *src_local, const double **xg)
{
    if (!ctx->use_comm) {
        *xg = src_local - ctx->row0;       // row0 == 0 here
        return 0;
    }
    if (ctx->use_halo) {
        if (int rc = exchange_strips(ctx, src_local, ctx->stream)) return rc;
        *xg = src_local - ctx->row0;       // global index g lives at src_local[g - row0]
        return 0;
    }
    if (int rc = comm_allgather(ctx, src_local, ctx->d_xg.p, (size_t)ctx->L, ctx->stream)) return rc;
    *xg = ctx->d_xg.p;
    return 0;
}

// After a generator was set: agree across ranks on the exchange mode.  Halo
// exchange needs every rank to hold a banded block whose reach max|delta| does
// not exceed one block length (only the two neighbours are involved then).
int setup_exchange(kfsp_ctx *ctx)
{
    ctx->use_halo = false;
    ctx->halo = 0;
    if (!ctx->use_comm) return 0;
    int64_t reach = 0;
    for (int d = 0; d < ctx->nd; ++d) reach = std::max<int64_t>(reach, std::llabs((long long)ctx->delta[d]));
    // a SELL generator with a bounded reach max |col - row| (known from its build; small under the internal
    // lexicographic state order) reads only boundary rows of its neighbours too
    const bool sell_ok = !ctx->use_dia && ctx->have_sell && ctx->sell_reach >= 0 && ctx->opt_halo_sell != 0;
    if (sell_ok) reach = ctx->sell_reach;
    // ranks without rows take part with neutral values
    const bool ok_local = ctx->opt_halo != 0 && (ctx->nloc == 0 || ctx->use_dia || sell_ok);
    double h[2] = {ok_local ? 0.0 : 1.0, (double)reach};          // max over ranks of (not ok, reach)
    double *st = ctx->d_stage.p;
    HIP_TRY(hipMemcpyAsync(st, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = comm_allreduce(ctx, st, 2, true, ctx->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int64_t H = round_up(std::max<int64_t>((int64_t)h[1], 1), 8);
    if (h[0] != 0.0 || H > ctx->L) return 0;                      // someone is not banded, or reach > one block
    ctx->halo = H;
    if (H + 2 * kChunk > ctx->margin) {
        // re-lay the basis with room for the strips (its contents are rebuilt by
        // the next begin_step anyway)
        // + 128: the banded kernel works on 128-row groups, whose padded rows read up
        // to one group beyond the block end
        ctx->margin = round_up(H + H / 4 + 2 * kChunk, 64);
        ctx->relayout = true;
        if (int rc = resize(ctx, ctx->n)) return rc;
    }
    HIP_TRY(ctx->d_strip.reserve((size_t)(2 * H) * (size_t)(ctx->nranks + 1), true));
    ctx->use_halo = true;
    return 0;
}

// (Re)size everything that depends on the number of states.
// M_MAX + 2 basis columns and one scratch column (option m_max caps it: at 10^8 states 105 columns are 90 GB)
// num_cols: the columns d_V HAS (option m_max only takes effect when the next generator re-lays the basis, so every
// bound and the scratch-column index follow v_mmax, the value the allocation was made with); mmax_now: the largest m
// a caller may use right now
inline int num_cols(const kfsp_ctx *c) { return (int)(c->v_mmax ? c->v_mmax : c->opt_mmax) + 3; }
inline int64_t mmax_now(const kfsp_ctx *c) { return c->v_mmax ? std::min(c->v_mmax, c->opt_mmax) : c->opt_mmax; }

// column j (0-based) of the basis: `margin` halo rows sit on either side of it
inline double *vcol(const kfsp_ctx *c, int j) { return c->d_V.p + (size_t)j * (size_t)c->ldv + (size_t)c->margin; }

int resize(kfsp_ctx *ctx, int64_t n)
{
    ctx->n = n;
    ctx->trip_order_n = 0;             // a trip order belongs to one generator
    ctx->L = round_up((n + ctx->nranks - 1) / ctx->nranks, kChunk);
    if (ctx->L == 0) ctx->L = kChunk;
    ctx->row0 = (int64_t)ctx->rank * ctx->L;
    ctx->nloc = std::max<int64_t>(0, std::min<int64_t>(ctx->L, n - ctx->row0));
    // The column stride only ever grows (with head room: the FSP usually keeps growing, and a drop
    // is followed by expansions): a changed stride would re-lay all 105 columns, and zero-filling
    // them (0.8 GB at 10^6 rows) on every FSP change is pure overhead - every pass rewrites the
    // rows [0, nchunks*64) of each column it uses, rows beyond them are never read, and w is
    // cleared by kfsp_set_vector.  Only fresh memory is zeroed.
    const int64_t need = round_up(ctx->L + 2 * ctx->margin, 256);
    int64_t ldv = ctx->ldv;
    if (need > ctx->ldv || ctx->relayout) {
        if (ctx->w_pending && !ctx->use_comm && !ctx->w_pending_full)
            return fail(ctx, -2, "the FSP grew between kfsp_drop_compact and the next generator");
        // (head room: half as much again, but no more than 2^24 rows - at 10^8 states 50 % would be 70 GB)
        ldv = ctx->relayout ? need : round_up(need + std::min<int64_t>(need / 2, (int64_t)1 << 24), 256);
        ctx->relayout = false;
        ctx->v_mmax = ctx->opt_mmax;
        HIP_TRY(ctx->d_V.reserve((size_t)ldv * num_cols(ctx), false));
        HIP_TRY(ctx->d_w.reserve((size_t)ldv, false));
        HIP_TRY(ctx->d_tmp.reserve((size_t)ldv, false));
        ctx->ldv = ldv;
        HIP_TRY(hipMemsetAsync(ctx->d_V.p, 0, (size_t)ldv * num_cols(ctx) * sizeof(double), ctx->stream));
        HIP_TRY(hipMemsetAsync(ctx->d_w.p, 0, (size_t)ldv * sizeof(double), ctx->stream));
        HIP_TRY(hipMemsetAsync(ctx->d_tmp.p, 0, (size_t)ldv * sizeof(double), ctx->stream));
    }
    const size_t xg = (size_t)std::max<int64_t>(ctx->L * ctx->nranks, ldv);
    if (xg > ctx->d_xg.cap) HIP_TRY(ctx->d_xg.reserve(xg + xg / 2, true));
    return 0;
}

// Upload gather rows given as per-row counts + a fill callback.
struct HostSell {
    std::vector<int64_t> off;
    std::vector<int32_t> col;
    std::vector<double> val, diag;
    int64_t nchunks = 0, nnz = 0;
};

void sell_layout(const std::vector<int32_t> &cnt, int64_t nloc, int64_t row0, HostSell &S)
{
    S.nchunks = (nloc + kChunk - 1) / kChunk;
    S.off.assign((size_t)S.nchunks + 1, 0);
    for (int64_t c = 0; c < S.nchunks; ++c) {
        int w = 0;
        const int64_t r1 = std::min<int64_t>(nloc, (c + 1) * kChunk);
        for (int64_t r = c * kChunk; r < r1; ++r) w = std::max(w, (int)cnt[(size_t)r]);
        S.off[(size_t)c + 1] = S.off[(size_t)c] + (int64_t)w * kChunk;
    }
    const size_t slots = (size_t)S.off[(size_t)S.nchunks];
    S.col.resize(slots);
    S.val.assign(slots, 0.0);
    // padded slots gather the row's own x (always a valid address)
    for (int64_t c = 0; c < S.nchunks; ++c) {
        const int64_t o = S.off[(size_t)c], w = (S.off[(size_t)c + 1] - o) / kChunk;
        for (int64_t k = 0; k < w; ++k)
            for (int l = 0; l < kChunk; ++l) {
                const int64_t r = std::min<int64_t>(c * kChunk + l, nloc > 0 ? nloc - 1 : 0);
                S.col[(size_t)kfsp::sell_pos(o, (int)w, (int)k, l)] = (int32_t)(row0 + r);
            }
    }
    S.diag.assign((size_t)S.nchunks * kChunk, 0.0);
}

int upload_sell(kfsp_ctx *ctx, const HostSell &S)
{
    ctx->nchunks = S.nchunks;
    ctx->slots = S.off[(size_t)S.nchunks];
    ctx->nnz = S.nnz;
    HIP_TRY(ctx->d_off.reserve(S.off.size(), false));
    HIP_TRY(ctx->d_col.reserve(std::max<size_t>(S.col.size(), 64), false));
    HIP_TRY(ctx->d_val.reserve(std::max<size_t>(S.val.size(), 64), false));
    HIP_TRY(ctx->d_diag.reserve(S.diag.size() + 2 * kChunk, false));
    HIP_TRY(hipMemsetAsync(ctx->d_diag.p, 0, (S.diag.size() + 2 * kChunk) * sizeof(double), ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_off.p, S.off.data(), S.off.size() * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    if (!S.col.empty()) {
        HIP_TRY(hipMemcpyAsync(ctx->d_col.p, S.col.data(), S.col.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->d_val.p, S.val.data(), S.val.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    if (!S.diag.empty())
        HIP_TRY(hipMemcpyAsync(ctx->d_diag.p, S.diag.data(), S.diag.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->have_sell = true;
    return kfsp::build_sell_code(ctx);
}

// Banded form: accepted when the local rows use at most kMaxDiag distinct
// column offsets and the diagonals are reasonably full.
int maybe_upload_dia(kfsp_ctx *ctx, const HostSell &S, const std::vector<int32_t> &cnt)
{
    ctx->use_dia = false;
    ctx->dia_masked = false;
    ctx->nd = 0;
    if (ctx->opt_format == 1 || ctx->nloc == 0) return 0;
    const int64_t nloc = ctx->nloc, row0 = ctx->row0;
    int nd = 0;
    int64_t delta[kMaxDiag];
    int64_t nnz_off = 0;
    for (int64_t r = 0; r < nloc; ++r) {
        const int64_t c = r / kChunk, l = r % kChunk, o = S.off[(size_t)c];
        const int w = (int)((S.off[(size_t)c + 1] - o) / kChunk);
        for (int k = 0; k < cnt[(size_t)r]; ++k) {
            const int64_t dl = (int64_t)S.col[(size_t)kfsp::sell_pos(o, w, k, (int)l)] - (row0 + r);
            int d = 0;
            while (d < nd && delta[d] != dl) ++d;
            if (d == nd) {
                if (nd == kMaxDiag) return 0;
                delta[nd++] = dl;
            }
            ++nnz_off;
        }
    }
    // a stored diagonal entry costs 8 bytes whether it is zero or not, a SELL slot 12: the
    // banded form moves fewer bytes as long as nd * rows < 1.5 * (off-diagonal entries)
    if (nd == 0 || (double)nd * (double)nloc > 1.5 * (double)nnz_off + 1024.0) return 0;
    std::sort(delta, delta + nd);
    const int64_t ld = round_up(ctx->nchunks * kChunk, 2 * kChunk);   // the banded kernel works on 128-row groups
    std::vector<double> val((size_t)nd * (size_t)ld, 0.0);
    for (int64_t r = 0; r < nloc; ++r) {
        const int64_t c = r / kChunk, l = r % kChunk, o = S.off[(size_t)c];
        const int w = (int)((S.off[(size_t)c + 1] - o) / kChunk);
        for (int k = 0; k < cnt[(size_t)r]; ++k) {
            const size_t pos = (size_t)kfsp::sell_pos(o, w, k, (int)l);
            const int64_t dl = (int64_t)S.col[pos] - (row0 + r);
            const int d = (int)(std::lower_bound(delta, delta + nd, dl) - delta);
            val[(size_t)d * (size_t)ld + (size_t)r] += S.val[pos];
        }
    }
    HIP_TRY(ctx->d_dia.reserve(val.size(), false));
    HIP_TRY(hipMemcpy(ctx->d_dia.p, val.data(), val.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->nd = nd;
    ctx->dia_ld = ld;
    for (int d = 0; d < nd; ++d) ctx->delta[d] = (int32_t)delta[d];
    ctx->use_dia = true;
    return kfsp::build_dia_mask(ctx);
}

// ---- the caller's state order <-> what the device keeps ----------------------------------------------
// Without an internal order and without a communicator a vector is a plain copy.  With an internal order
// (perm[new] = old, GLOBAL indices) a rank owns the block [row0, row0 + nloc) of the INTERNAL order, while
// the C ABI speaks of the same block of the CALLER's order: the two meet in a full-length vector that every
// rank assembles with one all-gather (set / get of vectors happen a few times per FSP change, not per product).

// full vector in the caller's order (device, n entries) -> this rank's block of a w-like vector
int scatter_from_full(kfsp_ctx *ctx, const double *full_caller, double *dev_local)
{
    if (ctx->nloc <= 0) return 0;
    if (ctx->perm_on)
        kfsp::launch_gather_index(ctx->nloc, ctx->d_perm.p + ctx->row0, full_caller, dev_local, ctx->stream);
    else
        HIP_TRY(hipMemcpyAsync(dev_local, full_caller + ctx->row0, (size_t)ctx->nloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// this rank's block of a w-like vector (readable for L entries) -> the full vector in the caller's order on
// every rank; *full points into d_xg or d_full (valid until either is used again)
int gather_to_full(kfsp_ctx *ctx, const double *dev_local, const double **full)
{
    const double *xg = dev_local;
    if (ctx->use_comm) {
        if (int rc = comm_allgather(ctx, dev_local, ctx->d_xg.p, (size_t)ctx->L, ctx->stream)) return rc;
        xg = ctx->d_xg.p;                                   // entry g of the internal order (blocks are contiguous: row0 = rank * L)
    }
    if (!ctx->perm_on) {
        *full = xg;
        return 0;
    }
    HIP_TRY(ctx->d_full.reserve((size_t)ctx->n + 64, false));
    kfsp::launch_gather_index(ctx->n, ctx->d_iperm.p, xg, ctx->d_full.p, ctx->stream);      // full[old] = xg[iperm[old]]
    *full = ctx->d_full.p;
    return 0;
}

// Host array (this rank's block of the caller's order) -> device vector and back.
int upload_states(kfsp_ctx *ctx, const double *host, double *dev, int64_t count)
{
    if (!ctx->perm_on) {
        if (count > 0)
            HIP_TRY(hipMemcpyAsync(dev, host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        return 0;
    }
    if (!ctx->use_comm) {
        if (count <= 0) return 0;
        HIP_TRY(ctx->d_pstage.reserve((size_t)count, false));
        HIP_TRY(hipMemcpyAsync(ctx->d_pstage.p, host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        kfsp::launch_gather_index(count, ctx->d_perm.p, ctx->d_pstage.p, dev, ctx->stream);
        return 0;
    }
    // blocks of the caller's order from all ranks, then this rank's block of the internal order
    HIP_TRY(ctx->d_pstage.reserve((size_t)ctx->L, false));
    HIP_TRY(hipMemsetAsync(ctx->d_pstage.p, 0, (size_t)ctx->L * sizeof(double), ctx->stream));
    if (count > 0)
        HIP_TRY(hipMemcpyAsync(ctx->d_pstage.p, host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = comm_allgather(ctx, ctx->d_pstage.p, ctx->d_xg.p, (size_t)ctx->L, ctx->stream)) return rc;
    return scatter_from_full(ctx, ctx->d_xg.p, dev);
}

int download_states(kfsp_ctx *ctx, const double *dev, double *host, int64_t count)
{
    if (!ctx->perm_on) {
        if (count > 0)
            HIP_TRY(hipMemcpyAsync(host, dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        return 0;
    }
    if (!ctx->use_comm) {
        if (count <= 0) return 0;
        HIP_TRY(ctx->d_pstage.reserve((size_t)count, false));
        kfsp::launch_gather_index(count, ctx->d_iperm.p, dev, ctx->d_pstage.p, ctx->stream);
        HIP_TRY(hipMemcpyAsync(host, ctx->d_pstage.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        return 0;
    }
    const double *full = nullptr;
    if (int rc = gather_to_full(ctx, dev, &full)) return rc;
    if (count > 0)
        HIP_TRY(hipMemcpyAsync(host, full + ctx->row0, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// A vector compacted on the device (kfsp_drop_compact) becomes the resident w of the generator
// that was just set: caller's order -> the order the device keeps this generator in.  With a communicator the
// pending vector is the FULL compacted vector (every rank holds it) and each rank takes its new block.
int adopt_pending_vector(kfsp_ctx *ctx)
{
    if (!ctx->w_pending) return 0;
    ctx->w_pending = false;
    if (ctx->w_pending_n != ctx->n) return fail(ctx, -2, "generator size does not match the vector compacted by kfsp_drop_compact");
    HIP_TRY(hipMemsetAsync(ctx->d_w.p, 0, (size_t)ctx->ldv * sizeof(double), ctx->stream));
    if (ctx->use_comm || ctx->w_pending_full) {
        ctx->w_pending_full = false;
        if (int rc = scatter_from_full(ctx, ctx->d_wfull.p, ctx->d_w.p)) return rc;
    } else if (ctx->perm_on)
        kfsp::launch_gather_index(ctx->n, ctx->d_perm.p, ctx->d_tmp.p, ctx->d_w.p, ctx->stream);
    else
        HIP_TRY(hipMemcpyAsync(ctx->d_w.p, ctx->d_tmp.p, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    // (one context works on one stream: whatever reads the vector next is ordered behind this)
    if (!ctx->opt_build_speculate || ctx->use_comm) HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// stream, events and the fixed-size buffers of a fresh context
int init_context(kfsp_ctx *ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));
    HIP_TRY(ctx->d_part.reserve((size_t)kNumPartial * kMaxGrid, true));
    HIP_TRY(ctx->d_stage.reserve(kNumStage, true));
    HIP_TRY(ctx->d_H.reserve((size_t)kMH * kMH + 2, true));
    HIP_TRY(ctx->d_sq.reserve(kMH + 2, true));
    HIP_TRY(ctx->d_g.reserve(kMH + 2, true));
    HIP_TRY(ctx->d_y.reserve(kMH, true));
    HIP_TRY(ctx->d_flag.reserve(4, true));
    HIP_TRY(ctx->d_zero.reserve(128, true));
    {
        int v = 0;
        HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device));
        ctx->lds_per_block = v;
    }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_H), ((size_t)kMH * kMH + 2) * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_pin), (size_t)(kMH + 8) * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_build), 8192, hipHostMallocDefault));
    std::memset(ctx->h_H, 0, ((size_t)kMH * kMH + 2) * sizeof(double));
    return 0;
}

}  // namespace

namespace {
// The order in which a product over a lexicographic box takes its 128-row trips when the box is too large for the caches
// (option "box_tile": -1 auto, 0 never, 1 always): rows are cut below the slowest stride that is still at most 16 K rows
// (cut), r = hi * S_cut + lo; blocks of B = 1024 rows of lo; per block ALL lines hi back to back.  A row's +-S neighbours
// for the strides above the cut are then whole lines away - one line = B rows = 8 KB of x - instead of whole strides:
// 22^6 with the cut below species 4: the +-22^3 / 22^4 / 22^5 neighbours sit 1 / 22 / 484 lines = 8 KB / 180 KB / 4 MB
// from the row in the order of the sweep.  Empty: keep the ascending order.
std::vector<int32_t> box_tile_order(int ns, const int32_t *dims, int64_t n, bool force)
{
    std::vector<int32_t> out;
    const int64_t trips = (n + 127) / 128;
    if (ns < 3 || trips < 2 || trips > INT32_MAX) return out;
    int64_t stride = 1, smax = 1;
    int cut = -1;
    int64_t scut = 1;
    for (int s = 0; s < ns; ++s) {
        if (stride <= 16384 && s > 0) {
            cut = s;
            scut = stride;
        }
        smax = stride;
        stride *= dims[s];
    }
    const double mall = 256.0 * 1024 * 1024;
    const bool large = (double)n * 8.0 > mall && 8.0 * 2.0 * (double)smax * 8.0 > mall;
    if (cut < 1 || scut < 2048 || !(force || large)) return out;
    const int64_t B = 1024, nhi = (n + scut - 1) / scut;
    std::vector<std::pair<int64_t, int32_t>> key((size_t)trips);
    for (int64_t c = 0; c < trips; ++c) {
        const int64_t r = c * 128, lo = r % scut, hi = r / scut;
        key[(size_t)c] = {((lo / B) * nhi + hi) * scut + lo, (int32_t)c};
    }
    std::sort(key.begin(), key.end());
    out.resize((size_t)trips);
    for (int64_t c = 0; c < trips; ++c) out[(size_t)c] = key[(size_t)c].second;
    return out;
}

}  // namespace

extern "C" {

int kfsp_abi_version(void) { return kAbiVersion; }

int kfsp_create(int device, kfsp_ctx **out)
{
    if (!out) return -2;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return 1000 + (int)(e == hipSuccess ? hipErrorNoDevice : e);
    if (device < 0 || device >= count) return -1;
    kfsp_ctx *ctx = new (std::nothrow) kfsp_ctx;
    if (!ctx) return 4001;
    ctx->device = device;
    const int rc = init_context(ctx);
    if (rc != 0) {
        (void)kfsp_destroy(ctx);        // releases whatever init_context got as far as creating
        return rc;
    }
    *out = ctx;
    return 0;
}

int kfsp_create_group(int nranks, const int *devices, kfsp_ctx **out)
{
    try {
        return kfsp::group_create(nranks, devices, out);
    } catch (...) {
        return 4000;
    }
}

int kfsp_group_selftest(int nranks, int failing_rank, int hanging_rank, int work_ms, int hang_ms, int timeout_ms, int grace_ms,
                        int settle_ms, int *rc_out, int *who_out, double *seconds, int *broken, int *stuck)
{
    try {
        return kfsp::group_selftest(nranks, failing_rank, hanging_rank, work_ms, hang_ms, timeout_ms, grace_ms, settle_ms, rc_out,
                                    who_out, seconds, broken, stuck);
    } catch (...) {
        return 4000;
    }
}

int kfsp_group_size(const kfsp_ctx *ctx, int *nranks)
{
    if (!ctx) return -1;
    if (!nranks) return -2;
    *nranks = ctx->group ? kfsp::group_size(ctx) : 1;
    return 0;
}

int kfsp_destroy(kfsp_ctx *ctx)
{
    if (!ctx) return 0;
    if (ctx->group) return kfsp::group_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    if (ctx->ev_src) (void)hipEventDestroy(ctx->ev_src);
    if (ctx->ev_halo) (void)hipEventDestroy(ctx->ev_halo);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    ctx->d_off.release(); ctx->d_col.release(); ctx->d_val.release(); ctx->d_diag.release();
    ctx->d_V.release(); ctx->d_w.release(); ctx->d_xg.release(); ctx->d_tmp.release();
    ctx->d_full.release(); ctx->d_wfull.release(); ctx->d_flagloc.release();
    ctx->d_dtab.release(); ctx->d_dtlen.release(); ctx->d_code.release(); ctx->d_codeoff.release(); ctx->d_trip_order.release();
    ctx->d_part.release(); ctx->d_stage.release(); ctx->d_H.release(); ctx->d_sq.release();
    ctx->d_y.release(); ctx->d_flag.release(); ctx->d_g.release(); ctx->d_dia.release();
    ctx->d_ell_adj.release(); ctx->d_ell_off.release(); ctx->d_ell_diag.release(); ctx->d_cnt.release();
    ctx->d_ticket.release(); ctx->d_slot.release(); ctx->d_scan.release(); ctx->d_strip.release();
    ctx->d_dropflag.release(); ctx->d_dropcnt.release(); ctx->d_box.release(); ctx->d_os1.release(); ctx->d_os2.release();
    ctx->d_pencil_order.release(); ctx->d_os3.release(); ctx->d_os4.release(); ctx->d_os5.release(); ctx->d_prop_i.release(); ctx->d_prop_d.release();
    ctx->d_prop_t2i.release(); ctx->d_prop_t2o.release(); ctx->d_prop_t2d.release(); ctx->d_prop_oob.release();
    ctx->d_perm.release(); ctx->d_iperm.release(); ctx->d_coords.release(); ctx->d_coords2.release(); ctx->d_ell_adj2.release();
    ctx->d_ell_off2.release(); ctx->d_ell_diag2.release(); ctx->d_pstage.release(); ctx->d_keys.release();
    ctx->d_sortidx.release(); ctx->d_sorttmp.release(); ctx->d_gmask.release(); ctx->d_zero.release();
    ctx->d_skeys.release(); ctx->d_skeys2.release(); ctx->d_perm2.release(); ctx->d_prop_fast_i.release(); ctx->d_prop_fast_d.release();
    if (ctx->h_loop) (void)hipHostFree(ctx->h_loop);
    if (ctx->h_H) (void)hipHostFree(ctx->h_H);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_build) (void)hipHostFree(ctx->h_build);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

const char *kfsp_last_error(const kfsp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int kfsp_comm_unique_id(void *id_bytes)
{
    if (!id_bytes) return -1;
    static_assert(sizeof(ncclUniqueId) == KFSP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return 2000 + (int)r;
    std::memcpy(id_bytes, &id, sizeof(id));
    return 0;
}

int kfsp_comm_init(kfsp_ctx *ctx, int nranks, int rank, const void *id_bytes)
{
    if (!ctx) return -1;
    if (nranks < 1) return fail(ctx, -2, "nranks < 1");
    if (rank < 0 || rank >= nranks) return fail(ctx, -3, "rank out of range");
    if (nranks > 1 && !id_bytes) return fail(ctx, -4, "null unique id");
    if (ctx->group) return fail(ctx, -9, "a group context makes its own communicator");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm) {
        // (abort, not destroy: the usual reason to come here twice is a communicator that returned an
        // error on some rank, and destroying one of those can wait for ever)
        (void)ncclCommAbort(ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->loop = nullptr;
    ctx->comm_aborted = false;
    ctx->nranks = nranks;
    ctx->rank = rank;
    // a unique id with nranks == 1 still creates a (one-rank) communicator, so the
    // collective code path can be exercised on a single GPU
    ctx->use_comm = false;
    if (id_bytes) {
        ncclUniqueId id;
        std::memcpy(&id, id_bytes, sizeof(id));
        NCCL_TRY(ncclCommInitRank(&ctx->comm, nranks, id, rank));
        ctx->use_comm = true;
        if (!ctx->comm_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_src, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
        }
    }
    ctx->relayout = true;   // new margins / block length: re-lay the basis on the next matrix
    ctx->use_halo = false;
    return 0;
}

int kfsp_loopback_create(int nranks, void **group)
{
    if (nranks < 1 || nranks > 64) return -1;
    if (!group) return -2;
    kfsp::LoopGroup *g = new (std::nothrow) kfsp::LoopGroup;
    if (!g) return 4001;
    g->n = nranks;
    g->slot.assign((size_t)nranks, nullptr);
    *group = g;
    return 0;
}

int kfsp_loopback_destroy(void *group)
{
    delete static_cast<kfsp::LoopGroup *>(group);
    return 0;
}

int kfsp_comm_init_loopback(kfsp_ctx *ctx, void *group, int rank)
{
    if (!ctx) return -1;
    if (!group) return fail(ctx, -2, "null group");
    kfsp::LoopGroup *g = static_cast<kfsp::LoopGroup *>(group);
    if (rank < 0 || rank >= g->n) return fail(ctx, -3, "rank out of range");
    if (ctx->group) return fail(ctx, -9, "a group context makes its own communicator");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm) {
        (void)ncclCommAbort(ctx->comm);   // as kfsp_comm_init: destroying a failed communicator can wait for ever
        ctx->comm = nullptr;
    }
    ctx->loop = g;
    ctx->comm_aborted = false;
    ctx->nranks = g->n;
    ctx->rank = rank;
    ctx->use_comm = true;
    if (!ctx->h_loop)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_loop), (size_t)(kLoopScratch + 8) * sizeof(double), hipHostMallocDefault));
    if (!ctx->comm_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_src, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
    }
    ctx->relayout = true;   // new margins / block length: re-lay the basis on the next matrix
    ctx->use_halo = false;
    return 0;
}

int kfsp_partition(int64_t n, int nranks, int rank, int64_t *row0, int64_t *nrows, int64_t *block_len)
{
    if (n < 0) return -1;
    if (nranks < 1) return -2;
    if (rank < 0 || rank >= nranks) return -3;
    int64_t L = round_up((n + nranks - 1) / nranks, kChunk);
    if (L == 0) L = kChunk;
    const int64_t r0 = (int64_t)rank * L;
    if (row0) *row0 = std::min(r0, n);
    if (nrows) *nrows = std::max<int64_t>(0, std::min<int64_t>(L, n - r0));
    if (block_len) *block_len = L;
    return 0;
}

int kfsp_row_block(const kfsp_ctx *ctx, int64_t n, int64_t *row0, int64_t *nrows)
{
    if (!ctx) return -1;
    if (n < 0) return -2;
    // (a group head owns all rows, like a one-rank context: nranks = 1, rank = 0)
    return kfsp_partition(n, ctx->nranks, ctx->rank, row0, nrows, nullptr) ? -2 : 0;
}

static int set_matrix_ell_impl(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, const int32_t *adj,
                               const double *offdiag, const double *diag, int64_t keep)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (n < 1) return fail(ctx, -2, "n < 1");
        if (bw < 1) return fail(ctx, -3, "bw < 1");
        if (ld < bw) return fail(ctx, -4, "ld < bw");
        if (!adj) return fail(ctx, -5, "null adj");
        if (!offdiag) return fail(ctx, -6, "null offdiag");
        if (!diag) return fail(ctx, -7, "null diag");
        if (ctx->group) return kfsp::group_update_matrix_ell(ctx, n, bw, ld, adj, offdiag, diag, (int32_t)keep);
        HIP_TRY(hipSetDevice(ctx->device));
        auto t0 = std::chrono::steady_clock::now();
        ctx->use_box = false;
        ctx->box_lds_bytes = 0;
        if (int rc = resize(ctx, n)) return rc;
        const int64_t row0 = ctx->row0, nloc = ctx->nloc;
        // coordinates handed over for exactly this generator switch the internal order on
        ctx->perm_on = ctx->perm_pending_n == n && !ctx->opt_host_build;
        ctx->perm_pending_n = 0;
        ctx->prod_last = ctx->prod_count;
        ctx->prod_count = 0;

        if (!ctx->opt_host_build) {
            // the arrays go to HBM verbatim and are transposed there (kfsp_build.hip)
            int rc = build_from_ell_device(ctx, n, bw, ld, adj, offdiag, diag, keep);
            if (!rc) rc = setup_exchange(ctx);
            if (!rc) rc = adopt_pending_vector(ctx);
            ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return rc;
        }

        // host transpose (kept for A/B checks of the device build)
        ctx->ell_cols = 0;
        // in-degree of every local row
        std::vector<int32_t> cnt((size_t)std::max<int64_t>(nloc, 1), 0);
        for (int64_t i = 0; i < n; ++i) {
            const int32_t *a = adj + (size_t)i * ld;
            for (int j = 0; j < bw; ++j) {
                const int64_t k = a[j];
                if (k > n) return fail(ctx, -5, "adj entry exceeds n");
                if (k >= 1) {
                    const int64_t r = k - 1 - row0;
                    if (r >= 0 && r < nloc) ++cnt[(size_t)r];
                }
            }
        }
        HostSell S;
        sell_layout(cnt, nloc, row0, S);
        std::vector<int32_t> fill((size_t)std::max<int64_t>(nloc, 1), 0);
        int64_t nnz = nloc;
        // sources in increasing order: each row's entries end up sorted by column,
        // the order in which FMATVEC (:598-604) accumulates them
        for (int64_t i = 0; i < n; ++i) {
            const int32_t *a = adj + (size_t)i * ld;
            const double *o = offdiag + (size_t)i * ld;
            for (int j = 0; j < bw; ++j) {
                const int64_t k = a[j];
                if (k < 1) continue;
                const int64_t r = k - 1 - row0;
                if (r < 0 || r >= nloc) continue;
                const int64_t c = r / kChunk, l = r % kChunk;
                const int64_t pos = kfsp::sell_pos(S.off[(size_t)c], (int)((S.off[(size_t)c + 1] - S.off[(size_t)c]) / kChunk), fill[(size_t)r]++, (int)l);
                S.col[(size_t)pos] = (int32_t)i;
                S.val[(size_t)pos] = o[j];
                ++nnz;
            }
        }
        for (int64_t r = 0; r < nloc; ++r) S.diag[(size_t)r] = diag[(size_t)(row0 + r)];
        S.nnz = nnz;
        if (int rc = upload_sell(ctx, S)) return rc;
        if (int rc = maybe_upload_dia(ctx, S, cnt)) return rc;
        if (int rc = setup_exchange(ctx)) return rc;
        if (int rc = adopt_pending_vector(ctx)) return rc;
        ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    });
}

int kfsp_set_matrix_ell(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, const int32_t *adj,
                        const double *offdiag, const double *diag)
{
    return set_matrix_ell_impl(ctx, n, bw, ld, adj, offdiag, diag, 0);
}

int kfsp_update_matrix_ell(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, const int32_t *adj,
                           const double *offdiag, const double *diag, int32_t n_unchanged)
{
    if (n_unchanged < 0 || n_unchanged > n) return ctx ? fail(ctx, -8, "0 <= n_unchanged <= n") : -1;
    return set_matrix_ell_impl(ctx, n, bw, ld, adj, offdiag, diag, n_unchanged);
}

int kfsp_set_matrix_csr(kfsp_ctx *ctx, int64_t n, int64_t row0, int64_t nrows, const int64_t *rowptr,
                        const int32_t *col, const double *val)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (n < 1 || n > 2147483647LL - 512) return fail(ctx, -2, "n out of range");
        if (ctx->group) return kfsp::group_set_matrix_csr(ctx, n, row0, nrows, rowptr, col, val);
        HIP_TRY(hipSetDevice(ctx->device));
        ctx->perm_on = false;
        ctx->perm_pending_n = 0;
        ctx->prod_last = ctx->prod_count;
        ctx->prod_count = 0;
        ctx->use_box = false;
        ctx->box_lds_bytes = 0;
        ctx->ell_cols = 0;
        if (int rc = resize(ctx, n)) return rc;
        if (row0 != std::min(ctx->row0, n)) return fail(ctx, -3, "row0 is not this rank's block start (kfsp_row_block)");
        if (nrows != ctx->nloc) return fail(ctx, -4, "nrows is not this rank's block size (kfsp_row_block)");
        if (!rowptr) return fail(ctx, -5, "null rowptr");
        if (nrows > 0 && (!col || !val)) return fail(ctx, -6, "null col/val");
        if (rowptr[0] != 0) return fail(ctx, -5, "rowptr[0] != 0");
        auto t0 = std::chrono::steady_clock::now();
        const int64_t nloc = ctx->nloc;
        std::vector<int32_t> cnt((size_t)std::max<int64_t>(nloc, 1), 0);
        for (int64_t r = 0; r < nloc; ++r) {
            if (rowptr[r + 1] < rowptr[r]) return fail(ctx, -5, "rowptr not monotone");
            int c = 0;
            for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                if (col[p] < 0 || col[p] >= n) return fail(ctx, -6, "column index out of range");
                if (col[p] != row0 + r) ++c;
            }
            cnt[(size_t)r] = c;
        }
        HostSell S;
        sell_layout(cnt, nloc, ctx->row0, S);
        for (int64_t r = 0; r < nloc; ++r) {
            const int64_t c = r / kChunk, l = r % kChunk;
            int k = 0;
            double d = 0.0;
            for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                if (col[p] == row0 + r) {
                    d += val[p];
                } else {
                    const int64_t pos = kfsp::sell_pos(S.off[(size_t)c], (int)((S.off[(size_t)c + 1] - S.off[(size_t)c]) / kChunk), k++, (int)l);
                    S.col[(size_t)pos] = col[p];
                    S.val[(size_t)pos] = val[p];
                }
            }
            S.diag[(size_t)r] = -d;   // kept positive like DIAG (StateSpace.f90:16)
        }
        S.nnz = rowptr[nloc];
        if (int rc = upload_sell(ctx, S)) return rc;
        if (int rc = maybe_upload_dia(ctx, S, cnt)) return rc;
        if (int rc = setup_exchange(ctx)) return rc;
        if (int rc = adopt_pending_vector(ctx)) return rc;
        ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    });
}

int kfsp_set_matrix_box(kfsp_ctx *ctx, int32_t ns, const int32_t *dims, int32_t nr, const int32_t *stoich,
                        const int32_t *ndep, const int32_t *dep_species, const double *tables)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ns < 1 || ns > kfsp::kBoxMaxS) return fail(ctx, -2, "1 <= ns <= 8");
        if (!dims) return fail(ctx, -3, "null dims");
        if (nr < 1 || nr > kfsp::kBoxMaxR) return fail(ctx, -4, "1 <= nr <= 16");
        if (!stoich || !ndep || !dep_species) return fail(ctx, -5, "null reaction description");
        if (!tables) return fail(ctx, -8, "null tables");
        if (ctx->group) return kfsp::group_set_matrix_box(ctx, ns, dims, nr, stoich, ndep, dep_species, tables);
        int64_t n = 1, stride[kfsp::kBoxMaxS];
        for (int s = 0; s < ns; ++s) {
            if (dims[s] < 1) return fail(ctx, -3, "dims must be positive");
            stride[s] = n;
            n *= dims[s];
            if (n > 2147483647LL - 512) return fail(ctx, -3, "box has more than 2^31 states");
        }
        if (n < 2) return fail(ctx, -3, "box has fewer than 2 states");
        HIP_TRY(hipSetDevice(ctx->device));
        auto t0 = std::chrono::steady_clock::now();
        ctx->perm_on = false;
        ctx->perm_pending_n = 0;
        ctx->prod_last = ctx->prod_count;
        ctx->prod_count = 0;
        if (int rc = resize(ctx, n)) return rc;
        // reactions sorted by the column offset of their entry (ascending, like the stored diagonals)
        struct R { int64_t delta; int k; };
        std::vector<R> order((size_t)nr);
        for (int k = 0; k < nr; ++k) {
            int64_t d = 0;
            for (int s = 0; s < ns; ++s) d -= (int64_t)stoich[(size_t)k * ns + s] * stride[s];
            order[(size_t)k] = R{d, k};
        }
        std::stable_sort(order.begin(), order.end(), [](const R &a, const R &b) { return a.delta < b.delta; });
        kfsp::BoxDev B;
        std::memset(&B, 0, sizeof(B));
        B.ns = ns;
        B.nr = nr;
        for (int s = 0; s < ns; ++s) {
            B.dims[s] = dims[s];
            B.inv_dim[s] = 1.0 / (double)dims[s];
        }
        // table offsets in the order the caller concatenated them: reaction by reaction, factor by factor
        std::vector<int32_t> toff((size_t)nr * kfsp::kBoxMaxDep, 0);
        int64_t ntab = 0;
        for (int k = 0; k < nr; ++k) {
            if (ndep[k] < 1 || ndep[k] > kfsp::kBoxMaxDep) return fail(ctx, -6, "1 <= ndep <= 3 factors per propensity");
            for (int i = 0; i < ndep[k]; ++i) {
                const int s = dep_species[(size_t)k * kfsp::kBoxMaxDep + i];
                if (s < 0 || s >= ns) return fail(ctx, -7, "factor species out of range");
                toff[(size_t)k * kfsp::kBoxMaxDep + i] = (int32_t)ntab;
                ntab += dims[s];
            }
        }
        const int head = kfsp::kBoxImageHead;              // image = [0.0, 0.0][factor tables][fast form's species tables]
        if (head + ntab > 6000) return fail(ctx, -8, "factor tables exceed the 48 KB they may take in LDS");
        for (int p = 0; p < nr; ++p) {
            const int k = order[(size_t)p].k;
            B.delta[p] = (int32_t)order[(size_t)p].delta;
            B.ndep[p] = (int8_t)ndep[k];
            for (int i = 0; i < ndep[k]; ++i) {
                const int s = dep_species[(size_t)k * kfsp::kBoxMaxDep + i];
                B.dep_s[p][i] = (int8_t)s;
                B.dep_nu[p][i] = (int8_t)stoich[(size_t)k * ns + s];
                B.dep_off[p][i] = head + toff[(size_t)k * kfsp::kBoxMaxDep + i];
            }
            int nm = 0;
            for (int s = 0; s < ns; ++s) {
                const int v = stoich[(size_t)k * ns + s];
                if (v == 0) continue;
                if (nm == kfsp::kBoxMaxDep) return fail(ctx, -5, "a reaction changes more than 3 species");
                if (v < -100 || v > 100) return fail(ctx, -5, "stoichiometry out of range");
                B.mov_s[p][nm] = (int8_t)s;
                B.mov_nu[p][nm] = (int8_t)v;
                B.mov_dim[p][nm] = dims[s];
                ++nm;
            }
            B.nmov[p] = (int8_t)nm;
            B.dorder[k] = p;                               // original reaction k sits at sorted position p
        }
        // single-factor fast form: every propensity one factor, no species changes by more than 2, at
        // most kBoxFastPer propensities per species, the entries of a row within 2^32 bytes of x
        kfsp::BoxFast F;
        std::memset(&F, 0, sizeof(F));
        bool fast = ns <= kfsp::kBoxFastS;
        int per_species[kfsp::kBoxMaxS] = {0};
        for (int k = 0; k < nr && fast; ++k) {
            fast = ndep[k] == 1;
            for (int s = 0; s < ns; ++s) fast = fast && std::abs(stoich[(size_t)k * ns + s]) <= 2;
            if (fast) fast = ++per_species[dep_species[(size_t)k * kfsp::kBoxMaxDep]] <= kfsp::kBoxFastPer;
        }
        const int64_t back = std::max<int64_t>(0, -order.front().delta), fwd = std::max<int64_t>(0, order.back().delta);
        if (back + fwd + 130 >= (1LL << 28)) fast = false;
        int per = 2, ns_inst = kfsp::kBoxFastS;
        int64_t df_at[kfsp::kBoxFastS] = {0};
        int64_t nimage = head + ntab;
        if (fast) {
            for (int s = 0; s < ns; ++s) per = std::max(per, per_species[s]);
            if (per > 2) per = kfsp::kBoxFastPer;
            // instantiations (launch_spmv): 2, 3 or 6 species with 2 slots each, else 6 species with 4 slots
            ns_inst = (per == 2 && ns <= 2) ? 2 : (per == 2 && ns == 3) ? 3 : kfsp::kBoxFastS;
            nimage += nimage & 1;
            for (int s = 0; s < ns_inst; ++s) {
                df_at[s] = nimage;
                nimage += 2 * (s < ns ? dims[s] : 1);
            }
            if (nimage > 8180) {                                  // 64 KB of LDS less the kernel's own few words
                fast = false;
                nimage = head + ntab;
            }
        }
        std::vector<double> image((size_t)nimage + (sizeof(kfsp::BoxFast) + 7) / 8, 0.0);
        std::memcpy(image.data() + head, tables, (size_t)ntab * sizeof(double));
        if (fast) {
            F.ns = ns_inst;
            F.per = per;
            F.bias8 = (int32_t)(8 * back);
            for (int s = 0; s < kfsp::kBoxFastS; ++s) {
                F.dims[s] = s < ns ? dims[s] : 1;                 // missing species: one population count, 0
                F.inv_dim[s] = 1.0 / (double)F.dims[s];
                F.df8[s] = (int32_t)(8 * df_at[s]);
            }
            struct DF { double dsum; uint32_t valid, pad; };
            static_assert(sizeof(DF) == 16, "layout of the kernel's BoxDF");
            int fill[kfsp::kBoxFastS] = {0};
            int entry_k[kfsp::kBoxFastS * kfsp::kBoxFastPer];
            for (int e = 0; e < ns_inst * per; ++e) entry_k[e] = -1;
            for (int p = 0; p < nr; ++p) {                        // ascending column offset within a species
                const int k = order[(size_t)p].k;
                const int s = dep_species[(size_t)k * kfsp::kBoxMaxDep];
                const int j = fill[s]++;
                F.koff8[s][j] = 8 * (head + toff[(size_t)k * kfsp::kBoxMaxDep] - stoich[(size_t)k * ns + s]);
                F.delta8[s][j] = (int32_t)(8 * order[(size_t)p].delta);
                entry_k[s * per + j] = k;
            }
            for (int s = 0; s < ns_inst; ++s) {
                const int d = s < ns ? dims[s] : 1;
                DF *df = reinterpret_cast<DF *>(image.data() + df_at[s]);
                for (int i = 0; i < d; ++i) {
                    double sum = 0.0;
                    uint32_t valid = 0;
                    for (int e = 0; e < ns_inst * per; ++e) {
                        const int k = entry_k[e];
                        if (k < 0) continue;                      // unused slot: never valid
                        const int v = s < ns ? stoich[(size_t)k * ns + s] : 0;   // source coordinate = i - v
                        if (i - v >= 0 && i - v < d) valid |= 1u << e;
                        if (e / per == s) sum += tables[(size_t)toff[(size_t)k * kfsp::kBoxMaxDep] + i];
                    }
                    df[i] = DF{sum, valid, 0u};
                }
            }
            B.pad = ns_inst * 16 + per;
        }
        B.ntab = (int32_t)nimage;
        std::memcpy(image.data() + nimage, &F, sizeof(F));
        HIP_TRY(ctx->d_box.reserve(image.size() + 8, false));
        HIP_TRY(hipMemcpy(ctx->d_box.p, image.data(), image.size() * sizeof(double), hipMemcpyHostToDevice));
        ctx->box = B;
        ctx->box_fast = fast;
        // Format 7 (pencils, kfsp_kernels.hip): the instantiation's species are the model's (no padding species), the slowest
        // species' own entries reach exactly one plane (or are unused slots), no other entry moves the slowest species, planes
        // have an even number of rows (16-byte pairs), one rank, and there are enough base trips to fill the chip.
        ctx->box_pencil = false;
        ctx->pencil_order_n = 0;
        if (fast && ns >= 3 && ns == ns_inst && ctx->nranks == 1 && !ctx->use_comm) {
            const int Ls = ns - 1;
            int64_t plane = 1;
            for (int s = 0; s < Ls; ++s) plane *= dims[s];
            // (1: also small boxes - tests.  Nothing else is required of the reactions: an entry that does not fit the register
            // scheme - it depends on the slowest species but moves another, or moves the slowest by two - is gathered from
            // memory as in format 4; entries of other species that move the slowest one get that species' valid bit per step)
            const bool ok = (plane % 2 == 0) && dims[Ls] >= 2 && ((plane + 127) / 128 >= 4096 || ctx->opt_box_pencil > 0);
            bool simple = true;
            for (int k = 0; k < nr; ++k)
                if (dep_species[(size_t)k * kfsp::kBoxMaxDep] != Ls && stoich[(size_t)k * ns + Ls] != 0) simple = false;
            ctx->pencil_simple = simple;
            // Format 8 (slabs): additionally the lines of the second-slowest species have an even number of rows, there are at
            // least two of them, and at least 256 workgroups' worth of slabs (or the option insists)
            ctx->box_slab = false;
            if (ok && ns >= 3) {
                int64_t line = 1;
                for (int s = 0; s < Ls - 1; ++s) line *= dims[s];
                const int lines = dims[Ls - 1];
                const int wmax = (int)std::max<int64_t>(1, std::min<int64_t>(12, ctx->opt_box_slab_waves));   // (<= 12 wavefronts per workgroup)
                const int groups = (lines + wmax - 1) / wmax, waves = (lines + groups - 1) / groups;
                const int64_t lo_trips = (line + 127) / 128;
                if (line % 2 == 0 && lines >= 2 && (lo_trips * groups >= 256 || ctx->opt_box_pencil == 2)) {
                    ctx->box_slab = true;
                    ctx->slab_line_rows = line;
                    ctx->slab_lines = lines;
                    ctx->slab_groups = groups;
                    ctx->slab_waves = waves;
                    ctx->slab_lo_trips = lo_trips;
                }
            }
            if (ok) {
                ctx->box_pencil = true;
                ctx->pencil_plane_rows = plane;
                ctx->pencil_planes = dims[Ls];
                ctx->pencil_trips = (plane + 127) / 128;
                // the base trips in small tiles of the plane (the same rule as box_tile_order, one species fewer)
                const std::vector<int32_t> po = box_tile_order(Ls, dims, plane, plane * 8 > (int64_t)(4 << 20));
                if (!po.empty()) {
                    HIP_TRY(ctx->d_pencil_order.reserve(po.size(), false));
                    HIP_TRY(hipMemcpy(ctx->d_pencil_order.p, po.data(), po.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                    ctx->pencil_order_n = (int64_t)po.size();
                }
            }
        }
        // format 6: the largest shift within opt_box_reach rows decides how much of x a workgroup stages in LDS;
        // the windows must fit beside the table image in the 64 KB a workgroup gets without asking for more
        ctx->box_reach = 0;
        if (fast) {
            int64_t reach = 0;
            for (int p = 0; p < nr; ++p) {
                const int64_t d = std::llabs((long long)order[(size_t)p].delta);
                if (d <= ctx->opt_box_reach) reach = std::max(reach, d);
            }
            reach = (reach + 1) & ~(int64_t)1;
            const size_t need = (((size_t)nimage * 8 + 15) & ~(size_t)15) + 2 * (size_t)(512 + 2 * reach) * 8 + 256;
            if (reach > 0 && need <= 64 * 1024) ctx->box_reach = (int)reach;
        }
        ctx->box_lds_bytes = (size_t)nimage * sizeof(double);
        // the same bookkeeping as a banded generator with one diagonal per reaction
        ctx->nchunks = (ctx->nloc + kChunk - 1) / kChunk;
        ctx->slots = 0;
        int64_t nnz = n;
        for (int k = 0; k < nr; ++k) {
            int64_t c = 1;
            for (int s = 0; s < ns; ++s) c *= std::max<int64_t>(0, dims[s] - std::abs(stoich[(size_t)k * ns + s]));
            nnz += c;
        }
        ctx->nnz = ctx->nranks == 1 ? nnz : 0;             // (per-rank counts are not tracked for boxes)
        ctx->use_dia = true;
        ctx->use_box = true;
        ctx->ell_cols = 0;
        ctx->dia_masked = false;
        ctx->have_sell = false;
        ctx->sell_coded = false;
        ctx->nd = nr;
        for (int p = 0; p < nr; ++p) ctx->delta[p] = B.delta[p];
        ctx->dia_ld = round_up(ctx->nchunks * kChunk, 2 * kChunk);
        HIP_TRY(ctx->d_diag.reserve((size_t)ctx->dia_ld + 2 * kChunk, true));
        if (ctx->opt_box_store) {
            // the same generator as STORED diagonals, written by the device from the tables: from here on an
            // ordinary banded generator (no host arrays of the size of the FSP ever exist)
            if (int rc = kfsp::box_materialize(ctx)) return rc;
            ctx->use_box = false;
            ctx->box_fast = false;
            ctx->box_lds_bytes = 0;
            if (int rc = kfsp::build_dia_mask(ctx)) return rc;
        }
        if (int rc = setup_exchange(ctx)) return rc;
        if (int rc = adopt_pending_vector(ctx)) return rc;
        ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    });
}

static int set_state_coords_impl(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state, int64_t keep)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (n < 1) return fail(ctx, -2, "n < 1");
        if (ns < 1) return fail(ctx, -3, "ns < 1");
        if (ld < ns) return fail(ctx, -4, "ld < ns");
        if (!state) return fail(ctx, -5, "null state");
        if (ctx->group) return kfsp::group_update_state_coords(ctx, n, ns, ld, state, (int32_t)keep);
        ctx->perm_pending_n = 0;
        const int64_t had = ctx->coords_n;                   // (coords_n is set again below if the coordinates are (partly) uploaded)
        ctx->coords_n = 0;
        // Sorting, relabelling and the extra upload cost about as much as 20 products (at 10^6 states)
        // save: worth it only while generators live that long.  The generator being
        // replaced is the best predictor there is.
        const bool order = ctx->opt_state_order && n >= ctx->opt_state_order_min && ctx->prod_count >= ctx->opt_state_order_products;
        if (!order && !ctx->opt_keep_coords) return 0;
        HIP_TRY(hipSetDevice(ctx->device));
        auto t0 = std::chrono::steady_clock::now();
        bool ok = false;
        ctx->coords_n = had;
        const int rc = kfsp::state_order_from_coords(ctx, n, ns, ld, state, &ok, keep, order);
        ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc) return rc;
        if (ok) ctx->perm_pending_n = n;
        return 0;
    });
}

int kfsp_set_state_coords(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state)
{
    return set_state_coords_impl(ctx, n, ns, ld, state, 0);
}

int kfsp_update_state_coords(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state, int32_t n_unchanged)
{
    if (n_unchanged < 0 || n_unchanged > n) return ctx ? fail(ctx, -6, "0 <= n_unchanged <= n") : -1;
    return set_state_coords_impl(ctx, n, ns, ld, state, n_unchanged);
}

int kfsp_state_order_active(const kfsp_ctx *ctx, int *active)
{
    if (!ctx) return -1;
    if (!active) return -2;
    if (ctx->group) return kfsp::group_state_order_active(ctx, active);
    *active = ctx->perm_on ? 1 : 0;
    return 0;
}

int kfsp_matrix_info(const kfsp_ctx *ctx, int64_t *nrows, int64_t *slots, int64_t *nnz)
{
    if (!ctx) return -1;
    if (ctx->group) return kfsp::group_matrix_info(ctx, nrows, slots, nnz);
    if (nrows) *nrows = ctx->nloc;
    if (slots) *slots = ctx->use_box ? 0 : (ctx->use_dia ? (int64_t)ctx->nd * ctx->dia_ld : ctx->slots);
    if (nnz) *nnz = ctx->nnz;
    return 0;
}

int kfsp_matrix_bytes(const kfsp_ctx *ctx, int force_sell, int64_t *bytes)
{
    if (!ctx) return -1;
    if (!bytes) return -3;
    if (ctx->ldv == 0) return -1;
    if (ctx->group) return kfsp::group_matrix_bytes(ctx, force_sell, bytes);
    const int64_t rows = ctx->nchunks * kChunk;
    int64_t b = rows * 24;                                    // diag, x (once), y
    if (ctx->use_box && !force_sell) {
        b = rows * 16 + (int64_t)ctx->box_lds_bytes;          // x once, y; the tables are read once per workgroup from cache
    } else if (ctx->use_dia && !force_sell) {
        b += (int64_t)ctx->nd * ctx->dia_ld * 8;
        if (ctx->dia_masked) b += (ctx->dia_ld >> 7) * 4 - ctx->dia_empty_segments * 128 * 8;
    } else {
        b += ctx->slots * 12 + (ctx->nchunks + 1) * 8;
        // dictionary-coded chunks: no column bytes, code words and offset tables instead (+ 12 B of chunk header)
        if (ctx->sell_coded && force_sell != 3)
            b += -4 * ctx->coded_slots + 8 * ctx->code_words + ctx->coded_tab_bytes + 12 * ctx->coded_chunks;
    }
    *bytes = b;
    return 0;
}

int kfsp_layout_info(const kfsp_ctx *ctx, int64_t *v)
{
    if (!ctx) return -1;
    if (!v) return -2;
    if (ctx->group) return kfsp::group_layout_info(ctx, v);
    v[0] = ctx->use_box ? (ctx->box_fast && !ctx->opt_box_generic ? (ctx->box_reach > 0 && ctx->opt_box_lds && !ctx->use_comm ? 6 :
                           (ctx->box_slab && ctx->opt_box_pencil == 2 && !ctx->use_comm ? 8 :
                            ctx->box_pencil && ctx->opt_box_pencil != 0 && !ctx->use_comm ? 7 : 4)) : 3) : ctx->use_dia ? (ctx->dia_masked ? 2 : 1) : (ctx->sell_coded ? 5 : 0);
    v[1] = !ctx->use_comm ? 0 : (ctx->use_halo ? 1 : 2);
    v[2] = ctx->halo;
    v[3] = ctx->use_dia ? -1 : ctx->sell_reach;
    v[4] = ctx->sell_coded ? ctx->coded_chunks : 0;
    v[5] = ctx->nchunks;
    v[6] = ctx->sell_coded ? ctx->code_words : 0;
    v[7] = ctx->perm_on ? 1 : 0;
    return 0;
}

int kfsp_build_info(const kfsp_ctx *ctx, int64_t *v)
{
    if (!ctx) return -1;
    if (!v) return -2;
    if (ctx->group) ctx = kfsp::group_rank0(ctx);
    v[0] = ctx->spec_builds;
    v[1] = ctx->spec_redone;
    v[2] = ctx->last_build_sell ? 1 : 0;
    v[3] = ctx->kc_ok ? 1 : 0;
    v[4] = ctx->order_merges;
    v[5] = 0;
    return 0;
}

int kfsp_set_trip_order(kfsp_ctx *ctx, int64_t ntrips, const int32_t *order)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ctx->group) return fail(ctx, -9, "not available on a group context");
        if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
        if (ntrips == 0) {
            ctx->trip_order_n = 0;
            return 0;
        }
        const int64_t trips = ctx->use_dia ? (ctx->nchunks + 1) / 2 : ctx->nchunks;
        if (ntrips != trips) return fail(ctx, -2, "ntrips is not the number of wavefront trips of the current generator");
        if (!order) return fail(ctx, -3, "null order");
        std::vector<uint8_t> seen((size_t)trips, 0);
        for (int64_t t = 0; t < trips; ++t) {
            if (order[t] < 0 || order[t] >= trips || seen[(size_t)order[t]]) return fail(ctx, -3, "order is not a permutation of the trips");
            seen[(size_t)order[t]] = 1;
        }
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(ctx->d_trip_order.reserve((size_t)trips, false));
        HIP_TRY(hipMemcpy(ctx->d_trip_order.p, order, (size_t)trips * sizeof(int32_t), hipMemcpyHostToDevice));
        ctx->trip_order_n = trips;
        return 0;
    });
}

int kfsp_num_states(const kfsp_ctx *ctx, int64_t *n)
{
    if (!ctx) return -1;
    if (!n) return -2;
    *n = ctx->n;
    return 0;
}

int kfsp_set_vector(kfsp_ctx *ctx, int64_t nlocal, const double *w)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (ctx->group) return kfsp::group_set_vector(ctx, nlocal, w);
    if (nlocal != ctx->nloc) return fail(ctx, -2, "nlocal is not this rank's block size");
    if (!w && nlocal > 0) return fail(ctx, -3, "null w");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->w_pending = false;
    ctx->w_pending_full = false;
    ctx->drop_planned = false;
    HIP_TRY(hipMemsetAsync(ctx->d_w.p, 0, (size_t)ctx->ldv * sizeof(double), ctx->stream));
    if (int rc = upload_states(ctx, w, ctx->d_w.p, nlocal)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int kfsp_get_vector(kfsp_ctx *ctx, int64_t nlocal, double *w)
{
    if (!ctx) return -1;
    if (ctx->group) return kfsp::group_get_vector(ctx, nlocal, w);
    if (nlocal != ctx->nloc) return fail(ctx, -2, "nlocal is not this rank's block size");
    if (!w && nlocal > 0) return fail(ctx, -3, "null w");
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = download_states(ctx, ctx->d_w.p, w, nlocal)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int kfsp_begin_step(kfsp_ctx *ctx, double *beta)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (!beta) return fail(ctx, -2, "null beta");
    if (ctx->group) return kfsp::group_begin_step(ctx, beta);
    PhaseTimer timer(ctx, KFSP_T_BEGIN);
    static const bool trace = std::getenv("KFSP_TRACE_BEGIN") != nullptr;   // diagnostics: profiles/begin_step_trace.sh
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(ctx->device));
    double *part = next_partial(ctx);
    const int g = vec_grid(ctx);
    launch_copy_nrm2(g, act_pairs(ctx), ctx->d_w.p, vcol(ctx, 0), part, ctx->stream);
    Pending s;
    if (int rc = publish(ctx, Pending{part, g}, &s)) return rc;
    double *hb = ctx->d_H.p + (size_t)kMH * kMH;   // scratch pair behind the H image
    launch_finalize(s, ctx->d_sq.p + 1, hb, ctx->stream);
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, hb, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    const auto t1 = std::chrono::steady_clock::now();
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *beta = ctx->h_pin[0];
    if (trace) {
        const auto t2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "KFSP_TRACE_BEGIN n=%lld enqueue_us=%.1f sync_us=%.1f\n", (long long)ctx->n,
                     std::chrono::duration<double, std::micro>(t1 - t0).count(),
                     std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    return 0;
}

int kfsp_arnoldi(kfsp_ctx *ctx, int m, int jold, int qiop, double break_tol, double *H, int ldh,
                 int *mbrkdwn, int *k1, double *avnorm)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    // m >= n is legal: the reference only caps M at N-1 when a step starts (:211),
    // a dimension change (:404) may exceed it and runs into a breakdown instead
    if (m < 1 || m > kMMax) return fail(ctx, -2, "bad m (need 1 <= m <= 100)");
    if (m > mmax_now(ctx)) return fail(ctx, -2, "m exceeds option m_max (the basis was allocated for fewer columns; a raised m_max applies from the next generator)");
    if (jold < 1 || jold > kMMax) return fail(ctx, -3, "bad jold");
    if (qiop < 0) return fail(ctx, -4, "bad qiop");
    if (!H) return fail(ctx, -6, "null H");
    if (ldh < m + 2) return fail(ctx, -7, "ldh < m+2");
    if (!mbrkdwn || !k1 || !avnorm) return fail(ctx, -8, "null output");
    if (ctx->group) return kfsp::group_arnoldi(ctx, m, jold, qiop, break_tol, H, ldh, mbrkdwn, k1, avnorm);
    PhaseTimer timer(ctx, KFSP_T_ARNOLDI);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t ldv = ctx->ldv;
    const int gv = vec_grid(ctx);
    double *V = vcol(ctx, 0), *Hd = ctx->d_H.p, *sq = ctx->d_sq.p;
    int *flag = ctx->d_flag.p;
    // (entries this pass will not write must not look like a breakdown)
    kfsp::launch_pass_reset(Hd, kMH * kMH, flag, st);

    const bool fused = (qiop == 2) && ctx->opt_fused != 0;
    double *gfin = ctx->d_g.p;
    // small state spaces: the whole pass in one launch of one workgroup
    const bool small = fused && !ctx->use_comm && ctx->opt_small != 0 && ctx->nchunks * kChunk <= kSmallRows &&
                       (ctx->use_dia || ctx->have_sell) && !ctx->use_box;
    if (small) {
        SmallArnoldiArgs sa;
        SpmvArgs tmp;
        set_matrix_args(ctx, tmp);
        sa.A = tmp.A;
        sa.D = tmp.D;
        sa.V = V;
        sa.ldv = ldv;
        sa.nact = ctx->nchunks * kChunk;
        sa.m = m;
        sa.jold = jold;
        sa.sq = sq;
        sa.gfin = gfin;
        sa.Hd = Hd;
        sa.break_tol = break_tol;
        sa.brk_flag = flag;
        sa.slots = ctx->use_dia ? 0 : ctx->slots;
        if (const int e = launch_arnoldi_small(sa, ctx->use_dia, ctx->opt_small_lds ? ctx->lds_per_block : 0, st))
            return hip_fail(ctx, (hipError_t)e, "hipFuncSetAttribute(k_arnoldi_small)");
    }
    Pending pend_sq{sq + jold, 1};
    Pending pend_g{gfin + jold, 1};      // u_jold . u_{jold-1}, finished by the pass that built column jold
    for (int j = jold; j <= m && !small; ++j) {
        const double *src = V + (size_t)(j - 1) * ldv;
        double *dst = V + (size_t)j * ldv;
        const int istart = (qiop > 0) ? std::max(1, j - qiop + 1) : 1;
        SpmvArgs a;
        set_matrix_args(ctx, a);
        a.y = dst;
        a.sq = pend_sq;
        a.sq_final = sq + j;
        a.h_sub = (j > jold) ? Hd + (size_t)(j - 2) * kMH + (j - 1) : nullptr;   // H(j,j-1)
        a.break_tol = (j > jold) ? break_tol : -1.0;
        a.brk_flag = flag;
        if (fused) {
            // one product kernel with both dot products, one update kernel
            const bool two = j >= 2;
            a.udot = two ? V + (size_t)(j - 2) * ldv : src;      // u_{j-1} (or u_1 for the first column)
            a.udot2 = two ? src : nullptr;                        // u_j
            Pending loc[2], pub[2];
            if (int rc = run_product(ctx, two ? 3 : 1, a, src, false, &loc[0], &loc[1])) return rc;
            if (int rc = publish_n(ctx, loc, two ? 2 : 1, pub)) return rc;
            Ortho2Args o;
            o.npairs = act_pairs(ctx);
            o.w = dst;
            o.u1 = two ? V + (size_t)(j - 2) * ldv : nullptr;
            o.u2 = src;
            o.a = two ? pub[0] : Pending{nullptr, 0};
            o.b = two ? pub[1] : pub[0];
            o.g = two ? pend_g : Pending{nullptr, 0};
            o.sq1 = two ? sq + (j - 1) : nullptr;
            o.sq2 = sq + j;
            o.partial_sq = next_partial(ctx);
            o.partial_g = next_partial(ctx);
            o.h1_out = two ? Hd + (size_t)(j - 1) * kMH + (j - 2) : nullptr;   // H(j-1,j)
            o.h2_out = Hd + (size_t)(j - 1) * kMH + (j - 1);                   // H(j,j)
            o.g_final = two ? gfin + j : nullptr;
            o.brk_flag = flag;
            launch_ortho2(gv, o, st);
            Pending loc2[2] = {Pending{o.partial_sq, gv}, Pending{o.partial_g, gv}}, pub2[2];
            if (int rc = publish_n(ctx, loc2, 2, pub2)) return rc;
            pend_sq = pub2[0];
            pend_g = pub2[1];
            continue;
        }
        a.udot = V + (size_t)(istart - 1) * ldv;
        Pending pend;
        if (int rc = run_product(ctx, 1, a, src, false, &pend, nullptr)) return rc;
        if (int rc = publish(ctx, pend, &pend)) return rc;
        for (int i = istart; i <= j; ++i) {
            OrthoArgs o;
            o.npairs = act_pairs(ctx);
            o.w = dst;
            o.ui = V + (size_t)(i - 1) * ldv;
            o.dot = pend;
            o.sq_i = sq + i;
            o.unext = (i < j) ? V + (size_t)i * ldv : nullptr;
            o.partial = next_partial(ctx);
            o.h_out = Hd + (size_t)(j - 1) * kMH + (i - 1);   // H(i,j)
            o.brk_flag = flag;
            launch_ortho(gv, o, st);
            if (int rc = publish(ctx, Pending{o.partial, gv}, &pend)) return rc;
        }
        pend_sq = pend;
    }
    // the extra product for AVNORM (:261-263); taken from column jold when the
    // dimension shrank below jold (J1V is only advanced inside the loop)
    const bool looped = jold <= m;
    const int jl = looped ? m + 1 : jold;
    if (!small) {
        const double *src = V + (size_t)(jl - 1) * ldv;
        SpmvArgs a;
        set_matrix_args(ctx, a);
        a.y = V + (size_t)jl * ldv;
        a.sq = looped ? pend_sq : Pending{sq + jold, 1};
        a.sq_final = sq + jl;
        a.h_sub = looped ? Hd + (size_t)(m - 1) * kMH + m : nullptr;   // H(m+1,m)
        a.udot = nullptr;
        a.break_tol = looped ? break_tol : -1.0;
        a.brk_flag = flag;
        Pending pend;
        if (int rc = run_product(ctx, 2, a, src, false, &pend, nullptr)) return rc;
        if (int rc = publish(ctx, pend, &pend)) return rc;
        launch_finalize(pend, Hd + (size_t)kMH * kMH, Hd + (size_t)kMH * kMH + 1, st);
        // u_{m+1} . u_m for a later restart at column m+1 is never needed (a
        // restart resumes at jold <= m), but finish it so that gfin stays final
        if (fused && looped) launch_finalize(pend_g, gfin + (m + 1), nullptr, st);
    }
    {
        // only what this pass wrote: columns jold..m (+ the sub-diagonal of column m) and the AVNORM pair
        const size_t first = (size_t)(jold - 1) * kMH, last = (size_t)kMH * kMH + 2;
        HIP_TRY(hipMemcpyAsync(ctx->h_H + first, Hd + first, (last - first) * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));

    const double *hh = ctx->h_H;
    *mbrkdwn = m;
    *k1 = 2;
    for (int j = jold; j <= m; ++j) {
        const int istart = (qiop > 0) ? std::max(1, j - qiop + 1) : 1;
        for (int i = istart; i <= j; ++i) H[(size_t)(j - 1) * ldh + (i - 1)] = hh[(size_t)(j - 1) * kMH + (i - 1)];
        const double hj1j = hh[(size_t)(j - 1) * kMH + j];
        if (!(hj1j > break_tol)) {   // :249-256
            *k1 = 0;
            *mbrkdwn = j;
            break;
        }
        H[(size_t)(j - 1) * ldh + j] = hj1j;
    }
    if (*k1 != 0) ctx->avnorm_last = hh[(size_t)kMH * kMH + 1];
    *avnorm = ctx->avnorm_last;
    H[(size_t)m * ldh + (m + 1)] = 1.0;   // :266
    return 0;
}

int kfsp_combine(kfsp_ctx *ctx, int mx, double beta, const double *y, double *wsum)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (mx < 1 || mx > mmax_now(ctx) + 2) return fail(ctx, -2, "bad mx");
    if (!y) return fail(ctx, -4, "null y");
    if (!wsum) return fail(ctx, -5, "null wsum");
    if (ctx->group) return kfsp::group_combine(ctx, mx, beta, y, wsum);
    PhaseTimer timer(ctx, KFSP_T_COMBINE);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::memcpy(ctx->h_pin + 8, y, (size_t)mx * sizeof(double));
    HIP_TRY(hipMemcpyAsync(ctx->d_y.p, ctx->h_pin + 8, (size_t)mx * sizeof(double), hipMemcpyHostToDevice, st));
    CombineArgs a;
    a.npairs = act_pairs(ctx);
    a.mx = mx;
    a.beta = beta;
    a.V = vcol(ctx, 0);
    a.ldv = ctx->ldv;
    a.sq = ctx->d_sq.p;
    a.y = ctx->d_y.p;
    a.w = ctx->d_w.p;
    a.partial = next_partial(ctx);
    // one pair per lane: each lane already streams mx columns
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, (act_pairs(ctx) + kBlock - 1) / kBlock));
    launch_combine(g, a, st);
    Pending s;
    if (int rc = publish(ctx, Pending{a.partial, g}, &s)) return rc;
    double *hb = ctx->d_H.p + (size_t)kMH * kMH;
    launch_finalize(s, hb, nullptr, st);
    HIP_TRY(hipMemcpyAsync(ctx->h_pin, hb, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *wsum = ctx->h_pin[0];
    return 0;
}

int kfsp_restore_w(kfsp_ctx *ctx, double beta)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (ctx->group) return kfsp::group_restore_w(ctx, beta);
    HIP_TRY(hipSetDevice(ctx->device));
    launch_scale_copy(vec_grid(ctx), act_pairs(ctx), vcol(ctx, 0), ctx->d_sq.p + 1, beta, ctx->d_w.p, ctx->stream);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int spmv_plain(kfsp_ctx *ctx, const double *src_local_or_full, bool src_is_full, double *y_dev,
                      bool force_sell = false, bool force_plain_sell = false)
{
    SpmvArgs a;
    set_matrix_args(ctx, a);
    a.y = y_dev;
    a.sq = Pending{nullptr, 0};
    a.sq_final = nullptr;
    a.h_sub = nullptr;
    a.udot = nullptr;
    a.break_tol = -1.0;
    a.brk_flag = ctx->d_flag.p;
    return run_product(ctx, 0, a, src_local_or_full, src_is_full, nullptr, nullptr, force_sell, force_plain_sell);
}

int kfsp_spmv(kfsp_ctx *ctx, const double *x, double *y)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (!x) return fail(ctx, -2, "null x");
    if (ctx->group) return kfsp::group_spmv(ctx, x, y);
    if (!y && ctx->nloc > 0) return fail(ctx, -3, "null y");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemsetAsync(ctx->d_xg.p, 0, ctx->d_xg.cap * sizeof(double), st));
    // x is the WHOLE vector in the caller's order on every rank; global index g of the internal order lives at d_xg[g]
    if (ctx->perm_on) {
        HIP_TRY(ctx->d_pstage.reserve((size_t)ctx->n, false));
        HIP_TRY(hipMemcpyAsync(ctx->d_pstage.p, x, (size_t)ctx->n * sizeof(double), hipMemcpyHostToDevice, st));
        kfsp::launch_gather_index(ctx->n, ctx->d_perm.p, ctx->d_pstage.p, ctx->d_xg.p, st);
    } else {
        HIP_TRY(hipMemcpyAsync(ctx->d_xg.p, x, (size_t)ctx->n * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (int rc = spmv_plain(ctx, ctx->d_xg.p, true, ctx->d_tmp.p)) return rc;
    if (int rc = download_states(ctx, ctx->d_tmp.p, y, ctx->nloc)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int kfsp_spmv_w(kfsp_ctx *ctx, double *y)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (ctx->group) return kfsp::group_spmv_w(ctx, y);
    if (!y && ctx->nloc > 0) return fail(ctx, -2, "null y");
    HIP_TRY(hipSetDevice(ctx->device));
    const double *src = ctx->d_w.p;
    if (ctx->use_halo) {
        // w carries no halo margins: stage it in the scratch column
        double *scratch = vcol(ctx, num_cols(ctx) - 1);
        HIP_TRY(hipMemcpyAsync(scratch, ctx->d_w.p, (size_t)ctx->L * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        src = scratch;
    }
    if (int rc = spmv_plain(ctx, src, false, ctx->d_tmp.p)) return rc;
    if (int rc = download_states(ctx, ctx->d_tmp.p, y, ctx->nloc)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int onestep_impl(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state, int32_t ld_state,
                        const int32_t *adj, int32_t ld_adj, int32_t max_count, int32_t capacity, int32_t *n_new, int32_t *state_new,
                        int32_t *adj_out, double *off_new, int32_t ld_off, double *diag_new)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ns < 1 || ns > 16) return fail(ctx, -2, "1 <= ns <= 16");
        if (nr < 1 || nr > 64) return fail(ctx, -3, "1 <= nr <= 64");
        if (!stoich) return fail(ctx, -4, "null stoich");
        if (n < 1) return fail(ctx, -5, "n < 1");
        if (!state || ld_state < ns) return fail(ctx, -6, "bad state / ld_state");
        if (!adj || ld_adj < nr) return fail(ctx, -8, "bad adj / ld_adj");
        if (max_count < 1) return fail(ctx, -10, "max_count < 1");
        if (capacity < n) return fail(ctx, -11, "capacity < n");
        if (!n_new || !state_new || !adj_out) return fail(ctx, -12, "null output");
        if (ctx->group) ctx = kfsp::group_rank0(ctx);      // integer work on the whole lists, no collective inside: one rank does it
        if (off_new || diag_new) {
            if (!off_new || !diag_new || ld_off < nr) return fail(ctx, -15, "bad offdiag_new / ld_off / diag_new");
            if (!ctx->prop_ready || ctx->prop_ns != ns || ctx->prop_nr != nr)
                return fail(ctx, -15, "no propensity program for this model (kfsp_set_propensity_program)");
        }
        HIP_TRY(hipSetDevice(ctx->device));
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = kfsp::onestep_device(ctx, ns, nr, stoich, n, state, ld_state, adj, ld_adj, max_count, capacity, n_new,
                                            state_new, adj_out, off_new, ld_off, diag_new);
        ctx->t_ms[KFSP_T_ONESTEP] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    });
}

int kfsp_onestep(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state, int32_t ld_state,
                 const int32_t *adj, int32_t ld_adj, int32_t max_count, int32_t capacity, int32_t *n_new, int32_t *state_new,
                 int32_t *adj_out)
{
    return onestep_impl(ctx, ns, nr, stoich, n, state, ld_state, adj, ld_adj, max_count, capacity, n_new, state_new, adj_out,
                        nullptr, 0, nullptr);
}

int kfsp_onestep_columns(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state, int32_t ld_state,
                         const int32_t *adj, int32_t ld_adj, int32_t max_count, int32_t capacity, int32_t *n_new, int32_t *state_new,
                         int32_t *adj_out, double *offdiag_new, int32_t ld_off, double *diag_new)
{
    if (ctx && (!offdiag_new || !diag_new)) return fail(ctx, -15, "null offdiag_new / diag_new");
    return onestep_impl(ctx, ns, nr, stoich, n, state, ld_state, adj, ld_adj, max_count, capacity, n_new, state_new, adj_out,
                        offdiag_new, ld_off, diag_new);
}

int kfsp_set_propensity_program(kfsp_ctx *ctx, int32_t ns, int32_t nr, int32_t nparams, const double *params, const int32_t *code_off,
                                const int32_t *code, const int32_t *imm_off, const double *imm, const int32_t *tab_species,
                                int32_t tab_len, const double *tab)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ns < 1 || ns > 16) return fail(ctx, -2, "1 <= ns <= 16");
        if (nr < 1 || nr > 64) return fail(ctx, -3, "1 <= nr <= 64");
        if (nparams < 0 || (nparams > 0 && !params)) return fail(ctx, -4, "bad parameters");
        if (!code_off || !code || !imm_off) return fail(ctx, -6, "null code");
        if (imm_off[nr] > 0 && !imm) return fail(ctx, -9, "null immediates");
        if (!tab_species) return fail(ctx, -10, "null tab_species");
        if (tab_len < 0 || (tab_len > 0 && !tab)) return fail(ctx, -11, "bad tables");
        if (ctx->group)
            return kfsp::group_set_propensity_program(ctx, ns, nr, nparams, params, code_off, code, imm_off, imm, tab_species, tab_len, tab);
        HIP_TRY(hipSetDevice(ctx->device));
        return kfsp::prop_set_program(ctx, ns, nr, nparams, params, code_off, code, imm_off, imm, tab_species, tab_len, tab);
    });
}

int kfsp_set_propensity_tables2(kfsp_ctx *ctx, int32_t nr, const int32_t *s1, const int32_t *s2, const int32_t *n1, const int32_t *n2,
                                const int64_t *off, int64_t len, const double *tab2)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (nr < 1 || nr > 64) return fail(ctx, -2, "1 <= nr <= 64");
        if (!s1 || !s2 || !n1 || !n2 || !off) return fail(ctx, -3, "null table description");
        if (len < 0 || (len > 0 && !tab2)) return fail(ctx, -8, "bad tables");
        if (ctx->group) return kfsp::group_set_propensity_tables2(ctx, nr, s1, s2, n1, n2, off, len, tab2);
        HIP_TRY(hipSetDevice(ctx->device));
        return kfsp::prop_set_tables2(ctx, nr, s1, s2, n1, n2, off, len, tab2);
    });
}

int kfsp_propensity_overflow(kfsp_ctx *ctx, int32_t ns, int32_t *max_missed)
{
    if (!ctx) return -1;
    if (ns < 1 || ns > 16 || !max_missed) return fail(ctx, -2, "1 <= ns <= 16, non-null max_missed");
    const kfsp_ctx *c = ctx->group ? kfsp::group_rank0(ctx) : ctx;     // (every rank works on the whole lists: same misses)
    for (int s = 0; s < ns; ++s) max_missed[s] = c->prop_missed[s];
    return 0;
}

int kfsp_ssa_streams(kfsp_ctx *ctx, double timestep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n,
                     const int32_t *state, int32_t ld_state, const int32_t *adj, const double *offdiag, int32_t ld_adj,
                     const double *diag, int32_t max_count, int32_t capacity_new, int32_t *n_found, int32_t *state_new,
                     double *offdiag_new, int32_t ld_off, double *diag_new)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ctx->group) ctx = kfsp::group_rank0(ctx);      // the whole lists, no collective inside: one rank does it
        if (!(timestep > 0.0)) return fail(ctx, -2, "timestep must be positive");
        if (ns < 1 || ns > 16) return fail(ctx, -4, "1 <= ns <= 16");
        if (nr < 1 || nr > 64) return fail(ctx, -5, "1 <= nr <= 64");
        if (!stoich) return fail(ctx, -6, "null stoich");
        if (n < 1) return fail(ctx, -7, "n < 1");
        if (!state || ld_state < ns) return fail(ctx, -8, "bad state / ld_state");
        if (!adj || !offdiag || ld_adj < nr || !diag) return fail(ctx, -10, "bad adj / offdiag / ld_adj / diag");
        if (max_count < 1) return fail(ctx, -14, "max_count < 1");
        if (capacity_new < 0 || !n_found || !state_new) return fail(ctx, -15, "bad capacity_new / outputs");
        if (!offdiag_new || ld_off < nr || !diag_new) return fail(ctx, -18, "bad offdiag_new / ld_off / diag_new");
        if (!ctx->prop_ready || ctx->prop_ns != ns || ctx->prop_nr != nr)
            return fail(ctx, -1, "no propensity program for this model (kfsp_set_propensity_program)");
        HIP_TRY(hipSetDevice(ctx->device));
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = kfsp::ssa_streams_device(ctx, timestep, seedmix, ns, nr, stoich, n, state, ld_state, adj, offdiag, ld_adj, diag,
                                                max_count, capacity_new, n_found, state_new, offdiag_new, ld_off, diag_new);
        ctx->t_ms[KFSP_T_ONESTEP] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    });
}

int kfsp_propensities(kfsp_ctx *ctx, int32_t n, const int32_t *state, int32_t ld_state, double *offdiag, int32_t ld_off, double *diag)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ctx->group) ctx = kfsp::group_rank0(ctx);
        if (!ctx->prop_ready) return fail(ctx, -1, "no propensity program (kfsp_set_propensity_program)");
        if (n < 0) return fail(ctx, -2, "n < 0");
        if (n == 0) return 0;
        if (!state || ld_state < ctx->prop_ns) return fail(ctx, -3, "bad state / ld_state");
        if (!offdiag || ld_off < ctx->prop_nr) return fail(ctx, -5, "bad offdiag / ld_off");
        if (!diag) return fail(ctx, -7, "null diag");
        HIP_TRY(hipSetDevice(ctx->device));
        return kfsp::prop_eval_host(ctx, n, state, ld_state, offdiag, ld_off, diag);
    });
}

int kfsp_drop_plan(kfsp_ctx *ctx, double dsum, double *droptol, int64_t *drop_count, int64_t *n_flagged)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (!(dsum > 0.0)) return fail(ctx, -2, "dsum must be positive (FIND_DROPTOL would not terminate)");
    if (!droptol || !drop_count || !n_flagged) return fail(ctx, -3, "null output");
    if (ctx->group) return kfsp::group_drop_plan(ctx, dsum, droptol, drop_count, n_flagged);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = ctx->n;
    const bool comm = ctx->use_comm;
    ctx->drop_planned = false;
    // FIND_DROPTOL: thresholds 1e-8, /10, /10, ... (the same divisions, so the same doubles), sixteen per pass.
    // With a communicator every rank sums its block and ONE all-reduce carries the sixteen sums.
    double *part = ctx->d_part.p;                       // kDropLevels * grid partials: the rotating buffers are idle between steps
    const int grid = (int)std::min<int64_t>(vec_grid(ctx), (int64_t)kNumPartial * kMaxGrid / kfsp::kDropLevels);
    double *sums_dev = ctx->d_H.p;                      // scratch: the H image is rewritten by the next pass anyway
    double tol = 1.0e-8, found = -1.0;
    for (int batch = 0; batch < 64 && found < 0.0; ++batch) {
        kfsp::DropLevels L;
        for (int k = 0; k < kfsp::kDropLevels; ++k) {
            L.tol[k] = tol;
            tol = tol / 10.0;
        }
        kfsp::launch_drop_sums(grid, act_pairs(ctx), ctx->d_w.p, L, part, sums_dev, st);
        if (comm)
            if (int rc = comm_allreduce(ctx, sums_dev, kfsp::kDropLevels, false, st)) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->h_pin, sums_dev, kfsp::kDropLevels * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int k = 0; k < kfsp::kDropLevels; ++k)
            if (ctx->h_pin[k] < dsum) {
                found = L.tol[k];
                break;
            }
    }
    if (found < 0.0) return fail(ctx, -2, "no threshold satisfies the mass bound");
    *droptol = found;
    // A w into a basis column nobody needs between two steps, then the flags
    double *aw = vcol(ctx, 1);
    {
        const double *src = ctx->d_w.p;
        if (ctx->use_halo) {                            // w carries no halo margins: stage it in the scratch column
            double *scratch = vcol(ctx, num_cols(ctx) - 1);
            HIP_TRY(hipMemcpyAsync(scratch, ctx->d_w.p, (size_t)ctx->L * sizeof(double), hipMemcpyDeviceToDevice, st));
            src = scratch;
        }
        if (int rc = spmv_plain(ctx, src, false, aw)) return rc;
    }
    HIP_TRY(ctx->d_dropflag.reserve(2 * (size_t)(n + 256), false));
    HIP_TRY(ctx->d_dropcnt.reserve(8, true));
    HIP_TRY(hipMemsetAsync(ctx->d_dropcnt.p, 0, 4 * sizeof(unsigned long long), st));
    unsigned long long cnt[4] = {0, 0, 0, 0};
    if (!comm) {
        if (ctx->perm_on) {
            // marks in the device's order (streaming reads, streaming byte stores), then the BYTES go to the caller's order:
            // gathering one byte per state from an array that fits the L2 costs a tenth of gathering w and A w
            HIP_TRY(ctx->d_flagloc.reserve((size_t)n + 256, false));
            kfsp::launch_drop_flags(n, ctx->d_w.p, aw, found, nullptr, ctx->d_flagloc.p, ctx->d_dropcnt.p, st);
            kfsp::launch_flags_to_caller(n, ctx->d_flagloc.p, ctx->d_iperm.p, ctx->d_dropflag.p, st);
        } else {
            kfsp::launch_drop_flags(n, ctx->d_w.p, aw, found, nullptr, ctx->d_dropflag.p, ctx->d_dropcnt.p, st);
        }
        HIP_TRY(hipMemcpyAsync(cnt, ctx->d_dropcnt.p, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        // flags of this rank's block (internal order), all blocks to every rank, then into the caller's order:
        // every rank ends up with the flags of ALL states, as the host that compacts its lists needs them
        const size_t Lb = (size_t)ctx->L;
        HIP_TRY(ctx->d_flagloc.reserve(Lb * (size_t)(ctx->nranks + 1) + 256, false));
        uint8_t *mine = ctx->d_flagloc.p, *all = ctx->d_flagloc.p + Lb;
        HIP_TRY(hipMemsetAsync(mine, 0, Lb, st));
        if (ctx->nloc > 0) kfsp::launch_drop_flags(ctx->nloc, ctx->d_w.p, aw, found, nullptr, mine, ctx->d_dropcnt.p, st);
        HIP_TRY(hipMemcpyAsync(cnt, ctx->d_dropcnt.p, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        double *stg = ctx->d_stage.p;                                     // counts as doubles (exact below 2^53)
        const double hc[3] = {(double)cnt[0], (double)cnt[1], (double)cnt[2]};
        HIP_TRY(hipMemcpyAsync(stg, hc, sizeof(hc), hipMemcpyHostToDevice, st));
        if (int rc = comm_allreduce(ctx, stg, 3, false, st)) return rc;
        double hr[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyAsync(hr, stg, sizeof(hr), hipMemcpyDeviceToHost, st));
        if (int rc = comm_allgather_bytes(ctx, mine, all, Lb, st)) return rc;
        kfsp::launch_flags_to_caller(n, all, ctx->perm_on ? ctx->d_iperm.p : nullptr, ctx->d_dropflag.p, st);
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < 3; ++i) cnt[i] = (unsigned long long)hr[i];
    }
    *drop_count = (int64_t)cnt[0] - (int64_t)cnt[1];     // the reference's DROP_COUNT (:476-495)
    *n_flagged = (int64_t)cnt[2];
    ctx->drop_planned = true;
    ctx->drop_n = n;
    ctx->drop_flagged = (int64_t)cnt[2];
    return 0;
}

int kfsp_drop_flags(kfsp_ctx *ctx, int64_t n, uint8_t *dropped)
{
    if (!ctx) return -1;
    if (ctx->group) return kfsp::group_drop_flags(ctx, n, dropped);
    if (!ctx->drop_planned || ctx->drop_n != ctx->n) return fail(ctx, -1, "no drop plan for the current FSP");
    if (n != ctx->n) return fail(ctx, -2, "n is not the size of the planned FSP");
    if (!dropped) return fail(ctx, -3, "null flags");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dropped, ctx->d_dropflag.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int kfsp_drop_compact(kfsp_ctx *ctx, int64_t *n_new)
{
    if (!ctx) return -1;
    if (ctx->group) return n_new ? kfsp::group_drop_compact(ctx, n_new) : -2;
    if (!ctx->drop_planned || ctx->drop_n != ctx->n) return fail(ctx, -1, "no drop plan for the current FSP");
    if (!n_new) return fail(ctx, -2, "null n_new");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = ctx->n;
    const double *src = ctx->d_w.p;
    double *dst = ctx->d_tmp.p;
    if (ctx->use_comm) {
        // the partition of the compacted FSP is another one: every rank assembles the WHOLE vector in the caller's
        // order, compacts it (all ranks hold all flags), and takes its new block when the next generator arrives
        if (int rc = gather_to_full(ctx, ctx->d_w.p, &src)) return rc;
        HIP_TRY(ctx->d_wfull.reserve((size_t)n + 64, false));
        dst = ctx->d_wfull.p;
    } else if (ctx->perm_on) {                           // back to the caller's order first
        HIP_TRY(ctx->d_pstage.reserve((size_t)n, false));
        kfsp::launch_gather_index(n, ctx->d_iperm.p, ctx->d_w.p, ctx->d_pstage.p, st);
        src = ctx->d_pstage.p;
    }
    if (!ctx->use_comm) HIP_TRY(hipMemsetAsync(ctx->d_tmp.p, 0, (size_t)ctx->ldv * sizeof(double), st));
    int *nk_dev = reinterpret_cast<int *>(ctx->d_dropcnt.p + 3);
    if (int rc = kfsp::drop_compact_vector(ctx, n, src, dst, nk_dev)) return fail(ctx, rc, "device compaction failed");
    int nk = 0;
    HIP_TRY(hipMemcpyAsync(&nk, nk_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((int64_t)nk != n - ctx->drop_flagged) return fail(ctx, 4000, "compaction count does not match the plan");
    ctx->drop_planned = false;
    ctx->drop_ell_cols = ctx->ell_cols;   // (kfsp_drop_rebuild can renumber the device's own copy)
    ctx->drop_bw = ctx->ell_bw;
    ctx->ell_cols = 0;                 // the columns are about to be renumbered
    ctx->w_pending = true;
    ctx->w_pending_n = nk;
    *n_new = nk;
    return 0;
}

int kfsp_drop_rebuild(kfsp_ctx *ctx)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ctx->group) return kfsp::group_drop_rebuild(ctx);
        if (!ctx->w_pending) return fail(ctx, -1, "no compaction pending (kfsp_drop_compact)");
        const int64_t n_old = ctx->drop_n, n_new = ctx->w_pending_n;
        if (ctx->drop_ell_cols != n_old || ctx->opt_host_build || n_new < 1)
            return fail(ctx, -9, "the reference arrays of this FSP are not resident on the device: upload the compacted generator");
        HIP_TRY(hipSetDevice(ctx->device));
        auto t0 = std::chrono::steady_clock::now();
        const int bw = ctx->drop_bw, ld = ctx->ell_ld;
        const uint8_t *keep = ctx->d_dropflag.p + ctx->d_dropflag.cap / 2;      // (made by kfsp_drop_compact, caller's order)
        const bool coords = ctx->coords_n == n_old;
        const int cld = ctx->coords_ld, cns = ctx->coords_ns;
        const bool want_order = ctx->opt_state_order && n_new >= ctx->opt_state_order_min && ctx->prod_count >= ctx->opt_state_order_products;
        if (want_order && !coords)     // the upload path would order this FSP, and the coordinates are not here: let it
            return fail(ctx, -9, "the state coordinates of this FSP are not resident on the device: upload the compacted generator");
        ctx->drop_ell_cols = 0;
        if (int rc = kfsp::compact_resident_ell(ctx, n_old, bw, ld, keep, n_new, coords, cld)) return rc;
        ctx->ell_cols = n_new;
        // the state order of the compacted FSP: the rule of kfsp_set_state_coords, on the coordinates that stayed here
        ctx->perm_pending_n = 0;
        ctx->coords_n = 0;
        bool ordered = false;
        const bool spec = ctx->opt_build_speculate != 0;    // (under a row partition too: order and build are rank-local, every rank repeats the same checks)
        if (coords && want_order) {
            // (the kept states keep their relative order: compact the order that is here, or make it from the coordinates)
            int rc = 0;
            const int32_t *scan = ctx->d_sortidx.p + n_old;             // (left there by compact_resident_ell)
            if (spec && cns == ctx->kc_ns && kfsp::state_order_after_drop(ctx, (int32_t)n_old, (int32_t)n_new, keep, scan, &rc)) {
                if (rc) return rc;
                ordered = true;
                ctx->coords_n = n_new;                                  // (as state_order_from_resident leaves them)
                ctx->coords_ld = cld;
                ctx->coords_ns = cns;
            } else if ((rc = kfsp::state_order_from_resident(ctx, (int32_t)n_new, cns, cld, &ordered, spec)))
                return rc;
        }
        if (coords && !ordered) {                              // (the coordinates are resident either way)
            ctx->coords_n = n_new;
            ctx->coords_ld = cld;
            ctx->coords_ns = cns;
        }
        ctx->use_box = false;
        ctx->box_lds_bytes = 0;
        if (int rc = resize(ctx, n_new)) return rc;
        ctx->perm_on = ordered;
        ctx->prod_last = ctx->prod_count;
        ctx->prod_count = 0;
        int rc = kfsp::build_from_resident_ell(ctx, (int32_t)n_new, bw, ld, spec);
        if (rc == kfsp::kRedoBuild) {                          // (a speculation did not hold: order and generator again, waiting for every number)
            ++ctx->spec_redone;
            if (coords && want_order) {
                if (int rc2 = kfsp::state_order_from_resident(ctx, (int32_t)n_new, cns, cld, &ordered)) return rc2;
                ctx->perm_on = ordered;
            }
            rc = kfsp::build_from_resident_ell(ctx, (int32_t)n_new, bw, ld);
        }
        if (!rc) rc = setup_exchange(ctx);
        if (!rc) rc = adopt_pending_vector(ctx);
        ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    });
}

int kfsp_expand_resident(kfsp_ctx *ctx, double t_ssa, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich,
                         int32_t max_count, int32_t capacity, int64_t *n_new, int64_t *n_from_ssa)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ns < 1 || ns > 16) return fail(ctx, -4, "1 <= ns <= 16");
        if (nr < 1 || nr > 64) return fail(ctx, -5, "1 <= nr <= 64");
        if (!stoich) return fail(ctx, -6, "null stoich");
        if (max_count < 1) return fail(ctx, -7, "max_count < 1");
        if (!n_new) return fail(ctx, -9, "null n_new");
        // (under a row partition every rank holds the WHOLE lists and expands them redundantly - same input, deterministic
        // kernels, same output - and rebuilds its own block; the vector is assembled and re-dealt through the communicator)
        if (ctx->group) return kfsp::group_expand_resident(ctx, t_ssa, seedmix, ns, nr, stoich, max_count, capacity, n_new, n_from_ssa);
        const int64_t n = ctx->n;
        if (ctx->use_box || ctx->opt_host_build || ctx->w_pending || n < 1 || ctx->ell_cols != n || ctx->ell_bw != nr)
            return fail(ctx, -9, "the reference arrays of the current FSP are not resident on the device (kfsp_set_matrix_ell)");
        if (ctx->coords_n != n || ctx->coords_ns != ns)
            return fail(ctx, -9, "the state coordinates of the current FSP are not resident on the device (option keep_coords, kfsp_set_state_coords)");
        if (capacity < n) return fail(ctx, -8, "capacity < n");
        if (!ctx->prop_ready || ctx->prop_ns != ns || ctx->prop_nr != nr)
            return fail(ctx, -1, "no propensity program for this model (kfsp_set_propensity_program)");
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        auto t0 = std::chrono::steady_clock::now();
        // the vector in the caller's order, before the order changes under it
        const double *full = nullptr;
        if (int rc = gather_to_full(ctx, ctx->d_w.p, &full)) return rc;
        int64_t n2 = n, nssa = 0;
        int rc = kfsp::expand_resident_lists(ctx, t_ssa, seedmix, ns, nr, stoich, max_count, capacity, &n2, &nssa);
        ctx->t_ms[KFSP_T_ONESTEP] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc) {
            // the lists may hold states appended by the walk: the caller's count (n) still describes a consistent prefix
            ctx->coords_n = n;
            return rc;
        }
        if (n_from_ssa) *n_from_ssa = nssa;
        *n_new = n2;
        if (n2 == n) return 0;                              // nothing was appended: generator, order and vector stay
        t0 = std::chrono::steady_clock::now();
        HIP_TRY(ctx->d_wfull.reserve((size_t)n2 + 64, false));
        HIP_TRY(hipMemcpyAsync(ctx->d_wfull.p, full, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        kfsp::launch_zero_pad(n, n2, ctx->d_wfull.p, st);
        ctx->w_pending = true;
        ctx->w_pending_full = true;
        ctx->w_pending_n = n2;
        ctx->ell_cols = n2;
        // the state order of the grown FSP: the rule of kfsp_set_state_coords, on the coordinates that are here
        const int cld = ctx->coords_ld;
        const bool want_order = ctx->opt_state_order && n2 >= ctx->opt_state_order_min && ctx->prod_count >= ctx->opt_state_order_products;
        ctx->perm_pending_n = 0;
        bool ordered = false;
        const bool spec = ctx->opt_build_speculate != 0;    // (under a row partition too: order and build are rank-local, every rank repeats the same checks)
        if (want_order)
            if (int rc2 = kfsp::state_order_from_resident(ctx, (int32_t)n2, ns, cld, &ordered, spec, (int32_t)n)) return rc2;
        ctx->coords_n = n2;
        ctx->coords_ld = cld;
        ctx->coords_ns = ns;
        if (int rc2 = resize(ctx, n2)) return rc2;
        ctx->perm_on = ordered;
        ctx->prod_last = ctx->prod_count;
        ctx->prod_count = 0;
        rc = kfsp::build_from_resident_ell(ctx, (int32_t)n2, nr, ctx->ell_ld, spec);
        if (rc == kfsp::kRedoBuild) {                          // (a speculation did not hold: order and generator again, waiting for every number)
            ++ctx->spec_redone;
            if (want_order) {
                if (int rc2 = kfsp::state_order_from_resident(ctx, (int32_t)n2, ns, cld, &ordered)) return rc2;
                ctx->coords_n = n2;
                ctx->perm_on = ordered;
            }
            rc = kfsp::build_from_resident_ell(ctx, (int32_t)n2, nr, ctx->ell_ld);
        }
        if (!rc) rc = setup_exchange(ctx);
        if (!rc) rc = adopt_pending_vector(ctx);
        ctx->t_ms[KFSP_T_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    });
}

int kfsp_download_fsp(kfsp_ctx *ctx, int32_t n, int32_t *state, int32_t ld_state, int32_t *adj, double *offdiag, int32_t ld_adj, double *diag)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ctx->group) ctx = kfsp::group_rank0(ctx);      // every rank holds the whole reference arrays
        if (n < 1 || n != ctx->ell_cols) return fail(ctx, -2, "n is not the number of states of the resident reference arrays");
        if (state && (ctx->coords_n != n || ld_state != ctx->coords_ld)) return fail(ctx, -3, "the state coordinates are not resident / ld_state differs");
        if ((adj || offdiag) && ld_adj != ctx->ell_ld) return fail(ctx, -7, "ld_adj differs from the resident arrays'");
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        if (state) HIP_TRY(hipMemcpyAsync(state, ctx->d_coords.p, (size_t)n * ld_state * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (adj) HIP_TRY(hipMemcpyAsync(adj, ctx->d_ell_adj.p, (size_t)n * ld_adj * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (offdiag) HIP_TRY(hipMemcpyAsync(offdiag, ctx->d_ell_off.p, (size_t)n * ld_adj * sizeof(double), hipMemcpyDeviceToHost, st));
        if (diag) HIP_TRY(hipMemcpyAsync(diag, ctx->d_ell_diag.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    });
}

static int reduce_w(kfsp_ctx *ctx, int squared, double *out)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (!out) return fail(ctx, -2, "null out");
    if (ctx->group) return kfsp::group_reduce_w(ctx, squared, out);
    HIP_TRY(hipSetDevice(ctx->device));
    double *part = next_partial(ctx);
    const int g = vec_grid(ctx);
    launch_reduce(g, act_pairs(ctx), ctx->d_w.p, squared, part, ctx->stream);
    Pending s;
    if (int rc = publish(ctx, Pending{part, g}, &s)) return rc;
    double *hb = ctx->d_H.p + (size_t)kMH * kMH;
    launch_finalize(s, hb, hb + 1, ctx->stream);
    HIP_TRY(hipMemcpyAsync(out, squared ? hb + 1 : hb, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int kfsp_nrm2_w(kfsp_ctx *ctx, double *out) { return reduce_w(ctx, 1, out); }
int kfsp_asum_w(kfsp_ctx *ctx, double *out) { return reduce_w(ctx, 0, out); }

int kfsp_get_basis(kfsp_ctx *ctx, int j, int64_t nlocal, double *v)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (ctx->group) return kfsp::group_get_basis(ctx, j, nlocal, v);
    if (j < 1 || j > mmax_now(ctx) + 2) return fail(ctx, -2, "bad column");
    if (nlocal != ctx->nloc) return fail(ctx, -3, "nlocal is not this rank's block size");
    if (!v && nlocal > 0) return fail(ctx, -4, "null v");
    HIP_TRY(hipSetDevice(ctx->device));
    double sq = 0.0;
    HIP_TRY(hipMemcpyAsync(&sq, ctx->d_sq.p + j, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (int rc = download_states(ctx, vcol(ctx, j - 1), v, nlocal)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const double s = 1.0 / std::sqrt(sq);
    for (int64_t i = 0; i < nlocal; ++i) v[i] *= s;
    return 0;
}

int kfsp_expv_fixed(kfsp_ctx *ctx, int m, double tau, int nsteps, double *wsums)
{
    return no_throw(ctx, [&]() -> int {
        if (!ctx) return -1;
        if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
        if (m < 1 || m > mmax_now(ctx) || (int64_t)m >= ctx->n) return fail(ctx, -2, "bad m");
        if (nsteps < 0) return fail(ctx, -4, "bad nsteps");
        const int mh = m + 2;
        std::vector<double> H((size_t)mh * mh), E((size_t)mh * mh);
        for (int s = 0; s < nsteps; ++s) {
            double beta = 0.0, avn = 0.0, hn = 0.0;
            int mb = m, k1 = 2, ns = 0;
            if (int rc = kfsp_begin_step(ctx, &beta)) return rc;
            std::fill(H.begin(), H.end(), 0.0);
            if (int rc = kfsp_arnoldi(ctx, m, 1, 2, 1.0e-7, H.data(), mh, &mb, &k1, &avn)) return rc;
            int mx = mb + k1;
            auto t0 = std::chrono::steady_clock::now();
            if (int rc = kfsp_padm(6, mx, tau, H.data(), mh, E.data(), &ns, &hn)) return fail(ctx, rc, "kfsp_padm failed");
            ctx->t_ms[KFSP_T_HOST_PADE] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            mx = mb + std::max(0, k1 - 1);
            double ws = 0.0;
            if (int rc = kfsp_combine(ctx, mx, beta, E.data(), &ws)) return rc;
            if (wsums) wsums[s] = ws;
        }
        return 0;
    });
}

int kfsp_spmv_bench(kfsp_ctx *ctx, int reps, int variant, float *ms_total)
{
    if (!ctx) return -1;
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (reps < 1) return fail(ctx, -2, "reps < 1");
    if (variant != 0 && variant != 2 && variant != 3) return fail(ctx, -3, "unknown variant (0 auto, 2 SELL, 3 SELL with plain columns)");
    if (!ms_total) return fail(ctx, -4, "null ms_total");
    if (ctx->group) return kfsp::group_spmv_bench(ctx, reps, variant, ms_total);
    if (variant >= 2 && !ctx->have_sell) return fail(ctx, -3, "no SELL image resident (banded matrix built on the device)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const double *src = vcol(ctx, 0);
    double *dst = vcol(ctx, 1);
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    for (int r = 0; r < reps; ++r) {
        if (int rc = spmv_plain(ctx, src, false, dst, variant >= 2, variant == 3)) return rc;
    }
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return 0;
}

int kfsp_exchange_bench(kfsp_ctx *ctx, int reps, float *ms_total, int64_t *bytes_in)
{
    if (!ctx) return -1;
    if (ctx->group) return fail(ctx, -9, "not available on a group context");
    if (ctx->ldv == 0) return fail(ctx, -1, "no matrix set");
    if (reps < 1) return fail(ctx, -2, "reps < 1");
    if (!ms_total || !bytes_in) return fail(ctx, -3, "null output");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    *ms_total = 0.f;
    *bytes_in = 0;
    if (!ctx->use_comm) return 0;
    const double *src = vcol(ctx, 0), *xg = nullptr;
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    for (int r = 0; r < reps; ++r)
        if (int rc = gather_source(ctx, src, &xg)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    const int64_t peers = ctx->nranks - 1;
    if (ctx->use_halo)
        *bytes_in = ctx->opt_halo_p2p ? 8 * ctx->halo * ((ctx->rank > 0) + (ctx->rank + 1 < ctx->nranks)) : 8 * 2 * ctx->halo * peers;
    else
        *bytes_in = 8 * ctx->L * peers;
    return 0;
}

int kfsp_selftest_stream(kfsp_ctx *ctx, int64_t nbytes, int elem_bytes, int reps, float *ms_total)
{
    if (!ctx) return -1;
    if (nbytes < 4096 || nbytes % 4096) return fail(ctx, -2, "nbytes must be a positive multiple of 4096");
    if (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return fail(ctx, -3, "elem_bytes must be 4, 8 or 16");
    if (reps < 1) return fail(ctx, -4, "reps < 1");
    if (ctx->group) return fail(ctx, -9, "not available on a group context");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<char> buf;
    HIP_TRY(buf.reserve((size_t)nbytes, true));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) launch_stream_read(kMaxGrid, elem_bytes, nbytes, buf.p, ctx->d_part.p, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (ms_total) *ms_total = ms;
    buf.release();
    return 0;
}

int kfsp_add_timer(kfsp_ctx *ctx, int phase, double ms)
{
    if (!ctx) return -1;
    if (phase < 0 || phase >= KFSP_T_COUNT) return -2;
    ctx->t_ms[phase] += ms;
    return 0;
}

int kfsp_get_timers(kfsp_ctx *ctx, double *ms, int reset)
{
    if (!ctx) return -1;
    if (!ms) return -2;
    if (ctx->group) return kfsp::group_get_timers(ctx, ms, reset);
    for (int i = 0; i < KFSP_T_COUNT; ++i) ms[i] = ctx->t_ms[i];
    if (reset)
        for (double &t : ctx->t_ms) t = 0.0;
    return 0;
}

int kfsp_set_option(kfsp_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx) return -1;
    if (!name) return -2;
    if (ctx->group) return kfsp::group_set_option(ctx, name, value);
    const std::string k(name);
    if (k == "grid_blocks") ctx->opt_grid = value;
    else if (k == "vec_grid_blocks") ctx->opt_vgrid = value;
    else if (k == "nt_loads") ctx->opt_nt = value;
    else if (k == "format") ctx->opt_format = value;
    else if (k == "fused_ortho") ctx->opt_fused = value;
    else if (k == "host_build") ctx->opt_host_build = value;
    else if (k == "dia_mask") ctx->opt_dia_mask = value;
    else if (k == "box_generic") ctx->opt_box_generic = value;
    else if (k == "box_lds") ctx->opt_box_lds = value;
    else if (k == "box_reach") ctx->opt_box_reach = value;
    else if (k == "box_store") ctx->opt_box_store = value;
    else if (k == "box_tile") ctx->opt_box_tile = value;
    else if (k == "box_pencil") ctx->opt_box_pencil = value;
    else if (k == "box_slab_waves") ctx->opt_box_slab_waves = value;
    else if (k == "sell_code") ctx->opt_sell_code = value;
    else if (k == "ssa_resident") ctx->opt_ssa_resident = value;
    else if (k == "keep_coords") ctx->opt_keep_coords = value;
    else if (k == "ssa_general") ctx->opt_ssa_general = value;
    else if (k == "ssa_partition") ctx->opt_ssa_partition = value;
    else if (k == "small_lds") ctx->opt_small_lds = value;
    else if (k == "state_order") ctx->opt_state_order = value;
    else if (k == "state_order_min") ctx->opt_state_order_min = value;
    else if (k == "state_order_products") ctx->opt_state_order_products = value;
    else if (k == "halo") ctx->opt_halo = value;
    else if (k == "halo_p2p") ctx->opt_halo_p2p = value;
    else if (k == "halo_sell") ctx->opt_halo_sell = value;
    else if (k == "sell_sigma") ctx->opt_sell_sigma = value;
    else if (k == "build_speculate") ctx->opt_build_speculate = value != 0;
    else if (k == "ssa_regs") ctx->opt_ssa_regs = value != 0;
    else if (k == "ssa_filter") ctx->opt_ssa_filter = value != 0;
    else if (k == "overlap") ctx->opt_overlap = value;
    else if (k == "small_kernel") ctx->opt_small = value;
    else if (k == "m_max") {
        if (value < 2 || value > kMMax) return fail(ctx, -3, "2 <= m_max <= 100");
        if (value != ctx->opt_mmax) ctx->relayout = true;      // the next generator re-lays the basis
        ctx->opt_mmax = value;
    }
    else return fail(ctx, -2, "unknown option");
    return 0;
}

}  // extern "C"
