// Generator maintenance on the device: the reference's column-oriented arrays
// (ADJ / OFFDIAG / DIAG of TYPE FSP_MATRIX, StateSpace.f90:13-17) are copied to
// HBM as they are and turned into the row-gather forms the product kernel reads
// - banded (DIA) when every reaction is a constant index shift, SELL-64
// otherwise - without touching the host again.  This is what runs after every
// MATRIX_STARTER / ONESTEP_EXTENDER / SSA_EXTENDER / DROP_STATES
// (KrylovSolver.f90:130-134, 511, 528-529), i.e. between most time steps of an
// adaptive solve, so it must cost milliseconds, not a host pass over nnz.
//
// SELL build = histogram of targets (integer atomics), per-chunk widths, one
// block-wide exclusive scan, scatter with per-row slot tickets, and a per-row
// sort by source index that makes the layout deterministic and identical to the
// order in which FMATVEC accumulates (KrylovSolver.f90:598-604).
#include "kfsp_ctx.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <climits>

namespace kfsp {

namespace {

constexpr int kMaxBw = 64;

struct ScanOut {            // device scratch, one instance
    int dmin[kMaxBw];       // per slot: min / max of (target - source) over valid links
    int dmax[kMaxBw];
    unsigned long long dcount[kMaxBw];   // valid links with the target in this rank's rows
    unsigned long long nnz_off;          // off-diagonal entries of the local rows
    int bad;                // an ADJ entry exceeded n
};

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// one lane per source state (grid-stride): link statistics per slot + in-degree of
// local rows.  The per-slot statistics are combined in LDS first: a global atomic
// per wavefront and slot on the same forty addresses costs 1.7 ms at 10^6 states.
__global__ __launch_bounds__(kBlock) void k_ell_scan(int64_t n, int bw, int ld, const int32_t *__restrict__ adj,
                                                     int64_t row0, int64_t nloc, int32_t *__restrict__ cnt,
                                                     ScanOut *__restrict__ out)
{
    __shared__ int smin[kMaxBw], smax[kMaxBw];
    __shared__ unsigned int scnt[kMaxBw];
    __shared__ int sbad;
    for (int j = threadIdx.x; j < kMaxBw; j += kBlock) {
        smin[j] = INT_MAX;
        smax[j] = INT_MIN;
        scnt[j] = 0;
    }
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < n;
        for (int j = 0; j < bw; ++j) {
            const int k = live ? adj[i * ld + j] : 0;
            if (k > n) sbad = 1;
            const bool valid = k >= 1 && k <= n;
            const int64_t r = (int64_t)k - 1 - row0;
            const bool local = valid && r >= 0 && r < nloc;
            if (local) atomicAdd(&cnt[r], 1);
            const int d = (int)((int64_t)k - 1 - i);
            const int lo = wave_min_i(valid ? d : INT_MAX);
            const int hi = wave_max_i(valid ? d : INT_MIN);
            const unsigned long long m = __ballot(local);
            if ((threadIdx.x & 63) == 0) {
                if (lo != INT_MAX) {
                    atomicMin(&smin[j], lo);
                    atomicMax(&smax[j], hi);
                }
                if (m) atomicAdd(&scnt[j], (unsigned int)__popcll(m));
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < bw; j += kBlock) {
        if (smin[j] != INT_MAX) {
            atomicMin(&out->dmin[j], smin[j]);
            atomicMax(&out->dmax[j], smax[j]);
        }
        if (scnt[j]) {
            atomicAdd(&out->dcount[j], (unsigned long long)scnt[j]);
            atomicAdd(&out->nnz_off, (unsigned long long)scnt[j]);
        }
    }
    if (threadIdx.x == 0 && sbad) out->bad = 1;
}

// ---- internal state order ------------------------------------------------------
// column i' of the relabelled arrays is the caller's column perm[i'], its links
// renumbered by iperm (entries <= 0 and out-of-range ones pass through: the scan
// reports the latter)
__global__ __launch_bounds__(kBlock) void k_ell_relabel(int64_t n, int bw, int ld, const int32_t *__restrict__ perm,
                                                        const int32_t *__restrict__ iperm,
                                                        const int32_t *__restrict__ adj, const double *__restrict__ off,
                                                        const double *__restrict__ diag, int32_t *__restrict__ adj2,
                                                        double *__restrict__ off2, double *__restrict__ diag2)
{
    // one lane per (state, slot): the ld slots of a state are read by neighbouring lanes
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n * ld) return;
    const int64_t i2 = t / ld;
    const int j = (int)(t - i2 * ld);
    const int64_t i = perm[i2];
    if (j == 0) diag2[i2] = diag[i];
    if (j < bw) {
        const int k = adj[i * ld + j];
        adj2[t] = (k >= 1 && k <= n) ? iperm[k - 1] + 1 : k;
        off2[t] = off[i * ld + j];
    } else {
        adj2[t] = 0;
        off2[t] = 0.0;
    }
}

// smallest and largest count of every species (mm[2k], mm[2k+1]); grid-stride,
// combined per workgroup in LDS before the global atomics
__global__ __launch_bounds__(kBlock) void k_coord_minmax(int64_t n, int ns, int ld, const int32_t *__restrict__ state,
                                                         int *__restrict__ mm)
{
    __shared__ int smm[32];
    if (threadIdx.x < 32) smm[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX;
    __syncthreads();
    for (int k = 0; k < ns; ++k) {
        int lo = INT_MAX, hi = INT_MIN;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
            const int v = state[i * ld + k];
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        lo = wave_min_i(lo);
        hi = wave_max_i(hi);
        if ((threadIdx.x & 63) == 0 && lo != INT_MAX) {
            atomicMin(&smm[2 * k], lo);
            atomicMax(&smm[2 * k + 1], hi);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * ns) {
        if (threadIdx.x & 1) atomicMax(&mm[threadIdx.x], smm[threadIdx.x]);
        else if (smm[threadIdx.x] != INT_MAX) atomicMin(&mm[threadIdx.x], smm[threadIdx.x]);
    }
}

// the pinned block of a context (h_build, 8 KB): where the numbers of a speculative rebuild land, and the constants it uploads
constexpr size_t kHbStats = 0;            // ScanOut: the link statistics of the build
constexpr size_t kHbSlots = 2048;         // int64: SELL slots
constexpr size_t kHbRanges = 2304;        // int[32]: min / max of every species' count over the states that were packed
constexpr size_t kHbRangesInit = 2560;    // int[32]: what those start from
constexpr size_t kHbStatsInit = 4096;     // ScanOut: what the statistics start from

struct KeyLayout {
    int ns;
    int lo[16];
    int shift[16];
};

// species 1 in the lowest bits: ascending keys = lexicographic order with the
// first species running fastest (the order of the benchmark boxes)
__global__ __launch_bounds__(kBlock) void k_pack_keys(int64_t n, int ld, const int32_t *__restrict__ state, KeyLayout L,
                                                      unsigned long long *__restrict__ keys, int32_t *__restrict__ idx, int32_t base,
                                                      unsigned long long mask)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0;
    for (int k = 0; k < L.ns; ++k) key |= (unsigned long long)(unsigned)(state[i * ld + k] - L.lo[k]) << L.shift[k];
    // (mask: the bits the sort looks at.  A state that has outgrown a speculative layout spills above them - cut off, its
    // key is wrong but the sorted list IS sorted, which the merge's binary searches rely on to hand out every place once)
    keys[i] = key & mask;
    idx[i] = base + (int32_t)i;
}

// Two sorted key lists into one: element i of list A lands i + (keys of B below it) places in, element j of B lands
// j + (keys of A not above it) - the stable merge, A first among equals.  The keys of an FSP are distinct, but a key packed
// under a layout its state has outgrown (the speculation state_order_check catches AFTER this has run) can collide with
// another: with this rule the places are a permutation whatever the keys are, so nothing downstream indexes through a hole.
// The launch covers na + nb lanes, the first na for A.  pa / pb: the caller's indices that travel with the keys.
__global__ __launch_bounds__(kBlock) void k_merge_sorted(int64_t na, const unsigned long long *__restrict__ ka, const int32_t *__restrict__ pa,
                                                         int64_t nb, const unsigned long long *__restrict__ kb, const int32_t *__restrict__ pb,
                                                         unsigned long long *__restrict__ kout, int32_t *__restrict__ pout)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= na + nb) return;
    const bool a = t < na;
    const int64_t i = a ? t : t - na;
    const unsigned long long key = a ? ka[i] : kb[i];
    const unsigned long long *__restrict__ other = a ? kb : ka;
    int64_t lo = 0, hi = a ? nb : na;
    while (lo < hi) {                                       // A: first element of B not below key; B: first of A above it
        const int64_t mid = (lo + hi) >> 1;
        if (a ? other[mid] < key : other[mid] <= key) lo = mid + 1;
        else hi = mid;
    }
    kout[i + lo] = key;
    pout[i + lo] = a ? pa[i] : pb[i];
}

// The order of the states a drop keeps: they stay in the order they had.  f[i'] = 1 when the state at internal place i' is
// kept (keep[] is in the caller's order); after the exclusive sum of f, the kept ones move up to pos[i'] and carry the
// caller's NEW index of their state (scan[]: the exclusive sum of keep[] the compaction of the lists used).
__global__ __launch_bounds__(kBlock) void k_order_keep_flags(int64_t n, const int32_t *__restrict__ perm, const uint8_t *__restrict__ keep,
                                                             int32_t *__restrict__ f)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) f[i] = keep[perm[i]] ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_order_keep_apply(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ f,
                                                             const int32_t *__restrict__ pos, const int32_t *__restrict__ scan,
                                                             const unsigned long long *__restrict__ keys, int32_t *__restrict__ perm2,
                                                             unsigned long long *__restrict__ keys2)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || !f[i]) return;
    perm2[pos[i]] = scan[perm[i]];
    keys2[pos[i]] = keys[i];
}

__global__ __launch_bounds__(kBlock) void k_invert_perm(int64_t n, const int32_t *__restrict__ perm,
                                                        int32_t *__restrict__ iperm)
{
    const int64_t i2 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i2 < n) iperm[perm[i2]] = (int32_t)i2;
}

__global__ __launch_bounds__(kBlock) void k_gather_index(int64_t n, const int32_t *__restrict__ index,
                                                         const double *__restrict__ src, double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

// one wavefront per 128-row group: bit d of mask[group] = diagonal d has an entry there
__global__ __launch_bounds__(kBlock) void k_dia_group_mask(int64_t ngroups, int nd, int64_t ld,
                                                           const double *__restrict__ val, uint32_t *__restrict__ mask,
                                                           unsigned long long *__restrict__ empty)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    uint32_t m = 0;
    for (int d = 0; d < nd; ++d) {
        const double *p = val + (int64_t)d * ld + (g << 7) + 2 * lane;
        const bool any = p[0] != 0.0 || p[1] != 0.0;
        if (__ballot(any)) m |= 1u << d;
    }
    if (lane == 0) {
        mask[g] = m;
        const int e = nd - __popc(m);
        if (e) atomicAdd(empty, (unsigned long long)e);
    }
}

// banded form: diagonal d is slot slot_of[d]; row r reads source r - shift
__global__ __launch_bounds__(kBlock) void k_ell_to_dia(int64_t n, int ld, const int32_t *__restrict__ adj,
                                                       const double *__restrict__ off,
                                                       const double *__restrict__ diag_in, int64_t row0, int64_t nloc,
                                                       int nd, DiaDev D, const int *__restrict__ slot_of,
                                                       double *__restrict__ val, double *__restrict__ diag_out)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= D.ld) return;
    const bool live = r < nloc;
    const int64_t g = row0 + r;
    for (int d = 0; d < nd; ++d) {
        double v = 0.0;
        if (live) {
            const int64_t src = g + D.delta[d];            // delta = source - target
            if (src >= 0 && src < n) {
                const int j = slot_of[d];
                if (adj[src * ld + j] == g + 1) v = off[src * ld + j];
            }
        }
        val[(int64_t)d * D.ld + r] = v;
    }
    diag_out[r] = live ? diag_in[g] : 0.0;
}

__global__ __launch_bounds__(kBlock) void k_chunk_width(int64_t nchunks, int64_t nloc, const int32_t *__restrict__ cnt,
                                                        int64_t *__restrict__ off)
{
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nchunks) return;
    int w = 0;
    const int64_t r1 = min(nloc, (c + 1) * kChunk);
    for (int64_t r = c * kChunk; r < r1; ++r) w = max(w, cnt[r]);
    off[c + 1] = (int64_t)w * kChunk;                      // scanned in place next
}

// exclusive scan of off[1..nchunks] by one workgroup (nchunks <= a few 1e5): every thread sums its stretch, the 1024 sums
// are scanned by wavefront shuffles (a single thread walking them through LDS took 20 of this kernel's 23 us), every thread
// writes its stretch.  Integers: the same offsets in whatever order they are added.
__global__ __launch_bounds__(1024) void k_scan_offsets(int64_t nchunks, int64_t *__restrict__ off)
{
    __shared__ int64_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t per = (nchunks + 1023) / 1024;
    const int64_t b = (int64_t)t * per, e = min(nchunks, b + per);
    int64_t s = 0;
    for (int64_t c = b; c < e; ++c) s += off[c + 1];
    int64_t v = s;                                          // inclusive over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int64_t run = v - s;                                    // exclusive: what lies in front of this thread's stretch
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (t == 0) off[0] = 0;
    for (int64_t c = b; c < e; ++c) {
        const int64_t x = off[c + 1];
        run += x;
        off[c + 1] = run;
    }
}

// padded slots: val = 0, column = the row itself (always a valid gather)
__global__ __launch_bounds__(kBlock) void k_sell_init(int64_t nchunks, int64_t nloc, int64_t row0,
                                                      const int64_t *__restrict__ off, int32_t *__restrict__ col,
                                                      double *__restrict__ val, const double *__restrict__ diag_in,
                                                      double *__restrict__ diag_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t o = off[c];
    const int w = (int)((off[c + 1] - o) >> 6);
    const int64_t r = c * kChunk + lane;
    const int64_t rr = r < nloc ? r : (nloc > 0 ? nloc - 1 : 0);
    for (int k = 0; k < w; ++k) {
        col[sell_pos(o, w, k, lane)] = (int32_t)(row0 + rr);
        val[sell_pos(o, w, k, lane)] = 0.0;
    }
    diag_out[r] = r < nloc ? diag_in[row0 + r] : 0.0;
}

__global__ __launch_bounds__(kBlock) void k_sell_fill(int64_t n, int bw, int ld, const int32_t *__restrict__ adj,
                                                      const double *__restrict__ offd, int64_t row0, int64_t nloc,
                                                      const int64_t *__restrict__ off, int32_t *__restrict__ ticket,
                                                      int32_t *__restrict__ col, double *__restrict__ val)
{
    // one lane per entry of the reference arrays (consecutive lanes, consecutive words)
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n * ld) return;
    const int64_t i = e / ld;
    if ((int)(e - i * ld) >= bw) return;
    const int k = adj[e];
    if (k < 1) return;
    const int64_t r = (int64_t)k - 1 - row0;
    if (r < 0 || r >= nloc) return;
    const int p = atomicAdd(&ticket[r], 1);
    const int64_t o = off[r >> 6];
    const int64_t pos = sell_pos(o, (int)((off[(r >> 6) + 1] - o) >> 6), p, (int)(r & 63));
    col[pos] = (int32_t)i;
    val[pos] = offd[e];
}

// each row's entries ascending by (source, value): deterministic, and the order
// in which the reference's scatter loop adds them.  Under the internal state order the key is the
// CALLER's index of the source (perm: internal -> caller), so that a row is summed in FMATVEC's
// order (KrylovSolver.f90:598-604) whatever order the device keeps the states in.
__global__ __launch_bounds__(kBlock) void k_sell_sort_rows(int64_t nloc, const int32_t *__restrict__ cnt,
                                                           const int64_t *__restrict__ off, int32_t *__restrict__ col,
                                                           double *__restrict__ val, const int32_t *__restrict__ perm)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= nloc) return;
    const int m = cnt[r];
    if (m < 2) return;
    const int64_t o = off[r >> 6];
    const int w = (int)((off[(r >> 6) + 1] - o) >> 6);
    const int l = (int)(r & 63);
    for (int a = 1; a < m; ++a) {                          // insertion sort, m <= #reactions
        const int32_t ca = col[sell_pos(o, w, a, l)];
        const double va = val[sell_pos(o, w, a, l)];
        const int32_t ka = perm ? perm[ca] : ca;
        int b = a - 1;
        while (b >= 0) {
            const int32_t cb = col[sell_pos(o, w, b, l)];
            const double vb = val[sell_pos(o, w, b, l)];
            const int32_t kb = perm ? perm[cb] : cb;
            if (kb < ka || (kb == ka && vb <= va)) break;
            col[sell_pos(o, w, b + 1, l)] = cb;
            val[sell_pos(o, w, b + 1, l)] = vb;
            --b;
        }
        col[sell_pos(o, w, b + 1, l)] = ca;
        val[sell_pos(o, w, b + 1, l)] = va;
    }
}

// The same order for rows of at most CAP entries, from registers: every entry is read once, ranked against the others
// (rank = how many entries sort before it under the insertion sort's rule - key, then value, then position: a stable
// ascending order, so the result is the insertion sort's, entry for entry) and written once to its place.  The insertion
// sort re-reads and re-writes the row from memory for every entry it places (96 us for the 10^6 rows of the Goutsias run,
// the longest kernel of a rebuild); loops over CAP are unrolled so nothing is indexed dynamically.
template <int CAP>
__global__ __launch_bounds__(kBlock) void k_sell_rank_rows(int64_t nloc, const int32_t *__restrict__ cnt,
                                                           const int64_t *__restrict__ off, int32_t *__restrict__ col,
                                                           double *__restrict__ val, const int32_t *__restrict__ perm)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= nloc) return;
    const int m = cnt[r];
    if (m < 2) return;
    const int64_t o = off[r >> 6];
    const int w = (int)((off[(r >> 6) + 1] - o) >> 6);
    const int l = (int)(r & 63);
    int32_t c[CAP], k[CAP];
    double v[CAP];
#pragma unroll
    for (int a = 0; a < CAP; ++a) {
        c[a] = 0;
        k[a] = 0;
        v[a] = 0.0;
        if (a < m) {
            c[a] = col[sell_pos(o, w, a, l)];
            v[a] = val[sell_pos(o, w, a, l)];
            k[a] = perm ? perm[c[a]] : c[a];
        }
    }
#pragma unroll
    for (int a = 0; a < CAP; ++a) {
        if (a < m) {
            int rank = 0;
#pragma unroll
            for (int b = 0; b < CAP; ++b) {
                if (b == a) continue;
                // b stays in front of a exactly when the insertion sort would stop a behind it: (k_b, v_b) <= (k_a, v_a)
                // for an earlier b, strictly smaller for a later one
                const bool before = b < a ? (k[b] < k[a] || (k[b] == k[a] && v[b] <= v[a]))
                                          : (k[b] < k[a] || (k[b] == k[a] && v[b] < v[a]));
                rank += (b < m && before) ? 1 : 0;
            }
            col[sell_pos(o, w, rank, l)] = c[a];
            val[sell_pos(o, w, rank, l)] = v[a];
        }
    }
}

void launch_sell_sort_rows(kfsp_ctx *ctx, int64_t nloc, int bw, hipStream_t st)
{
    if (nloc < 1) return;
    const dim3 grid((int)((nloc + kBlock - 1) / kBlock)), block(kBlock);
    const int32_t *perm = ctx->perm_on ? ctx->d_perm.p : (const int32_t *)nullptr;
    // (a row holds at most one entry per reaction)
    if (bw <= 8 && ctx->opt_build_speculate)
        hipLaunchKernelGGL(k_sell_rank_rows<8>, grid, block, 0, st, nloc, ctx->d_cnt.p, ctx->d_off.p, ctx->d_col.p, ctx->d_val.p, perm);
    else if (bw <= 12 && ctx->opt_build_speculate)
        hipLaunchKernelGGL(k_sell_rank_rows<12>, grid, block, 0, st, nloc, ctx->d_cnt.p, ctx->d_off.p, ctx->d_col.p, ctx->d_val.p, perm);
    else if (bw <= 16 && ctx->opt_build_speculate)
        hipLaunchKernelGGL(k_sell_rank_rows<16>, grid, block, 0, st, nloc, ctx->d_cnt.p, ctx->d_off.p, ctx->d_col.p, ctx->d_val.p, perm);
    else
        hipLaunchKernelGGL(k_sell_sort_rows, grid, block, 0, st, nloc, ctx->d_cnt.p, ctx->d_off.p, ctx->d_col.p, ctx->d_val.p, perm);
}

// SELL-sigma: inside windows of `sigma` rows of the internal order, the longest rows first (stable), so
// that the 64 rows of a chunk are about equally long and few slots are padding.
__global__ __launch_bounds__(kBlock) void k_sigma_keys(int64_t n, int sigma, const int32_t *__restrict__ cnt,
                                                       unsigned long long *__restrict__ keys, int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int c = cnt[i] < 127 ? cnt[i] : 127;
    keys[i] = (unsigned long long)(i / sigma) * 128ull + (unsigned long long)(127 - c);
    idx[i] = (int32_t)i;
}

__global__ __launch_bounds__(kBlock) void k_gather_i32(int64_t n, const int32_t *__restrict__ index,
                                                       const int32_t *__restrict__ src, int32_t *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

}  // namespace

#define HIP_TRY_B(expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return 1000 + (int)e_;                                                         \
        }                                                                                  \
    } while (0)

int build_from_ell_device(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, const int32_t *adj,
                          const double *offdiag, const double *diag, int64_t keep)
{
    if (bw > kMaxBw) {
        ctx->err = "more than 64 reaction slots";
        return -3;
    }
    hipStream_t st = ctx->stream;
    const size_t nent = (size_t)n * (size_t)ld;

    // the reference arrays, verbatim.  Propensities of a listed state never change while it stays listed
    // (OFFDIAG / DIAG columns are written once, StateSpace.f90:207-212): after an expansion only the
    // columns of the appended states travel; the links (ADJ) of old states do change and travel in full.
    if (keep > ctx->ell_cols || keep > n || ld != ctx->ell_ld) keep = 0;
    ctx->ell_cols = 0;
    HIP_TRY_B(ctx->d_ell_adj.reserve(nent, false));
    HIP_TRY_B(ctx->d_ell_off.reserve_keep(nent, (size_t)keep * (size_t)ld, st));
    HIP_TRY_B(ctx->d_ell_diag.reserve_keep((size_t)n, (size_t)keep, st));
    HIP_TRY_B(hipMemcpyAsync(ctx->d_ell_adj.p, adj, nent * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const size_t k0 = (size_t)keep * (size_t)ld;
    if (nent > k0)
        HIP_TRY_B(hipMemcpyAsync(ctx->d_ell_off.p + k0, offdiag + k0, (nent - k0) * sizeof(double), hipMemcpyHostToDevice, st));
    if (n > keep)
        HIP_TRY_B(hipMemcpyAsync(ctx->d_ell_diag.p + keep, diag + keep, (size_t)(n - keep) * sizeof(double), hipMemcpyHostToDevice, st));
    ctx->ell_cols = n;
    ctx->ell_ld = ld;
    ctx->ell_bw = bw;
    ctx->order_check = false;          // (an order made for an upload waited for its ranges: nothing is pending from an earlier call)
    return build_from_resident_ell(ctx, n, bw, ld);
}

// the gather form from the reference arrays that are ALREADY on the device (d_ell_adj / d_ell_off / d_ell_diag, n columns
// of leading dimension ld): what build_from_ell_device does after its upload, and all kfsp_drop_rebuild needs
bool state_order_check(kfsp_ctx *ctx);

// speculate (a resident FSP being rebuilt, kfsp_expand_resident / kfsp_drop_rebuild): when the last generator built
// here was SELL, assume this one is too and run scan, chunk widths, fill and row sort WITHOUT stopping for the link
// statistics or the slot count - the entry arrays are reserved for the bound nact * bw (a row has at most one link per
// reaction) - then wait once and check what the slow path would have looked at first: a bad link, the banded form the
// slow path would have chosen, a state order packed under stale ranges (state_order_check).  kRedoBuild: not held.
int build_from_resident_ell(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, bool speculate)
{
    hipStream_t st = ctx->stream;
    const int64_t row0 = ctx->row0, nloc = ctx->nloc;
    const int64_t nchunks = (nloc + kChunk - 1) / kChunk;
    const int64_t nact = nchunks * kChunk;
    const size_t nent = (size_t)n * (size_t)ld;
    const int32_t *ell_adj = ctx->d_ell_adj.p;
    const double *ell_off = ctx->d_ell_off.p, *ell_diag = ctx->d_ell_diag.p;
    if (ctx->perm_on) {
        // columns renumbered and reordered to the internal state order; the
        // transpose below then never knows about the caller's order
        HIP_TRY_B(ctx->d_ell_adj2.reserve(nent, false));
        HIP_TRY_B(ctx->d_ell_off2.reserve(nent, false));
        HIP_TRY_B(ctx->d_ell_diag2.reserve((size_t)n, false));
        hipLaunchKernelGGL(k_ell_relabel, dim3((int)(((int64_t)n * ld + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int64_t)n,
                           (int)bw, (int)ld, ctx->d_perm.p, ctx->d_iperm.p, ctx->d_ell_adj.p, ctx->d_ell_off.p,
                           ctx->d_ell_diag.p, ctx->d_ell_adj2.p, ctx->d_ell_off2.p, ctx->d_ell_diag2.p);
        ell_adj = ctx->d_ell_adj2.p;
        ell_off = ctx->d_ell_off2.p;
        ell_diag = ctx->d_ell_diag2.p;
    }

    HIP_TRY_B(ctx->d_cnt.reserve((size_t)std::max<int64_t>(nact, 64), false));
    HIP_TRY_B(ctx->d_scan.reserve(sizeof(ScanOut), false));
    HIP_TRY_B(hipMemsetAsync(ctx->d_cnt.p, 0, (size_t)std::max<int64_t>(nact, 64) * sizeof(int32_t), st));
    // (the initial statistics come from the pinned block: a copy from pageable memory makes the host wait for everything
    // enqueued before it - one more hidden synchronisation per rebuild)
    static_assert(sizeof(ScanOut) <= 2048, "h_build layout");
    ScanOut &init = *reinterpret_cast<ScanOut *>(ctx->h_build + kHbStatsInit);
    if (!ctx->h_build_ready) {
        for (int j = 0; j < kMaxBw; ++j) {
            init.dmin[j] = INT_MAX;
            init.dmax[j] = INT_MIN;
            init.dcount[j] = 0;
        }
        init.nnz_off = 0;
        init.bad = 0;
        ctx->h_build_ready = true;
    }
    HIP_TRY_B(hipMemcpyAsync(ctx->d_scan.p, &init, sizeof(init), hipMemcpyHostToDevice, st));
    ScanOut *dscan = reinterpret_cast<ScanOut *>(ctx->d_scan.p);
    const int gsrc = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_ell_scan, dim3(std::min(gsrc, 1024)), dim3(kBlock), 0, st, (int64_t)n, (int)bw, (int)ld, ell_adj,
                       row0, nloc, ctx->d_cnt.p, dscan);
    const bool spec = speculate && ctx->last_build_sell && ctx->h_build && nloc > 0 &&
                      !(ctx->perm_on && ctx->opt_sell_sigma >= 128) && (double)nact * (double)bw * 12.0 <= 16e9;
    ScanOut res_stack;
    ScanOut &res = spec ? *reinterpret_cast<ScanOut *>(ctx->h_build + kHbStats) : res_stack;
    HIP_TRY_B(hipMemcpyAsync(&res, dscan, sizeof(res), hipMemcpyDeviceToHost, st));
    if (!spec) {
        HIP_TRY_B(hipStreamSynchronize(st));
        if (!state_order_check(ctx)) return kRedoBuild;
        if (res.bad) {
            ctx->err = "adj entry exceeds n";
            return -5;
        }
    }
    ctx->nchunks = nchunks;
    const int64_t nact2 = round_up(nact, 2 * kChunk);      // the banded kernel works on 128-row groups
    HIP_TRY_B(ctx->d_diag.reserve((size_t)std::max<int64_t>(nact2, 128), false));
    ctx->last_build_sell = false;

    // banded?  every used slot is one constant shift, and the diagonals are full enough
    auto banded_form = [&](std::pair<int, int> *dl, int &nd) {
        nd = 0;
        bool banded = ctx->opt_format != 1 && nloc > 0;
        for (int j = 0; j < bw && banded; ++j) {
            if (res.dmin[j] == INT_MAX) continue;              // slot never links
            if (res.dmin[j] != res.dmax[j]) banded = false;
            else dl[nd++] = {-res.dmin[j], j};
        }
        if (nd == 0 || nd > kMaxDiag) banded = false;
        // (8 bytes per stored diagonal entry against 12 per SELL slot, cf. maybe_upload_dia)
        if (banded && (double)nd * (double)nloc > 1.5 * (double)res.nnz_off + 1024.0) banded = false;
        return banded;
    };
    if (spec) {
        ctx->use_dia = false;
        ctx->dia_masked = false;
        ctx->nd = 0;
        ctx->have_sell = false;
        ctx->sell_coded = false;
        ctx->sell_reach = -1;
        const int64_t bound = nact * (int64_t)bw;
        HIP_TRY_B(ctx->d_off.reserve((size_t)nchunks + 1, false));
        HIP_TRY_B(ctx->d_col.reserve((size_t)std::max<int64_t>(bound, 64), false));
        HIP_TRY_B(ctx->d_val.reserve((size_t)std::max<int64_t>(bound, 64), false));
        HIP_TRY_B(ctx->d_ticket.reserve((size_t)std::max<int64_t>(nact, 64), false));
        // (off[0] = 0 is k_scan_offsets' own first store)
        HIP_TRY_B(hipMemsetAsync(ctx->d_ticket.p, 0, (size_t)std::max<int64_t>(nact, 64) * sizeof(int32_t), st));
        hipLaunchKernelGGL(k_chunk_width, dim3((int)((nchunks + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, nchunks, nloc,
                           ctx->d_cnt.p, ctx->d_off.p);
        hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, st, nchunks, ctx->d_off.p);
        int64_t *hslots = reinterpret_cast<int64_t *>(ctx->h_build + kHbSlots);
        HIP_TRY_B(hipMemcpyAsync(hslots, ctx->d_off.p + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        hipLaunchKernelGGL(k_sell_init, dim3((int)((nchunks + 3) / 4)), dim3(kBlock), 0, st, nchunks, nloc, row0, ctx->d_off.p,
                           ctx->d_col.p, ctx->d_val.p, ell_diag, ctx->d_diag.p);
        hipLaunchKernelGGL(k_sell_fill, dim3((int)(((int64_t)n * ld + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int64_t)n,
                           (int)bw, (int)ld, ell_adj, ell_off, row0, nloc, ctx->d_off.p, ctx->d_ticket.p, ctx->d_col.p, ctx->d_val.p);
        launch_sell_sort_rows(ctx, nloc, bw, st);
        HIP_TRY_B(hipStreamSynchronize(st));
        ++ctx->spec_builds;
        if (!state_order_check(ctx)) return kRedoBuild;
        if (res.bad) {
            ctx->err = "adj entry exceeds n";
            return -5;
        }
        std::pair<int, int> dl[kMaxBw];
        int nd = 0;
        if (banded_form(dl, nd)) return kRedoBuild;           // (the slow path would store diagonals: let it)
        ctx->nnz = nloc + (int64_t)res.nnz_off;
        ctx->slots = *hslots;
        ctx->have_sell = true;
        ctx->last_build_sell = true;
        return build_sell_code(ctx);
    }
    ctx->nnz = nloc + (int64_t)res.nnz_off;
    std::pair<int, int> dl[kMaxBw];                        // (source - target, slot)
    int nd = 0;
    const bool banded = banded_form(dl, nd);

    ctx->use_dia = false;
    ctx->dia_masked = false;
    ctx->nd = 0;
    ctx->have_sell = false;
    ctx->sell_coded = false;
    ctx->sell_reach = -1;
    if (banded) {
        std::sort(dl, dl + nd);
        DiaDev D;
        D.nd = nd;
        int slot_of[kMaxDiag];
        for (int d = 0; d < nd; ++d) {
            D.delta[d] = dl[d].first;
            slot_of[d] = dl[d].second;
        }
        D.ld = nact2;
        D.n = n;
        D.nchunks = nchunks;
        D.val = nullptr;
        D.diag = nullptr;
        HIP_TRY_B(ctx->d_dia.reserve((size_t)nd * (size_t)nact2, false));
        HIP_TRY_B(ctx->d_slot.reserve(kMaxDiag, false));
        HIP_TRY_B(hipMemcpyAsync(ctx->d_slot.p, slot_of, sizeof(int) * (size_t)nd, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_ell_to_dia, dim3((int)((nact2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int64_t)n,
                           (int)ld, ell_adj, ell_off, ell_diag, row0, nloc, nd, D,
                           ctx->d_slot.p, ctx->d_dia.p, ctx->d_diag.p);
        HIP_TRY_B(hipStreamSynchronize(st));
        ctx->nd = nd;
        ctx->dia_ld = nact2;
        for (int d = 0; d < nd; ++d) ctx->delta[d] = D.delta[d];
        ctx->use_dia = true;
        ctx->slots = 0;
        return build_dia_mask(ctx);
    }

    // SELL-64
    if (ctx->perm_on && ctx->opt_sell_sigma >= 128 && nloc == n && n > ctx->opt_sell_sigma) {
        // Option sell_sigma (off by default): inside windows of sigma rows of the internal order the longest
        // rows come first (a stable sort: equal rows stay in lexicographic order, a row moves by less than
        // sigma positions).  On an SSA-grown Goutsias FSP of 10^6 states the padding of the 64-row chunks
        // falls from 38 % of the entries to 20 / 11 / 5 / 3 % with sigma = 128 / 256 / 512 / 1024 - and the
        // product takes 22.1 / 23.1 / 24.1 / 25.7 us instead of 22.4: the 64 rows of a chunk are no longer
        // neighbours, so every gather of a slot touches more lines of x, which costs more than the padded
        // slots (own x, zeros) did.  perm / iperm are replaced by the composed order and the reference
        // arrays relabelled once more; rows are still summed in the caller's column order.
        const int sigma = (int)ctx->opt_sell_sigma;
        ctx->order_n = 0;                                  // (d_perm is about to stop being the lexicographic order)
        const int grid = (int)(((int64_t)n + kBlock - 1) / kBlock);
        HIP_TRY_B(ctx->d_ticket.reserve((size_t)std::max<int64_t>(nact, 64), false));
        unsigned long long *kin = ctx->d_keys.p, *kout = ctx->d_keys.p + n;     // (reserved by state_order_from_coords)
        int32_t *ord = ctx->d_ticket.p;
        hipLaunchKernelGGL(k_sigma_keys, dim3(grid), dim3(kBlock), 0, st, (int64_t)n, sigma, ctx->d_cnt.p, kin, ctx->d_sortidx.p);
        int bits = 8;
        while ((1ull << bits) < ((unsigned long long)(n / sigma) + 1ull) * 128ull) ++bits;
        size_t tmp_bytes = 0;
        HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kin, kout, ctx->d_sortidx.p, ord, (int)n, 0, bits, st));
        HIP_TRY_B(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
        HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, kin, kout, ctx->d_sortidx.p, ord, (int)n, 0, bits, st));
        // perm'[i'] = perm[ord[i']] (internal -> caller), its inverse, the row lengths in the new order
        hipLaunchKernelGGL(k_gather_i32, dim3(grid), dim3(kBlock), 0, st, (int64_t)n, ord, ctx->d_perm.p, ctx->d_iperm.p);
        HIP_TRY_B(hipMemcpyAsync(ctx->d_perm.p, ctx->d_iperm.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_invert_perm, dim3(grid), dim3(kBlock), 0, st, (int64_t)n, ctx->d_perm.p, ctx->d_iperm.p);
        hipLaunchKernelGGL(k_gather_i32, dim3(grid), dim3(kBlock), 0, st, (int64_t)n, ord, ctx->d_cnt.p, ctx->d_sortidx.p);
        HIP_TRY_B(hipMemcpyAsync(ctx->d_cnt.p, ctx->d_sortidx.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_ell_relabel, dim3((int)(((int64_t)n * ld + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int64_t)n,
                           (int)bw, (int)ld, ctx->d_perm.p, ctx->d_iperm.p, ctx->d_ell_adj.p, ctx->d_ell_off.p,
                           ctx->d_ell_diag.p, ctx->d_ell_adj2.p, ctx->d_ell_off2.p, ctx->d_ell_diag2.p);
    }
    HIP_TRY_B(ctx->d_off.reserve((size_t)nchunks + 1, false));
    HIP_TRY_B(hipMemsetAsync(ctx->d_off.p, 0, sizeof(int64_t), st));
    if (nchunks > 0) {
        hipLaunchKernelGGL(k_chunk_width, dim3((int)((nchunks + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, nchunks,
                           nloc, ctx->d_cnt.p, ctx->d_off.p);
        hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, st, nchunks, ctx->d_off.p);
    }
    int64_t slots = 0;
    HIP_TRY_B(hipMemcpyAsync(&slots, ctx->d_off.p + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY_B(hipStreamSynchronize(st));
    ctx->slots = slots;
    HIP_TRY_B(ctx->d_col.reserve((size_t)std::max<int64_t>(slots, 64), false));
    HIP_TRY_B(ctx->d_val.reserve((size_t)std::max<int64_t>(slots, 64), false));
    HIP_TRY_B(ctx->d_ticket.reserve((size_t)std::max<int64_t>(nact, 64), false));
    HIP_TRY_B(hipMemsetAsync(ctx->d_ticket.p, 0, (size_t)std::max<int64_t>(nact, 64) * sizeof(int32_t), st));
    if (nchunks > 0) {
        hipLaunchKernelGGL(k_sell_init, dim3((int)((nchunks + 3) / 4)), dim3(kBlock), 0, st, nchunks, nloc, row0,
                           ctx->d_off.p, ctx->d_col.p, ctx->d_val.p, ell_diag, ctx->d_diag.p);
        hipLaunchKernelGGL(k_sell_fill, dim3((int)(((int64_t)n * ld + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int64_t)n, (int)bw, (int)ld, ell_adj,
                           ell_off, row0, nloc, ctx->d_off.p, ctx->d_ticket.p, ctx->d_col.p, ctx->d_val.p);
        launch_sell_sort_rows(ctx, nloc, bw, st);
    }
    HIP_TRY_B(hipStreamSynchronize(st));
    ctx->have_sell = true;
    ctx->last_build_sell = true;
    return build_sell_code(ctx);
}

int build_dia_mask(kfsp_ctx *ctx)
{
    ctx->dia_masked = false;
    ctx->dia_empty_segments = 0;
    if (!ctx->use_dia || !ctx->opt_dia_mask || ctx->nd < 1) return 0;
    hipStream_t st = ctx->stream;
    const int64_t ngroups = ctx->dia_ld >> 7;
    if (ngroups < 1) return 0;
    HIP_TRY_B(ctx->d_gmask.reserve((size_t)ngroups + 2, false));
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(ctx->d_scan.p);
    HIP_TRY_B(ctx->d_scan.reserve(sizeof(ScanOut), false));
    cnt = reinterpret_cast<unsigned long long *>(ctx->d_scan.p);
    HIP_TRY_B(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_dia_group_mask, dim3((int)((ngroups + 3) / 4)), dim3(kBlock), 0, st, ngroups, ctx->nd,
                       ctx->dia_ld, ctx->d_dia.p, ctx->d_gmask.p, cnt);
    unsigned long long empty = 0;
    HIP_TRY_B(hipMemcpyAsync(&empty, cnt, sizeof(empty), hipMemcpyDeviceToHost, st));
    HIP_TRY_B(hipStreamSynchronize(st));
    // the masked variant trades a little address arithmetic for the skipped bytes: worth it from ~3 % on
    ctx->dia_masked = (double)empty >= 0.03 * (double)ctx->nd * (double)ngroups;
    ctx->dia_empty_segments = (int64_t)empty;
    return 0;
}

// Dictionary-coded columns of a SELL chunk (kSellCode*, kfsp_internal.h): one wavefront per chunk collects the distinct
// offsets col - row of the chunk's entries, slot by slot in the rows' stored (FMATVEC) order, into a table held
// one entry per lane, and writes every entry's 6-bit index into the code words.  More than 64 distinct offsets:
// the chunk keeps its plain columns (dtlen = 0).
__global__ __launch_bounds__(kBlock) void k_sell_code(int64_t nchunks, int64_t row0, const int64_t *__restrict__ off,
                                                      const int32_t *__restrict__ col, const int64_t *__restrict__ codeoff,
                                                      int32_t *__restrict__ dtab, int32_t *__restrict__ dtlen,
                                                      unsigned long long *__restrict__ code, unsigned long long *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t o = off[c];
    const int w = (int)((off[c + 1] - o) >> 6);
    const int64_t g = row0 + (c << 6) + lane;
    const int64_t cbase = codeoff[c];
    int tab = 0, cnt = 0;
    bool overflow = false;
    unsigned long long word = 0;
    for (int q = 0; q < w && !overflow; ++q) {
        const int d = (int)((int64_t)col[sell_pos(o, w, q, lane)] - g);
        int id = 0;
        unsigned long long todo = __ballot(1);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const int dv = __builtin_amdgcn_readlane(d, src);
            const unsigned long long hit = __ballot(lane < cnt && tab == dv);
            int idx;
            if (hit) {
                idx = __ffsll((long long)hit) - 1;
            } else {
                if (cnt == 64) {
                    overflow = true;
                    break;
                }
                idx = cnt;
                if (lane == cnt) tab = dv;
                ++cnt;
            }
            const unsigned long long same = __ballot(d == dv);
            if (d == dv) id = idx;
            todo &= ~same;
        }
        word |= (unsigned long long)id << (kSellCodeBits * (q % kSellCodePerWord));
        if (q % kSellCodePerWord == kSellCodePerWord - 1 || q == w - 1) {
            code[cbase + (int64_t)(q / kSellCodePerWord) * 64 + lane] = word;
            word = 0;
        }
    }
    if (overflow || w == 0) cnt = 0;
    dtab[(c << 6) + lane] = lane < cnt ? tab : 0;
    if (lane == 0) {
        dtlen[c] = cnt;
        if (cnt > 0) {
            atomicAdd(stats + 0, 1ull);                                          // coded chunks
            atomicAdd(stats + 1, (unsigned long long)w * 64ull);                 // their entries
            atomicAdd(stats + 2, (unsigned long long)((cnt * 4 + 63) / 64) * 64ull);   // table bytes in 64-byte lines
            atomicAdd(stats + 3, (unsigned long long)((w + kSellCodePerWord - 1) / kSellCodePerWord) * 64ull);   // code words
        }
    }
}

// code words per chunk (then scanned in place): ceil(width / 10) words per lane
__global__ __launch_bounds__(kBlock) void k_code_width(int64_t nchunks, const int64_t *__restrict__ off, int64_t *__restrict__ codeoff)
{
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nchunks) return;
    const int w = (int)((off[c + 1] - off[c]) >> 6);
    codeoff[c + 1] = (int64_t)((w + kSellCodePerWord - 1) / kSellCodePerWord) * 64;
}

// max |col - row| over the stored entries (padding has col = row): the reach of the local rows
__global__ __launch_bounds__(kBlock) void k_sell_reach(int64_t nchunks, int64_t row0, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ col, unsigned long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t o = off[c];
    const int w = (int)((off[c + 1] - o) >> 6);
    const int64_t g = row0 + (c << 6) + lane;
    long long m = 0;
    for (int q = 0; q < w; ++q) {
        const long long d = (long long)col[sell_pos(o, w, q, lane)] - g;
        m = max(m, d < 0 ? -d : d);
    }
    for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s, 64));
    if (lane == 0) atomicMax(out, (unsigned long long)m);
}

int build_sell_code(kfsp_ctx *ctx)
{
    ctx->sell_coded = false;
    ctx->code_words = ctx->coded_chunks = ctx->coded_slots = ctx->coded_tab_bytes = 0;
    ctx->sell_reach = -1;
    if (!ctx->have_sell || ctx->nchunks < 1) return 0;
    hipStream_t st = ctx->stream;
    const int64_t nchunks = ctx->nchunks;
    HIP_TRY_B(ctx->d_scan.reserve(sizeof(ScanOut), false));
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(ctx->d_scan.p);
    // the reach of the rows (a bounded reach lets a partitioned product exchange halo strips instead of whole vectors):
    // only a communicator asks for it
    const bool want_reach = ctx->use_comm && ctx->opt_halo_sell != 0;
    // auto: under the internal state order AND only for generators that cannot stay in the 256 MiB Infinity Cache - a
    // cache-resident product is not bound by bytes, and coding costs about 30 products (Goutsias run, T = 300, N <= 1e6:
    // Arnoldi 1.02 -> 1.09 s, uploads + 0.26 s with the columns coded at every FSP change; profiles/r03_end_to_end_goutsias.txt)
    const bool want = ctx->opt_sell_code > 0 ||
                      (ctx->opt_sell_code < 0 && ctx->perm_on && (double)ctx->slots * 12.0 > 256.0 * 1024 * 1024);
    if (!want && !want_reach) return 0;
    HIP_TRY_B(hipMemsetAsync(stats, 0, 8 * sizeof(unsigned long long), st));
    if (want_reach)
        hipLaunchKernelGGL(k_sell_reach, dim3((int)((nchunks + 3) / 4)), dim3(kBlock), 0, st, nchunks, ctx->row0, ctx->d_off.p,
                           ctx->d_col.p, stats + 4);
    if (want) {
        HIP_TRY_B(ctx->d_codeoff.reserve((size_t)nchunks + 1, false));
        HIP_TRY_B(hipMemsetAsync(ctx->d_codeoff.p, 0, sizeof(int64_t), st));
        hipLaunchKernelGGL(k_code_width, dim3((int)((nchunks + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, nchunks, ctx->d_off.p,
                           ctx->d_codeoff.p);
        hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, st, nchunks, ctx->d_codeoff.p);
        int64_t words = 0;
        HIP_TRY_B(hipMemcpyAsync(&words, ctx->d_codeoff.p + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY_B(hipStreamSynchronize(st));
        HIP_TRY_B(ctx->d_code.reserve((size_t)std::max<int64_t>(words, 64), false));
        HIP_TRY_B(ctx->d_dtab.reserve((size_t)nchunks * 64, false));
        HIP_TRY_B(ctx->d_dtlen.reserve((size_t)nchunks, false));
        hipLaunchKernelGGL(k_sell_code, dim3((int)((nchunks + 3) / 4)), dim3(kBlock), 0, st, nchunks, ctx->row0, ctx->d_off.p,
                           ctx->d_col.p, ctx->d_codeoff.p, ctx->d_dtab.p, ctx->d_dtlen.p, ctx->d_code.p, stats);
    }
    unsigned long long h[8];
    HIP_TRY_B(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY_B(hipStreamSynchronize(st));
    if (want_reach) ctx->sell_reach = (int64_t)h[4];
    if (!want) return 0;
    ctx->coded_chunks = (int64_t)h[0];
    ctx->coded_slots = (int64_t)h[1];
    ctx->coded_tab_bytes = (int64_t)h[2];
    ctx->code_words = (int64_t)h[3];
    // worth it when the coded chunks save bytes overall: 4 B of column per entry against 8 B per code word + the tables
    const double saved = 4.0 * (double)ctx->coded_slots - 8.0 * (double)ctx->code_words - (double)ctx->coded_tab_bytes;
    ctx->sell_coded = ctx->opt_sell_code > 0 ? ctx->coded_chunks > 0 : saved > 0.02 * 12.0 * (double)ctx->slots;
    return 0;
}

// DROP_STATES on the device's own copy of the reference arrays (StateSpace.f90:500-546): the kept states move up in list order
// (scan[i] = kept states before i), links are renumbered through the same scan, a dropped target becomes 0 (:540-545), -1 stays.
// one lane per ENTRY of the link / propensity arrays (consecutive lanes read consecutive words; a lane per state walked its
// column with a stride of ld words), the first lane of a column also moves DIAG and the coordinates
__global__ __launch_bounds__(kBlock) void k_ell_compact(int64_t n, int bw, int ld, int ns, int lds, const uint8_t *__restrict__ keep,
                                                        const int32_t *__restrict__ scan, const int32_t *__restrict__ adj,
                                                        const double *__restrict__ off, const double *__restrict__ diag,
                                                        const int32_t *__restrict__ coords, int32_t *__restrict__ adj2,
                                                        double *__restrict__ off2, double *__restrict__ diag2, int32_t *__restrict__ coords2)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n * ld) return;
    const int64_t i = e / ld;
    if (!keep[i]) return;
    const int k = (int)(e - i * ld);
    const int64_t q = scan[i];
    int32_t a = adj[e];
    if (k < bw && a > 0) a = keep[a - 1] ? scan[a - 1] + 1 : 0;
    adj2[q * ld + k] = a;
    off2[q * ld + k] = off[e];
    if (k == 0) {
        diag2[q] = diag[i];
        if (coords)
            for (int s = 0; s < lds; ++s) coords2[q * lds + s] = coords[i * lds + s];
    }
}

__global__ __launch_bounds__(kBlock) void k_u8_to_i32(int64_t n, const uint8_t *__restrict__ a, int32_t *__restrict__ b)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) b[i] = a[i] ? 1 : 0;
}

// keep: one byte per state of the resident arrays (caller's order); with_coords: d_coords holds their coordinates (ns used,
// leading dimension lds).  Afterwards d_ell_* (and d_coords) hold the n_keep kept columns, renumbered.
int compact_resident_ell(kfsp_ctx *ctx, int64_t n, int bw, int ld, const uint8_t *keep, int64_t n_keep, bool with_coords, int lds)
{
    hipStream_t st = ctx->stream;
    const size_t nent2 = (size_t)n_keep * (size_t)ld;
    HIP_TRY_B(ctx->d_ell_adj2.reserve(std::max<size_t>(nent2, 64), false));
    HIP_TRY_B(ctx->d_ell_off2.reserve(std::max<size_t>(nent2, 64), false));
    HIP_TRY_B(ctx->d_ell_diag2.reserve(std::max<size_t>((size_t)n_keep, 64), false));
    if (with_coords) HIP_TRY_B(ctx->d_coords2.reserve((size_t)n_keep * (size_t)lds + 128, false));
    HIP_TRY_B(ctx->d_sortidx.reserve(2 * (size_t)n + 64, false));
    int32_t *flag32 = ctx->d_sortidx.p, *scan = ctx->d_sortidx.p + n;
    const int grid = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_u8_to_i32, dim3(grid), dim3(kBlock), 0, st, n, keep, flag32);
    size_t tmp_bytes = 0;
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flag32, scan, (int)n, st));
    HIP_TRY_B(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(ctx->d_sorttmp.p, tmp_bytes, flag32, scan, (int)n, st));
    const int grid_e = (int)((n * (int64_t)ld + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_ell_compact, dim3(grid_e), dim3(kBlock), 0, st, n, bw, ld, lds, lds, keep, scan, ctx->d_ell_adj.p,
                       ctx->d_ell_off.p, ctx->d_ell_diag.p, with_coords ? ctx->d_coords.p : (const int32_t *)nullptr, ctx->d_ell_adj2.p,
                       ctx->d_ell_off2.p, ctx->d_ell_diag2.p, with_coords ? ctx->d_coords2.p : (int32_t *)nullptr);
    // (everything that follows is ordered behind this on the same stream; the rebuild waits once, at its end)
    if (!ctx->opt_build_speculate || ctx->use_comm) HIP_TRY_B(hipStreamSynchronize(st));
    std::swap(ctx->d_ell_adj, ctx->d_ell_adj2);
    std::swap(ctx->d_ell_off, ctx->d_ell_off2);
    std::swap(ctx->d_ell_diag, ctx->d_ell_diag2);
    if (with_coords) std::swap(ctx->d_coords, ctx->d_coords2);
    return 0;
}

// A box generator (kfsp_set_matrix_box) written out as stored diagonals, on the device: row r of the block
// (global state g = row0 + r) gets, for the reaction at sorted position p, val[p * ld + r] = a_p(x - nu_p) when the
// source state x - nu_p lies in the box (else 0) - the product of the factor tables in the order the kernel of
// the matrix-free form multiplies them - and diag[r] = sum of ALL propensities at x in the model's reaction
// order (StateSpace.f90:207-212).  With one factor per propensity the entries are the table entries themselves,
// i.e. exactly what kfsp_set_matrix_csr would be given for this box.  One lane per row; runs once per generator.
__global__ __launch_bounds__(kBlock) void k_box_materialize(BoxDev B, const double *__restrict__ tab, int64_t row0,
                                                            int64_t nloc, int64_t ld, double *__restrict__ val,
                                                            double *__restrict__ diag)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= ld) return;
    if (r >= nloc) {                                        // padding rows of the last 128-row group
        for (int p = 0; p < B.nr; ++p) val[(int64_t)p * ld + r] = 0.0;
        diag[r] = 0.0;
        return;
    }
    int x[kBoxMaxS];
    uint64_t q = (uint64_t)(row0 + r);
    for (int s = 0; s < kBoxMaxS; ++s) {
        x[s] = 0;
        if (s < B.ns) {
            x[s] = (int)(q % (uint64_t)B.dims[s]);
            q /= (uint64_t)B.dims[s];
        }
    }
    double ax[kBoxMaxR];
    for (int p = 0; p < B.nr; ++p) {
        // a_p at x itself (for DIAG) and at the source state x - nu_p (the stored entry)
        double a0 = 1.0, a1 = 1.0;
        for (int i = 0; i < B.ndep[p]; ++i) {
            const int xs = x[B.dep_s[p][i]];
            const double f0 = tab[B.dep_off[p][i] + xs];
            a0 = i == 0 ? f0 : a0 * f0;
            const int xsrc = xs - B.dep_nu[p][i];
            const double f1 = (xsrc >= 0 && xsrc < B.dims[B.dep_s[p][i]]) ? tab[B.dep_off[p][i] + xsrc] : 0.0;
            a1 = i == 0 ? f1 : a1 * f1;
        }
        bool in = true;
        for (int i = 0; i < B.nmov[p]; ++i) {
            const int v = x[B.mov_s[p][i]] - B.mov_nu[p][i];
            in = in && v >= 0 && v < B.mov_dim[p][i];
        }
        ax[p] = a0;
        val[(int64_t)p * ld + r] = in ? a1 : 0.0;
    }
    double dsum = 0.0;
    for (int k = 0; k < B.nr; ++k) {
        const int p = B.dorder[k];
        double a = ax[0];
        for (int j = 1; j < kBoxMaxR; ++j) a = (p == j) ? ax[j] : a;
        dsum += a;
    }
    diag[r] = dsum;
}

int box_materialize(kfsp_ctx *ctx)
{
    hipStream_t st = ctx->stream;
    const int64_t ld = ctx->dia_ld;
    HIP_TRY_B(ctx->d_dia.reserve((size_t)ctx->box.nr * (size_t)ld, false));
    HIP_TRY_B(ctx->d_diag.reserve((size_t)ld + 2 * kChunk, true));
    if (ld > 0)
        hipLaunchKernelGGL(k_box_materialize, dim3((int)((ld + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ctx->box,
                           ctx->d_box.p, ctx->row0, ctx->nloc, ld, ctx->d_dia.p, ctx->d_diag.p);
    HIP_TRY_B(hipStreamSynchronize(st));
    return 0;
}

void launch_gather_index(int64_t n, const int32_t *index, const double *src, double *dst, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(k_gather_index, dim3((int)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, index, src,
                           dst);
}

int state_order_from_coords(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state, bool *ok, int64_t keep, bool order)
{
    *ok = false;
    // keep: leading states whose coordinates are resident and unchanged (the FSP only grew): only the rest travels
    if (keep > ctx->coords_n || keep > n || ld != ctx->coords_ld || ns != ctx->coords_ns) keep = 0;
    ctx->coords_n = 0;
    if (ns > 16) return 0;                                 // more species than the key layout holds
    hipStream_t st = ctx->stream;
    const size_t nent = (size_t)n * (size_t)ld, k0 = (size_t)keep * (size_t)ld;
    HIP_TRY_B(ctx->d_coords.reserve_keep(nent + 64, k0, st));
    if (nent > k0)
        HIP_TRY_B(hipMemcpyAsync(ctx->d_coords.p + k0, state + k0, (nent - k0) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!order) {                                          // (option keep_coords: resident for the expansion, the caller's order stays)
        HIP_TRY_B(hipStreamSynchronize(st));
        ctx->coords_n = n;
        ctx->coords_ld = ld;
        ctx->coords_ns = ns;
        return 0;
    }
    return state_order_from_resident(ctx, n, ns, ld, ok);
}

// the same from coordinates that are already in d_coords (n x ld, room for 64 more entries behind them).
// speculate: when the key layout of the last order is still here (kc_*), pack with IT - its fields are as wide as the bits
// they occupy, so it holds until a coordinate range crosses a power of two - and let the ranges of this FSP land in the
// pinned block: build_from_resident_ell checks them at its one synchronisation (order_check) and answers kRedoBuild when
// a coordinate fell outside.  The ORDER does not depend on the layout (lexicographic, first species fastest), only its
// validity does: the speculative order is the order the slow path would have made, bit for bit.
// n_prev > 0 (with speculate): the first n_prev states are the FSP the current order (d_perm, its sorted keys d_skeys) was
// made for and the rest were appended to it - only THEIR keys are packed, checked and sorted, and merged into the kept
// list (KrylovSolver.f90:528-529 appends; StateSpace.f90:136-246 never moves an old state).
int state_order_from_resident(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, bool *ok, bool speculate, int32_t n_prev)
{
    *ok = false;
    ctx->coords_n = 0;
    ctx->order_check = false;
    const int64_t had_order = ctx->order_n;
    ctx->order_n = 0;
    if (ns > 16) return 0;
    hipStream_t st = ctx->stream;
    const size_t nent = (size_t)n * (size_t)ld;
    const bool spec = speculate && ctx->kc_ok && ctx->kc_ns == ns && ctx->h_build && n > 0;
    const bool merge = spec && n_prev > 0 && n_prev < n && ctx->perm_on && had_order == n_prev && ctx->skeys_n == n_prev &&
                       ctx->opt_sell_sigma < 128;
    const int64_t first = merge ? n_prev : 0, count = (int64_t)n - first;      // the states whose keys are made now
    int mm_stack[32];
    int *mm = spec ? reinterpret_cast<int *>(ctx->h_build + kHbRanges) : mm_stack;      // (pinned: the copy back does not stop the host)
    int *mm_init = spec ? reinterpret_cast<int *>(ctx->h_build + kHbRangesInit) : mm_stack;
    for (int k = 0; k < ns; ++k) {
        mm_init[2 * k] = INT_MAX;
        mm_init[2 * k + 1] = INT_MIN;
    }
    int *dmm = reinterpret_cast<int *>(ctx->d_coords.p + nent);
    HIP_TRY_B(hipMemcpyAsync(dmm, mm_init, sizeof(int) * 2 * (size_t)ns, hipMemcpyHostToDevice, st));
    const int grid = (int)((count + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_coord_minmax, dim3(std::min(grid, 1024)), dim3(kBlock), 0, st, count, (int)ns, (int)ld,
                       ctx->d_coords.p + (size_t)first * (size_t)ld, dmm);
    HIP_TRY_B(hipMemcpyAsync(mm, dmm, sizeof(int) * 2 * (size_t)ns, hipMemcpyDeviceToHost, st));
    KeyLayout L;
    L.ns = ns;
    int bits = 0;
    if (spec) {
        ctx->order_check = true;
        for (int k = 0; k < ns; ++k) {
            L.lo[k] = ctx->kc_lo[k];
            L.shift[k] = ctx->kc_shift[k];
        }
        bits = ctx->kc_bits;
    } else {
        HIP_TRY_B(hipStreamSynchronize(st));
        ctx->kc_ok = false;
        for (int k = 0; k < ns; ++k) {
            const long long range = (long long)mm[2 * k + 1] - (long long)mm[2 * k];
            if (range < 0) return 0;                           // no states
            int b = 0;
            while ((1LL << b) <= range) ++b;
            L.lo[k] = mm[2 * k];
            L.shift[k] = bits;
            bits += b;
            ctx->kc_lo[k] = L.lo[k];
            ctx->kc_shift[k] = L.shift[k];
            ctx->kc_hi[k] = (int)std::min<long long>((long long)L.lo[k] + (1LL << b) - 1, (long long)INT_MAX);
        }
        if (bits > 64) return 0;                               // does not pack: keep the caller's order
        if (bits == 0) bits = 1;
        ctx->kc_ns = ns;
        ctx->kc_bits = bits;
        ctx->kc_ok = true;
    }
    ctx->coords_n = n;                                     // (the coordinates stay resident: kfsp_ssa_streams may use them)
    ctx->coords_ld = ld;
    ctx->coords_ns = ns;
    HIP_TRY_B(ctx->d_keys.reserve(2 * (size_t)n, false));
    HIP_TRY_B(ctx->d_sortidx.reserve(std::max((size_t)n, 2 * (size_t)count), false));
    HIP_TRY_B(ctx->d_iperm.reserve((size_t)n, false));
    const int gridn = (int)(((int64_t)n + kBlock - 1) / kBlock);
    unsigned long long *kin = ctx->d_keys.p;
    hipLaunchKernelGGL(k_pack_keys, dim3(grid), dim3(kBlock), 0, st, count, (int)ld, ctx->d_coords.p + (size_t)first * (size_t)ld, L, kin,
                       ctx->d_sortidx.p, (int32_t)first, bits >= 64 ? ~0ull : (1ull << bits) - 1ull);
    if (merge) {
        // the appended keys sorted behind kin (d_keys holds 2 n), then both lists into the second pair of buffers
        unsigned long long *knew = ctx->d_keys.p + count;
        int32_t *inew = ctx->d_sortidx.p + count;
        size_t tmp_bytes = 0;
        HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kin, knew, ctx->d_sortidx.p, inew, (int)count, 0, bits, st));
        HIP_TRY_B(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
        HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, kin, knew, ctx->d_sortidx.p, inew, (int)count, 0, bits, st));
        HIP_TRY_B(ctx->d_perm2.reserve((size_t)n, false));
        HIP_TRY_B(ctx->d_skeys2.reserve((size_t)n, false));
        hipLaunchKernelGGL(k_merge_sorted, dim3(gridn), dim3(kBlock), 0, st, (int64_t)n_prev, ctx->d_skeys.p, ctx->d_perm.p, count, knew, inew,
                           ctx->d_skeys2.p, ctx->d_perm2.p);
        std::swap(ctx->d_perm, ctx->d_perm2);
        std::swap(ctx->d_skeys, ctx->d_skeys2);
        ++ctx->order_merges;
    } else {
        HIP_TRY_B(ctx->d_perm.reserve((size_t)n, false));
        HIP_TRY_B(ctx->d_skeys.reserve((size_t)n, false));
        size_t tmp_bytes = 0;
        HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kin, ctx->d_skeys.p, ctx->d_sortidx.p, ctx->d_perm.p, (int)n,
                                                     0, bits, st));
        HIP_TRY_B(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
        HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, kin, ctx->d_skeys.p, ctx->d_sortidx.p, ctx->d_perm.p,
                                                     (int)n, 0, bits, st));
    }
    hipLaunchKernelGGL(k_invert_perm, dim3(gridn), dim3(kBlock), 0, st, (int64_t)n, ctx->d_perm.p, ctx->d_iperm.p);
    if (!spec) HIP_TRY_B(hipStreamSynchronize(st));
    ctx->order_n = n;
    ctx->skeys_n = n;
    *ok = true;
    return 0;
}

// The order after a drop (kfsp_drop_rebuild), from the order before it: the kept states keep their relative order, so
// the sorted keys and the permutation are compacted instead of made again - no key is packed, nothing is sorted, nothing
// needs a check (the kept states are a subset of states whose keys were checked).  keep / scan: one byte per OLD state and
// the exclusive sum of it, both in the caller's order (compact_resident_ell leaves them in d_dropflag / d_sortidx).
// false: the order of the old FSP is not here - make it from the coordinates.
bool state_order_after_drop(kfsp_ctx *ctx, int32_t n_old, int32_t n_new, const uint8_t *keep, const int32_t *scan, int *rc)
{
    *rc = 0;
    if (!(ctx->perm_on && ctx->order_n == n_old && ctx->skeys_n == n_old && ctx->kc_ok && ctx->opt_sell_sigma < 128 && n_new > 0)) return false;
    hipStream_t st = ctx->stream;
    auto bad = [&](hipError_t e) {
        if (e == hipSuccess) return false;
        ctx->err = std::string("state_order_after_drop: ") + hipGetErrorString(e);
        *rc = 1000 + (int)e;
        return true;
    };
    ctx->order_n = 0;
    ctx->order_check = false;
    if (bad(ctx->d_keys.reserve(2 * (size_t)n_old, false))) return true;
    int32_t *f = reinterpret_cast<int32_t *>(ctx->d_keys.p), *pos = f + n_old;          // (scratch: 2 n_old int32 of the 2 n_old keys)
    const int grid = (int)(((int64_t)n_old + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_order_keep_flags, dim3(grid), dim3(kBlock), 0, st, (int64_t)n_old, ctx->d_perm.p, keep, f);
    size_t tmp_bytes = 0;
    if (bad(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, f, pos, (int)n_old, st))) return true;
    if (bad(ctx->d_sorttmp.reserve(tmp_bytes + 256, false))) return true;
    if (bad(hipcub::DeviceScan::ExclusiveSum(ctx->d_sorttmp.p, tmp_bytes, f, pos, (int)n_old, st))) return true;
    if (bad(ctx->d_perm2.reserve((size_t)n_old, false)) || bad(ctx->d_skeys2.reserve((size_t)n_old, false)) ||
        bad(ctx->d_iperm.reserve((size_t)n_old, false)))
        return true;
    hipLaunchKernelGGL(k_order_keep_apply, dim3(grid), dim3(kBlock), 0, st, (int64_t)n_old, ctx->d_perm.p, f, pos, scan, ctx->d_skeys.p,
                       ctx->d_perm2.p, ctx->d_skeys2.p);
    std::swap(ctx->d_perm, ctx->d_perm2);
    std::swap(ctx->d_skeys, ctx->d_skeys2);
    hipLaunchKernelGGL(k_invert_perm, dim3((int)(((int64_t)n_new + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int64_t)n_new, ctx->d_perm.p,
                       ctx->d_iperm.p);
    ctx->order_n = n_new;
    ctx->skeys_n = n_new;
    ++ctx->order_merges;
    return true;
}

// the ranges the speculative order was packed under, once the stream has been waited for
bool state_order_check(kfsp_ctx *ctx)
{
    if (!ctx->order_check) return true;
    ctx->order_check = false;
    const int *mm = reinterpret_cast<const int *>(ctx->h_build + kHbRanges);
    for (int k = 0; k < ctx->kc_ns; ++k)
        if (mm[2 * k] < ctx->kc_lo[k] || mm[2 * k + 1] > ctx->kc_hi[k]) {
            ctx->kc_ok = false;
            return false;
        }
    return true;
}

}  // namespace kfsp
