// ONESTEP_EXTENDER on the device (the deterministic half of the state-space
// expansion; StateSpace.f90:347-396 with ADD_STATE :136-246).  All the integer work of one
// reachability sweep: every open link (ADJ = 0) of every listed state is followed; targets that are
// listed are linked, the others become new states - the DISTINCT targets in the order in which the
// reference's double loop (state by state, reaction by reaction) meets them first - and the link
// columns of old and new states are completed (every pair of listed states ends up linked, negative
// successors are -1, unlisted ones 0).  Propensities of the new states, the host's own look-up table
// and its lists stay with the host, which receives the new states and the complete link array.
//
// No hash table: states are packed into 64-bit keys (per-species bit fields sized from the
// populations present), listed keys are radix-sorted once and searched by bisection, duplicates
// among the targets are removed by a stable sort (candidates are generated in the reference's
// (state, reaction) order, so the first of equal keys is the first appearance) - every step is
// deterministic.  hipCUB does the sorts and scans; this is off the hot path (one call per expansion).
#include "kfsp_ctx.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>

namespace kfsp {

namespace {

constexpr int kOsMaxS = 16, kOsMaxR = 64;
constexpr unsigned long long kNoKey = ~0ULL;

struct OsModel {
    int ns, nr, max_count;
    int shift[kOsMaxS], bits[kOsMaxS];
    signed char nu[kOsMaxR][kOsMaxS];
};

__device__ __forceinline__ bool os_target(const OsModel &M, const int32_t *x, int k, unsigned long long *key, bool *negative)
{
    // y = x + nu_k: negative -> not a state (-1 links); above MAXNUMBERMOLECULES or beyond the key's
    // bit fields -> cannot be listed
    unsigned long long kk = 0;
    bool neg = false, ok = true;
    for (int s = 0; s < M.ns; ++s) {
        const int y = x[s] + M.nu[k][s];
        neg = neg || y < 0;
        ok = ok && y >= 0 && y <= M.max_count && (y >> M.bits[s]) == 0;
        kk |= (unsigned long long)(unsigned)(y < 0 ? 0 : y) << M.shift[s];
    }
    *key = kk;
    *negative = neg;
    return ok && !neg;
}

// position of key in the ascending array a[0..n), or -1
__device__ __forceinline__ int os_find(const unsigned long long *a, int n, unsigned long long key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && a[lo] == key) ? lo : -1;
}

// largest population of every species: a grid-stride sweep, one atomic per wavefront and species at the end (one atomic per
// state and species, as this kernel first did, serialises on ns words: 0.56 ms at 1e6 states)
__global__ void k_os_max(int n, int ns, int ld, const int32_t *__restrict__ state, int *__restrict__ mx)
{
    int m[kOsMaxS];
    for (int s = 0; s < kOsMaxS; ++s) m[s] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int s = 0; s < kOsMaxS; ++s)
            if (s < ns) m[s] = max(m[s], state[(size_t)i * ld + s]);
    for (int s = 0; s < kOsMaxS; ++s) {
        if (s >= ns) break;
        int v = m[s];
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(mx + s, v);
    }
}

__global__ void k_os_pack(int n, int ld, const int32_t *__restrict__ state, OsModel M, unsigned long long *__restrict__ key,
                          int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = 0;
    for (int s = 0; s < M.ns; ++s) k |= (unsigned long long)(unsigned)state[(size_t)i * ld + s] << M.shift[s];
    key[i] = k;
    idx[i] = i;
}

// open links of state j whose target is a legal state
__global__ void k_os_count(int n, int lds, int lda, const int32_t *__restrict__ state, const int32_t *__restrict__ adj, OsModel M,
                           int *__restrict__ cnt)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int c = 0;
    for (int k = 0; k < M.nr; ++k) {
        if (adj[(size_t)j * lda + k] != 0) continue;
        unsigned long long key;
        bool neg;
        c += os_target(M, state + (size_t)j * lds, k, &key, &neg);
    }
    cnt[j] = c;
}

// candidates in (state, reaction) order; a target that is already listed is linked at once
__global__ void k_os_fill(int n, int lds, int lda, const int32_t *__restrict__ state, const int32_t *__restrict__ adj, OsModel M,
                          const int *__restrict__ off, const unsigned long long *__restrict__ okey, const int *__restrict__ oidx,
                          unsigned long long *__restrict__ ckey, int *__restrict__ cord, int *__restrict__ cj, int *__restrict__ ck,
                          int32_t *__restrict__ adj_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int c = off[j];
    for (int k = 0; k < M.nr; ++k) {
        if (adj[(size_t)j * lda + k] != 0) continue;
        unsigned long long key;
        bool neg;
        if (!os_target(M, state + (size_t)j * lds, k, &key, &neg)) {
            // (a state whose column arrives unlinked - all zeros - gets its -1 links here; a column that was linked before has
            // them already and no open link with a negative target)
            if (neg) adj_out[(size_t)j * lda + k] = -1;
            continue;
        }
        const int p = os_find(okey, n, key);
        if (p >= 0) {
            adj_out[(size_t)j * lda + k] = oidx[p] + 1;
            key = kNoKey;
        }
        ckey[c] = key;
        cord[c] = c;
        cj[c] = j;
        ck[c] = k;
        ++c;
    }
}

// sorted candidates: head of every run of equal keys (the first appearance of that target)
__global__ void k_os_heads(int nc, const unsigned long long *__restrict__ key, int *__restrict__ head)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nc) return;
    head[p] = (key[p] != kNoKey && (p == 0 || key[p] != key[p - 1])) ? 1 : 0;
}

// gid = inclusive scan of head - 1; the unique targets with their first candidate
__global__ void k_os_unique(int nc, const unsigned long long *__restrict__ key, const int *__restrict__ ord, const int *__restrict__ head,
                            const int *__restrict__ gscan, unsigned long long *__restrict__ ukey, int *__restrict__ uord,
                            int *__restrict__ ugid)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nc || !head[p]) return;
    const int g = gscan[p];               // exclusive scan at a head = its group number
    ukey[g] = key[p];
    uord[g] = ord[p];
    ugid[g] = g;
}

// r-th new state (by first appearance) is group grp[r]
__global__ void k_os_rank(int nu, const int *__restrict__ grp, int n_old, int *__restrict__ newidx)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nu) newidx[grp[r]] = n_old + r;
}

__global__ void k_os_link_old(int nc, const unsigned long long *__restrict__ key, const int *__restrict__ ord, const int *__restrict__ head,
                              const int *__restrict__ gscan, const int *__restrict__ newidx, const int *__restrict__ cj,
                              const int *__restrict__ ck, int lda, int32_t *__restrict__ adj_out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nc || key[p] == kNoKey) return;
    const int g = head[p] ? gscan[p] : gscan[p] - 1;     // exclusive scan: non-heads belong to the group before
    const int o = ord[p];
    adj_out[(size_t)cj[o] * lda + ck[o]] = newidx[g] + 1;
}

// coordinates and link column of every new state
__global__ void k_os_new(int nu, int n_old, int lds, int lda, OsModel M, const int *__restrict__ grp,
                         const unsigned long long *__restrict__ ukey, const int *__restrict__ newidx,
                         const unsigned long long *__restrict__ okey, const int *__restrict__ oidx, int32_t *__restrict__ state_new,
                         int32_t *__restrict__ adj_out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nu) return;
    const unsigned long long key = ukey[grp[r]];
    int32_t x[kOsMaxS];
    for (int s = 0; s < M.ns; ++s) {
        x[s] = (int32_t)((key >> M.shift[s]) & ((1ULL << M.bits[s]) - 1ULL));
        state_new[(size_t)r * lds + s] = x[s];
    }
    int32_t *col = adj_out + (size_t)(n_old + r) * lda;
    for (int k = 0; k < M.nr; ++k) {
        unsigned long long yk;
        bool neg;
        int link = 0;
        if (os_target(M, x, k, &yk, &neg)) {
            int p = os_find(okey, n_old, yk);
            if (p >= 0) {
                link = oidx[p] + 1;
            } else {
                p = os_find(ukey, nu, yk);            // ukey is ascending: group number = position
                if (p >= 0) link = newidx[p] + 1;
            }
        } else if (neg) {
            link = -1;
        }
        col[k] = link;
    }
}

#define OS_TRY(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return 1000 + (int)e_;                                                         \
        }                                                                                  \
    } while (0)

// carve 256-byte aligned pieces out of one buffer
struct Carver {
    char *p;
    size_t used = 0;
    template <class T>
    T *take(size_t count)
    {
        T *r = reinterpret_cast<T *>(p + used);
        used += (count * sizeof(T) + 255) / 256 * 256;
        return r;
    }
};

inline int blocks(int64_t n) { return (int)std::max<int64_t>(1, (n + 255) / 256); }

}  // namespace

int onestep_device(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state, int32_t lds,
                   const int32_t *adj, int32_t lda, int32_t max_count, int32_t cap, int32_t *n_out, int32_t *state_new,
                   int32_t *adj_out, double *off_new, int32_t ldo, double *diag_new)
{
    hipStream_t st = ctx->stream;
    OsModel M;
    std::memset(&M, 0, sizeof(M));
    M.ns = ns;
    M.nr = nr;
    M.max_count = max_count;
    int maxnu[kOsMaxS] = {0};
    for (int k = 0; k < nr; ++k)
        for (int s = 0; s < ns; ++s) {
            const int v = stoich[(size_t)k * ns + s];
            if (v < -100 || v > 100) {
                ctx->err = "stoichiometry out of range";
                return -4;
            }
            M.nu[k][s] = (signed char)v;
            maxnu[s] = std::max(maxnu[s], v);
        }

    // phase 1 buffers: lists, keys, counts
    const size_t n1 = (size_t)n + 64;
    size_t need1 = (n1 * lds + n1 * lda) * 4 + 4 * n1 * 8 + 4 * n1 * 4 + 64 * 4 + 4096;
    OS_TRY(ctx->d_os1.reserve(need1, false));
    Carver c1{ctx->d_os1.p};
    int32_t *d_state = c1.take<int32_t>(n1 * lds);
    int32_t *d_adj = c1.take<int32_t>(n1 * lda);
    unsigned long long *d_key = c1.take<unsigned long long>(n1), *d_key2 = c1.take<unsigned long long>(n1);
    int *d_idx = c1.take<int>(n1), *d_idx2 = c1.take<int>(n1), *d_cnt = c1.take<int>(n1), *d_off = c1.take<int>(n1 + 1);
    int *d_mx = c1.take<int>(64);
    OS_TRY(hipMemcpyAsync(d_state, state, (size_t)n * lds * 4, hipMemcpyHostToDevice, st));
    OS_TRY(hipMemcpyAsync(d_adj, adj, (size_t)n * lda * 4, hipMemcpyHostToDevice, st));
    OS_TRY(hipMemsetAsync(d_mx, 0, 64 * 4, st));
    hipLaunchKernelGGL(k_os_max, dim3(std::min(blocks(n), 1024)), dim3(256), 0, st, n, ns, lds, d_state, d_mx);
    int mx[kOsMaxS];
    OS_TRY(hipMemcpyAsync(mx, d_mx, sizeof(int) * (size_t)ns, hipMemcpyDeviceToHost, st));
    OS_TRY(hipStreamSynchronize(st));
    int total_bits = 0;
    for (int s = 0; s < ns; ++s) {
        const long long top = (long long)mx[s] + maxnu[s];       // largest population a target can have
        int b = 1;
        while ((1LL << b) <= top) ++b;
        M.shift[s] = total_bits;
        M.bits[s] = b;
        total_bits += b;
    }
    if (total_bits > 63) {
        ctx->err = "state keys need more than 63 bits";
        return -9;
    }

    // listed states: packed, sorted by key
    hipLaunchKernelGGL(k_os_pack, dim3(blocks(n)), dim3(256), 0, st, n, lds, d_state, M, d_key, d_idx);
    size_t tmp_bytes = 0;
    OS_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_key2, d_idx, d_idx2, n, 0, total_bits, st));
    OS_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
    OS_TRY(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, d_key, d_key2, d_idx, d_idx2, n, 0, total_bits, st));
    const unsigned long long *okey = d_key2;
    const int *oidx = d_idx2;

    // open links with a legal target, per state, and their positions in (state, reaction) order
    hipLaunchKernelGGL(k_os_count, dim3(blocks(n)), dim3(256), 0, st, n, lds, lda, d_state, d_adj, M, d_cnt);
    OS_TRY(hipMemsetAsync(d_cnt + n, 0, 4, st));
    OS_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_off, n + 1, st));
    OS_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
    OS_TRY(hipcub::DeviceScan::ExclusiveSum(ctx->d_sorttmp.p, tmp_bytes, d_cnt, d_off, n + 1, st));
    int nc = 0;
    OS_TRY(hipMemcpyAsync(&nc, d_off + n, 4, hipMemcpyDeviceToHost, st));
    OS_TRY(hipStreamSynchronize(st));

    // phase 2 buffers: the link array being completed, candidates, unique targets
    // at most one new state per candidate: the arenas follow the work, not the caller's capacity (MAX_SIZE - 1 of the
    // Fortran host = 6.3e6 states whatever the FSP's size)
    const size_t ncp = (size_t)nc + 64, capn = (size_t)std::min<int64_t>(cap, (int64_t)n + nc) + 64;
    size_t need2 = capn * lda * 4 + capn * lds * 4 + 4 * ncp * 8 + 10 * ncp * 4 + 8192;
    OS_TRY(ctx->d_os2.reserve(need2, false));
    Carver c2{ctx->d_os2.p};
    int32_t *d_adj_out = c2.take<int32_t>(capn * lda);
    int32_t *d_state_new = c2.take<int32_t>(capn * lds);
    unsigned long long *d_ckey = c2.take<unsigned long long>(ncp), *d_ckey2 = c2.take<unsigned long long>(ncp);
    unsigned long long *d_ukey = c2.take<unsigned long long>(ncp);
    int *d_cord = c2.take<int>(ncp), *d_cord2 = c2.take<int>(ncp), *d_cj = c2.take<int>(ncp), *d_ck = c2.take<int>(ncp);
    int *d_head = c2.take<int>(ncp), *d_gscan = c2.take<int>(ncp + 1), *d_uord = c2.take<int>(ncp), *d_uord2 = c2.take<int>(ncp);
    int *d_ugid = c2.take<int>(ncp), *d_ugid2 = c2.take<int>(ncp), *d_newidx = c2.take<int>(ncp);
    OS_TRY(hipMemcpyAsync(d_adj_out, d_adj, (size_t)n * lda * 4, hipMemcpyDeviceToDevice, st));
    int nu = 0;
    // (also with no candidate at all: the -1 links of columns that arrive unlinked are written here)
    hipLaunchKernelGGL(k_os_fill, dim3(blocks(n)), dim3(256), 0, st, n, lds, lda, d_state, d_adj, M, d_off, okey, oidx, d_ckey, d_cord,
                       d_cj, d_ck, d_adj_out);
    if (nc > 0) {
        // equal targets side by side, in order of appearance (stable); already listed ones (kNoKey) last
        OS_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_ckey, d_ckey2, d_cord, d_cord2, nc, 0, 64, st));
        OS_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
        OS_TRY(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, d_ckey, d_ckey2, d_cord, d_cord2, nc, 0, 64, st));
        hipLaunchKernelGGL(k_os_heads, dim3(blocks(nc)), dim3(256), 0, st, nc, d_ckey2, d_head);
        OS_TRY(hipMemsetAsync(d_head + nc, 0, 4, st));
        OS_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_head, d_gscan, nc + 1, st));
        OS_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
        OS_TRY(hipcub::DeviceScan::ExclusiveSum(ctx->d_sorttmp.p, tmp_bytes, d_head, d_gscan, nc + 1, st));
        OS_TRY(hipMemcpyAsync(&nu, d_gscan + nc, 4, hipMemcpyDeviceToHost, st));
        OS_TRY(hipStreamSynchronize(st));
    }
    if ((int64_t)n + nu > cap) {
        ctx->err = "FSP SIZE EXCEEDS MEMORY LIMIT";
        return -11;
    }
    if (nu > 0) {
        hipLaunchKernelGGL(k_os_unique, dim3(blocks(nc)), dim3(256), 0, st, nc, d_ckey2, d_cord2, d_head, d_gscan, d_ukey, d_uord, d_ugid);
        // new states in the order their first candidate appears
        OS_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_uord, d_uord2, d_ugid, d_ugid2, nu, 0, 32, st));
        OS_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
        OS_TRY(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, d_uord, d_uord2, d_ugid, d_ugid2, nu, 0, 32, st));
        hipLaunchKernelGGL(k_os_rank, dim3(blocks(nu)), dim3(256), 0, st, nu, d_ugid2, n, d_newidx);
        hipLaunchKernelGGL(k_os_link_old, dim3(blocks(nc)), dim3(256), 0, st, nc, d_ckey2, d_cord2, d_head, d_gscan, d_newidx, d_cj,
                           d_ck, lda, d_adj_out);
        hipLaunchKernelGGL(k_os_new, dim3(blocks(nu)), dim3(256), 0, st, nu, n, lds, lda, M, d_ugid2, d_ukey, d_newidx, okey, oidx,
                           d_state_new, d_adj_out);
        OS_TRY(hipMemcpyAsync(state_new, d_state_new, (size_t)nu * lds * 4, hipMemcpyDeviceToHost, st));
        if (off_new) {
            // the propensity columns of the appended states, made where their coordinates already are (kfsp_prop.hip)
            const size_t ob = (size_t)nu * (size_t)ldo * 8;
            OS_TRY(ctx->d_os3.reserve(ob + (size_t)nu * 8 + 256, false));
            double *d_off = reinterpret_cast<double *>(ctx->d_os3.p), *d_dg = reinterpret_cast<double *>(ctx->d_os3.p + ob);
            if (int rc = prop_eval_device(ctx, nu, d_state_new, lds, d_off, ldo, d_dg)) return rc;
            OS_TRY(hipMemcpyAsync(off_new, d_off, ob, hipMemcpyDeviceToHost, st));
            OS_TRY(hipMemcpyAsync(diag_new, d_dg, (size_t)nu * 8, hipMemcpyDeviceToHost, st));
        }
    }
    OS_TRY(hipMemcpyAsync(adj_out, d_adj_out, (size_t)(n + nu) * lda * 4, hipMemcpyDeviceToHost, st));
    OS_TRY(hipStreamSynchronize(st));
    *n_out = n + nu;
    return 0;
}

}  // namespace kfsp
