// State-space expansion on lists that live on the device (the part that keeps the host out
// of the loop): the one-step reachability sweep without a sort, and the whole expansion step of the solver
// (KrylovSolver.f90:518-534: SSA_EXTENDER, then ONESTEP_EXTENDER) on the resident copy of the reference arrays.
//
// ONESTEP_EXTENDER (src/state_space/StateSpace.f90:136-246).  For every listed state j = 1..n, in this order, and every
// reaction k whose link ADJ(k, j) is open (0): the target y = x_j + nu_k
//   * has a negative population            -> ADJ(k, j) = -1
//   * exceeds MAXNUMBERMOLECULES           -> stays 0
//   * is listed                            -> ADJ(k, j) = its number
//   * is not                               -> y is appended (once: a later candidate with the same target finds it) and linked
// and afterwards the appended states get their own columns (successors that are listed by then, -1, or 0).  So the new
// states are the DISTINCT unlisted targets in the order in which the candidates (j, k) first name them.
//
// Round 2 found that order with two 64-bit radix sorts (states by packed key, candidates by key).  Here nothing is sorted
// and no key is packed (so no limit on the populations' bit widths either):
//   1. a table of the listed states (kfsp_hash_dev.h)
//   2. mark: every open link is resolved through it or marked as a candidate                       [one sweep over ADJ]
//   3. a second table of the candidates' targets; a slot remembers the SMALLEST candidate ordinal j nr + k that named
//      its target (atomic min: the survivor does not depend on the order of the insertions)          [one sweep]
//   4. a candidate whose ordinal is its slot's is the first to name its target: heads per state, an exclusive scan over
//      the states gives every head its number in order of first appearance                           [one sweep + scan]
//   5. heads write the new states; every candidate is linked to its slot's state; new states look their successors up
//      in both tables                                                                                  [two sweeps]
// The sweeps read ADJ (4 nr bytes per state) and gather a few table slots for the boundary states only.
#include "kfsp_prop_dev.h"
#include "kfsp_hash_dev.h"

#include <hipcub/hipcub.hpp>

#include <climits>
#include <cstring>

namespace kfsp {

namespace {

constexpr int kXMaxS = 16, kXMaxR = 64;
constexpr int kOpen = -2;                   // ADJ marker between the sweeps: open link, legal target, not listed
constexpr int kFree = 0x7f7f7f7f;           // empty slot of the candidates' table (what memset 0x7f leaves)

struct XDev {
    int ns, nr, n, lds, lda, max_count;
    const int32_t *state;                   // [n][lds]            the listed states
    int32_t *adj;                           // [n][lda]            completed in place
    int32_t *state_new;                     // [nu][lds]           the appended states (numbers n + 1 ..)
    int32_t *adj_new;                       // [nu][lda]
    const unsigned long long *tab;          // listed states, tagged table (kfsp_hash_dev.h): tag << 32 | index + 1, 0 = empty
    unsigned tmask;
    int32_t *tab2;                          // candidates' targets: smallest ordinal naming the target, kFree = empty
    int32_t *newidx;                        // per slot of tab2: number (1-based) of the appended state
    unsigned tmask2;
    signed char nu[kXMaxR][kXMaxS];
};

// y = x + nu_k; true if y is a state the FSP may hold
__device__ __forceinline__ bool x_target(const XDev &A, const int32_t *x, int k, int32_t *y, bool *neg)
{
    bool ng = false, ok = true;
    for (int s = 0; s < A.ns; ++s) {
        y[s] = x[s] + A.nu[k][s];
        ng = ng || y[s] < 0;
        ok = ok && y[s] <= A.max_count;
    }
    *neg = ng;
    return ok && !ng;
}

// is y the target of candidate c = j nr + k ?
__device__ __forceinline__ bool x_names(const XDev &A, int c, const int32_t *y)
{
    const int j = c / A.nr, k = c - j * A.nr;
    const int32_t *z = A.state + (int64_t)j * A.lds;
    bool same = true;
    for (int s = 0; s < A.ns; ++s) same = same && z[s] + A.nu[k][s] == y[s];
    return same;
}

// slot of y in the candidates' table, -1 if no candidate names it
__device__ __forceinline__ int x_find2(const XDev &A, const int32_t *y)
{
    unsigned slot = hash_state(y, A.ns) & A.tmask2;
    for (;;) {
        const int c = A.tab2[slot];
        if (c == kFree) return -1;
        if (x_names(A, c, y)) return (int)slot;
        slot = (slot + 1) & A.tmask2;
    }
}

__device__ __forceinline__ void x_insert2(const XDev &A, const int32_t *y, int c)
{
    unsigned slot = hash_state(y, A.ns) & A.tmask2;
    for (;;) {
        int cur = A.tab2[slot];
        if (cur == kFree) {
            cur = atomicCAS(&A.tab2[slot], kFree, c);
            if (cur == kFree) return;
        }
        // (whatever ordinal the slot holds now or later, it names the same target: atomicMin only swaps such ordinals)
        if (x_names(A, cur, y)) {
            atomicMin(&A.tab2[slot], c);
            return;
        }
        slot = (slot + 1) & A.tmask2;
    }
}

// The sweeps over the link array run one lane per ENTRY (j, k): consecutive lanes read consecutive words of ADJ (a lane per
// state walked its column with a stride of lda words: 120 us per sweep at 8e5 states instead of 25), and only the lanes
// that find an open link do anything more.
__device__ __forceinline__ bool x_entry(const XDev &A, int64_t e, int *j, int *k)
{
    if (e >= (int64_t)A.n * A.lda) return false;
    *j = (int)(e / A.lda);
    *k = (int)(e - (int64_t)*j * A.lda);
    return *k < A.nr;
}

// step 2: resolve or mark the open links; *ncand += the number of marks
__global__ __launch_bounds__(kBlock) void k_x_mark(XDev A, unsigned long long *__restrict__ ncand)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int j, k, c = 0;
    if (x_entry(A, e, &j, &k) && A.adj[e] == 0) {
        const int32_t *x = A.state + (int64_t)j * A.lds;
        int32_t y[kXMaxS];
        bool neg;
        if (!x_target(A, x, k, y, &neg)) {
            if (neg) A.adj[e] = -1;
        } else {
            const int f = table_find64(A.tab, A.tmask, A.state, A.lds, A.ns, y);
            if (f > 0) {
                A.adj[e] = f;
            } else {
                A.adj[e] = kOpen;
                c = 1;
            }
        }
    }
    // (one atomic per workgroup: adds on ONE address cost ~10 ns each)
    __shared__ int sc[kBlock / 64];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int v = 0; v < kBlock / 64; ++v) t += sc[v];
        if (t > 0) atomicAdd(ncand, (unsigned long long)t);
    }
}

// (a sweep that fails after the marks - capacity - takes them back, and with them the links to states beyond n_keep)
__global__ __launch_bounds__(kBlock) void k_x_unmark(XDev A, int n_keep)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < (int64_t)n_keep * A.lda && (A.adj[i] == kOpen || A.adj[i] > n_keep)) A.adj[i] = 0;
}

// step 3
__global__ __launch_bounds__(kBlock) void k_x_insert(XDev A)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int j, k;
    if (!x_entry(A, e, &j, &k) || A.adj[e] != kOpen) return;
    int32_t y[kXMaxS];
    bool neg;
    x_target(A, A.state + (int64_t)j * A.lds, k, y, &neg);
    x_insert2(A, y, j * A.nr + k);
}

// step 4: heads per state (cnt is zero on entry)
__global__ __launch_bounds__(kBlock) void k_x_heads(XDev A, int *__restrict__ cnt)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int j, k;
    if (!x_entry(A, e, &j, &k) || A.adj[e] != kOpen) return;
    int32_t y[kXMaxS];
    bool neg;
    x_target(A, A.state + (int64_t)j * A.lds, k, y, &neg);
    if (A.tab2[x_find2(A, y)] == j * A.nr + k) atomicAdd(&cnt[j], 1);
}

// step 5a: the heads append their targets; a head's place among the heads of its state is the number of heads before
// it in the state's column (few lanes get here, and a column holds nr entries)
__global__ __launch_bounds__(kBlock) void k_x_new(XDev A, const int *__restrict__ cnt, const int *__restrict__ off)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int j, k;
    if (!x_entry(A, e, &j, &k) || A.adj[e] != kOpen) return;
    const int32_t *x = A.state + (int64_t)j * A.lds;
    int32_t y[kXMaxS];
    bool neg;
    x_target(A, x, k, y, &neg);
    const int slot = x_find2(A, y);
    if (A.tab2[slot] != j * A.nr + k) return;
    int r = off[j];
    if (cnt[j] > 1) {
        const int32_t *a = A.adj + (int64_t)j * A.lda;
        for (int kk = 0; kk < k; ++kk) {
            if (a[kk] != kOpen) continue;
            int32_t yy[kXMaxS];
            x_target(A, x, kk, yy, &neg);
            r += A.tab2[x_find2(A, yy)] == j * A.nr + kk;
        }
    }
    int32_t *z = A.state_new + (int64_t)r * A.lds;
    for (int s = 0; s < A.lds; ++s) z[s] = s < A.ns ? y[s] : 0;
    A.newidx[slot] = A.n + r + 1;
}

// step 5b: every candidate is linked to the state its target became
__global__ __launch_bounds__(kBlock) void k_x_link(XDev A)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int j, k;
    if (!x_entry(A, e, &j, &k) || A.adj[e] != kOpen) return;
    int32_t y[kXMaxS];
    bool neg;
    x_target(A, A.state + (int64_t)j * A.lds, k, y, &neg);
    A.adj[e] = A.newidx[x_find2(A, y)];
}

// step 5c: the columns of the appended states (LINK_NEW without back links, StateSpace.f90:213-230)
__global__ __launch_bounds__(kBlock) void k_x_newcols(XDev A, int nu)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // one lane per entry: every one is two table look-ups
    if (e >= (int64_t)nu * A.lda) return;
    const int r = (int)(e / A.lda), k = (int)(e - (int64_t)r * A.lda);
    int link = 0;
    if (k < A.nr) {
        int32_t y[kXMaxS];
        bool neg;
        if (x_target(A, A.state_new + (int64_t)r * A.lds, k, y, &neg)) {
            link = table_find64(A.tab, A.tmask, A.state, A.lds, A.ns, y);
            if (link == 0) {
                const int slot = x_find2(A, y);
                if (slot >= 0) link = A.newidx[slot];
            }
        } else if (neg) {
            link = -1;
        }
    }
    A.adj_new[e] = link;
}

__global__ __launch_bounds__(kBlock) void k_x_zero_pad(int64_t n0, int64_t n1, double *__restrict__ w)
{
    const int64_t i = n0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n1) w[i] = 0.0;
}

struct Arena {
    char *p;
    template <class T>
    T *take(size_t n)
    {
        T *r = reinterpret_cast<T *>(p);
        p += (n * sizeof(T) + 255) / 256 * 256;
        return r;
    }
};

inline int blocks(int64_t n) { return (int)std::max<int64_t>(1, (n + kBlock - 1) / kBlock); }

#define X_TRY(expr)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return 1000 + (int)e_;                                                         \
        }                                                                                  \
    } while (0)

int fill_model(kfsp_ctx *ctx, XDev &A, int ns, int nr, const int32_t *stoich, int n, int lds, int lda, int max_count)
{
    std::memset(&A, 0, sizeof(A));
    A.ns = ns;
    A.nr = nr;
    A.n = n;
    A.lds = lds;
    A.lda = lda;
    A.max_count = max_count;
    for (int k = 0; k < nr; ++k)
        for (int s = 0; s < ns; ++s) {
            const int v = stoich[(size_t)k * ns + s];
            if (v < -100 || v > 100) {
                ctx->err = "stoichiometry out of range";
                return -4;
            }
            A.nu[k][s] = (signed char)v;
        }
    if ((int64_t)n * nr >= (int64_t)kFree) {               // (ordinals j nr + k stay below the empty-slot value)
        ctx->err = "more than 2^31 (state, reaction) pairs";
        return -9;
    }
    return 0;
}

// steps 1-4 on A.state / A.adj (n states): tables in d_os1 / d_os2, *nu_out = the number of states step 5 will append
int sweep_count(kfsp_ctx *ctx, XDev &A, int **cnt_out, int **off_out, int *nu_out)
{
    hipStream_t st = ctx->stream;
    const int n = A.n;
    unsigned slots = 64;
    while (slots < 2u * (unsigned)n) slots <<= 1;
    X_TRY(ctx->d_os1.reserve((size_t)slots * 8 + 2 * ((size_t)n + 1) * 4 + 4096, false));
    Arena a1{ctx->d_os1.p};
    unsigned long long *d_tab = a1.take<unsigned long long>(slots);
    int *d_cnt = a1.take<int>((size_t)n + 1), *d_off = a1.take<int>((size_t)n + 1);
    unsigned long long *d_ncand = a1.take<unsigned long long>(2);
    X_TRY(hipMemsetAsync(d_tab, 0, (size_t)slots * 8, st));
    X_TRY(hipMemsetAsync(d_ncand, 0, 16, st));
    launch_table_build64(n, A.ns, A.lds, A.state, d_tab, slots - 1, nullptr, 0, st);
    A.tab = d_tab;
    A.tmask = slots - 1;
    const int64_t nent = (int64_t)n * A.lda;
    hipLaunchKernelGGL(k_x_mark, dim3(blocks(nent)), dim3(kBlock), 0, st, A, d_ncand);
    unsigned long long nc = 0;
    X_TRY(hipMemcpyAsync(&nc, d_ncand, sizeof(nc), hipMemcpyDeviceToHost, st));
    X_TRY(hipStreamSynchronize(st));
    *nu_out = 0;
    *cnt_out = d_cnt;
    *off_out = d_off;
    if (nc == 0) return 0;
    unsigned slots2 = 64;
    while (slots2 < 2ull * nc) slots2 <<= 1;
    X_TRY(ctx->d_os2.reserve((size_t)slots2 * 8 + 1024, false));
    Arena a2{ctx->d_os2.p};
    A.tab2 = a2.take<int32_t>(slots2);
    A.newidx = a2.take<int32_t>(slots2);
    A.tmask2 = slots2 - 1;
    X_TRY(hipMemsetAsync(A.tab2, 0x7f, (size_t)slots2 * 4, st));
    X_TRY(hipMemsetAsync(d_cnt, 0, ((size_t)n + 1) * sizeof(int), st));      // (cnt[n] = 0: the scan's last element is the total)
    hipLaunchKernelGGL(k_x_insert, dim3(blocks(nent)), dim3(kBlock), 0, st, A);
    hipLaunchKernelGGL(k_x_heads, dim3(blocks(nent)), dim3(kBlock), 0, st, A, d_cnt);
    size_t tmp_bytes = 0;
    X_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_off, n + 1, st));
    X_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
    X_TRY(hipcub::DeviceScan::ExclusiveSum(ctx->d_sorttmp.p, tmp_bytes, d_cnt, d_off, n + 1, st));
    int nu = 0;
    X_TRY(hipMemcpyAsync(&nu, d_off + n, sizeof(int), hipMemcpyDeviceToHost, st));
    X_TRY(hipStreamSynchronize(st));
    *nu_out = nu;
    return 0;
}

// step 5 (A.state_new / A.adj_new have room for nu states), in two halves: the appended states' coordinates first - nothing
// of the listed states changes, so their propensity columns can be made and checked before the sweep commits -, then the links
void sweep_new_states(kfsp_ctx *ctx, const XDev &A, const int *cnt, const int *off, int nu)
{
    if (nu <= 0) return;
    hipLaunchKernelGGL(k_x_new, dim3(blocks((int64_t)A.n * A.lda)), dim3(kBlock), 0, ctx->stream, A, cnt, off);
}
void sweep_link(kfsp_ctx *ctx, const XDev &A, int nu)
{
    if (nu <= 0) return;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_x_link, dim3(blocks((int64_t)A.n * A.lda)), dim3(kBlock), 0, st, A);
    hipLaunchKernelGGL(k_x_newcols, dim3(blocks((int64_t)nu * A.lda)), dim3(kBlock), 0, st, A, nu);
}

}  // namespace

// ONESTEP_EXTENDER's integer work for lists in host memory (kfsp_onestep / kfsp_onestep_columns)
int onestep_device(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state, int32_t lds,
                   const int32_t *adj, int32_t lda, int32_t max_count, int32_t cap, int32_t *n_out, int32_t *state_new,
                   int32_t *adj_out, double *off_new, int32_t ldo, double *diag_new)
{
    hipStream_t st = ctx->stream;
    XDev A;
    if (int rc = fill_model(ctx, A, ns, nr, stoich, n, lds, lda, max_count)) return rc;
    X_TRY(ctx->d_os3.reserve(((size_t)n * lds + (size_t)n * lda) * 4 + 1024, false));
    Arena a3{ctx->d_os3.p};
    int32_t *d_state = a3.take<int32_t>((size_t)n * lds), *d_adj = a3.take<int32_t>((size_t)n * lda);
    X_TRY(hipMemcpyAsync(d_state, state, (size_t)n * lds * 4, hipMemcpyHostToDevice, st));
    X_TRY(hipMemcpyAsync(d_adj, adj, (size_t)n * lda * 4, hipMemcpyHostToDevice, st));
    A.state = d_state;
    A.adj = d_adj;
    int *d_cnt = nullptr, *d_off = nullptr, nu = 0;
    if (int rc = sweep_count(ctx, A, &d_cnt, &d_off, &nu)) return rc;
    if ((int64_t)n + nu > cap) {
        ctx->err = "FSP SIZE EXCEEDS MEMORY LIMIT";
        return -11;
    }
    if (nu > 0) {
        const size_t ob = off_new ? (size_t)nu * (size_t)ldo * 8 : 0;
        X_TRY(ctx->d_os4.reserve((size_t)nu * (lds + lda) * 4 + ob + (size_t)nu * 8 + 2048, false));
        Arena a4{ctx->d_os4.p};
        A.state_new = a4.take<int32_t>((size_t)nu * lds);
        A.adj_new = a4.take<int32_t>((size_t)nu * lda);
        sweep_new_states(ctx, A, d_cnt, d_off, nu);
        double *d_on = nullptr, *d_dn = nullptr;
        if (off_new) {
            // the propensity columns of the appended states, made where their coordinates already are (kfsp_prop.hip) - and
            // BEFORE anything is written to the caller's arrays: a state beyond a two-species table ends the call with -16
            d_on = a4.take<double>((size_t)nu * ldo);
            d_dn = a4.take<double>((size_t)nu);
            if (int rc = prop_eval_device(ctx, nu, A.state_new, lds, d_on, ldo, d_dn)) return rc;
            if (int rc = prop_check_overflow(ctx)) return rc;
        }
        sweep_link(ctx, A, nu);
        X_TRY(hipMemcpyAsync(state_new, A.state_new, (size_t)nu * lds * 4, hipMemcpyDeviceToHost, st));
        X_TRY(hipMemcpyAsync(adj_out + (size_t)n * lda, A.adj_new, (size_t)nu * lda * 4, hipMemcpyDeviceToHost, st));
        if (off_new) {
            X_TRY(hipMemcpyAsync(off_new, d_on, ob, hipMemcpyDeviceToHost, st));
            X_TRY(hipMemcpyAsync(diag_new, d_dn, (size_t)nu * 8, hipMemcpyDeviceToHost, st));
        }
    }
    X_TRY(hipMemcpyAsync(adj_out, d_adj, (size_t)n * lda * 4, hipMemcpyDeviceToHost, st));
    X_TRY(hipStreamSynchronize(st));
    *n_out = n + nu;
    return 0;
}

// The solver's expansion step on the RESIDENT lists (d_coords, d_ell_adj / d_ell_off / d_ell_diag; n = ell_cols = coords_n
// states): the independent-stream SSA walk, its states appended unlinked, the one-step sweep over all of them, the
// propensity columns of everything appended.  On return the lists hold *n_out states; the caller (kfsp_expand_resident)
// rebuilds the gather form.  Nothing crosses the bus but a few counters.
int expand_resident_lists(kfsp_ctx *ctx, double tstep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich,
                          int32_t max_count, int32_t cap, int64_t *n_out, int64_t *n_ssa)
{
    hipStream_t st = ctx->stream;
    const int32_t n = (int32_t)ctx->ell_cols;
    const int lds = ctx->coords_ld, lda = ctx->ell_ld;
    // -- SSA_EXTENDER (independent streams)
    int32_t n1 = n;
    *n_ssa = 0;
    if (tstep > 0.0) {
        int32_t nnew = 0, *d_sn = nullptr;
        double *d_on = nullptr, *d_dn = nullptr;
        if (int rc = ssa_streams_core(ctx, tstep, seedmix, ns, nr, stoich, n, ctx->d_coords.p, lds, ctx->d_ell_adj.p, ctx->d_ell_off.p,
                                      lda, ctx->d_ell_diag.p, max_count, cap - n, lda, &nnew, &d_sn, &d_on, &d_dn, true))
            return rc;
        if (nnew > 0) {
            n1 = n + nnew;
            X_TRY(ctx->d_coords.reserve_keep((size_t)n1 * lds + 64, (size_t)n * lds, st));
            X_TRY(ctx->d_ell_adj.reserve_keep((size_t)n1 * lda, (size_t)n * lda, st));
            X_TRY(ctx->d_ell_off.reserve_keep((size_t)n1 * lda, (size_t)n * lda, st));
            X_TRY(ctx->d_ell_diag.reserve_keep((size_t)n1, (size_t)n, st));
            X_TRY(hipMemcpyAsync(ctx->d_coords.p + (size_t)n * lds, d_sn, (size_t)nnew * lds * 4, hipMemcpyDeviceToDevice, st));
            X_TRY(hipMemcpyAsync(ctx->d_ell_off.p + (size_t)n * lda, d_on, (size_t)nnew * lda * 8, hipMemcpyDeviceToDevice, st));
            X_TRY(hipMemcpyAsync(ctx->d_ell_diag.p + n, d_dn, (size_t)nnew * 8, hipMemcpyDeviceToDevice, st));
            // (their links are left to the sweep: it completes every column that arrives as zeros)
            X_TRY(hipMemsetAsync(ctx->d_ell_adj.p + (size_t)n * lda, 0, (size_t)nnew * lda * 4, st));
            *n_ssa = nnew;
        }
    }
    // -- ONESTEP_EXTENDER
    XDev A;
    if (int rc = fill_model(ctx, A, ns, nr, stoich, n1, lds, lda, max_count)) return rc;
    A.state = ctx->d_coords.p;
    A.adj = ctx->d_ell_adj.p;
    int *d_cnt = nullptr, *d_off = nullptr, nu = 0;
    if (int rc = sweep_count(ctx, A, &d_cnt, &d_off, &nu)) return rc;
    if ((int64_t)n1 + nu > cap) {
        hipLaunchKernelGGL(k_x_unmark, dim3(blocks((int64_t)n * lda)), dim3(kBlock), 0, st, A, (int)n);
        X_TRY(hipStreamSynchronize(st));
        ctx->err = "FSP SIZE EXCEEDS MEMORY LIMIT";
        return -11;
    }
    if (nu > 0) {
        const int32_t n2 = n1 + nu;
        X_TRY(ctx->d_coords.reserve_keep((size_t)n2 * lds + 64, (size_t)n1 * lds, st));
        X_TRY(ctx->d_ell_adj.reserve_keep((size_t)n2 * lda, (size_t)n1 * lda, st));
        X_TRY(ctx->d_ell_off.reserve_keep((size_t)n2 * lda, (size_t)n1 * lda, st));
        X_TRY(ctx->d_ell_diag.reserve_keep((size_t)n2, (size_t)n1, st));
        A.state = ctx->d_coords.p;
        A.adj = ctx->d_ell_adj.p;
        A.state_new = ctx->d_coords.p + (size_t)n1 * lds;
        A.adj_new = ctx->d_ell_adj.p + (size_t)n1 * lda;
        sweep_new_states(ctx, A, d_cnt, d_off, nu);
        if (int rc = prop_eval_device(ctx, nu, A.state_new, lds, ctx->d_ell_off.p + (size_t)n1 * lda, lda, ctx->d_ell_diag.p + n1)) return rc;
        if (int rc = prop_check_overflow(ctx)) {
            // an appended state lies beyond a two-species table: the sweep is taken back (as for -11) - the first n columns are
            // what they were, the caller enlarges the table and repeats the whole step
            hipLaunchKernelGGL(k_x_unmark, dim3(blocks((int64_t)n * lda)), dim3(kBlock), 0, st, A, (int)n);
            X_TRY(hipStreamSynchronize(st));
            return rc;
        }
        sweep_link(ctx, A, nu);
    }
    // (kept although the caller's rebuild follows on the same stream: without it the run is 12 ms of 2 550 shorter, and the
    // trace's DEVICE_ONESTEP / UPLOAD lines no longer say where the device's time went)
    X_TRY(hipStreamSynchronize(st));
    *n_out = (int64_t)n1 + nu;
    return 0;
}

// w[n0 .. n1) = 0 (the appended states start with probability 0, KrylovSolver.f90:530-533)
void launch_zero_pad(int64_t n0, int64_t n1, double *w, hipStream_t st)
{
    if (n1 > n0) hipLaunchKernelGGL(k_x_zero_pad, dim3(blocks(n1 - n0)), dim3(kBlock), 0, st, n0, n1, w);
}

}  // namespace kfsp
