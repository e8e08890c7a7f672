// DROP_STATES on the device: the decision of
// StateSpace.f90:398-427 (FIND_DROPTOL) and :470-497 (marking, derivative guard, the
// counter with its quirk) and the compaction of the probability vector (:500-546 for W),
// on the resident w and A*w.  The state lists stay with the host, which receives one
// flag byte per state only when a compaction is due.
//
// All f64 streaming, HBM-bound: the threshold sums read w once (8 B/state) whatever the
// number of thresholds, the flag pass reads w and A*w (16 B/state) and writes 1 B/state.
#include "kfsp_ctx.h"

#include <hipcub/hipcub.hpp>

namespace kfsp {

namespace {

// sums over the block, result in thread 0..: red needs 4 doubles per value
__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// S_k = sum of the entries with 0 < w < tol[k], k = 0..kDropLevels-1, all thresholds in one
// pass.  tol is decreasing, so an entry below tol[k] is below tol[0..k]: it is added to every
// level it qualifies for - the same entries FIND_DROPTOL adds up sweep by sweep.  One partial per
// (level, block); fixed grid and fixed reduction order: reproducible to the bit run to run.
__global__ __launch_bounds__(kBlock) void k_drop_sums(int64_t npairs, const double2 *__restrict__ w, DropLevels L,
                                                      double *__restrict__ partial)
{
    __shared__ double red[kDropLevels][4];
    double acc[kDropLevels];
#pragma unroll
    for (int k = 0; k < kDropLevels; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock) {
        const double2 v = w[i];
        if (v.x > 0.0 && v.x < L.tol[0]) {
#pragma unroll
            for (int k = 0; k < kDropLevels; ++k)
                if (v.x < L.tol[k]) acc[k] += v.x;
        }
        if (v.y > 0.0 && v.y < L.tol[0]) {
#pragma unroll
            for (int k = 0; k < kDropLevels; ++k)
                if (v.y < L.tol[k]) acc[k] += v.y;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kDropLevels; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < kDropLevels) {
        const int k = threadIdx.x;
        partial[(size_t)k * gridDim.x + blockIdx.x] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    }
}

// one block: out[k] = sum over blocks of partial[k][*], every level in the same fixed order
__global__ __launch_bounds__(kBlock) void k_drop_finish(int nblocks, const double *__restrict__ partial, double *__restrict__ out)
{
    __shared__ double red[4];
    for (int k = 0; k < kDropLevels; ++k) {
        double a = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += kBlock) a += partial[(size_t)k * nblocks + i];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) out[k] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

// mark w < droptol, un-mark where (A w)_i > 1e-8 (StateSpace.f90:475-495); counters:
// cnt[0] = #(w < droptol), cnt[1] = #(A w > 1e-8) - DROP_COUNT is their DIFFERENCE in the
// reference, whether or not the guarded state was marked -, cnt[2] = states actually flagged.
// Flags are stored in the CALLER's state order (iperm: caller index -> internal, or null): the sweep runs over the caller's
// indices and GATHERS w and A w (scattered single-byte stores through the inverse map cost 140 us at 1e6 states).
__global__ __launch_bounds__(kBlock) void k_drop_flags(int64_t n, const double *__restrict__ w, const double *__restrict__ aw,
                                                       double droptol, const int32_t *__restrict__ iperm,
                                                       uint8_t *__restrict__ flag, unsigned long long *__restrict__ cnt)
{
    unsigned c0 = 0, c1 = 0, c2 = 0;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
        const int64_t i = iperm ? iperm[j] : j;
        const bool marked = w[i] < droptol;
        const bool guarded = aw[i] > 1.0e-8;
        const bool drop = marked && !guarded;
        flag[j] = drop ? 1 : 0;
        c0 += marked;
        c1 += guarded;
        c2 += drop;
    }
    // integer counts: any order gives the same result
    for (int o = 32; o > 0; o >>= 1) {
        c0 += __shfl_xor(c0, o, 64);
        c1 += __shfl_xor(c1, o, 64);
        c2 += __shfl_xor(c2, o, 64);
    }
    // one atomic per workgroup and counter (per wavefront they were 12 000 adds on three addresses: 120 us of a 140 us kernel)
    __shared__ unsigned sc[3][kBlock / 64];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sc[0][wv] = c0;
        sc[1][wv] = c1;
        sc[2][wv] = c2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
        for (int v = 0; v < kBlock / 64; ++v) t += sc[threadIdx.x][v];
        atomicAdd(cnt + threadIdx.x, t);
    }
}

__global__ __launch_bounds__(kBlock) void k_keep_from_drop(int64_t n, const uint8_t *__restrict__ drop, uint8_t *__restrict__ keep)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) keep[i] = drop[i] ? 0 : 1;
}

// flags of all blocks in the device's order (entry g at all[g]: blocks are contiguous) -> the caller's order
__global__ __launch_bounds__(kBlock) void k_flags_to_caller(int64_t n, const uint8_t *__restrict__ all, const int32_t *__restrict__ iperm,
                                                            uint8_t *__restrict__ flag)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) flag[j] = all[iperm ? iperm[j] : j];
}

}  // namespace

void launch_flags_to_caller(int64_t n, const uint8_t *all, const int32_t *iperm, uint8_t *flag, hipStream_t st)
{
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_flags_to_caller, dim3(grid), dim3(kBlock), 0, st, n, all, iperm, flag);
}

void launch_drop_sums(int grid, int64_t npairs, const double *w, const DropLevels &L, double *partial, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_drop_sums, dim3(grid), dim3(kBlock), 0, st, npairs, reinterpret_cast<const double2 *>(w), L, partial);
    hipLaunchKernelGGL(k_drop_finish, dim3(1), dim3(kBlock), 0, st, grid, partial, out);
}

void launch_drop_flags(int64_t n, const double *w, const double *aw, double droptol, const int32_t *iperm, uint8_t *flag,
                       unsigned long long *cnt, hipStream_t st)
{
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(512, (n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_drop_flags, dim3(grid), dim3(kBlock), 0, st, n, w, aw, droptol, iperm, flag, cnt);
}

// dst[0..n_keep) = the entries of src (caller order) whose flag is 0, order kept; *n_keep_dev receives the count
int drop_compact_vector(kfsp_ctx *ctx, int64_t n, const double *src, double *dst, int *n_keep_dev)
{
    hipStream_t st = ctx->stream;
    uint8_t *keep = ctx->d_dropflag.p + ctx->d_dropflag.cap / 2;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_keep_from_drop, dim3(grid), dim3(kBlock), 0, st, n, ctx->d_dropflag.p, keep);
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceSelect::Flagged(nullptr, bytes, src, keep, dst, n_keep_dev, (int)n, st);
    if (e != hipSuccess) return 1000 + (int)e;
    e = ctx->d_sorttmp.reserve(bytes + 256, false);
    if (e != hipSuccess) return 1000 + (int)e;
    e = hipcub::DeviceSelect::Flagged(ctx->d_sorttmp.p, bytes, src, keep, dst, n_keep_dev, (int)n, st);
    if (e != hipSuccess) return 1000 + (int)e;
    return 0;
}

}  // namespace kfsp
