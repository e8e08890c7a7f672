// gfx950 (MI355X, CDNA4) kernels of the exp(tA)v hot path.
//
// Everything here is bandwidth-bound f64 streaming (<= 0.17 flop/byte), so the
// design rules are: every HBM byte read once and in full 64-lane-wide
// contiguous pieces, the per-row scalar work fused into the pass that already
// holds the row, reductions finished by wavefront shuffles + one LDS hop, and
// no atomics (block partials in a fixed layout, summed by the consumer in a
// fixed order -> bit-reproducible runs).  No MFMA: there is no contraction.
//
// Reference call sites covered (voduchuy/KrylovFspSsa, src/fsp/KrylovSolver.f90):
//   k_spmv          FMATVEC :577-607 (as a row gather) fused with the first
//                   DDOT :243, or with the DNRM2 of :263, and with the DSCAL
//                   :258 of the source column (lazy normalisation)
//   k_ortho         DAXPY :244 fused with the next DDOT :243 / DNRM2 :247
//   k_ortho2        both DDOT/DAXPY pairs of an IOP(2) column + DNRM2 in one pass
//   k_combine       DGEMV :444 + clamp :447-449 + DASUM :450
//   k_copy_nrm2     DCOPY :176 / v1 = w/beta :223-226 + DNRM2 :177,:540
#include "kfsp_internal.h"

namespace kfsp {

// ---------------------------------------------------------------- reductions

// v of a lane CTRL away (DPP: a register-file crossbar move, no LDS traffic); lanes the
// masks exclude and lanes without a source get 0
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_get(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_allreduce_sum(double v)
{
    // Sum of the 64 lanes, the same bits in every lane.  Row-shift / row-broadcast DPP
    // steps (the ds_bpermute of a shuffle butterfly goes through the LDS pipeline and is
    // ~5x slower, which the one-launch Arnoldi kernel pays four times per column):
    // prefix sums inside each row of 16 lanes, then lane 15 of every row is folded into
    // the next rows; lane 63 ends with the total, which is read back as a scalar.
    const double v1 = v + dpp_get<0x111, 0xf, 0xf>(v);          // row_shr:1
    const double v2 = v1 + dpp_get<0x112, 0xf, 0xf>(v);         // row_shr:2
    double w = v2 + dpp_get<0x113, 0xf, 0xf>(v);                // row_shr:3 -> sums of 4
    w += dpp_get<0x114, 0xf, 0xe>(w);                           // row_shr:4, banks 1-3
    w += dpp_get<0x118, 0xf, 0xc>(w);                           // row_shr:8, banks 2-3 -> lane 15 = row total
    w += dpp_get<0x142, 0xa, 0xf>(w);                           // row_bcast:15 into rows 1 and 3
    w += dpp_get<0x143, 0xc, 0xf>(w);                           // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(w), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(w), 63);
    return __hiloint2double(hi, lo);
}

// Sum over the block, result in every thread.  red: 4 doubles of LDS.
__device__ __forceinline__ double block_allreduce_sum(double v, double *red)
{
    v = wave_allreduce_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

// Three sums with one barrier pair.  red: 12 doubles of LDS.
__device__ __forceinline__ void block_allreduce_sum3(double &a, double &b, double &c, double *red)
{
    a = wave_allreduce_sum(a);
    b = wave_allreduce_sum(b);
    c = wave_allreduce_sum(c);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave] = a;
        red[4 + wave] = b;
        red[8 + wave] = c;
    }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
    c = (red[8] + red[9]) + (red[10] + red[11]);
    __syncthreads();
}

// Finish a Pending scalar: every thread of every block performs the same
// additions in the same order.
__device__ __forceinline__ double finish_sum(Pending s, double *red)
{
    double a = 0.0;
    for (int i = threadIdx.x; i < s.n; i += kBlock) a += s.p[i];
    return block_allreduce_sum(a, red);
}

// Three Pending scalars at once (loads of all three in flight together).
__device__ __forceinline__ void finish_sum3(Pending p, Pending q, Pending r, double &a, double &b, double &c,
                                            double *red)
{
    a = 0.0;
    b = 0.0;
    c = 0.0;
    const int n = max(p.n, max(q.n, r.n));
    for (int i = threadIdx.x; i < n; i += kBlock) {
        if (i < p.n) a += p.p[i];
        if (i < q.n) b += q.p[i];
        if (i < r.n) c += r.p[i];
    }
    block_allreduce_sum3(a, b, c, red);
}

template <bool NT>
__device__ __forceinline__ double ld_stream(const double *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ int32_t ld_stream(const int32_t *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// ---------------------------------------------------------------------- SpMV
//
// One lane = one row, one wavefront = one 64-row chunk.
//   SELL: slot k of a chunk is one contiguous 256-B (col) + 512-B (val) read.
//   DIA : diagonal d of a chunk is one contiguous 512-B val read; x is read as
//         the same 64 rows shifted by delta[d] (no index traffic at all).
// Work distribution is XCD-aware: workgroups b, b+8, b+16.. share an XCD (and
// its 4 MiB L2), so XCD x sweeps the contiguous chunk range [x*CPX,(x+1)*CPX)
// with its workgroups advancing as one front; the gathered x window of a front
// (rows +- the generator's strides) then stays in that XCD's L2 instead of
// being fetched by all eight.
//
// MODE 0: y = A x                           (FMATVEC seam, DROP_STATES, bench)
// MODE 1: y = s A u ; partial = udot.y      (Arnoldi column, s = 1/||u||)
// MODE 2: y = s A u ; partial = y.y         (the AVNORM product)
// MODE 3: y = s A u ; partial = udot.y, partial2 = udot2.y   (IOP(2) column)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ d2 ld_stream2(const double *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
    return *reinterpret_cast<const d2 *>(p);
}
template <bool NT>
__device__ __forceinline__ i2 ld_stream2(const int32_t *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const i2 *>(p));
    return *reinterpret_cast<const i2 *>(p);
}

// One SELL row: the slots come in pairs (sell_pos), columns as one 8-byte and values as one 16-byte load
// per pair, summed slot by slot in FMATVEC's order.
template <bool NT>
__device__ __forceinline__ double row_sell(const SellDev &A, const double *__restrict__ xg, int64_t row0,
                                           int64_t c, int lane)
{
    const int64_t off = A.off[c];
    const int w = (int)((A.off[c + 1] - off) >> 6);
    const int w2 = w >> 1;                                   // slot pairs; an odd width ends in a single slot
    const int64_t r = (c << 6) + lane;
    const int32_t *cp = A.col + off + 2 * lane;
    const double *vp = A.val + off + 2 * lane;
    double sum = -ld_stream<NT>(A.diag + r) * xg[row0 + r];
    int k = 0;
    for (; k + 2 <= w2; k += 2) {
        const i2 c0 = ld_stream2<NT>(cp + (k + 0) * 128), c1 = ld_stream2<NT>(cp + (k + 1) * 128);
        const d2 v0 = ld_stream2<NT>(vp + (k + 0) * 128), v1 = ld_stream2<NT>(vp + (k + 1) * 128);
        sum += v0.x * xg[c0.x];
        sum += v0.y * xg[c0.y];
        sum += v1.x * xg[c1.x];
        sum += v1.y * xg[c1.y];
    }
    if (k < w2) {
        const i2 c0 = ld_stream2<NT>(cp + k * 128);
        const d2 v0 = ld_stream2<NT>(vp + k * 128);
        sum += v0.x * xg[c0.x];
        sum += v0.y * xg[c0.y];
    }
    if (w & 1) sum += ld_stream<NT>(A.val + off + w2 * 128 + lane) * xg[ld_stream<NT>(A.col + off + w2 * 128 + lane)];
    return sum;
}

// The same row with dictionary-coded columns (kSellCode*, kfsp_internal.h): the column of an entry is
// row + table[index], the table of the chunk's <= 64 distinct offsets sits one entry per lane in a register and
// an index is turned into its offset by ds_bpermute.  Same values, same order, same addresses as row_sell.
template <bool NT>
__device__ __forceinline__ double row_sell_coded(const SellDev &A, const double *__restrict__ xg, int64_t row0,
                                                 int64_t c, int lane)
{
    const int dtl = __builtin_amdgcn_readfirstlane(A.dtlen[c]);
    if (dtl == 0) return row_sell<NT>(A, xg, row0, c, lane);          // this chunk kept its plain columns (uniform branch)
    const int64_t off = A.off[c];
    const int w = (int)((A.off[c + 1] - off) >> 6);
    const int w2 = w >> 1;
    const int64_t r = (c << 6) + lane;
    const int tab = lane < dtl ? A.dtab[(c << 6) + lane] : 0;
    const unsigned long long *cp = A.code + A.codeoff[c] + lane;
    const double *vp = A.val + off + 2 * lane;
    const double *xr = xg + row0 + r;                                 // x of the row itself; entries are xr[offset]
    unsigned long long cw = __builtin_nontemporal_load(cp);
    int used = 0;
    double sum = -ld_stream<NT>(A.diag + r) * xr[0];
#define KFSP_NEXT_X(XV)                                                                     \
    {                                                                                       \
        if (used == kSellCodePerWord) {                                                     \
            cp += 64;                                                                       \
            cw = __builtin_nontemporal_load(cp);                                            \
            used = 0;                                                                       \
        }                                                                                   \
        const int d_ = __builtin_amdgcn_ds_bpermute((int)(cw & 63ull) << 2, tab);           \
        cw >>= kSellCodeBits;                                                               \
        ++used;                                                                             \
        XV = xr[d_];                                                                        \
    }
    int k = 0;
    for (; k + 2 <= w2; k += 2) {
        const d2 v0 = ld_stream2<NT>(vp + (k + 0) * 128), v1 = ld_stream2<NT>(vp + (k + 1) * 128);
        double x0, x1, x2, x3;
        KFSP_NEXT_X(x0)
        KFSP_NEXT_X(x1)
        KFSP_NEXT_X(x2)
        KFSP_NEXT_X(x3)
        sum += v0.x * x0;
        sum += v0.y * x1;
        sum += v1.x * x2;
        sum += v1.y * x3;
    }
    if (k < w2) {
        const d2 v0 = ld_stream2<NT>(vp + k * 128);
        double x0, x1;
        KFSP_NEXT_X(x0)
        KFSP_NEXT_X(x1)
        sum += v0.x * x0;
        sum += v0.y * x1;
    }
    if (w & 1) {
        double x0;
        KFSP_NEXT_X(x0)
        sum += ld_stream<NT>(A.val + off + w2 * 128 + lane) * x0;
    }
#undef KFSP_NEXT_X
    return sum;
}

// Banded form, TWO consecutive rows per lane (a wavefront covers 128 rows): the
// value streams - the bulk of the traffic - are read 16 B per lane, which is
// worth ~9 % of HBM rate over 8 B per lane (profiles/r01_stream_widths.log).
// x too is read 16 B per lane and diagonal (ld_xpair: shifted by delta, so 8-B aligned in general).
// x[p], x[p + 1] as ONE 16-byte load (8-byte aligned): a CU issues vector-memory instructions at a fixed rate
// whatever their width, so two 8-byte gathers cost twice one 16-byte gather.  p is clamped to [-1, n - 1]: the
// element before x and the one behind it are the guard words / column padding every device buffer carries
// (DevBuf, vcol): finite, and only ever multiplied by a stored 0.
typedef double __attribute__((ext_vector_type(2))) xpair_t;
typedef xpair_t __attribute__((aligned(8))) xpair_u;
__device__ __forceinline__ xpair_t ld_xpair(const double *__restrict__ xg, int64_t p, int64_t last)
{
    p = p < -1 ? -1 : (p > last ? last : p);
    return *reinterpret_cast<const xpair_u *>(xg + p);
}

template <bool NT, bool MASKED>
__device__ __forceinline__ d2 rows_dia(const DiaDev &D, const double *__restrict__ xg, int64_t row0,
                                       int64_t c, int lane, unsigned m)
{
    const int64_t r = (c << 7) + 2 * lane;
    const int64_t g = row0 + r;
    const int64_t last = D.n - 1;
    const double *vp = D.val + r;
    const double *zp = D.zero + 2 * lane;
    // m: which diagonals have entries in this group (same for the whole wavefront; the
    // caller fetched it one trip ahead, so no load sits in front of the address arithmetic)
    const d2 dg = ld_stream2<NT>(D.diag + r);
    const xpair_t xd = ld_xpair(xg, g, last);
    d2 sum;
    sum.x = -dg.x * xd.x;
    sum.y = -dg.y * xd.y;
    int d = 0;
    for (; d + 2 <= D.nd; d += 2) {
        const bool on0 = !MASKED || ((m >> d) & 1u), on1 = !MASKED || ((m >> (d + 1)) & 1u);
        // an empty segment costs no HBM traffic: zeros from a cached line, x from the row itself
        const d2 v0 = ld_stream2<NT>(on0 ? vp + (int64_t)(d + 0) * D.ld : zp);
        const d2 v1 = ld_stream2<NT>(on1 ? vp + (int64_t)(d + 1) * D.ld : zp);
        const xpair_t x0 = ld_xpair(xg, g + (on0 ? D.delta[d + 0] : 0), last);
        const xpair_t x1 = ld_xpair(xg, g + (on1 ? D.delta[d + 1] : 0), last);
        sum.x += v0.x * x0.x;
        sum.y += v0.y * x0.y;
        sum.x += v1.x * x1.x;
        sum.y += v1.y * x1.y;
    }
    for (; d < D.nd; ++d) {
        const bool on0 = !MASKED || ((m >> d) & 1u);
        const d2 v0 = ld_stream2<NT>(on0 ? vp + (int64_t)d * D.ld : zp);
        const xpair_t x0 = ld_xpair(xg, g + (on0 ? D.delta[d] : 0), last);
        sum.x += v0.x * x0.x;
        sum.y += v0.y * x0.y;
    }
    return sum;
}

// Matrix-free box generator, two consecutive rows per lane like the banded form.  The descriptor B
// is a kernel argument (uniform: scalar loads), the factor tables sit in LDS, the coordinates of a
// row in registers (NS compile-time for the benchmark models; species picked by select chains, a
// register array indexed by run-time data would spill).  NS = 0 / NR = 0: run-time sizes.
template <int NS>
__device__ __forceinline__ int box_pick(const int *x, int s)
{
    constexpr int N = NS ? NS : kBoxMaxS;
    int v = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = (s == i) ? x[i] : v;
    return v;
}

template <int NS, int NR>
__device__ __forceinline__ double row_box(const BoxDev &B, const double *tab, const int *x, const double *__restrict__ xg,
                                          int64_t g)
{
    constexpr int RMAX = NR ? NR : kBoxMaxR;
    const int nr = NR ? NR : B.nr;
    // a_k at x itself, reaction by reaction in sorted position
    double ax[RMAX];
#pragma unroll
    for (int k = 0; k < RMAX; ++k) {
        ax[k] = 0.0;
        if (k < nr) {
            double a = tab[B.dep_off[k][0] + box_pick<NS>(x, B.dep_s[k][0])];
#pragma unroll
            for (int i = 1; i < kBoxMaxDep; ++i)
                if (i < B.ndep[k]) a *= tab[B.dep_off[k][i] + box_pick<NS>(x, B.dep_s[k][i])];
            ax[k] = a;
        }
    }
    // DIAG = sum of all propensities at x, in the model's reaction order (StateSpace.f90:207-212)
    double dsum = 0.0;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
        if (q < nr) {
            const int k = B.dorder[q];
            double a = ax[0];
#pragma unroll
            for (int j = 1; j < RMAX; ++j) a = (k == j) ? ax[j] : a;
            dsum += a;
        }
    }
    double sum = -dsum * xg[g];
#pragma unroll
    for (int k = 0; k < RMAX; ++k) {                    // ascending column offset, as the banded kernel
        if (k < nr) {
            bool in = true;
            bool same = true;                           // the source state has the same factors as x itself
#pragma unroll
            for (int i = 0; i < kBoxMaxDep; ++i) {
                if (i < B.nmov[k]) {
                    const int s = B.mov_s[k][i];
                    const int v = box_pick<NS>(x, s) - B.mov_nu[k][i];
                    in = in && v >= 0 && v < B.mov_dim[k][i];
                }
                if (i < B.ndep[k]) same = same && B.dep_nu[k][i] == 0;
            }
            double a = ax[k];
            if (!same) {                                // uniform branch: B is the same for every lane
                a = 0.0;
                if (in) {
                    a = tab[B.dep_off[k][0] + box_pick<NS>(x, B.dep_s[k][0]) - B.dep_nu[k][0]];
#pragma unroll
                    for (int i = 1; i < kBoxMaxDep; ++i)
                        if (i < B.ndep[k]) a *= tab[B.dep_off[k][i] + box_pick<NS>(x, B.dep_s[k][i]) - B.dep_nu[k][i]];
                }
            }
            sum += (in ? a : 0.0) * xg[in ? g + B.delta[k] : g];
        }
    }
    return sum;
}

template <int NS, int NR>
__device__ __forceinline__ d2 rows_box(const BoxDev &B, const double *tab, const double *__restrict__ xg,
                                       int64_t row0, int64_t nloc, int64_t c, int lane)
{
    constexpr int SMAX = NS ? NS : kBoxMaxS;
    const int ns = NS ? NS : B.ns;
    d2 sum = {0.0, 0.0};
    // rows beyond this rank's block read as zero rows (with more ranks the last 128-row group of a
    // block may reach into the next rank's rows, which are real states of the box)
    const int64_t r0 = (c << 7) + 2 * lane;
    if (r0 >= nloc) return sum;
    const int64_t g = row0 + r0;
    // coordinates of row g: successive division by the box dimensions (exact: g < 2^31, one correction step)
    int x[SMAX];
    uint32_t q = (uint32_t)g;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
        x[s] = 0;
        if (s + 1 < ns) {
            const int d = B.dims[s];
            uint32_t t = (uint32_t)((double)q * B.inv_dim[s]);
            int r = (int)(q - t * (uint32_t)d);
            if (r < 0) {
                --t;
                r += d;
            } else if (r >= d) {
                ++t;
                r -= d;
            }
            x[s] = r;
            q = t;
        } else if (s + 1 == ns) {
            x[s] = (int)q;
        }
    }
    sum.x = row_box<NS, NR>(B, tab, x, xg, g);
    if (r0 + 1 < nloc) {
        bool carry = true;                               // next row: +1 with carry
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            if (s < ns && carry) {
                const int v = x[s] + 1;
                carry = v >= B.dims[s] && s + 1 < ns;
                x[s] = carry ? 0 : v;
            }
        }
        sum.y = row_box<NS, NR>(B, tab, x, xg, g + 1);
    }
    return sum;
}

// Single-factor fast path (BoxFast): NS species with PER entry slots each, both compile-time, so the
// species and slot of every entry are constants: straight-line code, no branches, all gathers of x and
// table look-ups of a row in flight together.  Per species ONE 16-byte LDS entry gives the sum of its
// propensities at x_s (its share of DIAG, StateSpace.f90:207-212) and a word with one bit per entry:
// "this coordinate lets the entry's source state x - nu lie inside the box"; the AND over the species
// is the row's set of valid entries.  An invalid entry is not branched around: its mask (0 / -1)
// redirects the look-up to the 0.0 at the head of the LDS image.
// A lane owns rows 2l and 2l+1 of the wave's 128, and ONE 16-byte load per entry fetches the source
// elements of both (x[g + delta], x[g + 1 + delta]): the kernel is bound by the number of vector-memory
// instructions a CU can issue, not by bytes.  If only one of the two rows has the entry, the other slot
// holds some other element of x, or the element just before / behind x (every device buffer carries
// guard words, DevBuf): finite, and multiplied by the 0.0.  x is addressed as wave base (scalar) +
// 32-bit lane offset (kfsp_set_matrix_box checks the reach).  The descriptor is read ONCE per wavefront
// into scalar registers (BoxRegs): left as loads inside the row code they would be re-issued for every
// row, as per-lane vector loads.
template <int NS, int PER>
struct BoxRegs {
    int koff8[NS][PER];    // byte offset in the LDS image of a_k(x_s - nu_k), less 8 x_s
    int delta8[NS][PER];   // byte offset in x of the source state relative to the row
    int df8[NS];           // byte offset of the species' {sum, valid bits} table
    int dims[NS];
    double inv_dim[NS];
    int bias8;             // bytes the wave's x base lies below its first row (>= the largest backward reach)
};

template <int NS, int PER>
__device__ __forceinline__ void box_load(const BoxFast *__restrict__ F, BoxRegs<NS, PER> &R)
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        R.dims[s] = __builtin_amdgcn_readfirstlane(F->dims[s]);
        R.df8[s] = __builtin_amdgcn_readfirstlane(F->df8[s]);
        const double inv = F->inv_dim[s];
        R.inv_dim[s] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(inv)),
                                        __builtin_amdgcn_readfirstlane(__double2loint(inv)));
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            R.koff8[s][j] = __builtin_amdgcn_readfirstlane(F->koff8[s][j]);
            R.delta8[s][j] = __builtin_amdgcn_readfirstlane(F->delta8[s][j]);
        }
    }
    R.bias8 = __builtin_amdgcn_readfirstlane(F->bias8);
}

extern __shared__ double box_lds[];   // matrix-free kernels: the image of the factor tables (dynamic LDS)
// byte pointers that keep their address space through integer arithmetic (LDS reads, scalar-base global loads)
typedef const __attribute__((address_space(3))) char *lds_bytes_t;
typedef const __attribute__((address_space(1))) char *global_bytes_t;
typedef double __attribute__((ext_vector_type(2))) box_pair_t;
// One population count of one species in the LDS image, 16 bytes: { double dsum - the sum of the
// propensities that depend on this species; unsigned valid - bit e: entry e's source coordinate of this
// species is inside the box; unsigned pad }.

template <int S, int NS, int PER>
__device__ __forceinline__ void box_species(const BoxRegs<NS, PER> &R, int xa, int xb, unsigned va, unsigned vb,
                                            global_bytes_t xw, unsigned voff, double &acca, double &accb)
{
    const unsigned lds0 = (unsigned)(size_t)(lds_bytes_t)box_lds;                  // LDS address of the image = of its 0.0
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int ma = __builtin_amdgcn_sbfe(va, S * PER + j, 1);                  // -1: source state of row A inside the box
        const int mb = __builtin_amdgcn_sbfe(vb, S * PER + j, 1);
        const unsigned ata = lds0 + (unsigned)(8 * xa + R.koff8[S][j]);           // a_k(x - nu_k) ...
        const unsigned atb = lds0 + (unsigned)(8 * xb + R.koff8[S][j]);
        const double a1a = *(const __attribute__((address_space(3))) double *)(size_t)((ma & ata) | (~ma & lds0));   // ... or 0
        const double a1b = *(const __attribute__((address_space(3))) double *)(size_t)((mb & atb) | (~mb & lds0));
        const unsigned vsrc = voff + (unsigned)R.delta8[S][j];                     // (the same in every trip of the lane)
        const unsigned at = (unsigned)__builtin_amdgcn_bitop3_b32(ma | mb, (int)vsrc, (int)voff, 0xCA);   // either ? vsrc : voff
        const box_pair_t xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + at);
        acca += a1a * xv.x;
        accb += a1b * xv.y;
    }
}

template <int NS, int PER>
__device__ __forceinline__ void box_df(const BoxRegs<NS, PER> &R, int c0, int c1, int c2, int c3, int c4, int c5,
                                       double &dsum, unsigned &valid)
{
    const lds_bytes_t lds = (lds_bytes_t)box_lds;
    {
        const lds_bytes_t f = lds + R.df8[0] + 16 * c0;
        dsum = *(const __attribute__((address_space(3))) double *)f;
        valid = *(const __attribute__((address_space(3))) unsigned *)(f + 8);
    }
#define KFSP_BOX_DF(S, VAR)                                                                            \
    if (NS > S) {                                                                                      \
        const lds_bytes_t f = lds + R.df8[NS > S ? S : 0] + 16 * VAR;                                  \
        dsum += *(const __attribute__((address_space(3))) double *)f;                                  \
        valid &= *(const __attribute__((address_space(3))) unsigned *)(f + 8);                         \
    }
    KFSP_BOX_DF(1, c1)
    KFSP_BOX_DF(2, c2)
    KFSP_BOX_DF(3, c3)
    KFSP_BOX_DF(4, c4)
    KFSP_BOX_DF(5, c5)
#undef KFSP_BOX_DF
}

template <int NS, int PER>
__device__ __forceinline__ d2 rows_box1(const BoxRegs<NS, PER> &R, const double *__restrict__ xg,
                                        int64_t row0, int64_t nloc, int64_t c, int lane)
{
    d2 sum = {0.0, 0.0};
    const int64_t r0 = (c << 7) + 2 * lane;
    if (r0 >= nloc) return sum;
    const int64_t g = row0 + r0;
    // x of this wave's 128 rows and everything an entry reaches from them: scalar base + unsigned lane offset
    const uint64_t xb = reinterpret_cast<uint64_t>(xg + (row0 + (c << 7))) - (uint64_t)(int64_t)R.bias8;
    const global_bytes_t xw = (global_bytes_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb) |
                                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32);
    const unsigned voff = (unsigned)(16 * lane + R.bias8);
    // coordinates of row g: successive division by the box dimensions (exact: g < 2^31, one correction step)
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    uint32_t q = (uint32_t)g;
#define KFSP_BOX_DEC(S, VAR)                                             \
    if (NS > S + 1) {                                                    \
        const int d = R.dims[NS > S ? S : 0];                            \
        uint32_t t = (uint32_t)((double)q * R.inv_dim[NS > S ? S : 0]);  \
        int r = (int)(q - t * (uint32_t)d);                              \
        const int lo = r < 0, hi = r >= d;                               \
        t = t - lo + hi;                                                 \
        r = r + (lo ? d : 0) - (hi ? d : 0);                             \
        VAR = r;                                                         \
        q = t;                                                           \
    } else if (NS == S + 1) {                                            \
        VAR = (int)q;                                                    \
    }
    KFSP_BOX_DEC(0, c0)
    KFSP_BOX_DEC(1, c1)
    KFSP_BOX_DEC(2, c2)
    KFSP_BOX_DEC(3, c3)
    KFSP_BOX_DEC(4, c4)
    KFSP_BOX_DEC(5, c5)
#undef KFSP_BOX_DEC
    // the next row: +1 with carry (the last species never wraps while g + 1 < n)
    int b0 = c0, b1 = c1, b2 = c2, b3 = c3, b4 = c4, b5 = c5, carry = 1;
#define KFSP_BOX_INC(S, VAR)                                             \
    if (NS > S) {                                                        \
        const int v = VAR + carry;                                       \
        const int wrap = (NS > S + 1) && v >= R.dims[NS > S ? S : 0];    \
        VAR = wrap ? 0 : v;                                              \
        carry = wrap;                                                    \
    }
    KFSP_BOX_INC(0, b0)
    KFSP_BOX_INC(1, b1)
    KFSP_BOX_INC(2, b2)
    KFSP_BOX_INC(3, b3)
    KFSP_BOX_INC(4, b4)
    KFSP_BOX_INC(5, b5)
#undef KFSP_BOX_INC
    const bool two = r0 + 1 < nloc;
    if (!two) b0 = b1 = b2 = b3 = b4 = b5 = 0;                 // (no such row: any coordinates inside the tables)
    double dsa, dsb;
    unsigned va, vb;
    box_df<NS, PER>(R, c0, c1, c2, c3, c4, c5, dsa, va);
    box_df<NS, PER>(R, b0, b1, b2, b3, b4, b5, dsb, vb);
    if (!two) vb = 0u;
    const box_pair_t xd = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + voff);
    double acca = 0.0, accb = 0.0;
    box_species<0, NS, PER>(R, c0, b0, va, vb, xw, voff, acca, accb);
    if (NS > 1) box_species<(NS > 1 ? 1 : 0), NS, PER>(R, c1, b1, va, vb, xw, voff, acca, accb);
    if (NS > 2) box_species<(NS > 2 ? 2 : 0), NS, PER>(R, c2, b2, va, vb, xw, voff, acca, accb);
    if (NS > 3) box_species<(NS > 3 ? 3 : 0), NS, PER>(R, c3, b3, va, vb, xw, voff, acca, accb);
    if (NS > 4) box_species<(NS > 4 ? 4 : 0), NS, PER>(R, c4, b4, va, vb, xw, voff, acca, accb);
    if (NS > 5) box_species<(NS > 5 ? 5 : 0), NS, PER>(R, c5, b5, va, vb, xw, voff, acca, accb);
    sum.x = acca - dsa * xd.x;
    sum.y = two ? accb - dsb * xd.y : 0.0;
    return sum;
}

// FMT: 0 SELL-64, 1 banded, 2 banded with group masks, 3 matrix-free box (descriptor interpreted at run time),
// 4 matrix-free box, single-factor fast path with NS species and NE entry slots per species
template <int MODE, bool NT, int FMT, int NS = 0, int NE = 0>
__global__ __launch_bounds__(kBlock) void k_spmv(SpmvArgs a)
{
    constexpr bool DIA = FMT != 0 && FMT != 5;
    constexpr bool BOX = FMT == 3 || FMT == 4;
    __shared__ double red[12];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE != 0) {
        if (*a.brk_flag) return;
    }
    const double *tab = nullptr;
    BoxRegs<(NS ? NS : 1), (NE ? NE : 1)> boxr;
    if (BOX) {
        for (int i = threadIdx.x; i < a.B.ntab; i += kBlock) box_lds[i] = a.box_tab[i];
        __syncthreads();
        tab = box_lds;
        if (FMT == 4) box_load(a.box_fast, boxr);
    }

    // SELL: one 64-row chunk per wavefront trip; DIA: one 128-row group
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int bx = gridDim.x >> 3;                        // workgroups per XCD
    const int64_t cpx = (a.trip_end - a.trip_begin + 7) >> 3;
    const int64_t cbeg = a.trip_begin + (int64_t)xcd * cpx;
    const int64_t cend = (cbeg + cpx < a.trip_end) ? cbeg + cpx : a.trip_end;
    const int64_t cstep = (int64_t)bx * 4;
    int64_t c = cbeg + (int64_t)slot * 4 + wave;

    // The first trip's row sums are started before the pending norm is
    // finished: the partial-sum round trip overlaps the first generator loads.
    d2 sum = {0.0, 0.0};
    int64_t ct = c < a.trip_split ? c : c + a.trip_jump;   // actual trip of linear index c
    if (a.trip_order && c < cend) ct = __builtin_amdgcn_readfirstlane(a.trip_order[c]);
    unsigned gm = 0xFFFFFFFFu;                              // group mask of the trip about to be computed
    if (FMT == 2 && c < cend) gm = __builtin_amdgcn_readfirstlane(a.D.gmask[ct]);
    if (c < cend) {
        if (FMT == 4) sum = rows_box1<(NS ? NS : 1), (NE ? NE : 1)>(boxr, a.xg, a.row0, a.A.nrows, ct, lane);
        else if (BOX) sum = rows_box<0, 0>(a.B, tab, a.xg, a.row0, a.A.nrows, ct, lane);
        else if (DIA) sum = rows_dia<NT, FMT == 2>(a.D, a.xg, a.row0, ct, lane, gm);
        else if (FMT == 5) sum.x = row_sell_coded<NT>(a.A, a.xg, a.row0, ct, lane);
            else sum.x = row_sell<NT>(a.A, a.xg, a.row0, ct, lane);
    }

    double s = 1.0;
    if (MODE != 0) {
        const double S = finish_sum(a.sq, red);
        const double nrm = sqrt(S);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.sq_final) *a.sq_final = S;
            if (a.h_sub) *a.h_sub = nrm;
        }
        if (a.break_tol >= 0.0 && !(nrm > a.break_tol)) {   // happy breakdown :249
            if (blockIdx.x == 0 && threadIdx.x == 0) *a.brk_flag = 1;
            return;
        }
        s = 1.0 / nrm;
    }

    double acc = 0.0, acc2 = 0.0;
    while (c < cend) {
        const int64_t cn = c + cstep;
        int64_t ctn = cn < a.trip_split ? cn : cn + a.trip_jump;
        if (a.trip_order && cn < cend) ctn = __builtin_amdgcn_readfirstlane(a.trip_order[cn]);
        if (FMT == 2 && cn < cend) gm = __builtin_amdgcn_readfirstlane(a.D.gmask[ctn]);
        if (DIA) {
            const int64_t r = (ct << 7) + 2 * lane;
            if (MODE != 0) {
                sum.x *= s;
                sum.y *= s;
            }
            *reinterpret_cast<d2 *>(a.y + r) = sum;
            if (MODE == 1 || MODE == 3) {
                const d2 u = *reinterpret_cast<const d2 *>(a.udot + r);
                acc += u.x * sum.x;
                acc += u.y * sum.y;
            }
            if (MODE == 2) {
                acc += sum.x * sum.x;
                acc += sum.y * sum.y;
            }
            if (MODE == 3) {
                const d2 u = *reinterpret_cast<const d2 *>(a.udot2 + r);
                acc2 += u.x * sum.x;
                acc2 += u.y * sum.y;
            }
        } else {
            const int64_t r = (ct << 6) + lane;
            double v = sum.x;
            if (MODE != 0) v *= s;
            a.y[r] = v;
            if (MODE == 1 || MODE == 3) acc += a.udot[r] * v;
            if (MODE == 2) acc += v * v;
            if (MODE == 3) acc2 += a.udot2[r] * v;
        }
        c = cn;
        ct = ctn;
        if (c < cend) {
            if (FMT == 4) sum = rows_box1<(NS ? NS : 1), (NE ? NE : 1)>(boxr, a.xg, a.row0, a.A.nrows, ct, lane);
            else if (BOX) sum = rows_box<0, 0>(a.B, tab, a.xg, a.row0, a.A.nrows, ct, lane);
            else if (DIA) sum = rows_dia<NT, FMT == 2>(a.D, a.xg, a.row0, ct, lane, gm);
            else if (FMT == 5) sum.x = row_sell_coded<NT>(a.A, a.xg, a.row0, ct, lane);
            else sum.x = row_sell<NT>(a.A, a.xg, a.row0, ct, lane);
        }
    }
    if (MODE == 3) {
        double dummy = 0.0;
        block_allreduce_sum3(acc, acc2, dummy, red);
        if (threadIdx.x == 0) {
            a.partial[blockIdx.x] = acc;
            a.partial2[blockIdx.x] = acc2;
        }
    } else if (MODE != 0) {
        const double t = block_allreduce_sum(acc, red);
        if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Format 6: the single-factor matrix-free product (format 4) with the NEAR part of x staged in LDS.
// Format 4 fetches every entry's x pair with its own 16-byte global gather: a row's x goes through the CU's vector L1 once
// per entry (7 x for the repressilator, 13 x for the 6-species network), and at ~18 wave-instructions per ns chip-wide -
// half the 64 B/clk/CU of the L1 - that, not HBM, is what the kernel waits for (profiles/r03_pmc_summary_boxes_and_fsp.txt:
// the same rate on c3x and on c5s).  The four wavefronts of a workgroup take four consecutive 128-row trips, i.e. 512
// consecutive rows: their x and the `reach` rows on either side are loaded ONCE per workgroup pass (16 B per lane,
// coalesced, two buffers so that the next pass's window is in flight while this one is computed; one barrier per
// pass), and every entry whose shift is within the reach - the diagonal, +-1, +-d1, +-d1 d2 when they fit - reads its
// pair from LDS (ds_read2_b64, conflict-free: consecutive lanes read consecutive pairs).  Far entries keep their
// gathers.  Same table look-ups, same products, same order of additions as format 4: bit-identical results.
template <int S, int NS, int PER>
__device__ __forceinline__ void box_species_lds(const BoxRegs<NS, PER> &R, int xa, int xb, unsigned va, unsigned vb,
                                                global_bytes_t xw, unsigned voff, const double *win, int wrow, int reach8,
                                                double &acca, double &accb)
{
    const unsigned lds0 = (unsigned)(size_t)(lds_bytes_t)box_lds;                  // LDS address of the image = of its 0.0
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int ma = __builtin_amdgcn_sbfe(va, S * PER + j, 1);                  // -1: source state of row A inside the box
        const int mb = __builtin_amdgcn_sbfe(vb, S * PER + j, 1);
        const unsigned ata = lds0 + (unsigned)(8 * xa + R.koff8[S][j]);           // a_k(x - nu_k) ...
        const unsigned atb = lds0 + (unsigned)(8 * xb + R.koff8[S][j]);
        const double a1a = *(const __attribute__((address_space(3))) double *)(size_t)((ma & ata) | (~ma & lds0));   // ... or 0
        const double a1b = *(const __attribute__((address_space(3))) double *)(size_t)((mb & atb) | (~mb & lds0));
        const int d8 = R.delta8[S][j];
        box_pair_t xv;
        if ((d8 < 0 ? -d8 : d8) <= reach8) {                                      // uniform: the descriptor sits in SGPRs
            const int sh = (ma | mb) ? (d8 >> 3) : 0;                              // neither row has the entry: their own x
            xv.x = win[wrow + sh];
            xv.y = win[wrow + sh + 1];
        } else {
            const unsigned vsrc = voff + (unsigned)d8;
            const unsigned at = (unsigned)__builtin_amdgcn_bitop3_b32(ma | mb, (int)vsrc, (int)voff, 0xCA);   // either ? vsrc : voff
            xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + at);
        }
        acca += a1a * xv.x;
        accb += a1b * xv.y;
    }
}

template <int NS, int PER>
__device__ __forceinline__ d2 rows_box_lds(const BoxRegs<NS, PER> &R, const double *__restrict__ xg, int64_t row0, int64_t nloc,
                                           int64_t c, int lane, const double *win, int wrow, int reach8)
{
    d2 sum = {0.0, 0.0};
    const int64_t r0 = (c << 7) + 2 * lane;
    if (r0 >= nloc) return sum;
    const int64_t g = row0 + r0;
    const uint64_t xb = reinterpret_cast<uint64_t>(xg + (row0 + (c << 7))) - (uint64_t)(int64_t)R.bias8;
    const global_bytes_t xw = (global_bytes_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb) |
                                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32);
    const unsigned voff = (unsigned)(16 * lane + R.bias8);
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    uint32_t q = (uint32_t)g;
#define KFSP_BOX_DEC(S, VAR)                                             \
    if (NS > S + 1) {                                                    \
        const int d = R.dims[NS > S ? S : 0];                            \
        uint32_t t = (uint32_t)((double)q * R.inv_dim[NS > S ? S : 0]);  \
        int r = (int)(q - t * (uint32_t)d);                              \
        const int lo = r < 0, hi = r >= d;                               \
        t = t - lo + hi;                                                 \
        r = r + (lo ? d : 0) - (hi ? d : 0);                             \
        VAR = r;                                                         \
        q = t;                                                           \
    } else if (NS == S + 1) {                                            \
        VAR = (int)q;                                                    \
    }
    KFSP_BOX_DEC(0, c0)
    KFSP_BOX_DEC(1, c1)
    KFSP_BOX_DEC(2, c2)
    KFSP_BOX_DEC(3, c3)
    KFSP_BOX_DEC(4, c4)
    KFSP_BOX_DEC(5, c5)
#undef KFSP_BOX_DEC
    int b0 = c0, b1 = c1, b2 = c2, b3 = c3, b4 = c4, b5 = c5, carry = 1;
#define KFSP_BOX_INC(S, VAR)                                             \
    if (NS > S) {                                                        \
        const int v = VAR + carry;                                       \
        const int wrap = (NS > S + 1) && v >= R.dims[NS > S ? S : 0];    \
        VAR = wrap ? 0 : v;                                              \
        carry = wrap;                                                    \
    }
    KFSP_BOX_INC(0, b0)
    KFSP_BOX_INC(1, b1)
    KFSP_BOX_INC(2, b2)
    KFSP_BOX_INC(3, b3)
    KFSP_BOX_INC(4, b4)
    KFSP_BOX_INC(5, b5)
#undef KFSP_BOX_INC
    const bool two = r0 + 1 < nloc;
    if (!two) b0 = b1 = b2 = b3 = b4 = b5 = 0;
    double dsa, dsb;
    unsigned va, vb;
    box_df<NS, PER>(R, c0, c1, c2, c3, c4, c5, dsa, va);
    box_df<NS, PER>(R, b0, b1, b2, b3, b4, b5, dsb, vb);
    if (!two) vb = 0u;
    const double xda = win[wrow], xdb = win[wrow + 1];
    double acca = 0.0, accb = 0.0;
    box_species_lds<0, NS, PER>(R, c0, b0, va, vb, xw, voff, win, wrow, reach8, acca, accb);
    if (NS > 1) box_species_lds<(NS > 1 ? 1 : 0), NS, PER>(R, c1, b1, va, vb, xw, voff, win, wrow, reach8, acca, accb);
    if (NS > 2) box_species_lds<(NS > 2 ? 2 : 0), NS, PER>(R, c2, b2, va, vb, xw, voff, win, wrow, reach8, acca, accb);
    if (NS > 3) box_species_lds<(NS > 3 ? 3 : 0), NS, PER>(R, c3, b3, va, vb, xw, voff, win, wrow, reach8, acca, accb);
    if (NS > 4) box_species_lds<(NS > 4 ? 4 : 0), NS, PER>(R, c4, b4, va, vb, xw, voff, win, wrow, reach8, acca, accb);
    if (NS > 5) box_species_lds<(NS > 5 ? 5 : 0), NS, PER>(R, c5, b5, va, vb, xw, voff, win, wrow, reach8, acca, accb);
    sum.x = acca - dsa * xda;
    sum.y = two ? accb - dsb * xdb : 0.0;
    return sum;
}

// reach: rows staged on either side of the workgroup's 512 (even); x is readable for global rows [0, a.D.n)
template <int MODE, int NS, int PER>
__global__ __launch_bounds__(kBlock) void k_spmv_boxlds(SpmvArgs a, int reach)
{
    __shared__ double red[12];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE != 0) {
        if (*a.brk_flag) return;
    }
    BoxRegs<NS, PER> boxr;
    for (int i = threadIdx.x; i < a.B.ntab; i += kBlock) box_lds[i] = a.box_tab[i];
    box_load(a.box_fast, boxr);
    const int W = 4 * 128 + 2 * reach;                     // doubles per window
    double *win0 = box_lds + ((a.B.ntab + 1) & ~1);        // two windows behind the table image (16-byte aligned)
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int bx = gridDim.x >> 3;                        // workgroups per XCD
    const int64_t trips = a.trip_end - a.trip_begin;
    const int64_t cpx = (((trips + 7) >> 3) + 3) & ~(int64_t)3;    // trips per XCD, a multiple of the 4 a workgroup takes per pass
    const int64_t cbeg = a.trip_begin + (int64_t)xcd * cpx;
    const int64_t cend = (cbeg + cpx < a.trip_end) ? cbeg + cpx : a.trip_end;
    const int64_t cstep = (int64_t)bx * 4;
    int64_t c0 = cbeg + (int64_t)slot * 4;                // the workgroup's first trip of this pass (uniform)
    const int64_t xn = a.D.n;
    // window of the pass that starts at trip cw: global rows [row0 + 128 cw - reach, row0 + 128 (cw + 4) + reach), zero outside x
    auto stage = [&](double *win, int64_t cw) {
        const int64_t gb = a.row0 + (cw << 7) - reach;
        for (int i = 2 * (int)threadIdx.x; i < W; i += 2 * kBlock) {
            const int64_t g = gb + i;
            d2 v;
            if (g >= 0 && g + 1 < xn) {
                v = *reinterpret_cast<const d2 *>(a.xg + g);
            } else {
                v.x = (g >= 0 && g < xn) ? a.xg[g] : 0.0;
                v.y = (g + 1 >= 0 && g + 1 < xn) ? a.xg[g + 1] : 0.0;
            }
            *reinterpret_cast<d2 *>(win + i) = v;
        }
    };
    if (c0 < cend) stage(win0, c0);
    __syncthreads();                                       // table image + first window
    double s = 1.0;
    if (MODE != 0) {
        const double S = finish_sum(a.sq, red);
        const double nrm = sqrt(S);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.sq_final) *a.sq_final = S;
            if (a.h_sub) *a.h_sub = nrm;
        }
        if (a.break_tol >= 0.0 && !(nrm > a.break_tol)) {   // happy breakdown :249
            if (blockIdx.x == 0 && threadIdx.x == 0) *a.brk_flag = 1;
            return;
        }
        s = 1.0 / nrm;
    }
    const int reach8 = 8 * reach;
    const int wrow = wave * 128 + 2 * lane + reach;        // this lane's first row inside a window
    double acc = 0.0, acc2 = 0.0;
    int buf = 0;
    while (c0 < cend) {
        const int64_t c0n = c0 + cstep;
        double *win = win0 + buf * W;
        if (c0n < cend) stage(win0 + (buf ^ 1) * W, c0n);  // the next window travels while this one is computed
        const int64_t c = c0 + wave;
        if (c < cend) {
            d2 sum = rows_box_lds<NS, PER>(boxr, a.xg, a.row0, a.A.nrows, c, lane, win, wrow, reach8);
            const int64_t r = (c << 7) + 2 * lane;
            if (MODE != 0) {
                sum.x *= s;
                sum.y *= s;
            }
            *reinterpret_cast<d2 *>(a.y + r) = sum;
            if (MODE == 1 || MODE == 3) {
                const d2 u = *reinterpret_cast<const d2 *>(a.udot + r);
                acc += u.x * sum.x;
                acc += u.y * sum.y;
            }
            if (MODE == 2) {
                acc += sum.x * sum.x;
                acc += sum.y * sum.y;
            }
            if (MODE == 3) {
                const d2 u = *reinterpret_cast<const d2 *>(a.udot2 + r);
                acc2 += u.x * sum.x;
                acc2 += u.y * sum.y;
            }
        }
        __syncthreads();                                   // next window complete, this one free
        c0 = c0n;
        buf ^= 1;
    }
    if (MODE == 3) {
        double dummy = 0.0;
        block_allreduce_sum3(acc, acc2, dummy, red);
        if (threadIdx.x == 0) {
            a.partial[blockIdx.x] = acc;
            a.partial2[blockIdx.x] = acc2;
        }
    } else if (MODE != 0) {
        const double t = block_allreduce_sum(acc, red);
        if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
    }
}

template <int NS, int NE>
static void launch_boxlds_mode(int mode, dim3 g, dim3 b, const SpmvArgs &a, hipStream_t st, size_t lds, int reach)
{
    if (mode == 0) hipLaunchKernelGGL((k_spmv_boxlds<0, NS, NE>), g, b, lds, st, a, reach);
    else if (mode == 1) hipLaunchKernelGGL((k_spmv_boxlds<1, NS, NE>), g, b, lds, st, a, reach);
    else if (mode == 2) hipLaunchKernelGGL((k_spmv_boxlds<2, NS, NE>), g, b, lds, st, a, reach);
    else hipLaunchKernelGGL((k_spmv_boxlds<3, NS, NE>), g, b, lds, st, a, reach);
}

// format 6: lds_bytes = table image (rounded to 16 B) + two windows of (512 + 2 reach) doubles
void launch_spmv_boxlds(int mode, int grid, const SpmvArgs &a, hipStream_t st, size_t lds_bytes, int reach)
{
    dim3 g(grid), b(kBlock);
    switch (a.B.pad) {
    case 2 * 16 + 2: launch_boxlds_mode<2, 2>(mode, g, b, a, st, lds_bytes, reach); break;
    case 3 * 16 + 2: launch_boxlds_mode<3, 2>(mode, g, b, a, st, lds_bytes, reach); break;
    case 6 * 16 + 2: launch_boxlds_mode<6, 2>(mode, g, b, a, st, lds_bytes, reach); break;
    default: launch_boxlds_mode<6, 4>(mode, g, b, a, st, lds_bytes, reach); break;
    }
}

template <bool NT, int FMT>
static void launch_spmv_mode(int mode, dim3 g, dim3 b, const SpmvArgs &a, hipStream_t st, size_t lds = 0)
{
    if (mode == 0) hipLaunchKernelGGL((k_spmv<0, NT, FMT>), g, b, lds, st, a);
    else if (mode == 1) hipLaunchKernelGGL((k_spmv<1, NT, FMT>), g, b, lds, st, a);
    else if (mode == 2) hipLaunchKernelGGL((k_spmv<2, NT, FMT>), g, b, lds, st, a);
    else hipLaunchKernelGGL((k_spmv<3, NT, FMT>), g, b, lds, st, a);
}

template <int FMT, int NS, int NE>
static void launch_box_mode(int mode, dim3 g, dim3 b, const SpmvArgs &a, hipStream_t st, size_t lds)
{
    if (mode == 0) hipLaunchKernelGGL((k_spmv<0, false, FMT, NS, NE>), g, b, lds, st, a);
    else if (mode == 1) hipLaunchKernelGGL((k_spmv<1, false, FMT, NS, NE>), g, b, lds, st, a);
    else if (mode == 2) hipLaunchKernelGGL((k_spmv<2, false, FMT, NS, NE>), g, b, lds, st, a);
    else hipLaunchKernelGGL((k_spmv<3, false, FMT, NS, NE>), g, b, lds, st, a);
}

void launch_spmv(int mode, int grid, const SpmvArgs &a, bool nt, int fmt, hipStream_t st, size_t lds_bytes)
{
    dim3 g(grid), b(kBlock);
    if (fmt == 3) {
        launch_box_mode<3, 0, 0>(mode, g, b, a, st, lds_bytes);
        return;
    }
    if (fmt == 4) {
        // a.B.pad = species count of the instantiation * 16 + slots per species (kfsp_set_matrix_box)
        switch (a.B.pad) {
        case 2 * 16 + 2: launch_box_mode<4, 2, 2>(mode, g, b, a, st, lds_bytes); break;
        case 3 * 16 + 2: launch_box_mode<4, 3, 2>(mode, g, b, a, st, lds_bytes); break;
        case 6 * 16 + 2: launch_box_mode<4, 6, 2>(mode, g, b, a, st, lds_bytes); break;
        default: launch_box_mode<4, 6, 4>(mode, g, b, a, st, lds_bytes); break;
        }
        return;
    }
    if (nt) {
        if (fmt == 2) launch_spmv_mode<true, 2>(mode, g, b, a, st);
        else if (fmt == 1) launch_spmv_mode<true, 1>(mode, g, b, a, st);
        else if (fmt == 5) launch_spmv_mode<true, 5>(mode, g, b, a, st);
        else launch_spmv_mode<true, 0>(mode, g, b, a, st);
    } else {
        if (fmt == 2) launch_spmv_mode<false, 2>(mode, g, b, a, st);
        else if (fmt == 1) launch_spmv_mode<false, 1>(mode, g, b, a, st);
        else if (fmt == 5) launch_spmv_mode<false, 5>(mode, g, b, a, st);
        else launch_spmv_mode<false, 0>(mode, g, b, a, st);
    }
}

// --------------------------------------------------- orthogonalisation step
// h = (u_i . w) s_i ;  w -= h s_i u_i ;  partial = unext . w  (or w . w)
__global__ __launch_bounds__(kBlock) void k_ortho(OrthoArgs a)
{
    __shared__ double red[4];
    if (*a.brk_flag) return;
    const double si = 1.0 / sqrt(*a.sq_i);
    const double h = finish_sum(a.dot, red) * si;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.h_out = h;
    const double coef = h * si;
    double2 *w2 = reinterpret_cast<double2 *>(a.w);
    const double2 *u2 = reinterpret_cast<const double2 *>(a.ui);
    const double2 *n2 = reinterpret_cast<const double2 *>(a.unext);
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.npairs; i += (int64_t)gridDim.x * kBlock) {
        double2 w = w2[i];
        const double2 u = u2[i];
        w.x -= coef * u.x;
        w.y -= coef * u.y;
        w2[i] = w;
        if (n2) {
            const double2 v = n2[i];
            acc += v.x * w.x;
            acc += v.y * w.y;
        } else {
            acc += w.x * w.x;
            acc += w.y * w.y;
        }
    }
    const double t = block_allreduce_sum(acc, red);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
}

void launch_ortho(int grid, const OrthoArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_ortho, dim3(grid), dim3(kBlock), 0, st, a);
}

// both updates of an IOP(2) column in one pass (see Ortho2Args)
__global__ __launch_bounds__(kBlock) void k_ortho2(Ortho2Args a)
{
    __shared__ double red[12];
    if (*a.brk_flag) return;
    double2 *w2 = reinterpret_cast<double2 *>(a.w);
    const double2 *p1 = reinterpret_cast<const double2 *>(a.u1);
    const double2 *p2 = reinterpret_cast<const double2 *>(a.u2);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    // first trip's operands are requested before the scalars are finished, so
    // the partial-sum round trip overlaps the first vector loads
    const bool have0 = i < a.npairs, have1 = i + stride < a.npairs;
    const double2 zero = make_double2(0.0, 0.0);
    double2 wa = have0 ? w2[i] : zero, wb = have1 ? w2[i + stride] : zero;
    double2 ya = have0 ? p2[i] : zero, yb = have1 ? p2[i + stride] : zero;
    double2 xa = zero, xb = zero;
    if (p1) {
        xa = have0 ? p1[i] : zero;
        xb = have1 ? p1[i + stride] : zero;
    }

    const double s2 = 1.0 / sqrt(*a.sq2);
    double c1 = 0.0, h1 = 0.0, h2;
    if (a.u1) {
        const double s1 = 1.0 / sqrt(*a.sq1);
        double sa, sb, g;
        finish_sum3(a.a, a.b, a.g, sa, sb, g, red);
        h1 = sa * s1;
        c1 = h1 * s1;
        h2 = (sb - c1 * g) * s2;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *a.h1_out = h1;
            if (a.g_final) *a.g_final = g;
        }
    } else {
        h2 = finish_sum(a.b, red) * s2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.h2_out = h2;
    const double c2 = h2 * s2;
    double asq = 0.0, ag = 0.0;
    for (;;) {
        wa.x -= c1 * xa.x; wa.y -= c1 * xa.y;
        wb.x -= c1 * xb.x; wb.y -= c1 * xb.y;
        wa.x -= c2 * ya.x; wa.y -= c2 * ya.y;
        wb.x -= c2 * yb.x; wb.y -= c2 * yb.y;
        if (have0) w2[i] = wa;
        if (i + stride < a.npairs) w2[i + stride] = wb;
        asq += wa.x * wa.x; asq += wa.y * wa.y;
        asq += wb.x * wb.x; asq += wb.y * wb.y;
        ag += wa.x * ya.x; ag += wa.y * ya.y;
        ag += wb.x * yb.x; ag += wb.y * yb.y;
        i += 2 * stride;
        if (i >= a.npairs) break;
        const bool h1b = i + stride < a.npairs;
        wa = w2[i];
        ya = p2[i];
        wb = h1b ? w2[i + stride] : zero;
        yb = h1b ? p2[i + stride] : zero;
        if (p1) {
            xa = p1[i];
            xb = h1b ? p1[i + stride] : zero;
        }
    }
    double dummy = 0.0;
    block_allreduce_sum3(asq, ag, dummy, red);
    if (threadIdx.x == 0) {
        a.partial_sq[blockIdx.x] = asq;
        a.partial_g[blockIdx.x] = ag;
    }
}

void launch_ortho2(int grid, const Ortho2Args &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_ortho2, dim3(grid), dim3(kBlock), 0, st, a);
}

__global__ __launch_bounds__(kBlock) void k_pass_reset(double *__restrict__ H, int count, int *__restrict__ flag)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < count) H[i] = 0.0;
    if (i == 0) *flag = 0;
}

void launch_pass_reset(double *H, int count, int *flag, hipStream_t st)
{
    hipLaunchKernelGGL(k_pass_reset, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, st, H, count, flag);
}

// ------------------------------------------------------------------------------------------------
// Format 7 (round 4): the PENCIL product of a matrix-free box (single-factor fast form, one rank).
//
// Format 4 takes the rows 128 at a time in (some) order of the row index and gathers every entry's source element of x from
// memory.  On the 6-species boxes (BASELINE config 5: 22^6) that is where the time goes: the two entries of the SLOWEST species
// reach +-22^5 rows = 41 MB, no cache holds anything that far, and x crosses the fabric ~6 times per product
// (profiles/r04_pmc_trip_order_tiles.txt: 5.2-5.6 GB read for a 0.9 GB vector, at 0.73-0.80 of the HBM peak - the kernel is bound
// by that traffic; ordering the trips in small tiles moves it by 7 %).
// Here a wavefront owns the 128 rows of a "base trip" of ONE plane of the slowest species and walks the planes: rows
// lo + p * plane_rows, p = 0 .. planes - 1.  Along that walk
//   * the coordinates of the other species, their shares of DIAG, every entry's table value a_k(x - nu) and its validity do not
//     change: they are worked out once per pencil (coordinate decoding, 10 table look-ups) instead of once per 128 rows;
//   * the source elements of the slowest species' own entries are the lane's OWN rows of the previous / next plane: the pair
//     the lane loaded one step ago and the pair it loads one step ahead - registers, no memory access at all;
//   * the slowest species' coordinate is the step number: its table values and valid bits are wave-uniform.
// Per step and lane: 10 gathers + 1 streaming pair load and ~40 vector instructions, against 13 gathers and ~300; and what
// crosses the fabric is x once plus whatever the second-slowest stride misses in the L2 (the base trips are taken in small tiles,
// box_tile_order, so that those neighbours are in flight on the same XCD at the same step).
// Every row is the same sequence of fused multiply-adds over the same operands as in format 4: products are bit-identical
// (tests/test_gpu_box.py, tests/test_gpu_configs.py).  Eligibility (kfsp_set_matrix_box): the slowest species' entries reach
// exactly one plane, no other entry moves that species, planes have an even number of rows, enough base trips to fill the chip.
// SIMPLE: no entry of another species moves the slowest one (every birth-death / mass-action box whose reactions change one
// species each): the entries served from memory keep their validity along the pencil.  Otherwise their valid bit of the
// slowest species (wave-uniform, from its table at the step's population) is applied step by step.
template <int MODE, int NS, int PER, bool SIMPLE>
__global__ __launch_bounds__(kBlock) void k_spmv_pencil(SpmvArgs a, int64_t plane_rows, int planes, int64_t base_trips,
                                                        const int32_t *__restrict__ order)
{
    constexpr int L = NS - 1;                       // the slowest species
    constexpr int NE = L * PER;                     // entries whose source comes from memory
    __shared__ double red[12];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE != 0) {
        if (*a.brk_flag) return;
    }
    for (int i = threadIdx.x; i < a.B.ntab; i += kBlock) box_lds[i] = a.box_tab[i];
    __syncthreads();
    BoxRegs<NS, PER> R;
    box_load(a.box_fast, R);
    double s = 1.0;
    if (MODE != 0) {
        const double S = finish_sum(a.sq, red);
        const double nrm = sqrt(S);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.sq_final) *a.sq_final = S;
            if (a.h_sub) *a.h_sub = nrm;
        }
        if (a.break_tol >= 0.0 && !(nrm > a.break_tol)) {   // happy breakdown :249
            if (blockIdx.x == 0 && threadIdx.x == 0) *a.brk_flag = 1;
            return;
        }
        s = 1.0 / nrm;
    }
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int bx = gridDim.x >> 3;
    const int64_t cpx = (base_trips + 7) >> 3;
    const int64_t cbeg = (int64_t)xcd * cpx;
    const int64_t cend = (cbeg + cpx < base_trips) ? cbeg + cpx : base_trips;
    const int64_t cstep = (int64_t)bx * 4;
    const lds_bytes_t lds = (lds_bytes_t)box_lds;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int64_t plane_bytes = plane_rows * 8;
    double acc = 0.0, acc2 = 0.0;
    for (int64_t c = cbeg + (int64_t)slot * 4 + wave; c < cend; c += cstep) {
        const int64_t ct = order ? (int64_t)__builtin_amdgcn_readfirstlane(order[c]) : c;
        const int64_t r0 = (ct << 7) + 2 * lane;                  // the lane's rows r0, r0 + 1 of every plane
        if (r0 < plane_rows) {                                     // (plane_rows is even: both rows or neither)
            // ---- what does not change along the pencil
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
            uint32_t q = (uint32_t)r0;
#define KFSP_PEN_DEC(S, VAR)                                             \
    if (L > S + 1) {                                                     \
        const int d = R.dims[L > S ? S : 0];                             \
        uint32_t t = (uint32_t)((double)q * R.inv_dim[L > S ? S : 0]);   \
        int r = (int)(q - t * (uint32_t)d);                              \
        const int lo_ = r < 0, hi_ = r >= d;                             \
        t = t - lo_ + hi_;                                               \
        r = r + (lo_ ? d : 0) - (hi_ ? d : 0);                           \
        VAR = r;                                                         \
        q = t;                                                           \
    } else if (L == S + 1) {                                             \
        VAR = (int)q;                                                    \
    }
            KFSP_PEN_DEC(0, c0)
            KFSP_PEN_DEC(1, c1)
            KFSP_PEN_DEC(2, c2)
            KFSP_PEN_DEC(3, c3)
            KFSP_PEN_DEC(4, c4)
#undef KFSP_PEN_DEC
            int b0 = c0, b1 = c1, b2 = c2, b3 = c3, b4 = c4, carry = 1;
#define KFSP_PEN_INC(S, VAR)                                             \
    if (L > S) {                                                         \
        const int v = VAR + carry;                                       \
        const int wrap = (L > S + 1) && v >= R.dims[L > S ? S : 0];      \
        VAR = wrap ? 0 : v;                                              \
        carry = wrap;                                                    \
    }
            KFSP_PEN_INC(0, b0)
            KFSP_PEN_INC(1, b1)
            KFSP_PEN_INC(2, b2)
            KFSP_PEN_INC(3, b3)
            KFSP_PEN_INC(4, b4)
#undef KFSP_PEN_INC
            // shares of DIAG and valid bits of the species below the slowest, in species order (the order format 4 adds them in)
            double dsa0, dsb0;
            unsigned va0, vb0;
            {
                const lds_bytes_t fa = lds + R.df8[0] + 16 * c0, fb = lds + R.df8[0] + 16 * b0;
                dsa0 = *(const __attribute__((address_space(3))) double *)fa;
                va0 = *(const __attribute__((address_space(3))) unsigned *)(fa + 8);
                dsb0 = *(const __attribute__((address_space(3))) double *)fb;
                vb0 = *(const __attribute__((address_space(3))) unsigned *)(fb + 8);
            }
#define KFSP_PEN_DF(S, CA, CB)                                                                       \
    if (L > S) {                                                                                     \
        const lds_bytes_t fa = lds + R.df8[L > S ? S : 0] + 16 * CA, fb = lds + R.df8[L > S ? S : 0] + 16 * CB; \
        dsa0 += *(const __attribute__((address_space(3))) double *)fa;                               \
        va0 &= *(const __attribute__((address_space(3))) unsigned *)(fa + 8);                        \
        dsb0 += *(const __attribute__((address_space(3))) double *)fb;                               \
        vb0 &= *(const __attribute__((address_space(3))) unsigned *)(fb + 8);                        \
    }
            KFSP_PEN_DF(1, c1, b1)
            KFSP_PEN_DF(2, c2, b2)
            KFSP_PEN_DF(3, c3, b3)
            KFSP_PEN_DF(4, c4, b4)
#undef KFSP_PEN_DF
            // table values of the entries served from memory (0.0 where the source state lies outside the box) and where their
            // source pair sits relative to the lane's own pair
            double ta[NE > 0 ? NE : 1], tb[NE > 0 ? NE : 1];
            unsigned off[NE > 0 ? NE : 1];
            const unsigned voff = (unsigned)(16 * lane + R.bias8);
#define KFSP_PEN_ENT(S, CA, CB)                                                                      \
    if (L > S) {                                                                                     \
        _Pragma("unroll") for (int j = 0; j < PER; ++j) {                                            \
            constexpr int e = (L > S ? S : 0) * PER;                                                 \
            const int ma = __builtin_amdgcn_sbfe(va0, e + j, 1), mb = __builtin_amdgcn_sbfe(vb0, e + j, 1); \
            const unsigned ata = lds0 + (unsigned)(8 * CA + R.koff8[L > S ? S : 0][j]);              \
            const unsigned atb = lds0 + (unsigned)(8 * CB + R.koff8[L > S ? S : 0][j]);              \
            ta[e + j] = *(const __attribute__((address_space(3))) double *)(size_t)((ma & ata) | (~ma & lds0)); \
            tb[e + j] = *(const __attribute__((address_space(3))) double *)(size_t)((mb & atb) | (~mb & lds0)); \
            off[e + j] = (ma | mb) ? voff + (unsigned)R.delta8[L > S ? S : 0][j] : voff;             \
        }                                                                                            \
    }
            KFSP_PEN_ENT(0, c0, b0)
            KFSP_PEN_ENT(1, c1, b1)
            KFSP_PEN_ENT(2, c2, b2)
            KFSP_PEN_ENT(3, c3, b3)
            KFSP_PEN_ENT(4, c4, b4)
#undef KFSP_PEN_ENT
            // ---- the walk over the planes
            uint64_t xb = reinterpret_cast<uint64_t>(a.xg + (ct << 7)) - (uint64_t)(int64_t)R.bias8;
            box_pair_t xprev = {0.0, 0.0}, xcur, xnext = {0.0, 0.0};
            {
                const global_bytes_t xw = (global_bytes_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb) |
                                                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32);
                xcur = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + voff);
            }
            int64_t row = r0;
            for (int p = 0; p < planes; ++p) {
                const global_bytes_t xw = (global_bytes_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb) |
                                                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32);
                if (p + 1 < planes) xnext = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + plane_bytes + voff);
                // the slowest species at population p: its share of DIAG and its valid bits are the same for every lane
                const lds_bytes_t fl = lds + R.df8[L] + 16 * p;
                const double dsl = *(const __attribute__((address_space(3))) double *)fl;
                const unsigned vl = *(const __attribute__((address_space(3))) unsigned *)(fl + 8);
                double acca = 0.0, accb = 0.0;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (SIMPLE) {
                        const box_pair_t xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + off[e]);
                        acca += ta[e] * xv.x;
                        accb += tb[e] * xv.y;
                    } else {
                        // (the entry moves the slowest species too: whether its source plane exists is this step's bit)
                        const bool here = (vl >> e) & 1u;
                        const box_pair_t xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + (here ? off[e] : voff));
                        acca += (here ? ta[e] : 0.0) * xv.x;
                        accb += (here ? tb[e] : 0.0) * xv.y;
                    }
                }
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int ma = __builtin_amdgcn_sbfe(va0 & vl, L * PER + j, 1), mb = __builtin_amdgcn_sbfe(vb0 & vl, L * PER + j, 1);
                    const unsigned at = lds0 + (unsigned)(8 * p + R.koff8[L][j]);
                    const double a1a = *(const __attribute__((address_space(3))) double *)(size_t)((ma & at) | (~ma & lds0));
                    const double a1b = *(const __attribute__((address_space(3))) double *)(size_t)((mb & at) | (~mb & lds0));
                    // The source pair.  An entry that only moves the slowest species by one: the lane's own rows one plane
                    // down / up - the pair it loaded a step ago / loads a step ahead (format 4 gathers x[g + delta] from
                    // memory).  Any other entry of this species (a propensity that depends on it but moves others): from memory,
                    // as in format 4.  Where neither row has the entry: the own pair (format 4 reads it and multiplies by 0.0).
                    const int d = R.delta8[L][j];
                    box_pair_t xv;
                    if ((int64_t)d == plane_bytes) xv = xnext;
                    else if ((int64_t)d == -plane_bytes) xv = xprev;
                    else xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + ((ma | mb) ? voff + (unsigned)d : voff));
                    if (!(ma | mb)) xv = xcur;
                    acca += a1a * xv.x;
                    accb += a1b * xv.y;
                }
                d2 sum;
                sum.x = acca - (dsa0 + dsl) * xcur.x;
                sum.y = accb - (dsb0 + dsl) * xcur.y;
                if (MODE != 0) {
                    sum.x *= s;
                    sum.y *= s;
                }
                *reinterpret_cast<d2 *>(a.y + row) = sum;
                if (MODE == 1 || MODE == 3) {
                    const d2 u = *reinterpret_cast<const d2 *>(a.udot + row);
                    acc += u.x * sum.x;
                    acc += u.y * sum.y;
                }
                if (MODE == 2) {
                    acc += sum.x * sum.x;
                    acc += sum.y * sum.y;
                }
                if (MODE == 3) {
                    const d2 u = *reinterpret_cast<const d2 *>(a.udot2 + row);
                    acc2 += u.x * sum.x;
                    acc2 += u.y * sum.y;
                }
                xprev = xcur;
                xcur = xnext;
                xb += (uint64_t)plane_bytes;
                row += plane_rows;
            }
        }
    }
    if (MODE == 3) {
        double dummy = 0.0;
        block_allreduce_sum3(acc, acc2, dummy, red);
        if (threadIdx.x == 0) {
            a.partial[blockIdx.x] = acc;
            a.partial2[blockIdx.x] = acc2;
        }
    } else if (MODE != 0) {
        const double t = block_allreduce_sum(acc, red);
        if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Format 8 (round 4): pencils in SLABS.  Format 7 removed the slowest species' streams; what was left of the traffic on the
// 6-species boxes was the SECOND-slowest species (22^4 rows = 1.9 MB away: the L2 does not hold it across a pencil's 41 MB
// steps).  Here a WORKGROUP owns the same 128 rows (of the index below the second-slowest stride) in W consecutive lines of
// that species - wavefront w walks the pencil of line g W + w - and all its wavefronts take the planes in step, one
// workgroup barrier per step: the second-slowest species' own +-1 entries are then the pair wavefront w -+ 1 holds for the
// same plane, handed over through LDS (16 bytes per lane, two buffers).  Only the first and the last line of a workgroup
// gather such a neighbour from memory (22 lines in two workgroups of 11: 2 of 44 neighbour pairs).  Everything else is
// format 7: invariants hoisted out of the walk, the slowest species' entries from the lane's own previous / next pair.
// Same fused multiply-adds over the same operands: bit-identical products.
template <int MODE, int NS, int PER, bool SIMPLE>
__global__ __launch_bounds__(768) void k_spmv_slab(SpmvArgs a, int64_t plane_rows, int planes, int64_t line_rows, int lines, int groups,
                                                    int64_t lo_trips)
{
    constexpr int L = NS - 1;                       // the slowest species: planes
    constexpr int M = NS - 2;                       // the second slowest: lines
    constexpr int NE = M * PER;                     // entries of the species below them: always gathered from memory
    __shared__ double red[12];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = (int)(blockDim.x >> 6);
    if (MODE != 0) {
        if (*a.brk_flag) return;
    }
    for (int i = threadIdx.x; i < a.B.ntab; i += blockDim.x) box_lds[i] = a.box_tab[i];
    __syncthreads();
    BoxRegs<NS, PER> R;
    box_load(a.box_fast, R);
    // the exchange area behind the table image: two buffers of W x 64 pairs
    box_pair_t *xchg = reinterpret_cast<box_pair_t *>(box_lds + ((a.B.ntab + 1) & ~1));
    double s = 1.0;
    if (MODE != 0) {
        // (finish_sum for a workgroup of W wavefronts: the first four sum the partials exactly as the 256-thread kernels do)
        double t = 0.0;
        if (threadIdx.x < kBlock)
            for (int i = threadIdx.x; i < a.sq.n; i += kBlock) t += a.sq.p[i];
        t = wave_allreduce_sum(t);
        if (lane == 0 && wave < 4) red[wave] = t;
        __syncthreads();
        const double S = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        const double nrm = sqrt(S);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.sq_final) *a.sq_final = S;
            if (a.h_sub) *a.h_sub = nrm;
        }
        if (a.break_tol >= 0.0 && !(nrm > a.break_tol)) {   // happy breakdown :249
            if (blockIdx.x == 0 && threadIdx.x == 0) *a.brk_flag = 1;
            return;
        }
        s = 1.0 / nrm;
    }
    const lds_bytes_t lds = (lds_bytes_t)box_lds;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int64_t plane_bytes = plane_rows * 8, line_bytes = line_rows * 8;
    const int64_t total = lo_trips * groups;
    double acc = 0.0, acc2 = 0.0;
    for (int64_t b = blockIdx.x; b < total; b += gridDim.x) {
        const int g = (int)(b % groups);
        const int64_t ct = b / groups;
        const int sl = g * W + wave;                               // this wavefront's line
        const int64_t r0 = (ct << 7) + 2 * lane;                   // the lane's rows r0, r0 + 1 of the line (line_rows is even)
        const bool live = sl < lines && r0 < line_rows;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        double dsa0 = 0.0, dsb0 = 0.0;
        unsigned va0 = 0u, vb0 = 0u;
        double ta[NE > 0 ? NE : 1], tb[NE > 0 ? NE : 1];
        unsigned off[NE > 0 ? NE : 1];
        const unsigned voff = (unsigned)(16 * lane + R.bias8);
#pragma unroll
        for (int e = 0; e < (NE > 0 ? NE : 1); ++e) {
            ta[e] = tb[e] = 0.0;
            off[e] = voff;
        }
        if (live) {
            uint32_t q = (uint32_t)r0;
#define KFSP_SLB_DEC(S, VAR)                                             \
    if (M > S + 1) {                                                     \
        const int d = R.dims[M > S ? S : 0];                             \
        uint32_t t = (uint32_t)((double)q * R.inv_dim[M > S ? S : 0]);   \
        int r = (int)(q - t * (uint32_t)d);                              \
        const int lo_ = r < 0, hi_ = r >= d;                             \
        t = t - lo_ + hi_;                                               \
        r = r + (lo_ ? d : 0) - (hi_ ? d : 0);                           \
        VAR = r;                                                         \
        q = t;                                                           \
    } else if (M == S + 1) {                                             \
        VAR = (int)q;                                                    \
    }
            KFSP_SLB_DEC(0, c0)
            KFSP_SLB_DEC(1, c1)
            KFSP_SLB_DEC(2, c2)
            KFSP_SLB_DEC(3, c3)
#undef KFSP_SLB_DEC
            int b0 = c0, b1 = c1, b2 = c2, b3 = c3, carry = 1;
#define KFSP_SLB_INC(S, VAR)                                             \
    if (M > S) {                                                         \
        const int v = VAR + carry;                                       \
        const int wrap = (M > S + 1) && v >= R.dims[M > S ? S : 0];      \
        VAR = wrap ? 0 : v;                                              \
        carry = wrap;                                                    \
    }
            KFSP_SLB_INC(0, b0)
            KFSP_SLB_INC(1, b1)
            KFSP_SLB_INC(2, b2)
            KFSP_SLB_INC(3, b3)
#undef KFSP_SLB_INC
            {
                const lds_bytes_t fa = lds + R.df8[0] + 16 * c0, fb = lds + R.df8[0] + 16 * b0;
                dsa0 = *(const __attribute__((address_space(3))) double *)fa;
                va0 = *(const __attribute__((address_space(3))) unsigned *)(fa + 8);
                dsb0 = *(const __attribute__((address_space(3))) double *)fb;
                vb0 = *(const __attribute__((address_space(3))) unsigned *)(fb + 8);
            }
#define KFSP_SLB_DF(S, CA, CB)                                                                       \
    if (M > S) {                                                                                     \
        const lds_bytes_t fa = lds + R.df8[M > S ? S : 0] + 16 * CA, fb = lds + R.df8[M > S ? S : 0] + 16 * CB; \
        dsa0 += *(const __attribute__((address_space(3))) double *)fa;                               \
        va0 &= *(const __attribute__((address_space(3))) unsigned *)(fa + 8);                        \
        dsb0 += *(const __attribute__((address_space(3))) double *)fb;                               \
        vb0 &= *(const __attribute__((address_space(3))) unsigned *)(fb + 8);                        \
    }
            KFSP_SLB_DF(1, c1, b1)
            KFSP_SLB_DF(2, c2, b2)
            KFSP_SLB_DF(3, c3, b3)
#undef KFSP_SLB_DF
            {
                // the second-slowest species at this wavefront's line: uniform
                const lds_bytes_t fm = lds + R.df8[M] + 16 * sl;
                const double dsm = *(const __attribute__((address_space(3))) double *)fm;
                const unsigned vm = *(const __attribute__((address_space(3))) unsigned *)(fm + 8);
                dsa0 += dsm;
                dsb0 += dsm;
                va0 &= vm;
                vb0 &= vm;
            }
#define KFSP_SLB_ENT(S, CA, CB)                                                                      \
    if (M > S) {                                                                                     \
        _Pragma("unroll") for (int j = 0; j < PER; ++j) {                                            \
            constexpr int e = (M > S ? S : 0) * PER;                                                 \
            const int ma = __builtin_amdgcn_sbfe(va0, e + j, 1), mb = __builtin_amdgcn_sbfe(vb0, e + j, 1); \
            const unsigned ata = lds0 + (unsigned)(8 * CA + R.koff8[M > S ? S : 0][j]);              \
            const unsigned atb = lds0 + (unsigned)(8 * CB + R.koff8[M > S ? S : 0][j]);              \
            ta[e + j] = *(const __attribute__((address_space(3))) double *)(size_t)((ma & ata) | (~ma & lds0)); \
            tb[e + j] = *(const __attribute__((address_space(3))) double *)(size_t)((mb & atb) | (~mb & lds0)); \
            off[e + j] = (ma | mb) ? voff + (unsigned)R.delta8[M > S ? S : 0][j] : voff;             \
        }                                                                                            \
    }
            KFSP_SLB_ENT(0, c0, b0)
            KFSP_SLB_ENT(1, c1, b1)
            KFSP_SLB_ENT(2, c2, b2)
            KFSP_SLB_ENT(3, c3, b3)
#undef KFSP_SLB_ENT
        }
        // ---- the walk over the planes, all wavefronts of the workgroup in step
        uint64_t xb = reinterpret_cast<uint64_t>(a.xg + (int64_t)(sl < lines ? sl : 0) * line_rows + (ct << 7)) - (uint64_t)(int64_t)R.bias8;
        box_pair_t xprev = {0.0, 0.0}, xcur = {0.0, 0.0}, xnext = {0.0, 0.0};
        if (live) {
            const global_bytes_t xw = (global_bytes_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb) |
                                                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32);
            xcur = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + voff);
        }
        int64_t row = (int64_t)sl * line_rows + r0;
        for (int p = 0; p < planes; ++p) {
            box_pair_t *buf = xchg + (size_t)(p & 1) * (size_t)W * 64;
            buf[wave * 64 + lane] = xcur;
            __syncthreads();
            if (live) {
                const global_bytes_t xw = (global_bytes_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb) |
                                                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32);
                if (p + 1 < planes) xnext = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + plane_bytes + voff);
                const lds_bytes_t fl = lds + R.df8[L] + 16 * p;
                const double dsl = *(const __attribute__((address_space(3))) double *)fl;
                const unsigned vl = *(const __attribute__((address_space(3))) unsigned *)(fl + 8);
                double acca = 0.0, accb = 0.0;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (SIMPLE) {
                        const box_pair_t xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + off[e]);
                        acca += ta[e] * xv.x;
                        accb += tb[e] * xv.y;
                    } else {
                        const bool here = (vl >> e) & 1u;
                        const box_pair_t xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + (here ? off[e] : voff));
                        acca += (here ? ta[e] : 0.0) * xv.x;
                        accb += (here ? tb[e] : 0.0) * xv.y;
                    }
                }
                // the second-slowest species' entries: its population is this wavefront's line, so the table value is uniform;
                // an entry that moves only this species by one reads the pair of the wavefront one line down / up - from LDS
                // when that line is in this workgroup, else from memory (as format 4 does for every entry)
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int ma = __builtin_amdgcn_sbfe(va0 & vl, M * PER + j, 1), mb = __builtin_amdgcn_sbfe(vb0 & vl, M * PER + j, 1);
                    const unsigned at = lds0 + (unsigned)(8 * sl + R.koff8[M][j]);
                    const double a1a = *(const __attribute__((address_space(3))) double *)(size_t)((ma & at) | (~ma & lds0));
                    const double a1b = *(const __attribute__((address_space(3))) double *)(size_t)((mb & at) | (~mb & lds0));
                    const int d = R.delta8[M][j];
                    box_pair_t xv;
                    if ((int64_t)d == line_bytes && wave + 1 < W) xv = buf[(wave + 1) * 64 + lane];
                    else if ((int64_t)d == -line_bytes && wave > 0) xv = buf[(wave - 1) * 64 + lane];
                    else xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + ((ma | mb) ? voff + (unsigned)d : voff));
                    if (!(ma | mb)) xv = xcur;
                    acca += a1a * xv.x;
                    accb += a1b * xv.y;
                }
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int ma = __builtin_amdgcn_sbfe(va0 & vl, L * PER + j, 1), mb = __builtin_amdgcn_sbfe(vb0 & vl, L * PER + j, 1);
                    const unsigned at = lds0 + (unsigned)(8 * p + R.koff8[L][j]);
                    const double a1a = *(const __attribute__((address_space(3))) double *)(size_t)((ma & at) | (~ma & lds0));
                    const double a1b = *(const __attribute__((address_space(3))) double *)(size_t)((mb & at) | (~mb & lds0));
                    const int d = R.delta8[L][j];
                    box_pair_t xv;
                    if ((int64_t)d == plane_bytes) xv = xnext;
                    else if ((int64_t)d == -plane_bytes) xv = xprev;
                    else xv = *(const __attribute__((address_space(1), aligned(8))) box_pair_t *)(xw + ((ma | mb) ? voff + (unsigned)d : voff));
                    if (!(ma | mb)) xv = xcur;
                    acca += a1a * xv.x;
                    accb += a1b * xv.y;
                }
                d2 sum;
                sum.x = acca - (dsa0 + dsl) * xcur.x;
                sum.y = accb - (dsb0 + dsl) * xcur.y;
                if (MODE != 0) {
                    sum.x *= s;
                    sum.y *= s;
                }
                *reinterpret_cast<d2 *>(a.y + row) = sum;
                if (MODE == 1 || MODE == 3) {
                    const d2 u = *reinterpret_cast<const d2 *>(a.udot + row);
                    acc += u.x * sum.x;
                    acc += u.y * sum.y;
                }
                if (MODE == 2) {
                    acc += sum.x * sum.x;
                    acc += sum.y * sum.y;
                }
                if (MODE == 3) {
                    const d2 u = *reinterpret_cast<const d2 *>(a.udot2 + row);
                    acc2 += u.x * sum.x;
                    acc2 += u.y * sum.y;
                }
                xprev = xcur;
                xcur = xnext;
            }
            xb += (uint64_t)plane_bytes;
            row += plane_rows;
        }
        __syncthreads();                        // (the next slab's first step writes the buffer the last step may still be read from)
    }
    if (MODE != 0) {
        // W wavefront sums, added in wavefront order by every thread: one partial per workgroup
        __shared__ double wred[2 * 16];   // (W <= 12)
        acc = wave_allreduce_sum(acc);
        acc2 = wave_allreduce_sum(acc2);
        if (lane == 0) {
            wred[wave] = acc;
            wred[16 + wave] = acc2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0, t2 = 0.0;
            for (int w = 0; w < W; ++w) {
                t += wred[w];
                t2 += wred[16 + w];
            }
            a.partial[blockIdx.x] = t;
            if (MODE == 3) a.partial2[blockIdx.x] = t2;
        }
    }
}

template <int NS, int NE, bool SIMPLE>
static void launch_slab_mode(int mode, dim3 g, dim3 b, const SpmvArgs &a, hipStream_t st, size_t lds, int64_t plane_rows, int planes,
                             int64_t line_rows, int lines, int groups, int64_t lo_trips)
{
    if (mode == 0) hipLaunchKernelGGL((k_spmv_slab<0, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, line_rows, lines, groups, lo_trips);
    else if (mode == 1) hipLaunchKernelGGL((k_spmv_slab<1, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, line_rows, lines, groups, lo_trips);
    else if (mode == 2) hipLaunchKernelGGL((k_spmv_slab<2, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, line_rows, lines, groups, lo_trips);
    else hipLaunchKernelGGL((k_spmv_slab<3, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, line_rows, lines, groups, lo_trips);
}

// format 8: waves = wavefronts per workgroup (<= 12: three per SIMD, 170 vector registers each); lds_bytes = table image (rounded to 16 B) + 2 x waves x 1 KB
void launch_spmv_slab(int mode, int grid, int waves, const SpmvArgs &a, hipStream_t st, size_t lds_bytes, int64_t plane_rows, int planes,
                      int64_t line_rows, int lines, int groups, int64_t lo_trips, bool simple)
{
    dim3 g(grid), b(64 * waves);
    if (simple) {
        switch (a.B.pad) {
        case 3 * 16 + 2: launch_slab_mode<3, 2, true>(mode, g, b, a, st, lds_bytes, plane_rows, planes, line_rows, lines, groups, lo_trips); break;
        case 6 * 16 + 2: launch_slab_mode<6, 2, true>(mode, g, b, a, st, lds_bytes, plane_rows, planes, line_rows, lines, groups, lo_trips); break;
        default: launch_slab_mode<6, 4, true>(mode, g, b, a, st, lds_bytes, plane_rows, planes, line_rows, lines, groups, lo_trips); break;
        }
        return;
    }
    switch (a.B.pad) {
    case 3 * 16 + 2: launch_slab_mode<3, 2, false>(mode, g, b, a, st, lds_bytes, plane_rows, planes, line_rows, lines, groups, lo_trips); break;
    case 6 * 16 + 2: launch_slab_mode<6, 2, false>(mode, g, b, a, st, lds_bytes, plane_rows, planes, line_rows, lines, groups, lo_trips); break;
    default: launch_slab_mode<6, 4, false>(mode, g, b, a, st, lds_bytes, plane_rows, planes, line_rows, lines, groups, lo_trips); break;
    }
}

template <int NS, int NE, bool SIMPLE>
static void launch_pencil_mode(int mode, dim3 g, dim3 b, const SpmvArgs &a, hipStream_t st, size_t lds, int64_t plane_rows, int planes,
                               int64_t base_trips, const int32_t *order)
{
    if (mode == 0) hipLaunchKernelGGL((k_spmv_pencil<0, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, base_trips, order);
    else if (mode == 1) hipLaunchKernelGGL((k_spmv_pencil<1, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, base_trips, order);
    else if (mode == 2) hipLaunchKernelGGL((k_spmv_pencil<2, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, base_trips, order);
    else hipLaunchKernelGGL((k_spmv_pencil<3, NS, NE, SIMPLE>), g, b, lds, st, a, plane_rows, planes, base_trips, order);
}

// format 7 (the instantiations of format 4 whose species count is the model's: 3 and 6 species with two slots, 6 with four)
void launch_spmv_pencil(int mode, int grid, const SpmvArgs &a, hipStream_t st, size_t lds_bytes, int64_t plane_rows, int planes,
                        int64_t base_trips, const int32_t *order, bool simple)
{
    dim3 g(grid), b(kBlock);
    if (simple) {
        switch (a.B.pad) {
        case 3 * 16 + 2: launch_pencil_mode<3, 2, true>(mode, g, b, a, st, lds_bytes, plane_rows, planes, base_trips, order); break;
        case 6 * 16 + 2: launch_pencil_mode<6, 2, true>(mode, g, b, a, st, lds_bytes, plane_rows, planes, base_trips, order); break;
        default: launch_pencil_mode<6, 4, true>(mode, g, b, a, st, lds_bytes, plane_rows, planes, base_trips, order); break;
        }
        return;
    }
    switch (a.B.pad) {
    case 3 * 16 + 2: launch_pencil_mode<3, 2, false>(mode, g, b, a, st, lds_bytes, plane_rows, planes, base_trips, order); break;
    case 6 * 16 + 2: launch_pencil_mode<6, 2, false>(mode, g, b, a, st, lds_bytes, plane_rows, planes, base_trips, order); break;
    default: launch_pencil_mode<6, 4, false>(mode, g, b, a, st, lds_bytes, plane_rows, planes, base_trips, order); break;
    }
}

// ----------------------------------------------- small-N Arnoldi pass, one launch
// One workgroup of 1024 lanes owns every row (<= 4 per lane).  Columns are
// separated by workgroup barriers instead of kernel boundaries, the current
// source vector lives in LDS (32 KB: the x gathers of the product never leave
// the CU), each lane keeps its own rows of the new column in registers between
// the product and the update, and scalars never leave registers.  Per column
// the only global round trip left is the stream of generator entries.  H, the
// squared norms and the breakdown flag are written exactly as the multi-launch
// path writes them, so the host side is unchanged.
constexpr int kSmallBlock = 1024;
constexpr int kSmallTrips = kSmallRows / kSmallBlock;      // rows per lane

// Two block sums with ONE barrier: the caller alternates between the two halves of
// red (64 doubles), so a half is rewritten only after another barrier has separated
// the writers from the last readers.
__device__ __forceinline__ void small_reduce2(double &a, double &b, double *red_half)
{
    a = wave_allreduce_sum(a);
    b = wave_allreduce_sum(b);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red_half[wave] = a;
        red_half[16 + wave] = b;
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < kSmallBlock / 64; ++w) {
        sa += red_half[w];
        sb += red_half[16 + w];
    }
    a = sa;
    b = sb;
}

__device__ __forceinline__ double row_dia1(const DiaDev &D, const double *x, int64_t r)
{
    const int64_t last = D.n - 1;
    double sum = -D.diag[r] * x[r < last ? r : last];
    for (int d = 0; d < D.nd; ++d) {
        int64_t i = r + D.delta[d];
        i = i < 0 ? 0 : (i > last ? last : i);
        sum += D.val[(int64_t)d * D.ld + r] * x[i];
    }
    return sum;
}

// row t of this lane: SELL lanes own (chunk = wave + 16 t, lane), DIA lanes own tid + 1024 t
template <bool DIA>
__device__ __forceinline__ int64_t small_row(int t)
{
    if (DIA) return (int64_t)threadIdx.x + (int64_t)t * kSmallBlock;
    return (((int64_t)(threadIdx.x >> 6) + (int64_t)t * (kSmallBlock / 64)) << 6) + (threadIdx.x & 63);
}

// SELL row with the chunk's offset and width already in registers (they do not
// change from column to column, so the dependent off[] load is paid once per pass)
__device__ __forceinline__ double row_sell_pre(const SellDev &A, const double *xs, int64_t r, int64_t off, int w,
                                               double dg)
{
    const int lane = (int)(r & 63);
    const int32_t *cp = A.col + off + 2 * lane;
    const double *vp = A.val + off + 2 * lane;
    double sum = -dg * xs[r];
    for (int k = 0; k < (w >> 1); ++k) {                  // slot pairs (sell_pos)
        const i2 c0 = *reinterpret_cast<const i2 *>(cp + k * 128);
        const d2 v0 = *reinterpret_cast<const d2 *>(vp + k * 128);
        sum += v0.x * xs[c0.x];
        sum += v0.y * xs[c0.y];
    }
    if (w & 1) sum += A.val[off + (w >> 1) * 128 + lane] * xs[A.col[off + (w >> 1) * 128 + lane]];
    return sum;
}

// the same with the chunk's slots copied to LDS (values, and columns as 16-bit numbers:
// there are at most 4096 rows)
__device__ __forceinline__ double row_sell_lds(const double *vs, const unsigned short *cs, double dg,
                                               const double *xs, int64_t r, int64_t off, int w)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const int lane = (int)(r & 63);
    const unsigned short *cp = cs + off + 2 * lane;
    const double *vp = vs + off + 2 * lane;
    double sum = -dg * xs[r];
    for (int k = 0; k < (w >> 1); ++k) {
        const us2 c0 = *reinterpret_cast<const us2 *>(cp + k * 128);
        const d2 v0 = *reinterpret_cast<const d2 *>(vp + k * 128);
        sum += v0.x * xs[c0.x];
        sum += v0.y * xs[c0.y];
    }
    if (w & 1) sum += vs[off + (w >> 1) * 128 + lane] * xs[cs[off + (w >> 1) * 128 + lane]];
    return sum;
}

// FMT: 0 SELL from global memory, 1 banded, 2 SELL from LDS
template <int FMT>
__device__ __forceinline__ double small_product_row(const SmallArnoldiArgs &a, const double *xs, const double *vs,
                                                    const unsigned short *cs, int64_t r, int64_t off, int w, double dg)
{
    if (FMT == 1) return row_dia1(a.D, xs, r);
    if (FMT == 2) return row_sell_lds(vs, cs, dg, xs, r, off, w);
    return row_sell_pre(a.A, xs, r, off, w, dg);
}

template <int FMT>
__global__ __launch_bounds__(kSmallBlock) void k_arnoldi_small(SmallArnoldiArgs a)
{
    constexpr bool DIA = FMT == 1;
    // dynamic LDS: the source column u_j [nact doubles]; FMT 2 also the generator's slots
    // [slots doubles | slots 16-bit columns]
    extern __shared__ double dyn_lds[];
    double *xs = dyn_lds;
    double *vs = dyn_lds + a.nact;
    unsigned short *cs = reinterpret_cast<unsigned short *>(vs + (FMT == 2 ? a.slots : 0));
    __shared__ double red[64];
    int rb = 0;                                        // which half of red the next reduction uses
    const int tid = threadIdx.x;
    if (FMT == 2) {
        for (int64_t i = tid; i < a.slots; i += kSmallBlock) {
            vs[i] = a.A.val[i];
            cs[i] = (unsigned short)a.A.col[i];
        }
    }
    double *Hd = a.Hd;
    double S = a.sq[a.jold];          // squared norm of the column about to be multiplied
    double g = a.gfin[a.jold];        // u_jold . u_{jold-1}
    double s1 = a.jold >= 2 ? 1.0 / sqrt(a.sq[a.jold - 1]) : 0.0;   // 1/||u_{j-1}||, carried in a register
    int64_t offs[kSmallTrips];
    int wid[kSmallTrips];
    double dg[kSmallTrips];                            // DIAG of this lane's rows: the same in every column
#pragma unroll
    for (int t = 0; t < kSmallTrips; ++t) {
        const int64_t r = small_row<DIA>(t);
        offs[t] = 0;
        wid[t] = 0;
        dg[t] = 0.0;
        if (!DIA && r < a.nact) {
            offs[t] = a.A.off[r >> 6];
            wid[t] = (int)((a.A.off[(r >> 6) + 1] - offs[t]) >> 6);
            dg[t] = a.A.diag[r];
        }
    }
    {
        const double *src = a.V + (size_t)(a.jold - 1) * a.ldv;
        for (int64_t r = tid; r < a.nact; r += kSmallBlock) xs[r] = src[r];
    }
    __syncthreads();
    bool broke = false;
    // u_{j-1} on this lane's rows: read once for the first column, then carried over (it is
    // the source column of the previous iteration)
    double v1[kSmallTrips];
#pragma unroll
    for (int t = 0; t < kSmallTrips; ++t) {
        const int64_t r = small_row<DIA>(t);
        v1[t] = (a.jold >= 2 && r < a.nact) ? a.V[(size_t)(a.jold - 2) * a.ldv + r] : 0.0;
    }
    for (int j = a.jold; j <= a.m; ++j) {
        const bool u1 = j >= 2;
        double *dst = a.V + (size_t)j * a.ldv;
        const double nrm = sqrt(S);
        if (tid == 0) {
            a.sq[j] = S;
            if (j > a.jold) Hd[(size_t)(j - 2) * kMH + (j - 1)] = nrm;     // H(j,j-1)
        }
        if (j > a.jold && !(nrm > a.break_tol)) {      // happy breakdown :249 (S is the same in every lane)
            broke = true;
            break;
        }
        const double s2 = 1.0 / nrm;
        double y[kSmallTrips];
        double pa = 0.0, pb = 0.0;
#pragma unroll
        for (int t = 0; t < kSmallTrips; ++t) {
            const int64_t r = small_row<DIA>(t);
            y[t] = 0.0;
            if (r < a.nact) {
                y[t] = s2 * small_product_row<FMT>(a, xs, vs, cs, r, offs[t], wid[t], dg[t]);
                pa += v1[t] * y[t];
                pb += xs[r] * y[t];
            }
        }
        small_reduce2(pa, pb, red + rb);                // (also: every lane is past its gathers from xs)
        rb ^= 32;
        double c1 = 0.0, h2;
        if (u1) {
            const double h1 = pa * s1;
            c1 = h1 * s1;
            h2 = (pb - c1 * g) * s2;
            if (tid == 0) {
                Hd[(size_t)(j - 1) * kMH + (j - 2)] = h1;                  // H(j-1,j)
                a.gfin[j] = g;
            }
        } else {
            h2 = pb * s2;
        }
        if (tid == 0) Hd[(size_t)(j - 1) * kMH + (j - 1)] = h2;            // H(j,j)
        const double c2 = h2 * s2;
        double asq = 0.0, ag = 0.0;
#pragma unroll
        for (int t = 0; t < kSmallTrips; ++t) {
            const int64_t r = small_row<DIA>(t);
            if (r < a.nact) {
                const double v2 = xs[r];
                const double w = y[t] - c1 * v1[t] - c2 * v2;
                v1[t] = v2;              // next column's u_{j-1}
                dst[r] = w;
                xs[r] = w;               // own rows only; nobody gathers before the barrier below
                asq += w * w;
                ag += w * v2;
            }
        }
        small_reduce2(asq, ag, red + rb);               // its barrier also publishes the new source column
        rb ^= 32;
        S = asq;
        g = ag;
        s1 = s2;
    }
    if (broke) {
        if (tid == 0) *a.brk_flag = 1;
        return;
    }
    // the extra product for AVNORM (:261-263); from column jold when the loop did not run
    const bool looped = a.jold <= a.m;
    const int jl = looped ? a.m + 1 : a.jold;
    double *dst = a.V + (size_t)jl * a.ldv;
    const double nrm = sqrt(S);
    if (tid == 0) {
        a.sq[jl] = S;
        if (looped) {
            Hd[(size_t)(a.m - 1) * kMH + a.m] = nrm;                       // H(m+1,m)
            a.gfin[a.m + 1] = g;
        }
    }
    if (looped && !(nrm > a.break_tol)) {
        if (tid == 0) *a.brk_flag = 1;
        return;
    }
    const double s2 = 1.0 / nrm;
    double av = 0.0, dummy = 0.0;
#pragma unroll
    for (int t = 0; t < kSmallTrips; ++t) {
        const int64_t r = small_row<DIA>(t);
        if (r < a.nact) {
            const double yv = s2 * small_product_row<FMT>(a, xs, vs, cs, r, offs[t], wid[t], dg[t]);
            dst[r] = yv;
            av += yv * yv;
        }
    }
    small_reduce2(av, dummy, red + rb);
    if (tid == 0) {
        Hd[(size_t)kMH * kMH] = av;
        Hd[(size_t)kMH * kMH + 1] = sqrt(av);
    }
}

// bytes of dynamic LDS the generator-in-LDS variant needs
size_t small_lds_bytes(int64_t nact, int64_t slots, bool with_matrix)
{
    return (size_t)nact * 8 + (with_matrix ? (size_t)slots * 10 + 16 : 0);
}

int launch_arnoldi_small(const SmallArnoldiArgs &a, bool dia, int64_t lds_limit, hipStream_t st)
{
    static bool raised = false;
    const size_t want = small_lds_bytes(a.nact, a.slots, true);
    const bool lmat = !dia && a.slots > 0 && (int64_t)want + 512 <= lds_limit;
    if (lmat && !raised) {
        // more than the default 64 KiB of dynamic LDS has to be asked for once per kernel
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_arnoldi_small<2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit - 512);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    if (dia) hipLaunchKernelGGL((k_arnoldi_small<1>), dim3(1), dim3(kSmallBlock), small_lds_bytes(a.nact, 0, false), st, a);
    else if (lmat) hipLaunchKernelGGL((k_arnoldi_small<2>), dim3(1), dim3(kSmallBlock), want, st, a);
    else hipLaunchKernelGGL((k_arnoldi_small<0>), dim3(1), dim3(kSmallBlock), small_lds_bytes(a.nact, 0, false), st, a);
    return 0;
}

// ------------------------------------------------------------------ combine
// w = beta * sum_j y_j s_j u_j ; clamp ; partial = sum |w|
__global__ __launch_bounds__(kBlock) void k_combine(CombineArgs a)
{
    __shared__ double red[4];
    __shared__ double coef[kMH];
    for (int j = threadIdx.x; j < a.mx; j += kBlock)
        coef[j] = a.beta * a.y[j] / sqrt(a.sq[j + 1]);
    __syncthreads();
    double2 *w2 = reinterpret_cast<double2 *>(a.w);
    const int64_t ld2 = a.ldv >> 1;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.npairs; i += (int64_t)gridDim.x * kBlock) {
        const double2 *v = reinterpret_cast<const double2 *>(a.V) + i;
        double sx = 0.0, sy = 0.0;
        int j = 0;
        for (; j + 4 <= a.mx; j += 4) {
            const double2 v0 = v[(int64_t)(j + 0) * ld2], v1 = v[(int64_t)(j + 1) * ld2];
            const double2 v2 = v[(int64_t)(j + 2) * ld2], v3 = v[(int64_t)(j + 3) * ld2];
            sx += coef[j] * v0.x; sy += coef[j] * v0.y;
            sx += coef[j + 1] * v1.x; sy += coef[j + 1] * v1.y;
            sx += coef[j + 2] * v2.x; sy += coef[j + 2] * v2.y;
            sx += coef[j + 3] * v3.x; sy += coef[j + 3] * v3.y;
        }
        for (; j < a.mx; ++j) {
            const double2 vj = v[(int64_t)j * ld2];
            sx += coef[j] * vj.x; sy += coef[j] * vj.y;
        }
        sx = sx < 0.0 ? 0.0 : sx;       // FSP non-negativity :447-449 (NaN stays NaN)
        sy = sy < 0.0 ? 0.0 : sy;
        w2[i] = make_double2(sx, sy);
        acc += fabs(sx);
        acc += fabs(sy);
    }
    const double t = block_allreduce_sum(acc, red);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
}

void launch_combine(int grid, const CombineArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_combine, dim3(grid), dim3(kBlock), 0, st, a);
}

// --------------------------------------------------------- small streaming ops
__global__ __launch_bounds__(kBlock) void k_copy_nrm2(int64_t npairs, const double2 *__restrict__ w,
                                                      double2 *__restrict__ u, double *__restrict__ partial)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock) {
        const double2 v = w[i];
        u[i] = v;
        acc += v.x * v.x;
        acc += v.y * v.y;
    }
    const double t = block_allreduce_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

void launch_copy_nrm2(int grid, int64_t npairs, const double *w, double *u1, double *partial, hipStream_t st)
{
    hipLaunchKernelGGL(k_copy_nrm2, dim3(grid), dim3(kBlock), 0, st, npairs,
                       reinterpret_cast<const double2 *>(w), reinterpret_cast<double2 *>(u1), partial);
}

__global__ __launch_bounds__(kBlock) void k_reduce(int64_t npairs, const double2 *__restrict__ w, int squared,
                                                   double *__restrict__ partial)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock) {
        const double2 v = w[i];
        if (squared) {
            acc += v.x * v.x;
            acc += v.y * v.y;
        } else {
            acc += fabs(v.x);
            acc += fabs(v.y);
        }
    }
    const double t = block_allreduce_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

void launch_reduce(int grid, int64_t npairs, const double *w, int squared, double *partial, hipStream_t st)
{
    hipLaunchKernelGGL(k_reduce, dim3(grid), dim3(kBlock), 0, st, npairs, reinterpret_cast<const double2 *>(w),
                       squared, partial);
}

__global__ __launch_bounds__(kBlock) void k_finalize(Pending p, double *out, double *out_sqrt)
{
    __shared__ double red[4];
    const double t = finish_sum(p, red);
    if (threadIdx.x == 0) {
        if (out) *out = t;
        if (out_sqrt) *out_sqrt = sqrt(t);
    }
}

void launch_finalize(Pending p, double *out, double *out_sqrt, hipStream_t st)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(kBlock), 0, st, p, out, out_sqrt);
}

__global__ __launch_bounds__(kBlock) void k_scale_copy(int64_t npairs, const double2 *__restrict__ u,
                                                       const double *__restrict__ sq_u, double beta,
                                                       double2 *__restrict__ w)
{
    const double c = beta / sqrt(*sq_u);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock) {
        const double2 v = u[i];
        w[i] = make_double2(c * v.x, c * v.y);
    }
}

void launch_scale_copy(int grid, int64_t npairs, const double *u, const double *sq_u, double beta, double *w,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k_scale_copy, dim3(grid), dim3(kBlock), 0, st, npairs,
                       reinterpret_cast<const double2 *>(u), sq_u, beta, reinterpret_cast<double2 *>(w));
}

// ------------------------------------------------------- counter calibration
// Reads n elements of width W bytes per lane (4, 8 or 16) in the SpMV's own
// access shape (one contiguous 64-lane piece per wave instruction) and folds
// them into one value per block, so the traffic is exactly n*W bytes read.
template <class T>
__global__ __launch_bounds__(kBlock) void k_stream_read(int64_t n, const T *__restrict__ p, double *__restrict__ sink)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const T v = __builtin_nontemporal_load(p + i);
        if constexpr (sizeof(T) == 16) acc += v.x + v.y;
        else acc += (double)v;
    }
    if (acc == 0.12345) sink[blockIdx.x] = acc;   // never true for the zero-filled buffer: keeps the loads alive
}

void launch_stream_read(int grid, int elem_bytes, int64_t nbytes, const void *p, double *sink, hipStream_t st)
{
    if (elem_bytes == 4)
        hipLaunchKernelGGL((k_stream_read<int32_t>), dim3(grid), dim3(kBlock), 0, st, nbytes / 4, (const int32_t *)p, sink);
    else if (elem_bytes == 8)
        hipLaunchKernelGGL((k_stream_read<double>), dim3(grid), dim3(kBlock), 0, st, nbytes / 8, (const double *)p, sink);
    else
        hipLaunchKernelGGL((k_stream_read<d2>), dim3(grid), dim3(kBlock), 0, st, nbytes / 16, (const d2 *)p, sink);
}

}  // namespace kfsp
