// SSA paths on the device - the INDEPENDENT-STREAM expansion ("parallel SSA paths with a
// counter-based generator").
//
// The reference's SSA_EXTENDER (src/state_space/StateSpace.f90:550-630) walks one stochastic path of length
// TIMESTEP from every listed state on ONE random stream, each path seeing the states the earlier ones added: the
// walk is sequential by definition and stays on the host.  The host's opt-in variant SSA_EXTENDER_STREAMS
// (krylovfspssa_amd/fortran/kfsp_statespace.f90, KFSP_SSA_STREAMS=1) gives every path a stream of its own - a
// Lehmer generator seeded from (one number per call, index of the seed state) - walks the FSP as it stood at the
// call, passes through unlisted states by evaluating their propensities on the fly, and appends the states met in
// (seed state, position on the path) order, first occurrence first.  THAT variant is what runs here, one lane per
// path, and it is defined so that host and device produce the same bits:
//   * the generator, the mixing of the seed and the choice of the reaction are integer / single IEEE operations
//   * the exponential waiting time uses plog() below instead of the math library's log: a fixed sequence of IEEE
//     operations that the Fortran host (KFSP_PLOG) performs identically
//   * propensities of unlisted states come from the model's program (kfsp_prop_dev.h): host-made tables or exact
//     + - * / code for every shipped model (an expression with library functions of several species may differ
//     from the host's in the last bits, and a path that hinges on it with them - documented in kfsp.h)
// Two passes over the paths (count the records, then write them at their final offsets) make the record list
// deterministic without a sort; duplicates are removed through a hash table with an atomic MIN on the record
// index (the set of survivors does not depend on the order of the insertions).
#include "kfsp_prop_dev.h"
#include "kfsp_hash_dev.h"

#include <hipcub/hipcub.hpp>

#include <climits>
#include <cstring>

#pragma clang fp contract(off)

namespace kfsp {

namespace {

constexpr int kSsaMaxS = 16, kSsaMaxR = 64;

struct SsaDev {
    int ns, nr, n0, max_count, lds, lda;
    double tstep;
    unsigned long long seedmix;
    const int32_t *state;      // [n0][lds]
    const int32_t *adj;        // [n0][lda], the reference's encoding
    const double *off;         // [n0][lda]
    const double *diag;        // [n0]
    const int32_t *nu;         // [nr][ns]
    const int32_t *tab;        // hash table of the listed states: index + 1, 0 = empty
    unsigned tmask;
    // the same table with a tag (the register-resident kernels): (upper half of the state's hash) << 32 | index + 1, 0 = empty
    const unsigned long long *tab64;
    // a filter in front of it (option ssa_filter): one bit per listed state in a map of bmask + 1 words, addressed by the
    // tag half of the hash; a clear bit means "not listed" without touching the table
    const unsigned *bitmap;
    unsigned bmask;
    // the wavefronts of this launch are wavefronts [wave0, wave1) of the walk (a partitioned expansion: this rank's share;
    // else all of them, [0, nwaves))
    long long wave0, wave1;
    PropDev P;
    const int32_t *fast_i;     // descriptors of the register path (kfsp_prop.hip prop_set_program), one per reaction
    const double *fast_d;
};

// index (1-based) of state y among the listed ones, 0 = not listed
__device__ __forceinline__ int lookup_state(const SsaDev &A, const int32_t *y) { return table_find(A.tab, A.tmask, A.state, A.lds, A.ns, y); }

__global__ __launch_bounds__(kBlock) void k_ht_build(int n, int ns, int lds, const int32_t *__restrict__ state, int32_t *tab, unsigned mask)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    unsigned slot = hash_state(state + (int64_t)j * lds, ns) & mask;
    for (;;) {
        if (atomicCAS(&tab[slot], 0, j + 1) == 0) return;
        slot = (slot + 1) & mask;
    }
}

// (the tagged table and its hash: kfsp_hash_dev.h)
__global__ __launch_bounds__(kBlock) void k_ht_build64(int n, int ns, int lds, const int32_t *__restrict__ state, unsigned long long *tab,
                                                       unsigned mask, unsigned *bitmap, unsigned bmask)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const unsigned long long h = hash_state64(state + (int64_t)j * lds, ns);
    const unsigned long long entry = (h & 0xFFFFFFFF00000000ull) | (unsigned long long)(unsigned)(j + 1);
    if (bitmap) {
        const unsigned t = (unsigned)(h >> 32);
        atomicOr(&bitmap[(t >> 5) & bmask], 1u << (t & 31));
    }
    unsigned slot = (unsigned)h & mask;
    for (;;) {
        if (atomicCAS(&tab[slot], 0ull, entry) == 0ull) return;
        slot = (slot + 1) & mask;
    }
}

// log of 0 < x <= 1 by a fixed sequence of IEEE operations (no library call, no contraction): x = m 2^e with m in
// [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^22/23), log x = e ln2 + log m.
// KFSP_PLOG of the Fortran host is the same sequence; ~1e-16 relative.
__device__ __forceinline__ double plog(double x)
{
    int e;
    double m = frexp(x, &e);                      // x = m 2^e, m in [0.5, 1)
    if (m < 0.70710678118654752440) {
        m = m + m;
        e = e - 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.0 / 23.0;
    for (int k = 21; k >= 3; k -= 2) {
        p = p * z;
        p = p + 1.0 / (double)k;
    }
    p = p * z;
    const double two_s = s + s;
    const double r = two_s + two_s * p;
    const double de = (double)e;
    const double hi = de * 6.93147180369123816490e-01;
    const double lo = de * 1.90821492927058770002e-10;
    return hi + (lo + r);
}

constexpr unsigned long long kLcgA = 48271ull, kLcgM = 2147483647ull, kLcgLow = 1073741823ull;

// rs a mod (2^31 - 1) for 1 <= rs < 2^31 - 1 without a division: 2^31 = 1 (mod m), so the product's high part is folded
// onto its low 31 bits, twice (rs a < 2^47), and one conditional subtraction lands in [0, m) - the value MOD() gives
// the Fortran host, in a dozen instructions instead of the 64-bit remainder's sixty (four of these per jump)
__device__ __forceinline__ unsigned long long lcg_next(unsigned long long rs)
{
    const unsigned long long p = rs * kLcgA;
    unsigned long long y = (p & kLcgM) + (p >> 31);
    y = (y & kLcgM) + (y >> 31);
    return y >= kLcgM ? y - kLcgM : y;
}

__device__ __forceinline__ double lcg_uniform(unsigned long long &rs)
{
    const unsigned long long g1 = rs;
    rs = lcg_next(rs);
    const unsigned long long g2 = rs;
    rs = lcg_next(rs);
    return (double)(((g1 << 30) | ((g2 - 1ull) & kLcgLow)) >> 7) * 0x1p-54;
}

// The paths (STREAM_PATH of the host), one lane per path, ONE pass.  Paths differ wildly in length (a handful of jumps
// for most - they fall back onto an earlier seed -, thousands for a few), and a path is a chain of dependent memory
// round trips, so the kernel lasts as long as its longest path: everything a jump needs from the current state (DIAG,
// the OFFDIAG column, the ADJ column) is requested together, one round trip per jump.  A wavefront owns kSsaSeedsPerWave
// consecutive seed states and a lane that finishes its path takes the next of them (a wave-uniform register counter
// handed out by ballot / prefix count), so the whole wavefront reconverges once per jump - which is where the unlisted
// states met are appended to the record list: one atomic add per wavefront and jump, key = (seed state, position on
// the path).  Sorting the records by that key afterwards makes their order independent of how the hardware
// scheduled the paths.
constexpr int kSsaSeedsPerWave = 256;
constexpr int kSsaLdsCode = 512, kSsaLdsDbl = 64;      // a light program's code words / immediates and parameters held in LDS
constexpr int kSsaPosBits = 22;
constexpr int kSsaChunk = 64;                          // record slots a wavefront takes from the list at a time
// (an empty slot keeps the key the list was filled with, all ones: it sorts behind every record)

__global__ __launch_bounds__(kBlock) void k_ssa_walk_any(SsaDev A, unsigned long long *__restrict__ nrec_total, long long cap,
                                                     unsigned long long *__restrict__ keys, int32_t *__restrict__ rec)
{
    const int lane = threadIdx.x & 63;
    const long long wave = A.wave0 + (((long long)blockIdx.x * kBlock + threadIdx.x) >> 6);
    // kSsaSeedsPerWave seeds per wavefront, dealt in blocks of 64 consecutive seed states ROUND ROBIN over the wavefronts:
    // the long paths start from neighbouring states (measured: the LAST states of the list, the FSP's rim - they step
    // outside and walk through unlisted states to the horizon, hundreds of jumps, while a path from an inner state falls
    // back onto an earlier seed after a handful), and a wavefront that owned 256 consecutive ones of them walked four
    // long paths per lane, one after the other
    const long long nwaves = ((long long)A.n0 + kSsaSeedsPerWave - 1) / kSsaSeedsPerWave;
    if (wave >= A.wave1) return;
    int next = 0;                                                  // wave-uniform: seeds handed out so far
    bool active = false;
    int j0 = 0, j = 0, npos = 0;
    bool virt = false;
    unsigned long long rs = 0;
    double tt = 0.0, a0 = 0.0;
    int32_t x[kSsaMaxS], y[kSsaMaxS];
    double pr[kSsaMaxR];
    int32_t aj[kSsaMaxR];
    // the row of a listed state - DIAG, the OFFDIAG column, the ADJ column - is REQUESTED as soon as the state is known
    // (when a path starts, and at the end of the jump that reaches it) and USED after the jump's random numbers and
    // logarithm have been computed: the arithmetic that does not depend on the row runs while the row travels
#define KFSP_SSA_LOAD_ROW()                                           \
    do {                                                              \
        const int64_t row_ = (int64_t)(j - 1) * A.lda;                \
        a0 = A.diag[j - 1];                                           \
        for (int k_ = 0; k_ < A.nr; ++k_) {                           \
            pr[k_] = A.off[row_ + k_];                                \
            aj[k_] = A.adj[row_ + k_];                                \
        }                                                             \
    } while (0)
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle) {
            if (!active) {
                const int t = next + __popcll(idle & ((1ull << lane) - 1ull));
                const long long seed = ((((long long)(t >> 6)) * nwaves + wave) << 6) + (t & 63) + 1;
                if (t < kSsaSeedsPerWave && seed <= A.n0) {
                    j0 = (int)seed;
                    // the path's own stream: a 64-bit mix of (call, seed state) folded into the generator's range
                    rs = (A.seedmix * 2654435761ull) ^ ((unsigned long long)j0 * 40503ull + 12345ull);
                    rs = ((rs ^ (rs >> 29)) & 4294967295ull) * 1181783497ull;
                    rs = 1ull + (((rs ^ (rs >> 32)) & 9223372036854775807ull) % (kLcgM - 1ull));
                    j = j0;
                    virt = false;
                    for (int s = 0; s < A.ns; ++s) x[s] = A.state[(int64_t)(j - 1) * A.lds + s];
                    KFSP_SSA_LOAD_ROW();
                    tt = 0.0;
                    npos = 0;
                    active = true;
                }
            }
            next += __popcll(idle);
        }
        if (!__ballot(active)) break;                              // no path left in this wave
        // ---- one jump of this lane's path (lanes without one wait at the end of this block: the ballots are always
        // executed by the whole wavefront)
        bool ended = active, record = false;
        if (active) do {
            double r1 = lcg_uniform(rs);
            const double r2 = lcg_uniform(rs);
            if (r1 <= 0.0) r1 = 0x1p-54;
            const double wait = -plog(r1);
            if (virt) {
                a0 = 0.0;
                for (int k = 0; k < A.nr; ++k) {
                    pr[k] = prop_eval(A.P, k, x);
                    a0 = a0 + pr[k];
                    aj[k] = 0;
                }
            }
            if (!(a0 > 0.0)) break;                                // absorbing state
            tt = fmin(A.tstep, tt + (wait / a0));
            double acc = pr[0];
            int k = 0;
            const double r2a = fmin(r2 * a0, a0);
            while (acc < r2a && k < A.nr - 1) {
                ++k;
                acc = acc + pr[k];
            }
            bool neg = false;
            for (int s = 0; s < A.ns; ++s) {
                y[s] = x[s] + A.nu[k * A.ns + s];
                neg = neg || y[s] < 0;
            }
            if (neg) break;
            int idx = max(aj[k], 0);
            if (idx == 0) {
                bool legal = true;
                for (int s = 0; s < A.ns; ++s) legal = legal && y[s] <= A.max_count;
                if (!legal) break;
                idx = lookup_state(A, y);
            }
            for (int s = 0; s < A.ns; ++s) x[s] = y[s];
            if (idx > 0) {
                j = idx;
                virt = false;
                if (j < j0) break;                                 // fell back onto an earlier seed
            } else {
                virt = true;
                record = true;                                     // x is an unlisted state: recorded below
            }
            if (!(tt < A.tstep)) break;
            if (!virt) KFSP_SSA_LOAD_ROW();                        // (the next jump's row: on its way while the records are written)
            ended = false;
        } while (false);
        const unsigned long long rmask = __ballot(record);
        if (rmask) {
            unsigned long long first = 0;
            const int leader = __ffsll((long long)rmask) - 1;
            if (lane == leader) {
                first = atomicAdd(nrec_total, (unsigned long long)__popcll(rmask));
                atomicAdd(nrec_total + 1, (unsigned long long)__popcll(rmask));      // (no empty slots in this kernel's list)
            }
            first = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(first >> 32), leader) << 32) |
                    (unsigned)__builtin_amdgcn_readlane((int)first, leader);
            if (record) {
                const unsigned long long slot = first + (unsigned long long)__popcll(rmask & ((1ull << lane) - 1ull));
                if ((long long)slot < cap) {
                    keys[slot] = ((unsigned long long)(unsigned)j0 << kSsaPosBits) | (unsigned long long)min(npos, (1 << kSsaPosBits) - 1);
                    for (int s = 0; s < A.ns; ++s) rec[slot * A.ns + s] = x[s];
                }
                ++npos;
            }
        }
        if (active && ended) active = false;
    }
}

#undef KFSP_SSA_LOAD_ROW

// The same walk for models of at most NS species and NR reactions (8 and 16 cover every shipped model): all the per-path
// state - coordinates, the row of the current state - is indexed with compile-time constants only, so it lives in
// registers (the general kernel above indexes pr[k] / aj[k] / x[s] with run-time values and therefore keeps them in
// scratch memory: every access a trip to the cache, on the critical path of every jump).  The reaction vectors sit in
// LDS, one 64-bit word of signed bytes per reaction.  Same arithmetic in the same order: same paths, same records.
template <int NS>
__device__ __forceinline__ unsigned long long hash_regs64(const int32_t (&x)[NS], int ns)
{
    static_assert(NS <= 8, "eight constants per sum");
    constexpr unsigned ca[8] = KFSP_MIX_A, cb[8] = KFSP_MIX_B;                 // (hash_state64 on registers)
    unsigned a = kMixSeedA, b = kMixSeedB;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < ns) {
            a += (unsigned)x[s] * ca[s];
            b += (unsigned)x[s] * cb[s];
        }
    return mix_finish(a, b);
}

template <int NS>
__device__ __forceinline__ int lookup_regs(const SsaDev &A, const int32_t (&y)[NS])
{
    const unsigned long long h = hash_regs64<NS>(y, A.ns);
    const unsigned tag = (unsigned)(h >> 32);
    if (A.bitmap && !((A.bitmap[(tag >> 5) & A.bmask] >> (tag & 31)) & 1u)) return 0;
    unsigned slot = (unsigned)h & A.tmask;
    for (;;) {
        const unsigned long long e = A.tab64[slot];
        if (e == 0ull) return 0;
        if ((unsigned)(e >> 32) == tag) {                           // (else: another state's slot - no need to look at it)
            const int idx = (int)(unsigned)e;
            const int32_t *z = A.state + (int64_t)(idx - 1) * A.lds;
            bool same = true;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (s < A.ns) same = same && z[s] == y[s];
            if (same) return idx;
        }
        slot = (slot + 1) & A.tmask;
    }
}

// x[s] for a run-time s (the same in every lane) out of the registers: a chain of selects, no memory
template <int NS>
__device__ __forceinline__ int x_select(const int32_t (&x)[NS], int s)
{
    int v = x[0];
#pragma unroll
    for (int i = 1; i < NS; ++i) v = s == i ? x[i] : v;
    return v;
}

// REGS: every reaction of the program is a product chain of at most three operands (one constant among them at most) or
// reads its one-species table, and the wavefront keeps one descriptor word and one constant per reaction in registers
// (SsaDev::fast_i / fast_d).  An unlisted state - every jump of a path once it has left the FSP at its rim, the paths
// a launch waits for - is then evaluated without a trip to memory except the tables': the interpreter reads the
// program word by word from LDS, one dependent load after the other, and the state through scratch memory because it
// indexes it at run time.  The chain ((o1 * o2) * o3) is multiplied in the code's order, the table entry is the table
// entry, a population beyond the table goes to the interpreter as before: same bits, same paths, same records.
template <int NS, int NR, bool LIGHT, bool REGS = false>
__global__ __launch_bounds__(kBlock) void k_ssa_walk(SsaDev A, unsigned long long *__restrict__ nrec_total, long long cap,
                                                     unsigned long long *__restrict__ keys, int32_t *__restrict__ rec)
{
    static_assert(NS <= 8, "one 64-bit word of signed bytes per reaction");
    __shared__ unsigned long long s_nu[NR];
    if (threadIdx.x < NR) {
        unsigned long long w = 0;
        if ((int)threadIdx.x < A.nr)
            for (int s = 0; s < A.ns; ++s) w |= (unsigned long long)(unsigned char)(signed char)A.nu[threadIdx.x * A.ns + s] << (8 * s);
        s_nu[threadIdx.x] = w;
    }
    // a light program is small (kSsaLds* bound it): its code, immediates and parameters are read from LDS - an unlisted
    // state costs nr passes through the interpreter, every opcode a dependent load, and the whole wavefront waits for it
    __shared__ int32_t s_code[LIGHT ? kSsaLdsCode : 1], s_ioff[LIGHT ? 3 * (NR + 1) : 1];
    __shared__ double s_dbl[LIGHT ? 2 * kSsaLdsDbl + NR * kPropMonoOps : 1];
    __shared__ int32_t s_mono[LIGHT ? NR * (1 + kPropMonoOps) : 1];
    PropDev P = A.P;
    if (LIGHT) {
        const int ncode = A.P.code_off[A.nr], nimm = A.P.imm_off[A.nr];
        for (int i = threadIdx.x; i < ncode; i += kBlock) s_code[i] = A.P.code[i];
        for (int i = threadIdx.x; i <= A.nr; i += kBlock) {
            s_ioff[i] = A.P.code_off[i];
            s_ioff[NR + 1 + i] = A.P.imm_off[i];
            if (i < A.nr) s_ioff[2 * (NR + 1) + i] = A.P.tab_species[i];
        }
        for (int i = threadIdx.x; i < nimm; i += kBlock) s_dbl[i] = A.P.imm[i];
        for (int i = threadIdx.x; i < A.P.np; i += kBlock) s_dbl[kSsaLdsDbl + i] = A.P.params[i];
        for (int i = threadIdx.x; i < A.nr * (1 + kPropMonoOps); i += kBlock) s_mono[i] = A.P.mono[i];
        for (int i = threadIdx.x; i < A.nr * kPropMonoOps; i += kBlock) s_dbl[2 * kSsaLdsDbl + i] = A.P.mono_c[i];
        P.mono = s_mono;
        P.mono_c = s_dbl + 2 * kSsaLdsDbl;
        P.code = s_code;
        P.code_off = s_ioff;
        P.imm_off = s_ioff + (NR + 1);
        P.tab_species = s_ioff + 2 * (NR + 1);
        P.imm = s_dbl;
        P.params = s_dbl + kSsaLdsDbl;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long wave = A.wave0 + (((long long)blockIdx.x * kBlock + threadIdx.x) >> 6);
    // kSsaSeedsPerWave seeds per wavefront, dealt in blocks of 64 consecutive seed states ROUND ROBIN over the wavefronts:
    // the long paths start from neighbouring states (measured: the LAST states of the list, the FSP's rim - they step
    // outside and walk through unlisted states to the horizon, hundreds of jumps, while a path from an inner state falls
    // back onto an earlier seed after a handful), and a wavefront that owned 256 consecutive ones of them walked four
    // long paths per lane, one after the other
    const long long nwaves = ((long long)A.n0 + kSsaSeedsPerWave - 1) / kSsaSeedsPerWave;
    if (wave >= A.wave1) return;
    int next = 0;                                                  // wave-uniform: seeds handed out so far
    unsigned long long cbase = 0;                                  // wave-uniform: the wavefront's chunk of the record list
    int cleft = 0, nvalid = 0;
    bool active = false;
    int j0 = 0, j = 0, npos = 0;
    bool virt = false;
    unsigned long long rs = 0;
    double tt = 0.0, a0 = 0.0;
    int32_t x[NS];
    double pr[NR];
    int32_t aj[NR];
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        pr[k] = 0.0;
        aj[k] = 0;
    }
    int32_t fd[REGS ? NR : 1];
    double fc[REGS ? NR : 1];
    if (REGS) {
#pragma unroll
        for (int k = 0; k < (REGS ? NR : 1); ++k) {
            fd[k] = 0;
            fc[k] = 0.0;
            if (k < A.nr) {
                fd[k] = A.fast_i[k];
                fc[k] = A.fast_d[k];
            }
        }
    }
    // the row of a listed state is REQUESTED as soon as the state is known (when a path starts, and at the end of the
    // jump that reaches it) and USED after the next jump's random numbers and logarithm have been computed
#define KFSP_SSA_ROW_REGS()                                                     \
    do {                                                                        \
        const int64_t row_ = (int64_t)(j - 1) * A.lda;                          \
        a0 = A.diag[j - 1];                                                     \
        _Pragma("unroll") for (int k_ = 0; k_ < NR; ++k_) if (k_ < A.nr) {      \
            pr[k_] = A.off[row_ + k_];                                          \
            aj[k_] = A.adj[row_ + k_];                                          \
        }                                                                       \
    } while (0)
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle) {
            if (!active) {
                const int t = next + __popcll(idle & ((1ull << lane) - 1ull));
                const long long seed = ((((long long)(t >> 6)) * nwaves + wave) << 6) + (t & 63) + 1;
                if (t < kSsaSeedsPerWave && seed <= A.n0) {
                    j0 = (int)seed;
                    rs = (A.seedmix * 2654435761ull) ^ ((unsigned long long)j0 * 40503ull + 12345ull);
                    rs = ((rs ^ (rs >> 29)) & 4294967295ull) * 1181783497ull;
                    rs = 1ull + (((rs ^ (rs >> 32)) & 9223372036854775807ull) % (kLcgM - 1ull));
                    j = j0;
                    virt = false;
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (s < A.ns) x[s] = A.state[(int64_t)(j - 1) * A.lds + s];
                    KFSP_SSA_ROW_REGS();
                    tt = 0.0;
                    npos = 0;
                    active = true;
                }
            }
            next += __popcll(idle);
        }
        if (!__ballot(active)) break;                              // no path left in this wave
        bool ended = active, record = false;
        if (active) do {
            double r1 = lcg_uniform(rs);
            const double r2 = lcg_uniform(rs);
            if (r1 <= 0.0) r1 = 0x1p-54;
            const double wait = -plog(r1);
            if (virt) {
                // (an unlisted state: its propensities come from the program)
                double ps[NR];
                if (REGS) {
#pragma unroll
                    for (int k = 0; k < (REGS ? NR : 1); ++k) {
                        ps[k] = 0.0;
                        if (k < A.nr) {
                            const int d = __builtin_amdgcn_readfirstlane(fd[k]);
                            const int n = d & 7;
                            if (n == 0) {
                                const int v = x_select<NS>(x, (d >> 4) & 15);
                                if (v >= 0 && v < P.tab_len) {
                                    ps[k] = A.P.tab[(int64_t)k * P.tab_len + v];
                                } else {
                                    int32_t xs[NS];
#pragma unroll
                                    for (int s = 0; s < NS; ++s) xs[s] = x[s];
                                    ps[k] = LIGHT ? prop_eval_light(P, k, xs) : prop_eval(A.P, k, xs);
                                }
                            } else {
                                const int s0 = (d >> 4) & 15, s1 = (d >> 8) & 15, s2 = (d >> 12) & 15;
                                double v = s0 == 15 ? fc[k] : (double)x_select<NS>(x, s0);
                                if (n > 1) v = v * (s1 == 15 ? fc[k] : (double)x_select<NS>(x, s1));
                                if (n > 2) v = v * (s2 == 15 ? fc[k] : (double)x_select<NS>(x, s2));
                                ps[k] = v;
                            }
                        }
                    }
                } else {
                    int32_t xs[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) xs[s] = x[s];
                    for (int k = 0; k < A.nr; ++k) ps[k] = LIGHT ? prop_eval_light(P, k, xs) : prop_eval(A.P, k, xs);
                }
                a0 = 0.0;
#pragma unroll
                for (int k = 0; k < NR; ++k)
                    if (k < A.nr) {
                        pr[k] = ps[k];
                        a0 = a0 + pr[k];
                        aj[k] = 0;
                    }
            }
            if (!(a0 > 0.0)) break;                                // absorbing state
            tt = fmin(A.tstep, tt + (wait / a0));
            // the reaction whose cumulative propensity first reaches r2 a0 (the while loop of the general kernel, unrolled)
            double acc = pr[0];
            int k = 0;
            const double r2a = fmin(r2 * a0, a0);
#pragma unroll
            for (int i = 1; i < NR; ++i)
                if (i < A.nr && k == i - 1 && acc < r2a) {
                    k = i;
                    acc = acc + pr[i];
                }
            int ajk = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) ajk = i == k ? aj[i] : ajk;
            const unsigned long long nuw = s_nu[k];
            int32_t y[NS];
            bool neg = false;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                y[s] = x[s] + (int)(signed char)(nuw >> (8 * s));   // (species beyond ns: 0 + 0)
                neg = neg || y[s] < 0;
            }
            if (neg) break;
            int idx = max(ajk, 0);
            if (idx == 0) {
                bool legal = true;
#pragma unroll
                for (int s = 0; s < NS; ++s) legal = legal && y[s] <= A.max_count;
                if (!legal) break;
                idx = lookup_regs<NS>(A, y);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) x[s] = y[s];
            if (idx > 0) {
                j = idx;
                virt = false;
                if (j < j0) break;                                 // fell back onto an earlier seed
            } else {
                virt = true;
                record = true;                                     // x is an unlisted state: recorded below
            }
            if (!(tt < A.tstep)) break;
            if (!virt) KFSP_SSA_ROW_REGS();                        // (the next jump's row: on its way while the records are written)
            ended = false;
        } while (false);
        const unsigned long long rmask = __ballot(record);
        if (rmask) {
            // room for this jump's records: the wavefront takes slots of the list kSsaChunk at a time (ONE counter serves
            // all wavefronts, and an atomic per recording jump - 1e5 of them per call on one address - was what the kernel
            // waited for); what is left of a chunk when the next one is taken, or at the end, stays empty (its key keeps the all-ones fill)
            const int c = __popcll(rmask);
            if (c > cleft) {
                unsigned long long first = 0;
                if (lane == 0) first = atomicAdd(nrec_total, (unsigned long long)kSsaChunk);
                cbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)first);
                cleft = kSsaChunk;
            }
            const unsigned long long first = cbase;
            cbase += (unsigned long long)c;
            cleft -= c;
            nvalid += c;
            if (record) {
                const unsigned long long slot = first + (unsigned long long)__popcll(rmask & ((1ull << lane) - 1ull));
                if ((long long)slot < cap) {
                    keys[slot] = ((unsigned long long)(unsigned)j0 << kSsaPosBits) | (unsigned long long)min(npos, (1 << kSsaPosBits) - 1);
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (s < A.ns) rec[slot * A.ns + s] = x[s];
                }
                ++npos;
            }
        }
        if (active && ended) active = false;
    }
    if (lane == 0 && nvalid > 0) atomicAdd(nrec_total + 1, (unsigned long long)nvalid);
#undef KFSP_SSA_ROW_REGS
}

// duplicates among the records, visited in the order of their keys (perm[r] = the record of rank r): one table slot
// per distinct state, minrank[slot] = the rank of its first record
__global__ __launch_bounds__(kBlock) void k_rec_insert(long long nrec, int ns, const int32_t *__restrict__ rec, const int32_t *__restrict__ perm,
                                                       int32_t *tab2, int32_t *minrank, unsigned mask)
{
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= nrec) return;
    const int32_t *y = rec + (long long)perm[r] * ns;
    unsigned slot = hash_state(y, ns) & mask;
    for (;;) {
        int cur = tab2[slot];
        if (cur == 0) cur = atomicCAS(&tab2[slot], 0, (int)r + 1);
        if (cur == 0) {                                            // claimed: this record names the slot
            atomicMin(&minrank[slot], (int)r);
            return;
        }
        const int32_t *z = rec + (long long)perm[cur - 1] * ns;
        bool same = true;
        for (int s = 0; s < ns; ++s) same = same && z[s] == y[s];
        if (same) {
            atomicMin(&minrank[slot], (int)r);
            return;
        }
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(kBlock) void k_rec_first(unsigned slots, const int32_t *__restrict__ tab2, const int32_t *__restrict__ minrank,
                                                      uint8_t *__restrict__ first)
{
    const unsigned s = blockIdx.x * kBlock + threadIdx.x;
    if (s < slots && tab2[s] != 0) first[minrank[s]] = 1;
}

// out[i] = coordinates of the record whose rank is sel[i]
__global__ __launch_bounds__(kBlock) void k_rec_gather(int nnew, int ns, int lds, const int32_t *__restrict__ sel, const int32_t *__restrict__ perm,
                                                       const int32_t *__restrict__ rec, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nnew) return;
    const int32_t *y = rec + (int64_t)perm[sel[i]] * ns;
    for (int s = 0; s < lds; ++s) out[(int64_t)i * lds + s] = s < ns ? y[s] : 0;
}

__global__ __launch_bounds__(kBlock) void k_iota(long long n, int32_t *v)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
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

}  // namespace

#define SSA_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return 1000 + (int)e_;                                                         \
        }                                                                                  \
    } while (0)

void launch_table_build64(int n, int ns, int lds, const int32_t *state, unsigned long long *tab, unsigned mask, unsigned *bitmap, unsigned bmask,
                          hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_ht_build64, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, ns, lds, state, tab, mask, bitmap, bmask);
}

void launch_table_build(int n, int ns, int lds, const int32_t *state, int32_t *tab, unsigned mask, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_ht_build, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, ns, lds, state, tab, mask);
}

// The walk on lists that are on the device (n states: d_state / d_adj / d_off / d_diag).  The states met, in (seed state,
// position on the path) order of their first occurrence, stay on the device with their propensity columns:
// *sn (nnew x lds), *on (nnew x ldo), *dn (nnew) point into ctx->d_pstage until that buffer is used again.
int ssa_streams_core(kfsp_ctx *ctx, double tstep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n,
                     const int32_t *d_state, int32_t lds, const int32_t *d_adj, const double *d_off, int32_t lda, const double *d_diag,
                     int32_t max_count, int32_t cap_new, int32_t ldo, int32_t *n_found, int32_t **sn, double **on, double **dn,
                     bool partitioned)
{
    hipStream_t st = ctx->stream;
    const auto blocks = [](long long k) { return (int)std::max<long long>(1, (k + kBlock - 1) / kBlock); };
    unsigned slots = 64;
    while (slots < 2u * (unsigned)n) slots <<= 1;
    bool fast = ns <= 8 && nr <= 16 && !ctx->opt_ssa_general;
    // (the register-resident kernels keep a reaction vector as signed bytes: a coefficient outside them takes the general kernel)
    for (size_t i = 0; i < (size_t)nr * ns; ++i)
        if (stoich[i] < -128 || stoich[i] > 127) fast = false;
    // arena 1: the table of the listed states (tagged, 8 bytes per slot, for the register-resident kernels), the reaction
    // vectors, the record counter
    // (the filter: 4 bits per slot = 8 to 16 per listed state, one word per 32)
    const unsigned bwords = slots / 8;
    const bool filter = fast && ctx->opt_ssa_filter != 0;
    const size_t need1 = (size_t)nr * ns * 4 + (size_t)slots * 8 + (size_t)bwords * 4 + 4096;
    SSA_TRY(ctx->d_os2.reserve(need1, false));
    Arena a1{ctx->d_os2.p};
    int32_t *d_nu = a1.take<int32_t>((size_t)nr * ns);
    unsigned long long *d_tab64 = a1.take<unsigned long long>(slots);
    unsigned *d_bitmap = a1.take<unsigned>(bwords);
    int32_t *d_tab = reinterpret_cast<int32_t *>(d_tab64);
    unsigned long long *d_total = a1.take<unsigned long long>(2);
    SSA_TRY(hipMemcpyAsync(d_nu, stoich, (size_t)nr * ns * 4, hipMemcpyHostToDevice, st));
    // (table and filter are neighbours in the arena: one memset)
    SSA_TRY(hipMemsetAsync(d_tab64, 0, fast ? (size_t)((char *)(d_bitmap + bwords) - (char *)d_tab64) : (size_t)slots * 4, st));
    if (fast) {
        launch_table_build64(n, ns, lds, d_state, d_tab64, slots - 1, filter ? d_bitmap : (unsigned *)nullptr, bwords - 1, st);
    } else {
        launch_table_build(n, ns, lds, d_state, d_tab, slots - 1, st);
    }
    SsaDev A;
    A.ns = ns;
    A.nr = nr;
    A.n0 = n;
    A.max_count = max_count;
    A.lds = lds;
    A.lda = lda;
    A.tstep = tstep;
    A.seedmix = (unsigned long long)seedmix;
    A.state = d_state;
    A.adj = d_adj;
    A.off = d_off;
    A.diag = d_diag;
    A.nu = d_nu;
    A.tab = d_tab;
    A.tab64 = d_tab64;
    A.bitmap = filter ? d_bitmap : nullptr;
    A.bmask = bwords - 1;
    A.tmask = slots - 1;
    A.P = prop_dev(ctx);
    // one wavefront per kSsaSeedsPerWave seed states (4 wavefronts per workgroup); the record list is sized by a guess
    // and, should the paths meet more unlisted states than that, by the count the first attempt returns
    // Under a row partition (partitioned: every rank is here, with the same lists) rank p walks wavefronts
    // [nwaves p / P, nwaves (p + 1) / P) of the walk only - a wavefront's seeds are blocks of 64 dealt round robin over ALL
    // wavefronts, so every share holds rim and inner seeds alike - and the records are all-gathered below: keys are
    // (seed state, position), the union is the unpartitioned walk's record set, and everything after it (sort, first
    // occurrences, columns) runs on every rank on identical input.
    const long long nwaves = ((long long)n + kSsaSeedsPerWave - 1) / kSsaSeedsPerWave;
    const bool share = partitioned && ctx->use_comm && ctx->nranks > 1 && ctx->opt_ssa_partition != 0;
    A.wave0 = share ? nwaves * ctx->rank / ctx->nranks : 0;
    A.wave1 = share ? nwaves * (ctx->rank + 1) / ctx->nranks : nwaves;
    const int wgrid = (int)std::max<long long>(1, (A.wave1 - A.wave0 + 3) / 4);
    // (the register-resident kernel leaves up to kSsaChunk - 1 slots empty per wavefront and chunk change)
    long long cap = std::max<long long>((long long)1 << 18, (long long)n), nrec = 0, nvalid = 0;
    unsigned long long *d_keys = nullptr;
    int32_t *d_rec = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        SSA_TRY(ctx->d_os4.reserve((size_t)cap * 8 + (size_t)cap * ns * 4 + 1024, false));
        Arena a2{ctx->d_os4.p};
        d_keys = a2.take<unsigned long long>((size_t)cap);
        d_rec = a2.take<int32_t>((size_t)cap * ns);
        SSA_TRY(hipMemsetAsync(d_total, 0, 2 * sizeof(unsigned long long), st));
        SSA_TRY(hipMemsetAsync(d_keys, 0xff, (size_t)cap * 8, st));
        // (the program's library functions, if any, sit behind tables that cover every population a legal state can have)
        const bool light = (ctx->prop_light || (ctx->prop_light_tab && ctx->prop_tab_len > max_count)) && ctx->prop_ncode <= kSsaLdsCode &&
                           ctx->prop_nimm <= kSsaLdsDbl && ctx->prop_np <= kSsaLdsDbl;
        // (the unrolled per-path code is as long as its bounds: the smallest instance that holds the model)
        // (product chains and one-species tables only, no two-species table in the way: descriptors in registers)
        const bool regs = fast && light && ctx->prop_fast && !ctx->prop_has_tab2 && ctx->opt_ssa_regs != 0;
        A.fast_i = ctx->d_prop_fast_i.p;
        A.fast_d = ctx->d_prop_fast_d.p;
        if (regs && ns <= 2 && nr <= 4)
            hipLaunchKernelGGL((k_ssa_walk<2, 4, true, true>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else if (regs && ns <= 6 && nr <= 12)
            hipLaunchKernelGGL((k_ssa_walk<6, 12, true, true>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else if (regs)
            hipLaunchKernelGGL((k_ssa_walk<8, 16, true, true>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else if (fast && light && ns <= 2 && nr <= 4)
            hipLaunchKernelGGL((k_ssa_walk<2, 4, true>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else if (fast && light && ns <= 6 && nr <= 12)
            hipLaunchKernelGGL((k_ssa_walk<6, 12, true>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else if (fast && light)
            hipLaunchKernelGGL((k_ssa_walk<8, 16, true>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else if (fast)
            hipLaunchKernelGGL((k_ssa_walk<8, 16, false>), dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        else
            hipLaunchKernelGGL(k_ssa_walk_any, dim3(wgrid), dim3(kBlock), 0, st, A, d_total, cap, d_keys, d_rec);
        unsigned long long got[2] = {0, 0};
        SSA_TRY(hipMemcpyAsync(got, d_total, sizeof(got), hipMemcpyDeviceToHost, st));
        SSA_TRY(hipStreamSynchronize(st));
        int over = prop_check_overflow(ctx);                        // (a path left a two-species table: nothing was changed)
        if (over && over != -16) return over;
        nrec = (long long)got[0];                                   // slots taken (empty ones included)
        nvalid = (long long)got[1];                                 // records
        if (share) {
            // what every rank must agree on before the next collective: the largest slot count (the list's size, and the
            // length of the pieces gathered below), the records in all, and whether ANY rank's paths left a table
            const int P = ctx->nranks;
            SSA_TRY(ctx->d_os1.reserve((size_t)(P + 1) * 20 * sizeof(double) + 256, false));
            double *d_s = reinterpret_cast<double *>(ctx->d_os1.p), *d_r = d_s + 20, hs[20], hr[64 * 20];
            hs[0] = (double)nrec;
            hs[1] = (double)nvalid;
            hs[2] = over ? 1.0 : 0.0;
            for (int s_ = 0; s_ < 16; ++s_) hs[3 + s_] = over ? (double)ctx->prop_missed[s_] : 0.0;
            SSA_TRY(hipMemcpyAsync(d_s, hs, sizeof(hs), hipMemcpyHostToDevice, st));
            if (int rc = comm_gather_doubles(ctx, d_s, d_r, 20, st)) return rc;
            SSA_TRY(hipMemcpyAsync(hr, d_r, (size_t)P * 20 * sizeof(double), hipMemcpyDeviceToHost, st));
            SSA_TRY(hipStreamSynchronize(st));
            long long maxrec = 0, sumvalid = 0;
            bool any_over = false;
            for (int p = 0; p < P; ++p) {
                maxrec = std::max<long long>(maxrec, (long long)hr[20 * p]);
                sumvalid += (long long)hr[20 * p + 1];
                any_over = any_over || hr[20 * p + 2] != 0.0;
            }
            if (any_over) {
                for (int s_ = 0; s_ < 16; ++s_) {
                    int32_t m_ = 0;
                    for (int p = 0; p < P; ++p) m_ = std::max<int32_t>(m_, (int32_t)hr[20 * p + 3 + s_]);
                    ctx->prop_missed[s_] = m_;
                }
                ctx->err = "a population lies beyond a two-species propensity table (kfsp_propensity_overflow says which; enlarge and repeat)";
                return -16;
            }
            if (maxrec > cap) {                                      // some rank's list was too short: all repeat with the same size
                if (attempt == 1 || maxrec > 2000000000LL / P) {
                    ctx->err = "SSA paths met more unlisted states than the record list holds";
                    return -11;
                }
                cap = maxrec + 1024;
                continue;
            }
            // the pieces: maxrec slots of every rank (slots a rank did not take keep the empty key), rank after rank
            const long long tot = maxrec * P;
            SSA_TRY(ctx->d_os5.reserve((size_t)tot * 8 + (size_t)tot * ns * 4 + 1024, false));
            Arena a5{ctx->d_os5.p};
            unsigned long long *d_keys_all = a5.take<unsigned long long>((size_t)tot);
            int32_t *d_rec_all = a5.take<int32_t>((size_t)tot * ns);
            if (maxrec > 0) {
                if (int rc = comm_gather_bytes(ctx, d_keys, d_keys_all, (size_t)maxrec * 8, st)) return rc;
                if (int rc = comm_gather_bytes(ctx, d_rec, d_rec_all, (size_t)maxrec * ns * 4, st)) return rc;
            }
            d_keys = d_keys_all;
            d_rec = d_rec_all;
            nrec = tot;
            nvalid = sumvalid;
            break;
        }
        if (over) return over;
        if (nrec <= cap) break;
        if (attempt == 1 || nrec > 2000000000LL) {
            ctx->err = "SSA paths met more unlisted states than the record list holds";
            return -11;
        }
        cap = nrec + 1024;
    }
    *n_found = 0;
    *sn = nullptr;
    *on = *dn = nullptr;
    if (nvalid == 0) return 0;
    // the records in (seed state, position) order (empty slots sort behind them), duplicates removed (first occurrence stays)
    unsigned slots2 = 64;
    while (slots2 < 2u * (unsigned)nvalid) slots2 <<= 1;
    const size_t need3 = (size_t)nrec * (8 + 4 + 4) + (size_t)nvalid * (4 + 1) + (size_t)slots2 * 8 + 4096 + 10 * 256;
    SSA_TRY(ctx->d_os3.reserve(need3, false));
    Arena a3x{ctx->d_os3.p};
    unsigned long long *d_keys2 = a3x.take<unsigned long long>((size_t)nrec);
    int32_t *d_iota = a3x.take<int32_t>((size_t)nrec), *d_perm = a3x.take<int32_t>((size_t)nrec), *d_sel = a3x.take<int32_t>((size_t)nvalid);
    int32_t *d_tab2 = a3x.take<int32_t>(slots2), *d_min = a3x.take<int32_t>(slots2);
    uint8_t *d_first = a3x.take<uint8_t>((size_t)nvalid);
    int *d_nsel = a3x.take<int>(4);
    hipLaunchKernelGGL(k_iota, dim3(blocks(nrec)), dim3(kBlock), 0, st, nrec, d_iota);
    size_t tmp_bytes = 0;
    SSA_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys2, d_iota, d_perm, (int)nrec, 0, 31 + kSsaPosBits, st));
    SSA_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
    SSA_TRY(hipcub::DeviceRadixSort::SortPairs(ctx->d_sorttmp.p, tmp_bytes, d_keys, d_keys2, d_iota, d_perm, (int)nrec, 0, 31 + kSsaPosBits, st));
    SSA_TRY(hipMemsetAsync(d_tab2, 0, (size_t)slots2 * 4, st));
    SSA_TRY(hipMemsetAsync(d_min, 0x7f, (size_t)slots2 * 4, st));
    SSA_TRY(hipMemsetAsync(d_first, 0, (size_t)nvalid, st));
    hipLaunchKernelGGL(k_rec_insert, dim3(blocks(nvalid)), dim3(kBlock), 0, st, nvalid, ns, d_rec, d_perm, d_tab2, d_min, slots2 - 1);
    hipLaunchKernelGGL(k_rec_first, dim3(blocks(slots2)), dim3(kBlock), 0, st, slots2, d_tab2, d_min, d_first);
    SSA_TRY(hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, d_iota, d_first, d_sel, d_nsel, (int)nvalid, st));
    SSA_TRY(ctx->d_sorttmp.reserve(tmp_bytes + 256, false));
    SSA_TRY(hipcub::DeviceSelect::Flagged(ctx->d_sorttmp.p, tmp_bytes, d_iota, d_first, d_sel, d_nsel, (int)nvalid, st));
    int nnew = 0;
    SSA_TRY(hipMemcpyAsync(&nnew, d_nsel, sizeof(int), hipMemcpyDeviceToHost, st));
    SSA_TRY(hipStreamSynchronize(st));
    if (nnew > cap_new) {
        ctx->err = "FSP SIZE EXCEEDS MEMORY LIMIT";
        return -11;
    }
    // the new states in (seed state, position on the path) order of their first occurrence, and their columns
    const size_t sb = (size_t)nnew * lds * 4;
    SSA_TRY(ctx->d_pstage.reserve(((size_t)nnew * ldo + (size_t)nnew + (sb + 7) / 8) + 256, false));
    double *d_on = ctx->d_pstage.p, *d_dn = d_on + (size_t)nnew * ldo;
    int32_t *d_sn = reinterpret_cast<int32_t *>(d_dn + nnew);
    hipLaunchKernelGGL(k_rec_gather, dim3(blocks(nnew)), dim3(kBlock), 0, st, nnew, ns, lds, d_sel, d_perm, d_rec, d_sn);
    if (int rc = prop_eval_device(ctx, nnew, d_sn, lds, d_on, ldo, d_dn)) return rc;
    if (int rc = prop_check_overflow(ctx)) return rc;
    *n_found = nnew;
    *sn = d_sn;
    *on = d_on;
    *dn = d_dn;
    return 0;
}

// host lists in, host lists out (kfsp_ssa_streams)
int ssa_streams_device(kfsp_ctx *ctx, double tstep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n,
                       const int32_t *state, int32_t lds, const int32_t *adj, const double *offdiag, int32_t lda, const double *diag,
                       int32_t max_count, int32_t cap_new, int32_t *n_found, int32_t *state_new, double *off_new, int32_t ldo,
                       double *diag_new)
{
    hipStream_t st = ctx->stream;
    // Option ssa_resident: the caller vouches that these arrays are what it uploaded last (kfsp_update_matrix_ell keeps the
    // reference arrays verbatim, kfsp_set_state_coords the coordinates) - 152 MB per call stay where they are at 1e6 states.
    const bool res_gen = ctx->opt_ssa_resident != 0 && ctx->ell_cols == n && ctx->ell_ld == lda;
    const bool res_st = ctx->opt_ssa_resident != 0 && ctx->coords_n == n && ctx->coords_ld == lds;
    const size_t need1 = (res_st ? 0 : (size_t)n * lds * 4) + (res_gen ? 0 : (size_t)n * lda * 12 + (size_t)n * 8) + 4096;
    SSA_TRY(ctx->d_os1.reserve(need1, false));
    Arena a1{ctx->d_os1.p};
    const int32_t *d_state = ctx->d_coords.p, *d_adj = ctx->d_ell_adj.p;
    const double *d_off = ctx->d_ell_off.p, *d_diag = ctx->d_ell_diag.p;
    if (!res_st) {
        int32_t *p = a1.take<int32_t>((size_t)n * lds);
        SSA_TRY(hipMemcpyAsync(p, state, (size_t)n * lds * 4, hipMemcpyHostToDevice, st));
        d_state = p;
    }
    if (!res_gen) {
        int32_t *pa = a1.take<int32_t>((size_t)n * lda);
        double *po = a1.take<double>((size_t)n * lda), *pd = a1.take<double>((size_t)n);
        SSA_TRY(hipMemcpyAsync(pa, adj, (size_t)n * lda * 4, hipMemcpyHostToDevice, st));
        SSA_TRY(hipMemcpyAsync(po, offdiag, (size_t)n * lda * 8, hipMemcpyHostToDevice, st));
        SSA_TRY(hipMemcpyAsync(pd, diag, (size_t)n * 8, hipMemcpyHostToDevice, st));
        d_adj = pa;
        d_off = po;
        d_diag = pd;
    }
    int32_t nnew = 0, *d_sn = nullptr;
    double *d_on = nullptr, *d_dn = nullptr;
    *n_found = 0;
    if (int rc = ssa_streams_core(ctx, tstep, seedmix, ns, nr, stoich, n, d_state, lds, d_adj, d_off, lda, d_diag, max_count, cap_new,
                                  ldo, &nnew, &d_sn, &d_on, &d_dn))
        return rc;
    if (nnew == 0) return 0;
    SSA_TRY(hipMemcpyAsync(state_new, d_sn, (size_t)nnew * lds * 4, hipMemcpyDeviceToHost, st));
    SSA_TRY(hipMemcpyAsync(off_new, d_on, (size_t)nnew * ldo * 8, hipMemcpyDeviceToHost, st));
    SSA_TRY(hipMemcpyAsync(diag_new, d_dn, (size_t)nnew * 8, hipMemcpyDeviceToHost, st));
    SSA_TRY(hipStreamSynchronize(st));
    *n_found = nnew;
    return 0;
}

}  // namespace kfsp
