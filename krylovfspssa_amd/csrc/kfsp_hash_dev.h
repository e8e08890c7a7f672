// Open-addressing tables over lists of states (coordinates, n x ld int32), shared by the SSA walk (kfsp_ssa.hip)
// and the one-step sweep / resident expansion (kfsp_expand.hip).  The host's table of the same lists is
// StateSpace.f90:36-66 (a hashed key per state); nothing of its layout is observable, only "is this state listed,
// and under which number", so the device keeps its own: 32-bit slots holding index + 1, linear probing, a load of
// at most one half.
#pragma once

#include "kfsp_ctx.h"

namespace kfsp {

__device__ __forceinline__ unsigned hash_state(const int32_t *x, int ns)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    for (int s = 0; s < ns; ++s) {
        h ^= (unsigned long long)(unsigned)x[s] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    return (unsigned)(h ^ (h >> 32));
}

// 1-based index of y in state[0..)(the table's list), 0 = not listed
__device__ __forceinline__ int table_find(const int32_t *__restrict__ tab, unsigned mask, const int32_t *__restrict__ state, int lds, int ns,
                                          const int32_t *y)
{
    unsigned slot = hash_state(y, ns) & mask;
    for (;;) {
        const int e = tab[slot];
        if (e == 0) return 0;
        const int32_t *z = state + (int64_t)(e - 1) * lds;
        bool same = true;
        for (int s = 0; s < ns; ++s) same = same && z[s] == y[s];
        if (same) return e;
        slot = (slot + 1) & mask;
    }
}

// tab (zeroed, mask + 1 slots, mask + 1 >= 2 n) <- the n states of the list
void launch_table_build(int n, int ns, int lds, const int32_t *state, int32_t *tab, unsigned mask, hipStream_t st);

// The TAGGED table (round 4): 8-byte slots, (upper half of the state's hash) << 32 | index + 1, 0 = empty.  A probe that
// meets another state's slot sees it from the tag and moves on without fetching that state's coordinates - a look-up of
// an UNLISTED target ends at an empty slot after one cache miss instead of one per occupied slot on its way plus one per
// coordinate row behind them.  Its hash is made of 32-bit multiplications that do not wait for each other - two sums of
// x_s times odd constants, one finished into the slot, one into the tag - instead of hash_state's chain of dependent
// 64-bit products.  Any hash gives the same answers; nothing of a table's layout is observable.
__device__ __forceinline__ unsigned long long mix_finish(unsigned a, unsigned b)
{
    a ^= a >> 16;
    a *= 0x7FEB352Du;
    a ^= a >> 15;
    a *= 0x846CA68Bu;
    a ^= a >> 16;
    b ^= b >> 15;
    b *= 0x2C1B3C6Du;
    b ^= b >> 13;
    return ((unsigned long long)b << 32) | a;                      // tag | slot bits
}

#define KFSP_MIX_A {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u}
#define KFSP_MIX_B {0x7FEB352Du, 0x846CA68Bu, 0xE6546B65u, 0x9E485565u, 0xAF836E39u, 0xC5A308D3u, 0x2C1B3C6Du, 0x297A2D39u}
constexpr unsigned kMixSeedA = 0x68E31DA4u, kMixSeedB = 0xB5297A4Du;

__device__ __forceinline__ unsigned long long hash_state64(const int32_t *x, int ns)
{
    constexpr unsigned ca[8] = KFSP_MIX_A, cb[8] = KFSP_MIX_B;
    unsigned a = kMixSeedA, b = kMixSeedB;
    for (int s = 0; s < ns; ++s) {
        a += (unsigned)x[s] * ca[s & 7];
        b += (unsigned)x[s] * cb[s & 7];
    }
    return mix_finish(a, b);
}

__device__ __forceinline__ int table_find64(const unsigned long long *__restrict__ tab, unsigned mask, const int32_t *__restrict__ state, int lds,
                                            int ns, const int32_t *y)
{
    const unsigned long long h = hash_state64(y, ns);
    const unsigned tag = (unsigned)(h >> 32);
    unsigned slot = (unsigned)h & mask;
    for (;;) {
        const unsigned long long e = tab[slot];
        if (e == 0ull) return 0;
        if ((unsigned)(e >> 32) == tag) {
            const int idx = (int)(unsigned)e;
            const int32_t *z = state + (int64_t)(idx - 1) * lds;
            bool same = true;
            for (int s = 0; s < ns; ++s) same = same && z[s] == y[s];
            if (same) return idx;
        }
        slot = (slot + 1) & mask;
    }
}

// tab (zeroed, mask + 1 eight-byte slots) <- the n states; bitmap (zeroed, bmask + 1 words; null: none) gets one bit per
// state, addressed by the tag (the walk's filter in front of the table)
void launch_table_build64(int n, int ns, int lds, const int32_t *state, unsigned long long *tab, unsigned mask, unsigned *bitmap, unsigned bmask,
                          hipStream_t st);

}  // namespace kfsp
