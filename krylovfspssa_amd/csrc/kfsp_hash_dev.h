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

}  // namespace kfsp
