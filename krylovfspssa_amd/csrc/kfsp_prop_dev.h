// Device side of the propensity program (kfsp_prop.hip), shared with the SSA walk (kfsp_ssa.hip): the program's
// view in device memory and the interpreter.  See kfsp_prop.hip for the opcodes and the bit-exactness rules.
#pragma once

#include "kfsp_ctx.h"

#pragma clang fp contract(off)

namespace kfsp {

constexpr int kPropStack = 32;
constexpr int kPropLightStack = 8;      // deepest stack of a program that may run through prop_eval_light

constexpr int kPropMonoOps = 4;         // a product chain of at most this many operands is evaluated without the interpreter

struct PropDev {
    int ns, nr, np, tab_len;
    const int32_t *code_off, *code, *imm_off, *tab_species;
    const double *imm, *params, *tab;
    // reaction k is a plain product  o_1 * o_2 * ... * o_n  (mass action: c X Y): mono[k][0] = n (0: it is not),
    // mono[k][i] = species of operand i or -1 for a constant, whose value is mono_c[k][i - 1]
    const int32_t *mono;                // [nr][1 + kPropMonoOps]
    const double *mono_c;               // [nr][kPropMonoOps]
    // two-species tables (null: none): reaction k reads t2d[t2o[k] + x[s2] * n1 + x[s1]] with (s1, s2, n1, n2) = t2i[4k ..]
    const int32_t *t2i;
    const long long *t2o;
    const double *t2d;
    int32_t *oob;                       // [0] raised by a population beyond a table, [1 + s] largest such population of species s
};

// a_k(x) from the two-species table of reaction k (the caller saw t2i[4k] >= 0); a miss is recorded and gives 0 - the host
// discards whatever the operation produced, enlarges the table and repeats it (include/kfsp.h kfsp_set_propensity_tables2)
__device__ __forceinline__ double prop_tab2(const PropDev &P, int k, const int32_t *__restrict__ x)
{
    const int32_t *t = P.t2i + 4 * k;
    const int v1 = x[t[0]], v2 = x[t[1]];
    if (v1 >= 0 && v2 >= 0 && v1 < t[2] && v2 < t[3]) return P.t2d[P.t2o[k] + (long long)v2 * t[2] + v1];
    if (v1 >= t[2]) atomicMax(&P.oob[1 + t[0]], v1);
    if (v2 >= t[3]) atomicMax(&P.oob[1 + t[1]], v2);
    P.oob[0] = 1;
    return 0.0;
}

// ((o_1 * o_2) * o_3) ... : the multiplications of the postfix code  o_1 o_2 MUL o_3 MUL ...  in its order - same bits
__device__ __forceinline__ double prop_mono(const PropDev &P, int k, int n, const int32_t *__restrict__ x)
{
    const int32_t *m = P.mono + k * (1 + kPropMonoOps);
    const double *c = P.mono_c + k * kPropMonoOps;
    double v = m[1] < 0 ? c[0] : (double)x[m[1]];
    for (int i = 1; i < n; ++i) v = v * (m[1 + i] < 0 ? c[i] : (double)x[m[1 + i]]);
    return v;
}

__device__ inline double prop_eval(const PropDev &P, int k, const int32_t *__restrict__ x)
{
    // (a product chain first: c X from two registers beats the host-made table's trip to memory, and is the same bits -
    // the table entry IS that product, made by the host's interpreter)
    {
        const int n = P.mono[k * (1 + kPropMonoOps)];
        if (n > 0) return prop_mono(P, k, n, x);
    }
    if (P.t2i && P.t2i[4 * k] >= 0) return prop_tab2(P, k, x);
    const int ts = P.tab_species[k];
    if (ts >= 0) {
        const int v = x[ts];
        if (v >= 0 && v < P.tab_len) return P.tab[(int64_t)k * P.tab_len + v];
    }
    double st[kPropStack];
    int sp = 0;
    const double *imm = P.imm + P.imm_off[k];
    for (int ip = P.code_off[k]; ip < P.code_off[k + 1]; ++ip) {
        const int c = P.code[ip];
        switch (c) {
        case 1: st[sp++] = *imm++; break;
        case 2: st[sp - 1] = -st[sp - 1]; break;
        case 3: st[sp - 2] = st[sp - 2] + st[sp - 1]; --sp; break;
        case 4: st[sp - 2] = st[sp - 2] - st[sp - 1]; --sp; break;
        case 5: st[sp - 2] = st[sp - 2] * st[sp - 1]; --sp; break;
        case 6:
            if (st[sp - 1] == 0.0) return 0.0;
            st[sp - 2] = st[sp - 2] / st[sp - 1];
            --sp;
            break;
        case 7: st[sp - 2] = pow(st[sp - 2], st[sp - 1]); --sp; break;
        case 11: st[sp - 1] = fabs(st[sp - 1]); break;
        case 12: st[sp - 1] = exp(st[sp - 1]); break;
        case 13:
            if (st[sp - 1] <= 0.0) return 0.0;
            st[sp - 1] = log10(st[sp - 1]);
            break;
        case 14:
            if (st[sp - 1] <= 0.0) return 0.0;
            st[sp - 1] = log(st[sp - 1]);
            break;
        case 15:
            if (st[sp - 1] < 0.0) return 0.0;
            st[sp - 1] = sqrt(st[sp - 1]);
            break;
        case 16: st[sp - 1] = sinh(st[sp - 1]); break;
        case 17: st[sp - 1] = cosh(st[sp - 1]); break;
        case 18: st[sp - 1] = tanh(st[sp - 1]); break;
        case 19: st[sp - 1] = sin(st[sp - 1]); break;
        case 20: st[sp - 1] = cos(st[sp - 1]); break;
        case 21: st[sp - 1] = tan(st[sp - 1]); break;
        case 22:
            if (fabs(st[sp - 1]) > 1.0) return 0.0;
            st[sp - 1] = asin(st[sp - 1]);
            break;
        case 23:
            if (fabs(st[sp - 1]) > 1.0) return 0.0;
            st[sp - 1] = acos(st[sp - 1]);
            break;
        case 24: st[sp - 1] = atan(st[sp - 1]); break;
        default: {
            const int v = c - 101;                                   // 0-based variable
            st[sp++] = v < P.ns ? (double)x[v] : P.params[v - P.ns];
        }
        }
    }
    return sp >= 1 ? st[0] : 0.0;
}

// The same for a program made of + - * / NEG, immediates and variables only (mass action, and everything whose other
// reactions travel as tables) with a stack of at most kPropLightStack: none of the math library is pulled into the
// caller, whose register budget stays that of its own loop (the SSA walk runs four times the wavefronts per SIMD with
// it).  Same operations in the same order: same bits.  prop_set_program decides whether a program qualifies.
__device__ inline double prop_eval_light(const PropDev &P, int k, const int32_t *__restrict__ x)
{
    // (a product chain first: c X from two registers beats the host-made table's trip to memory, and is the same bits -
    // the table entry IS that product, made by the host's interpreter)
    {
        const int n = P.mono[k * (1 + kPropMonoOps)];
        if (n > 0) return prop_mono(P, k, n, x);
    }
    if (P.t2i && P.t2i[4 * k] >= 0) return prop_tab2(P, k, x);
    const int ts = P.tab_species[k];
    if (ts >= 0) {
        const int v = x[ts];
        if (v >= 0 && v < P.tab_len) return P.tab[(int64_t)k * P.tab_len + v];
    }
    double st[kPropLightStack];
    int sp = 0;
    const double *imm = P.imm + P.imm_off[k];
    for (int ip = P.code_off[k]; ip < P.code_off[k + 1]; ++ip) {
        const int c = P.code[ip];
        switch (c) {
        case 1: st[sp++] = *imm++; break;
        case 2: st[sp - 1] = -st[sp - 1]; break;
        case 3: st[sp - 2] = st[sp - 2] + st[sp - 1]; --sp; break;
        case 4: st[sp - 2] = st[sp - 2] - st[sp - 1]; --sp; break;
        case 5: st[sp - 2] = st[sp - 2] * st[sp - 1]; --sp; break;
        case 6:
            if (st[sp - 1] == 0.0) return 0.0;
            st[sp - 2] = st[sp - 2] / st[sp - 1];
            --sp;
            break;
        default: {
            const int v = c - 101;                                   // 0-based variable
            st[sp++] = v < P.ns ? (double)x[v] : P.params[v - P.ns];
        }
        }
    }
    return sp >= 1 ? st[0] : 0.0;
}

inline PropDev prop_dev(const kfsp_ctx *ctx)
{
    PropDev P;
    P.ns = ctx->prop_ns;
    P.nr = ctx->prop_nr;
    P.np = ctx->prop_np;
    P.tab_len = ctx->prop_tab_len;
    const int32_t *ib = ctx->d_prop_i.p;
    P.code_off = ib;
    P.imm_off = ib + (P.nr + 1);
    P.tab_species = ib + 2 * (P.nr + 1);
    P.code = ib + 2 * (P.nr + 1) + P.nr;
    const double *db = ctx->d_prop_d.p;
    P.params = db;
    P.imm = db + ctx->prop_np_pad;
    P.tab = db + ctx->prop_np_pad + ctx->prop_nimm_pad;
    P.mono = ib + ctx->prop_mono_off;
    P.mono_c = db + ctx->prop_monoc_off;
    P.t2i = ctx->prop_has_tab2 ? ctx->d_prop_t2i.p : nullptr;
    P.t2o = ctx->d_prop_t2o.p;
    P.t2d = ctx->d_prop_t2d.p;
    P.oob = ctx->d_prop_oob.p;
    return P;
}


}  // namespace kfsp
