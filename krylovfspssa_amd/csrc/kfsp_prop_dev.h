// Device side of the propensity program (kfsp_prop.hip), shared with the SSA walk (kfsp_ssa.hip): the program's
// view in device memory and the interpreter.  See kfsp_prop.hip for the opcodes and the bit-exactness rules.
#pragma once

#include "kfsp_ctx.h"

#pragma clang fp contract(off)

namespace kfsp {

constexpr int kPropStack = 32;
constexpr int kPropLightStack = 8;      // deepest stack of a program that may run through prop_eval_light

struct PropDev {
    int ns, nr, np, tab_len;
    const int32_t *code_off, *code, *imm_off, *tab_species;
    const double *imm, *params, *tab;
};

__device__ inline double prop_eval(const PropDev &P, int k, const int32_t *__restrict__ x)
{
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
    return P;
}


}  // namespace kfsp
