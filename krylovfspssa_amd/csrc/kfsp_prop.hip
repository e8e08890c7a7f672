// Propensities on the device (second half): a_k(x) of the model's .input expressions
// for whole lists of states, so that the OFFDIAG / DIAG columns of appended states are made where the state
// lists already are.
//
// Reference: MODEL%PROPENSITY (src/model/ModelModule.f90:163-199) evaluates the parsed expression of
// reaction k through the stack machine of src/parser/FortranParser.f90:187-302 over (species counts,
// parameter values).  The program handed over here is the postfix code of the host's own expression type
// (krylovfspssa_amd/fortran/kfsp_expr.f90, pinned against the reference parser by tests/golden/exprtable.npz):
//   1 IMM (next immediate)   2 NEG   3 ADD   4 SUB   5 MUL   6 DIV   7 POW   10+k function k (abs exp log10 log
//   sqrt sinh cosh tanh sin cos tan asin acos atan)   100+i variable i (1..ns species, then the parameters)
// with the reference's rules: x / 0, log / log10 of x <= 0, sqrt of x < 0, asin / acos outside [-1, 1] make the
// WHOLE expression 0.
//
// Bit-exactness.  + - * / and NEG are IEEE operations, evaluated one by one as the host's interpreter does (no
// contraction: the code is data) - same bits.  pow and the functions come from the device's math library, which
// is not the host's: up to ~2 ulp apart.  To keep the columns bit-identical to the host's wherever possible, a
// propensity that depends on ONE species only (every Hill function, every x (x - 1) / 2 of the shipped
// models) is not interpreted at all: the host tabulates it with ITS evaluator at every population count
// 0 .. tab_len - 1 and the device looks it up; populations beyond the table fall back to the interpreter.
// Propensities of several species built from + - * / alone (mass action c X Y) are exact either way.
#include "kfsp_prop_dev.h"

#include <cstring>

#pragma clang fp contract(off)

namespace kfsp {

namespace {

// OFFDIAG(k, i) = a_k(x_i), DIAG(i) = their sum in reaction order (StateSpace.f90:207-212); one lane per state
__global__ __launch_bounds__(kBlock) void k_propensities(PropDev P, int64_t n, const int32_t *__restrict__ state, int lds,
                                                         double *__restrict__ offdiag, int ldo, double *__restrict__ diag)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t *x = state + i * lds;
    double d = 0.0;
    for (int k = 0; k < P.nr; ++k) {
        const double a = prop_eval(P, k, x);
        offdiag[i * ldo + k] = a;
        d += a;
    }
    diag[i] = d;
}

}  // namespace

#define HIP_TRY_P(expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return 1000 + (int)e_;                                                         \
        }                                                                                  \
    } while (0)

int prop_set_program(kfsp_ctx *ctx, int32_t ns, int32_t nr, int32_t np, const double *params, const int32_t *code_off,
                     const int32_t *code, const int32_t *imm_off, const double *imm, const int32_t *tab_species, int32_t tab_len,
                     const double *tab)
{
    ctx->prop_ready = false;
    ctx->prop_has_tab2 = false;           // (two-species tables belong to one program: kfsp_set_propensity_tables2 follows this call)
    // the code is checked once, here: every operand exists, the stack never exceeds the kernel's, variables are in range
    const int ncode = code_off[nr], nimm = imm_off[nr];
    bool light = true, light_tab = true;         // (+ - * / NEG only, shallow: prop_eval_light may run it / all but tabulated reactions)
    for (int k = 0; k < nr; ++k) {
        if (code_off[k + 1] < code_off[k] || imm_off[k + 1] < imm_off[k]) {
            ctx->err = "propensity program: offsets not monotone";
            return -6;
        }
        int sp = 0, ni = 0;
        for (int ip = code_off[k]; ip < code_off[k + 1]; ++ip) {
            const int c = code[ip];
            if (c == 7 || (c >= 11 && c <= 24)) {
                light = false;
                if (tab_species[k] < 0 || tab_len <= 0) light_tab = false;
            }
            if (c == 1) {
                ++sp;
                ++ni;
            } else if (c == 2 || (c >= 11 && c <= 24)) {
                if (sp < 1) sp = -1000;
            } else if (c >= 3 && c <= 7) {
                if (sp < 2) sp = -1000;
                --sp;
            } else if (c >= 101 && c <= 100 + ns + np) {
                ++sp;
            } else {
                ctx->err = "propensity program: unknown opcode";
                return -7;
            }
            if (sp > kPropLightStack) light = light_tab = false;
            if (sp < 0 || sp > kPropStack) {
                ctx->err = "propensity program: malformed expression or stack deeper than 32";
                return -7;
            }
        }
        if (ni != imm_off[k + 1] - imm_off[k]) {
            ctx->err = "propensity program: immediates do not match the code";
            return -9;
        }
        if (tab_species[k] >= ns) {
            ctx->err = "propensity program: table species out of range";
            return -10;
        }
    }
    // product chains  o_1 o_2 MUL o_3 MUL ...  of species, parameters and immediates (mass action) skip the interpreter
    std::vector<int32_t> mono((size_t)nr * (1 + kPropMonoOps), 0);
    std::vector<double> mono_c((size_t)nr * kPropMonoOps, 0.0);
    for (int k = 0; k < nr; ++k) {
        int32_t *m = mono.data() + (size_t)k * (1 + kPropMonoOps);
        double *mc = mono_c.data() + (size_t)k * kPropMonoOps;
        int n = 0, ii = imm_off[k];
        bool ok = code_off[k + 1] > code_off[k];
        for (int ip = code_off[k]; ok && ip < code_off[k + 1]; ++ip) {
            const int c = code[ip];
            // positions: 0 operand, 1 operand, 2 MUL, 3 operand, 4 MUL, ...
            const int pos = ip - code_off[k];
            if (pos == 0 || pos % 2 == 1) {
                if (n >= kPropMonoOps) ok = false;
                else if (c == 1) {
                    m[1 + n] = -1;
                    mc[n] = imm[ii++];
                    ++n;
                } else if (c >= 101 && c <= 100 + ns) {
                    m[1 + n] = c - 101;
                    ++n;
                } else if (c > 100 + ns && c <= 100 + ns + np) {
                    m[1 + n] = -1;
                    mc[n] = params[c - 101 - ns];
                    ++n;
                } else ok = false;
            } else if (c != 5) ok = false;
        }
        const int len = code_off[k + 1] - code_off[k];
        if (ok && (len == 1 || (len >= 3 && len % 2 == 1)) && n == (len + 1) / 2) m[0] = n;
        else m[0] = 0;
    }
    // descriptors for the walk's register path: bits 0-2 operands of a chain (0: the reaction reads its one-species table),
    // bits 4-7 / 8-11 / 12-15 the species of operand 1 / 2 / 3 (15: the chain's one constant), table: bits 4-7 its species
    std::vector<int32_t> fast_i((size_t)std::max(nr, 1), 0);
    std::vector<double> fast_d((size_t)std::max(nr, 1), 0.0);
    bool fast_ok = ns <= 8;
    for (int k = 0; k < nr && fast_ok; ++k) {
        const int32_t *m = mono.data() + (size_t)k * (1 + kPropMonoOps);
        const double *mc = mono_c.data() + (size_t)k * kPropMonoOps;
        if (m[0] >= 1 && m[0] <= 3) {
            int d = m[0], nconst = 0;
            for (int i = 0; i < m[0]; ++i) {
                if (m[1 + i] < 0) {
                    ++nconst;
                    fast_d[(size_t)k] = mc[i];
                    d |= 15 << (4 + 4 * i);
                } else
                    d |= m[1 + i] << (4 + 4 * i);
            }
            if (nconst > 1) fast_ok = false;
            fast_i[(size_t)k] = d;
        } else if (m[0] == 0 && tab_species[k] >= 0 && tab_len > 0) {
            fast_i[(size_t)k] = tab_species[k] << 4;
        } else
            fast_ok = false;
    }
    hipStream_t st = ctx->stream;
    std::vector<int32_t> ib((size_t)(2 * (nr + 1) + nr + std::max(ncode, 1)));
    std::memcpy(ib.data(), code_off, sizeof(int32_t) * (size_t)(nr + 1));
    std::memcpy(ib.data() + (nr + 1), imm_off, sizeof(int32_t) * (size_t)(nr + 1));
    std::memcpy(ib.data() + 2 * (nr + 1), tab_species, sizeof(int32_t) * (size_t)nr);
    if (ncode > 0) std::memcpy(ib.data() + 2 * (nr + 1) + nr, code, sizeof(int32_t) * (size_t)ncode);
    const int np_pad = std::max(np, 1), nimm_pad = std::max(nimm, 1);
    const size_t ntab = tab_len > 0 ? (size_t)nr * (size_t)tab_len : 0;
    std::vector<double> db((size_t)np_pad + (size_t)nimm_pad + std::max<size_t>(ntab, 1), 0.0);
    if (np > 0) std::memcpy(db.data(), params, sizeof(double) * (size_t)np);
    if (nimm > 0) std::memcpy(db.data() + np_pad, imm, sizeof(double) * (size_t)nimm);
    if (ntab > 0) std::memcpy(db.data() + np_pad + nimm_pad, tab, sizeof(double) * ntab);
    ctx->prop_mono_off = ib.size();
    ib.insert(ib.end(), mono.begin(), mono.end());
    ctx->prop_monoc_off = db.size();
    db.insert(db.end(), mono_c.begin(), mono_c.end());
    HIP_TRY_P(ctx->d_prop_i.reserve(ib.size(), false));
    HIP_TRY_P(ctx->d_prop_d.reserve(db.size(), false));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_i.p, ib.data(), ib.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_d.p, db.data(), db.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY_P(ctx->d_prop_fast_i.reserve(fast_i.size(), false));
    HIP_TRY_P(ctx->d_prop_fast_d.reserve(fast_d.size(), false));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_fast_i.p, fast_i.data(), fast_i.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_fast_d.p, fast_d.data(), fast_d.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY_P(hipStreamSynchronize(st));
    ctx->prop_fast = fast_ok;
    ctx->prop_ns = ns;
    ctx->prop_nr = nr;
    ctx->prop_np = np;
    ctx->prop_np_pad = np_pad;
    ctx->prop_nimm_pad = nimm_pad;
    ctx->prop_tab_len = tab_len > 0 ? tab_len : 0;
    ctx->prop_ncode = ncode;
    ctx->prop_nimm = nimm;
    ctx->prop_light = light;
    ctx->prop_light_tab = light_tab;
    ctx->prop_ready = true;
    return 0;
}

// Two-species tables of the program just set (a compiled-in CUSTOMPROP the host tabulated, include/kfsp.h)
int prop_set_tables2(kfsp_ctx *ctx, int32_t nr, const int32_t *s1, const int32_t *s2, const int32_t *n1, const int32_t *n2,
                     const int64_t *off, int64_t len, const double *tab2)
{
    if (!ctx->prop_ready || nr != ctx->prop_nr) {
        ctx->err = "two-species tables: no propensity program with this many reactions";
        return -2;
    }
    std::vector<int32_t> ti((size_t)nr * 4);
    std::vector<long long> to((size_t)nr, 0);
    bool any = false;
    for (int k = 0; k < nr; ++k) {
        ti[4 * k] = ti[4 * k + 1] = -1;
        ti[4 * k + 2] = ti[4 * k + 3] = 0;
        if (s1[k] < 0) continue;
        if (s1[k] >= ctx->prop_ns || s2[k] < 0 || s2[k] >= ctx->prop_ns || s1[k] == s2[k] || n1[k] < 1 || n2[k] < 1 || off[k] < 0 ||
            off[k] + (int64_t)n1[k] * n2[k] > len) {
            ctx->err = "two-species tables: species / extent / offset out of range";
            return -3;
        }
        ti[4 * k] = s1[k];
        ti[4 * k + 1] = s2[k];
        ti[4 * k + 2] = n1[k];
        ti[4 * k + 3] = n2[k];
        to[(size_t)k] = off[k];
        any = true;
    }
    if (!any) return 0;
    hipStream_t st = ctx->stream;
    HIP_TRY_P(ctx->d_prop_t2i.reserve(ti.size(), false));
    HIP_TRY_P(ctx->d_prop_t2o.reserve(to.size(), false));
    HIP_TRY_P(ctx->d_prop_t2d.reserve((size_t)std::max<int64_t>(len, 1), false));
    HIP_TRY_P(ctx->d_prop_oob.reserve(32, false));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_t2i.p, ti.data(), ti.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_t2o.p, to.data(), to.size() * sizeof(long long), hipMemcpyHostToDevice, st));
    HIP_TRY_P(hipMemcpyAsync(ctx->d_prop_t2d.p, tab2, (size_t)len * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY_P(hipMemsetAsync(ctx->d_prop_oob.p, 0, 32 * sizeof(int32_t), st));
    HIP_TRY_P(hipStreamSynchronize(st));
    ctx->prop_has_tab2 = true;
    return 0;
}

int prop_check_overflow(kfsp_ctx *ctx)
{
    if (!ctx->prop_has_tab2) return 0;
    int32_t h[17];
    hipStream_t st = ctx->stream;
    HIP_TRY_P(hipMemcpyAsync(h, ctx->d_prop_oob.p, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY_P(hipStreamSynchronize(st));
    if (!h[0]) return 0;
    std::memcpy(ctx->prop_missed, h + 1, sizeof(ctx->prop_missed));
    HIP_TRY_P(hipMemsetAsync(ctx->d_prop_oob.p, 0, 32 * sizeof(int32_t), st));
    ctx->err = "a population lies beyond a two-species propensity table (kfsp_propensity_overflow says which; enlarge and repeat)";
    return -16;
}

// states already on the device (n x lds int32) -> columns on the device
int prop_eval_device(kfsp_ctx *ctx, int64_t n, const int32_t *d_state, int lds, double *d_off, int ldo, double *d_diag)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_propensities, dim3((int)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, prop_dev(ctx), n, d_state,
                       lds, d_off, ldo, d_diag);
    return 0;
}

// host arrays in, host arrays out
int prop_eval_host(kfsp_ctx *ctx, int32_t n, const int32_t *state, int32_t lds, double *offdiag, int32_t ldo, double *diag)
{
    hipStream_t st = ctx->stream;
    const size_t ns_b = (size_t)n * (size_t)lds * 4, no_b = (size_t)n * (size_t)ldo * 8, nd_b = (size_t)n * 8;
    HIP_TRY_P(ctx->d_os1.reserve(ns_b + no_b + nd_b + 1024, false));
    char *base = ctx->d_os1.p;
    double *d_off = reinterpret_cast<double *>(base);
    double *d_diag = reinterpret_cast<double *>(base + no_b);
    int32_t *d_state = reinterpret_cast<int32_t *>(base + no_b + nd_b);
    HIP_TRY_P(hipMemcpyAsync(d_state, state, ns_b, hipMemcpyHostToDevice, st));
    if (int rc = prop_eval_device(ctx, n, d_state, lds, d_off, ldo, d_diag)) return rc;
    HIP_TRY_P(hipMemcpyAsync(offdiag, d_off, no_b, hipMemcpyDeviceToHost, st));
    HIP_TRY_P(hipMemcpyAsync(diag, d_diag, nd_b, hipMemcpyDeviceToHost, st));
    HIP_TRY_P(hipStreamSynchronize(st));
    return prop_check_overflow(ctx);
}

}  // namespace kfsp
