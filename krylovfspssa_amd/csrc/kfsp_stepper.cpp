// kfsp_dgexpv: the adaptive Krylov-FSP time loop (host control of the hot path).
//
// A pure client of the C ABI in include/kfsp.h: every O(N) operation is one of
// the device entry points (begin_step / arnoldi / combine / restore_w), the
// (m+2)^2 Pade exponential is the host routine kfsp_padm, and everything else
// here is scalar bookkeeping that decides the trajectory.  It restates
// DGEXPV_FSP of the reference (src/fsp/KrylovSolver.f90:151-573): EXPOKIT's
// step-size control, the Niesen-Wright choice between a new step size and a
// new Krylov dimension, and the FSP mass criterion with its step shrinking and
// the hand-over to the state-space code (drop / expand callbacks).
//
// kfsp_dgexpv_replay runs the same loop in lock step with a recorded run: every
// decision is computed here as usual, compared with the record, and the record's
// choice is the one carried out; differences are reported as kfsp_fork entries.
#include "../../include/kfsp.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int kMMax = KFSP_M_MAX;   // M_MAX  :47
constexpr int kMMin = 10;           // M_MIN  :47
constexpr int kIdeg = 6;            // IDEG   :82
constexpr double kDelta = 1.2;      // DELTA  :85
constexpr double kGamma = 0.9;      // GAMMA  :87
constexpr int kQiop = 2;            // QIOP   :137
constexpr double kBreakTol = 1.0e-7;   // BREAK_TOL :173
constexpr int64_t kRefNmax = 6291469;  // NMAX, StateSpace.f90:10

// x**k for an INTEGER k as Fortran compilers expand it
double int_power(double x, int k)
{
    const bool inv = k < 0;
    unsigned e = inv ? (unsigned)(-(long)k) : (unsigned)k;
    double r = 1.0;
    for (double b = x; e; e >>= 1, b *= b)
        if (e & 1u) r *= b;
    return inv ? 1.0 / r : r;
}

// keep two significant digits (:186-187 with bias 0.55, :344-345 with none)
double two_digits(double t, double bias)
{
    static const double sqr1 = std::sqrt(0.1);
    const double unit = int_power(10.0, (int)std::round(std::log10(t) - sqr1) - 1);
    return std::trunc(t / unit + bias) * unit;
}

double clamp_step(double remaining, double t_step, double proposal)
{
    return std::min(remaining, std::max(t_step / 5.0, std::min(5.0 * t_step, proposal)));
}

struct Emit {
    const kfsp_fsp_ops *ops;
    void operator()(int ev, std::initializer_list<double> v) const
    {
        if (ops && ops->log) ops->log(ops->user, ev, v.begin(), (int)v.size());
    }
};

struct Stepper {
    kfsp_ctx *ctx;
    const kfsp_fsp_ops *ops;
    Emit emit;
    double fsptol, krytol, t_out, sgn;
    int n_reactions;
    int64_t n = 0;        // current FSP size
    int64_t nnz = 0;      // the reference's estimate (M+1)*N  :196,:537

    // flops of one choice of (tau, m), KRYLOV_COST :618-639.  The reference
    // evaluates the integer factors in default INTEGER; within its own capacity
    // (N <= NMAX) that 32-bit wrap-around is reproduced so that the comparison
    // at :362 takes the same branch, beyond it the arithmetic is exact.
    double cost(double t_now, double tau, int m, double hnorm) const
    {
        // INT() of a non-finite quotient (HNORM = 0, underflow) is unspecified in Fortran and
        // undefined in C++; the reference's build lands on a negative number, i.e. MAX(0, ..) = 0
        const double q = std::log(tau * hnorm) / std::log(2.0);
        const int lg = (std::isfinite(q) && std::fabs(q) < 1.0e9) ? 2 + (int)q : 0;
        const double nom = 25.0 / 3.0 + (double)std::max(0, lg);
        const int64_t a = 2 * (int64_t)(m + 1) * nnz;
        const int64_t b = (int64_t)(5 * m + 4 * kQiop * m + 2 * kQiop - 2 * kQiop * kQiop + 7) * n;
        double ab;
        if (n <= kRefNmax)
            ab = (double)(int32_t)((uint32_t)(uint64_t)a + (uint32_t)(uint64_t)b);
        else
            ab = (double)(a + b);
        const double per_step = ab + 2.0 * nom * (m + 2) * (m + 2) * (m + 2);
        return std::round((t_out - t_now) / tau) * per_step;
    }
};

}  // namespace

namespace {

// lock-step state: the script cursor and the fork list (see kfsp_replay in kfsp.h)
struct Lockstep {
    kfsp_replay *rp;
    bool out_of_step = false;
    const double *next(double kind)
    {
        if (rp->rows_used >= rp->n_rows) {
            out_of_step = true;
            return nullptr;
        }
        const double *r = rp->script + 4 * rp->rows_used;
        if (r[0] != kind) {
            out_of_step = true;
            return nullptr;
        }
        ++rp->rows_used;
        return r;
    }
    void fork(int step, int kind, double own0, double own1, double f0, double f1, double lhs, double rhs)
    {
        if (rp->forks && rp->n_forks < rp->max_forks) {
            kfsp_fork &k = rp->forks[rp->n_forks];
            k.step = step;
            k.kind = kind;
            k.own[0] = own0;
            k.own[1] = own1;
            k.forced[0] = f0;
            k.forced[1] = f1;
            k.lhs = lhs;
            k.rhs = rhs;
        }
        ++rp->n_forks;
    }
};

int dgexpv_impl(kfsp_ctx *ctx, double t, double fsptol, double krytol, int n_reactions,
                const kfsp_fsp_ops *ops, kfsp_stats *stats, kfsp_replay *rp)
{
    if (!ctx) return -1;
    if (!(t != 0.0) || !std::isfinite(t)) return -2;
    if (!(fsptol > 0.0)) return -3;
    if (!(krytol >= 0.0)) return -4;
    if (n_reactions < 1) return -5;

    Stepper S{ctx, ops, Emit{ops}, fsptol, krytol, std::fabs(t), t < 0 ? -1.0 : 1.0, n_reactions};
    int rc = kfsp_num_states(ctx, &S.n);
    if (rc) return rc;
    if (S.n < 2) return -1;
    S.nnz = (int64_t)(n_reactions + 1) * S.n;
    const Emit &emit = S.emit;
    Lockstep ls{rp};
    bool diverged = false;
    if (rp) {
        if (!rp->script || rp->n_rows < 1) return -8;
        rp->rows_used = 0;
        rp->n_forks = 0;
        rp->n_safe_extensions = 0;
        rp->max_wsum_diff = 0.0;
    }

    kfsp_stats st;
    std::memset(&st, 0, sizeof(st));
    st.step_min = S.t_out;

    // machine epsilon by the 4/3 trick (:166-170) and the tolerance floor (:171)
    double eps;
    {
        volatile double a = 4.0 / 3.0, b, c;
        do {
            b = a - 1.0;
            c = b + b + b;
            eps = std::fabs(c - 1.0);
        } while (eps == 0.0);
    }
    if (S.krytol <= eps) S.krytol = std::sqrt(eps);
    const double rndoff = eps;   // ANORM = 1  :129,:172
    const double t_out = S.t_out;

    const int mhmax = kMMax + 2;
    std::vector<double> H((size_t)mhmax * mhmax, 0.0), Hold((size_t)mhmax * mhmax, 0.0), E((size_t)mhmax * mhmax, 0.0);

    int m = kMMin, m_new = kMMin, m_old = 0, mh = m + 2;
    double beta = 0.0;
    if ((rc = kfsp_begin_step(ctx, &beta))) return rc;   // BETA = ||w||  :177
    const double vnorm = beta;
    st.hump = beta;
    emit(KFSP_EV_READY, {0.0, 0.0, beta, (double)S.n});

    // the very first step size (:182-187)
    double t_new;
    {
        const double p1 = S.krytol * int_power((m + 1) / 2.72, m + 1) * std::sqrt(2.0 * 3.14 * (m + 1));
        t_new = std::pow(p1 / (4.0 * beta), 1.0 / (double)m);
        t_new = two_digits(t_new, 0.55);
    }

    double t_now = 0.0, t_step = 0.0, t_old = 0.0;
    double omega = 0.0, omega_old = 0.0;    // used before set in the reference (:313); zero here
    double order = 0.0, kappa = 2.0, hnorm = 0.0, err_loc = 0.0, avnorm = 0.0;
    double wsum = 0.0, wsum_old = 1.0, fsp_err = 0.0, fsp_err_old = 0.0, tau_old = 0.0, fsp_order = 2.0;
    bool order_is_default = true, kappa_is_default = true, need_expand = false, m_changed = false;
    int jold = 1, dim_rejects = 0, step_rejects = 0, k1 = 2, mbrkdwn = m, mx = 0, ns = 0;
    bool first_begin_done = true;   // begin_step above already prepared v1 for step 1

    while (t_now < t_out) {                                   // label 100
        t_step = std::min(t_out - t_now, t_new);
        m = (int)std::min<int64_t>(S.n - 1, m_new);
        bool rec_breakdown = false;
        if (rp) {
            const double *row = ls.next(1.0);
            if (!row) break;
            rec_breakdown = row[1] == 0.0;
            if (row[1] > 0.0 && row[1] != t_step) {
                ls.fork(st.nstep + 1, KFSP_FORK_BEGIN_TAU, t_step, 0.0, row[1], 0.0, t_new, t_out - t_now);
                t_step = row[1];
            }
            if ((int)row[2] != m) ls.fork(st.nstep + 1, KFSP_FORK_BEGIN_M, (double)m, 0.0, row[2], 0.0, (double)m_new, (double)(S.n - 1));
            m = (int)row[2];
            if (m < 1 || m > kMMax) return -8;
        }
        mbrkdwn = m;
        k1 = 2;
        mh = m + 2;
        ++st.nstep;
        if (!first_begin_done)
            if ((rc = kfsp_begin_step(ctx, &beta))) return rc;   // v1 = w/beta  :223-226
        first_begin_done = false;
        std::fill(H.begin(), H.begin() + (size_t)mh * mh, 0.0);
        step_rejects = 0;
        emit(KFSP_EV_BEGIN_IOP, {});

        bool redo_arnoldi = true;
        for (;;) {
            if (redo_arnoldi) {                                // label 101
                const int cols = (jold <= m) ? m - jold + 1 : 0;
                if ((rc = kfsp_arnoldi(ctx, m, jold, kQiop, kBreakTol, H.data(), mh, &mbrkdwn, &k1, &avnorm))) return rc;
                st.nmult += (k1 != 0) ? cols + 1 : mbrkdwn - jold + 1;
                if (rp && (k1 == 0) != rec_breakdown && jold == 1) {
                    // a happy breakdown (h(j+1,j) <= 1e-7, :249) on one side only: the step sizes are no
                    // longer comparable (T_STEP becomes T_OUT - T_NOW, :254)
                    ls.fork(st.nstep, KFSP_FORK_BREAKDOWN, k1 == 0 ? (double)mbrkdwn : 0.0, 0.0, rec_breakdown ? 1.0 : 0.0, 0.0,
                            H[(size_t)(std::max(mbrkdwn, 1) - 1) * mh + std::max(mbrkdwn, 1)], kBreakTol);
                    diverged = true;
                    break;
                }
                if (k1 == 0) {                                 // happy breakdown :250-254
                    st.ibrkflag = 1;
                    st.tbrkdwn = t_now;
                    t_step = t_out - t_now;
                }
                redo_arnoldi = false;
            }
            // label 401: exp(t_step * H) of order MBRKDWN + K1   :270-277
            ++st.nexph;
            mx = mbrkdwn + k1;
            {
                const auto t0 = std::chrono::steady_clock::now();
                rc = kfsp_padm(kIdeg, mx, S.sgn * t_step, H.data(), mh, E.data(), &ns, &hnorm);
                kfsp_add_timer(ctx, KFSP_T_HOST_PADE, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                if (rc) return 3000 - rc;
            }
            st.nscale += ns;
            // local error estimate :290-305
            if (k1 == 0) {
                err_loc = S.krytol;
            } else {
                const double p1 = std::fabs(E[(size_t)m]) * beta;
                const double p2 = std::fabs(E[(size_t)m + 1]) * beta * avnorm;
                if (p1 > 10.0 * p2) err_loc = p2;
                else if (p1 > p2) err_loc = (p1 * p2) / (p1 - p2);
                else err_loc = p1;
            }
            if (std::isnan(err_loc)) {                         // :307-310
                t_step /= 5.0;
                if (rp) {
                    const double *row = ls.next(2.0);
                    if (!row) break;
                    if (row[1] != 1.0 || row[2] != t_step) ls.fork(st.nstep, KFSP_FORK_KRYLOV_VALUE, t_step, 0.0, row[2], row[1], err_loc, 0.0);
                    t_step = row[2];
                }
                continue;
            }
            omega_old = omega;
            omega = err_loc / (S.krytol * t_step);             // :314
            // order of the method, from two step sizes at equal m  :316-324
            if (m == m_old && t_step != t_old && step_rejects >= 1) {
                order = std::max(1.0, std::log(omega / omega_old) / std::log(t_step / t_old));
                order_is_default = false;
            } else if (order_is_default || step_rejects == 0) {
                order = (double)m / 4.0;
                order_is_default = true;
            } else {
                order_is_default = true;
            }
            // error reduction per extra basis vector, from two m at equal step  :326-334
            if (m != m_old && t_step == t_old && step_rejects >= 1) {
                kappa = std::max(1.1, std::pow(omega / omega_old, 1.0 / (double)(m_old - m)));
                kappa_is_default = false;
            } else if (kappa_is_default || step_rejects == 0) {
                kappa = 2.0;
                kappa_is_default = true;
            } else {
                kappa_is_default = true;
            }
            t_old = t_step;
            m_old = m;
            const double remaining = t_out - t_now;
            const double t_opt = clamp_step(remaining, t_step, kGamma * t_step * std::pow(omega, -1.0 / order));
            float c_step = 0.0f, c_dim = 0.0f;
            if ((m == kMMax && omega > kDelta) || dim_rejects > 4) {   // :339-346
                t_new = two_digits(t_opt, 0.0);
                m_changed = false;
            } else {                                                   // :348-372
                int m_opt = std::max({kMMin, 3 * m / 4, m + (int)std::ceil(std::log(omega) / std::log(kappa))});
                m_opt = std::min({m_opt, kMMax, (int)std::ceil(4.0 * m / 3.0) + 1});
                // COST1/COST2 are default REAL in the reference (:109)
                c_step = (float)S.cost(t_now, t_opt, m, hnorm);
                c_dim = (float)S.cost(t_now, t_step, m_opt, hnorm);
                if (c_step <= c_dim) {
                    t_new = two_digits(t_opt, 0.0);
                    m_new = m;
                    m_changed = false;
                } else {
                    m_new = m_opt;
                    t_new = t_step;
                    m_changed = true;
                }
            }
            // what this run chooses: 0 accept, 1 new step size (same basis), 2 new dimension  :375-433
            int code = 0, next_m = m;
            double next_t = t_step;
            if (k1 != 0 && omega > kDelta) {
                if (!m_changed) {
                    code = 1;
                    next_t = two_digits(clamp_step(remaining, t_step, t_new), 0.55);
                } else {
                    code = 2;
                    next_m = m_new;
                    next_t = std::min(remaining, t_new);
                }
            }
            if (rp) {
                const double *row = ls.next(2.0);
                if (!row) break;
                const int rcode = (int)row[1], rm = (int)row[3];
                const double rt = row[2];
                if ((rcode != 0) != (code != 0))
                    ls.fork(st.nstep, KFSP_FORK_KRYLOV_TEST, (double)code, code == 2 ? (double)next_m : next_t, (double)rcode,
                            rcode == 2 ? (double)rm : rt, omega, kDelta);
                else if (rcode != code)
                    ls.fork(st.nstep, KFSP_FORK_KRYLOV_CHOICE, (double)code, code == 2 ? (double)next_m : next_t, (double)rcode,
                            rcode == 2 ? (double)rm : rt, (double)c_step, (double)c_dim);
                else if ((code == 1 && rt != next_t) || (code == 2 && (rm != next_m || rt != next_t)))
                    ls.fork(st.nstep, KFSP_FORK_KRYLOV_VALUE, next_t, (double)next_m, rt, (double)rm, omega, kDelta);
                if (rp->safe && rcode == 0 && code != 0 && m >= kMMax)
                    ls.fork(st.nstep, KFSP_FORK_UNSAFE_ACCEPT, (double)code, next_t, 0.0, t_step, omega, kDelta);
                if (rp->safe && rcode == 0 && code != 0 && m < kMMax) {
                    // The recorded run accepted here, this run's own error test does not: carrying the
                    // step out would inject an error the record does not have.  Keep the recorded step
                    // size (the time grid stays aligned) and enlarge the basis until the test passes;
                    // the accept row is read again after the extension.
                    --rp->rows_used;
                    ++rp->n_safe_extensions;
                    const int want = (code == 2 && m_new > m) ? m_new : m + std::max(4, m / 8);
                    code = 2;
                    next_m = std::min(kMMax, want);
                    next_t = t_step;
                    m_new = next_m;
                    t_new = t_step;
                    m_changed = true;
                } else {
                    // keep the proposals consistent with the branch the record took
                    if (rcode == 1 && code == 2) {
                        m_new = m;
                        t_new = two_digits(t_opt, 0.0);
                        m_changed = false;
                    }
                    if (rcode == 2) {
                        if (code != 2) t_new = t_step;
                        m_new = rm;
                        m_changed = true;
                        if (rm < 1 || rm > kMMax) return -8;
                    }
                    code = rcode;
                    next_t = rt;
                    next_m = rm;
                }
            }
            if (code == 1) {                                   // new step size, same basis  :377-399
                ++st.nreject;
                emit(KFSP_EV_REJECT_STEP, {t_old, err_loc, kDelta * t_old * S.krytol, next_t});
                t_step = next_t;
                ++step_rejects;
                continue;
            }
            if (code == 2) {
                // new dimension: keep the basis, re-lay H, resume at column m_old  :400-432
                ++st.nreject;
                ++dim_rejects;
                std::copy(H.begin(), H.begin() + (size_t)mh * mh, Hold.begin());
                m = next_m;
                mbrkdwn = m;
                k1 = 2;
                mh = m + 2;
                t_step = next_t;
                std::fill(H.begin(), H.begin() + (size_t)mh * mh, 0.0);
                for (int j = 1; j <= m_old; ++j)
                    for (int i = 1; i <= j + 1; ++i) {
                        const size_t dst = (size_t)(j - 1) * (m + 2) + (i - 1);
                        if (dst < H.size()) H[dst] = Hold[(size_t)(j - 1) * (m_old + 2) + (i - 1)];
                    }
                jold = m_old;
                emit(KFSP_EV_DIM_CHANGE, {err_loc, kDelta * t_old * S.krytol, (double)m});
                redo_arnoldi = true;
                continue;
            }
            break;   // Krylov step accepted
        }
        if (rp && (ls.out_of_step || diverged)) break;
        dim_rejects = 0;                                       // :435-439
        jold = 1;
        if (err_loc < 1.0e-16) t_new = std::max(t_new, 2.0 * t_step);
        mx = mbrkdwn + std::max(0, k1 - 1);

        // FSP criterion: enough probability mass must survive  :442-495
        bool to_ssa = false;
        for (int fsp_rejects = 0;;) {
            if ((rc = kfsp_combine(ctx, mx, beta, E.data(), &wsum))) return rc;
            ++st.n_wsum;
            emit(KFSP_EV_WSUM, {wsum});
            fsp_err = wsum_old - wsum;
            const double bound = 1.0 - (t_now + t_step) * S.fsptol / t_out;   // :458 with FERRORBOUND :615
            // 0 accept, 1 retry with a smaller step, 2 give up shrinking and expand now (:466-470)
            int code = wsum >= bound ? 0 : (fsp_rejects + 1 >= 5 ? 2 : 1);
            const double *row = nullptr;
            if (rp) {
                row = ls.next(3.0);
                if (!row) break;
                rp->max_wsum_diff = std::max(rp->max_wsum_diff, std::fabs(wsum - row[3]));
                if ((int)row[1] != code) ls.fork(st.nstep, KFSP_FORK_FSP_TEST, (double)code, wsum, row[1], row[3], wsum, bound);
                code = (int)row[1];
            }
            if (code == 0) break;
            need_expand = true;
            ++fsp_rejects;
            if (code == 2) {
                if ((rc = kfsp_restore_w(ctx, beta))) return rc;
                --st.nstep;
                to_ssa = true;
                break;
            }
            fsp_order = (fsp_rejects == 1) ? 2.0
                                           : std::log(fsp_err / fsp_err_old) / std::log(t_step / tau_old) - 1.0;
            const double t_fsp = kGamma * t_step * std::pow(S.fsptol * t_step / (fsp_err * t_out), 1.0 / fsp_order);
            fsp_err_old = fsp_err;
            tau_old = t_step;
            t_step = std::min(t_out - t_now, std::max(t_step / 5.0, std::min(0.9 * t_step, t_fsp)));
            t_step = two_digits(t_step, 0.55);
            if (row) {
                if (row[2] != t_step) ls.fork(st.nstep, KFSP_FORK_FSP_TAU, t_step, 0.0, row[2], 0.0, fsp_err, fsp_order);
                t_step = row[2];
            }
            ++st.nexph;                                        // :489-493, order MX this time
            {
                const auto t0 = std::chrono::steady_clock::now();
                rc = kfsp_padm(kIdeg, mx, S.sgn * t_step, H.data(), mh, E.data(), &ns, nullptr);
                kfsp_add_timer(ctx, KFSP_T_HOST_PADE, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                if (rc) return 3000 - rc;
            }
            st.nscale += ns;
        }
        if (rp && ls.out_of_step) break;
        int64_t n_expected = -1;
        if (rp) {
            const double *row = ls.next(4.0);
            if (!row) break;
            n_expected = (int64_t)row[1];
            // T_NEW as the recorded run printed it with this step (:647)
            if (!to_ssa && row[2] > 0.0 && row[2] != t_new) {
                ls.fork(st.nstep, KFSP_FORK_T_NEW, t_new, 0.0, row[2], 0.0, omega, order);
                t_new = row[2];
            }
        }

        if (!to_ssa) {
            t_now += t_step;                                   // :498-499
            wsum_old = wsum;
            emit(KFSP_EV_STEP, {(double)st.nstep, (double)S.n, t_step, t_new, t_now, (double)m});
            if (t_now >= t_out) break;                         // :506 (STEP_MIN/MAX :543-544 are not updated for the last step)
            if (st.nstep > 1 && !need_expand) {                // DROP_STATES  :509-512
                const double dsum = wsum - (1.0 - t_now * S.fsptol / t_out);
                if (dsum > 0.0 && ops && ops->drop) {
                    int64_t nn = S.n;
                    ++st.n_drop_calls;
                    const auto t0 = std::chrono::steady_clock::now();
                    rc = ops->drop(ops->user, dsum, &nn);
                    kfsp_add_timer(ctx, KFSP_T_CALLBACKS, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                    if (rc) return rc;
                    S.n = nn;
                }
            }
        }
        if (need_expand && t_now < t_out) {                    // label 404, :518-534
            if (st.nstep == 1) t_new = t_step;
            const double t_ssa = std::min(t_new, t_out - t_now);
            emit(KFSP_EV_CALL_SSA, {t_ssa});
            if (!ops || !ops->expand) {
                st.t_now = t_now;
                if (stats) *stats = st;
                return 10;
            }
            int64_t nn = S.n;
            ++st.n_expand;
            {
                const auto t0 = std::chrono::steady_clock::now();
                rc = ops->expand(ops->user, t_ssa, &nn);
                kfsp_add_timer(ctx, KFSP_T_CALLBACKS, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                if (rc) return rc;
            }
            S.n = nn;
            need_expand = false;
        }
        // bookkeeping for the next step  :537-548
        if ((rc = kfsp_num_states(ctx, &S.n))) return rc;
        S.nnz = (int64_t)(n_reactions + 1) * S.n;
        if (rp && n_expected >= 0 && n_expected != S.n) {
            ls.fork(st.nstep, KFSP_FORK_FSP_SIZE, (double)S.n, 0.0, (double)n_expected, 0.0, 0.0, 0.0);
            if (rp->safe == 2) {
                // the state lists have parted: from here on the record describes another problem
                st.t_now = S.sgn * t_now;
                if (stats) *stats = st;
                return 21;
            }
        }
        if ((rc = kfsp_begin_step(ctx, &beta))) return rc;     // BETA = ||w||  :540 (and v1 of the next step)
        first_begin_done = true;
        emit(KFSP_EV_READY, {(double)st.nstep, t_now, beta, (double)S.n});
        st.hump = std::max(st.hump, beta);
        err_loc = std::max(err_loc, rndoff);
        st.step_min = std::min(st.step_min, t_step);
        st.step_max = std::max(st.step_max, t_step);
        st.s_error += err_loc;
        st.x_error = std::max(st.x_error, err_loc);
        t_new = two_digits(t_new, 0.55);
    }

    st.mbrkdwn = mbrkdwn;
    st.t_now = S.sgn * t_now;
    st.beta = beta / vnorm;
    st.hump = st.hump / vnorm;
    if (stats) *stats = st;
    if (rp && ls.out_of_step) return 20;   // the record ended early or does not describe this run
    if (diverged) return 21;               // a happy breakdown on one side only
    return 0;
}

}  // namespace

extern "C" int kfsp_dgexpv(kfsp_ctx *ctx, double t, double fsptol, double krytol, int n_reactions,
                           const kfsp_fsp_ops *ops, kfsp_stats *stats)
try {
    return dgexpv_impl(ctx, t, fsptol, krytol, n_reactions, ops, stats, nullptr);
} catch (...) {
    return 4000;   // host allocation failure or an exception out of a callback: never across the C boundary
}

extern "C" int kfsp_dgexpv_replay(kfsp_ctx *ctx, double t, double fsptol, double krytol, int n_reactions,
                                  const kfsp_fsp_ops *ops, kfsp_stats *stats, kfsp_replay *replay)
try {
    if (!replay) return -8;
    return dgexpv_impl(ctx, t, fsptol, krytol, n_reactions, ops, stats, replay);
} catch (...) {
    return 4000;
}
