// kfsp_padm: exp(t*H) of the small Hessenberg matrix on the HOST (by design the
// tiny (m+2)^2 exponential never goes to the GPU): (ideg,ideg) Pade approximant
// with scaling and squaring, the semantics of DGPADM / DGPADMnorm
// (src/expokit/dgpadm.f:2-169, :171-339) with our own products and LU.
#include "../../include/kfsp.h"

#include <sched.h>
#include <xmmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// An m x m column-major matrix over storage the caller of kfsp_padm keeps between calls (Workspace below): five fresh
// 83 KB vectors per exponential cost more than its arithmetic - glibc hands blocks of that size back to the system on
// free, so every call paid ~100 page faults again (137 us of a 980 us call in the build container, 420 of 690 us on the
// MI355X boxes' host; found with kfsp_padm_profile in round 4).
struct Span {
    double *p = nullptr;
    size_t n = 0;
    double *data() { return p; }
    const double *data() const { return p; }
    size_t size() const { return n; }
    double &operator[](size_t i) { return p[i]; }
    double operator[](size_t i) const { return p[i]; }
    double *begin() { return p; }
    double *end() { return p + n; }
    const double *begin() const { return p; }
    const double *end() const { return p + n; }
};
struct Dense {
    int m;
    Span a;
    Dense(int m_, double *store, bool zero) : m(m_)
    {
        a.p = store;
        a.n = (size_t)m_ * m_;
        if (zero) std::memset(store, 0, a.n * sizeof(double));
    }
    double &operator()(int i, int j) { return a[(size_t)j * m + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * m + i]; }
};
inline void swap_storage(Dense &x, Dense &y) { std::swap(x.a.p, y.a.p); }

// C = alpha * A * B, column major.  Column j of C is a linear combination of
// the columns of A; four of them are folded per sweep over the column and zero
// coefficients are skipped (the Hessenberg matrices of the IOP process are
// banded, only the squaring phase is dense).  The same body is compiled twice:
// baseline x86-64 and AVX2+FMA, picked once at run time.
#define KFSP_MATMUL_BODY                                                                      \
    const int m = A.m;                                                                        \
    const double *a = A.a.data();                                                             \
    const double *b = B.a.data();                                                             \
    double *c = C.a.data();                                                                   \
    for (int j = jbeg; j < jend; ++j) {                                                       \
        double *__restrict__ cj = c + (size_t)j * m;                                          \
        for (int i = 0; i < m; ++i) cj[i] = 0.0;                                              \
        int idx[4];                                                                           \
        double coef[4];                                                                       \
        int nk = 0;                                                                           \
        for (int k = 0; k <= m; ++k) {                                                        \
            if (k < m) {                                                                      \
                const double v = alpha * b[(size_t)j * m + k];                                \
                if (v == 0.0) continue;                                                       \
                idx[nk] = k;                                                                  \
                coef[nk] = v;                                                                 \
                ++nk;                                                                         \
            }                                                                                 \
            if (nk == 4 || (k == m && nk > 0)) {                                              \
                for (int q = nk; q < 4; ++q) {                                                \
                    idx[q] = idx[0];                                                          \
                    coef[q] = 0.0;                                                            \
                }                                                                             \
                const double *__restrict__ a0 = a + (size_t)idx[0] * m;                       \
                const double *__restrict__ a1 = a + (size_t)idx[1] * m;                       \
                const double *__restrict__ a2 = a + (size_t)idx[2] * m;                       \
                const double *__restrict__ a3 = a + (size_t)idx[3] * m;                       \
                const double c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];          \
                for (int i = 0; i < m; ++i) cj[i] += (c0 * a0[i] + c1 * a1[i]) + (c2 * a2[i] + c3 * a3[i]); \
                nk = 0;                                                                       \
            }                                                                                 \
        }                                                                                     \
    }

// (columns jbeg .. jend - 1 of the product: a column is made by the same expressions whoever makes it)
void matmul_base(double alpha, const Dense &A, const Dense &B, Dense &C, int jbeg, int jend) { KFSP_MATMUL_BODY }
__attribute__((target("avx2,fma"))) void matmul_avx2(double alpha, const Dense &A, const Dense &B, Dense &C, int jbeg, int jend)
{
    KFSP_MATMUL_BODY
}
__attribute__((target("avx512f,avx512vl,fma"))) void matmul_avx512(double alpha, const Dense &A, const Dense &B, Dense &C, int jbeg, int jend)
{
    KFSP_MATMUL_BODY
}
#undef KFSP_MATMUL_BODY

// Dense case (the squaring phase): two output columns share every load of A and
// eight columns of A are folded per sweep - 12 memory operations per 16 fused
// multiply-adds instead of 6 per 4.
#define KFSP_MATMUL_DENSE_BODY                                                                \
    const int m = A.m;                                                                        \
    const double *a = A.a.data();                                                             \
    const double *b = B.a.data();                                                             \
    double *c = C.a.data();                                                                   \
    int j = jbeg;                                                                             \
    /* four output columns share every load of A (8 loads + 8 column accesses per 32 FMAs); every element is still */ \
    /* the same expression as in the two-column sweep below, so the bits do not depend on which sweep made it */      \
    for (; j + 4 <= jend; j += 4) {                                                           \
        double *__restrict__ c0 = c + (size_t)j * m;                                          \
        double *__restrict__ c1 = c0 + m;                                                     \
        double *__restrict__ c2 = c1 + m;                                                     \
        double *__restrict__ c3 = c2 + m;                                                     \
        const double *b0 = b + (size_t)j * m, *b1 = b0 + m, *b2 = b1 + m, *b3 = b2 + m;      \
        for (int i = 0; i < m; ++i) {                                                         \
            c0[i] = 0.0;                                                                      \
            c1[i] = 0.0;                                                                      \
            c2[i] = 0.0;                                                                      \
            c3[i] = 0.0;                                                                      \
        }                                                                                     \
        int k = 0;                                                                            \
        for (; k + 8 <= m; k += 8) {                                                          \
            const double *__restrict__ ak = a + (size_t)k * m;                                \
            double p[8], q[8], r[8], s[8];                                                    \
            for (int t = 0; t < 8; ++t) {                                                     \
                p[t] = alpha * b0[k + t];                                                     \
                q[t] = alpha * b1[k + t];                                                     \
                r[t] = alpha * b2[k + t];                                                     \
                s[t] = alpha * b3[k + t];                                                     \
            }                                                                                 \
            for (int i = 0; i < m; ++i) {                                                     \
                const double x0 = ak[i], x1 = ak[i + m], x2 = ak[i + 2 * m], x3 = ak[i + 3 * m];       \
                const double x4 = ak[i + 4 * m], x5 = ak[i + 5 * m], x6 = ak[i + 6 * m], x7 = ak[i + 7 * m]; \
                c0[i] += ((p[0] * x0 + p[1] * x1) + (p[2] * x2 + p[3] * x3)) + ((p[4] * x4 + p[5] * x5) + (p[6] * x6 + p[7] * x7)); \
                c1[i] += ((q[0] * x0 + q[1] * x1) + (q[2] * x2 + q[3] * x3)) + ((q[4] * x4 + q[5] * x5) + (q[6] * x6 + q[7] * x7)); \
                c2[i] += ((r[0] * x0 + r[1] * x1) + (r[2] * x2 + r[3] * x3)) + ((r[4] * x4 + r[5] * x5) + (r[6] * x6 + r[7] * x7)); \
                c3[i] += ((s[0] * x0 + s[1] * x1) + (s[2] * x2 + s[3] * x3)) + ((s[4] * x4 + s[5] * x5) + (s[6] * x6 + s[7] * x7)); \
            }                                                                                 \
        }                                                                                     \
        for (; k < m; ++k) {                                                                  \
            const double *__restrict__ ak = a + (size_t)k * m;                                \
            const double p0 = alpha * b0[k], q0 = alpha * b1[k], r0 = alpha * b2[k], s0 = alpha * b3[k]; \
            for (int i = 0; i < m; ++i) {                                                     \
                c0[i] += p0 * ak[i];                                                          \
                c1[i] += q0 * ak[i];                                                          \
                c2[i] += r0 * ak[i];                                                          \
                c3[i] += s0 * ak[i];                                                          \
            }                                                                                 \
        }                                                                                     \
    }                                                                                         \
    if (jend != m) return; /* (a range that ends inside the matrix is whole groups of four) */ \
    for (; j + 2 <= m; j += 2) {                                                              \
        double *__restrict__ c0 = c + (size_t)j * m;                                          \
        double *__restrict__ c1 = c0 + m;                                                     \
        const double *b0 = b + (size_t)j * m, *b1 = b0 + m;                                   \
        for (int i = 0; i < m; ++i) {                                                         \
            c0[i] = 0.0;                                                                      \
            c1[i] = 0.0;                                                                      \
        }                                                                                     \
        int k = 0;                                                                            \
        for (; k + 8 <= m; k += 8) {                                                          \
            const double *__restrict__ ak = a + (size_t)k * m;                                \
            double p[8], q[8];                                                                \
            for (int r = 0; r < 8; ++r) {                                                     \
                p[r] = alpha * b0[k + r];                                                     \
                q[r] = alpha * b1[k + r];                                                     \
            }                                                                                 \
            for (int i = 0; i < m; ++i) {                                                     \
                const double x0 = ak[i], x1 = ak[i + m], x2 = ak[i + 2 * m], x3 = ak[i + 3 * m];       \
                const double x4 = ak[i + 4 * m], x5 = ak[i + 5 * m], x6 = ak[i + 6 * m], x7 = ak[i + 7 * m]; \
                c0[i] += ((p[0] * x0 + p[1] * x1) + (p[2] * x2 + p[3] * x3)) + ((p[4] * x4 + p[5] * x5) + (p[6] * x6 + p[7] * x7)); \
                c1[i] += ((q[0] * x0 + q[1] * x1) + (q[2] * x2 + q[3] * x3)) + ((q[4] * x4 + q[5] * x5) + (q[6] * x6 + q[7] * x7)); \
            }                                                                                 \
        }                                                                                     \
        for (; k < m; ++k) {                                                                  \
            const double *__restrict__ ak = a + (size_t)k * m;                                \
            const double p0 = alpha * b0[k], q0 = alpha * b1[k];                              \
            for (int i = 0; i < m; ++i) {                                                     \
                c0[i] += p0 * ak[i];                                                          \
                c1[i] += q0 * ak[i];                                                          \
            }                                                                                 \
        }                                                                                     \
    }                                                                                         \
    for (; j < m; ++j) {                                                                      \
        double *__restrict__ c0 = c + (size_t)j * m;                                          \
        for (int i = 0; i < m; ++i) c0[i] = 0.0;                                              \
        for (int k = 0; k < m; ++k) {                                                         \
            const double p0 = alpha * b[(size_t)j * m + k];                                   \
            const double *__restrict__ ak = a + (size_t)k * m;                                \
            for (int i = 0; i < m; ++i) c0[i] += p0 * ak[i];                                  \
        }                                                                                     \
    }

// (output columns [jbeg, jend): jbeg a multiple of 4, jend a multiple of 4 or m - the ranges the worker threads take)
void matmul_dense_base(double alpha, const Dense &A, const Dense &B, Dense &C, int jbeg, int jend) { KFSP_MATMUL_DENSE_BODY }
__attribute__((target("avx2,fma"))) void matmul_dense_avx2(double alpha, const Dense &A, const Dense &B, Dense &C, int jbeg, int jend)
{
    KFSP_MATMUL_DENSE_BODY
}
// 512-bit lanes where the host has them (EPYC 9575F of the MI355X boxes: 0.67 -> 0.51 ms at order 102).  The fused
// operations per element are those of the 256-bit build: the bits do not change with the width - checked bitwise against
// the previous library on 400 random matrices on both an AVX-512 Xeon and the EPYC (the check is how any change to these
// bodies has to be accepted: which product of "a b + c d" the compiler fuses is its choice, and a restructured
// body - a template over the column count was tried - can come out with other bits).
__attribute__((target("avx512f,avx512vl,fma"))) void matmul_dense_avx512(double alpha, const Dense &A, const Dense &B, Dense &C, int jbeg,
                                                                         int jend)
{
    KFSP_MATMUL_DENSE_BODY
}
#undef KFSP_MATMUL_DENSE_BODY

// The dense products of the squaring phase (13 or so per exponential, 2 m^3 flops each, m <= 102) are the host's share of a
// step - 19 % of the resident Goutsias run in round 3.  Output columns are independent and every element is the same
// expression whichever thread makes it, so the columns are dealt to a few threads in groups of four: the bits cannot change
// (tests/test_padm_bits.py pins them).  The workers spin for a moment after a product (the next one follows at once inside a
// call), then sleep until the next call.  KFSP_PADE_THREADS sets their number (default 4 on hosts with >= 16 CPUs, else none).
// Measured inside the resident Goutsias run on the MI355X boxes' host (profiles/r04_padm_profile.txt): the dense products
// 355 -> 259 -> 167 ms with 1 / 2 / 4 threads, the whole run 3.03 -> 2.95 -> 2.86 s.
// The CPUs this process may use, as they were when the library was loaded: by the time a solve runs, the host's OpenMP
// runtime has bound the calling thread to ONE core (OMP_PROC_BIND=close, fortran/kfsp_statespace.f90) and a thread started
// from it inherits that mask - four workers spinning on the caller's core turned a 0.65 s Pade share into 11 s
// (profiles/r04_padm_profile.txt).  Workers take this mask instead.
cpu_set_t g_initial_cpus;
bool g_have_initial_cpus = false;
__attribute__((constructor)) void remember_initial_cpus()
{
    g_have_initial_cpus = sched_getaffinity(0, sizeof(g_initial_cpus), &g_initial_cpus) == 0;
}

class PadePool {
public:
    static PadePool &get()
    {
        static PadePool *p = new PadePool;      // (never destroyed: its threads are detached and may sleep past exit)
        return *p;
    }
    int threads() const { return nthreads_; }
    // fn(t) for t = 0 .. threads() - 1, t = 0 on the caller
    template <class F>
    void run(const F &fn)
    {
        if (nthreads_ <= 1) {
            fn(0);
            return;
        }
        // (the caller runs with subnormals flushed - kfsp_padm sets FTZ / DAZ for its duration; a worker must round the same way)
        const unsigned csr = _mm_getcsr();
        job_ = [&fn, csr](int t) {
            const unsigned saved = _mm_getcsr();
            _mm_setcsr(csr);
            fn(t);
            _mm_setcsr(saved);
        };
        remaining_.store(nthreads_ - 1, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        if (sleepers_.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
        fn(0);
        while (remaining_.load(std::memory_order_acquire) != 0) _mm_pause();
    }

private:
    PadePool()
    {
        // (default: 4 threads on a host with >= 16 CPUs - the MI355X boxes give a one-GPU job 16; the resident Goutsias run
        // spends 0.61 / 0.37 s in here with 1 / 4, and between 0.32 and 0.59 s with 6 depending on what else the host is
        // doing: spinning workers on a shared machine -, none on a smaller one)
        const int hw = (int)std::thread::hardware_concurrency();
        int want = hw >= 16 ? 4 : 1;
        if (const char *e = std::getenv("KFSP_PADE_THREADS")) want = std::atoi(e);
        if (hw > 0) want = std::min(want, std::max(1, hw / 2));
        nthreads_ = std::max(1, std::min(want, 16));
        if (const char *e = std::getenv("KFSP_PADE_SPIN_US")) spin_us_ = std::max(0.0, std::atof(e));
        for (int t = 1; t < nthreads_; ++t) {
            try {
                std::thread(&PadePool::worker, this, t).detach();
            } catch (...) {
                nthreads_ = t;
                break;
            }
        }
    }
    void worker(int t)
    {
        if (g_have_initial_cpus) (void)sched_setaffinity(0, sizeof(g_initial_cpus), &g_initial_cpus);
        uint64_t seen = 0;
        for (;;) {
            // a worker spins for spin_us_ after its last job before it goes to sleep: the exponentials of a solve follow each
            // other every 1-2 ms (an Arnoldi pass on the device lies between them), and a sleeping worker costs the next
            // product a futex wake-up
            int spins = 0;
            auto t_idle = std::chrono::steady_clock::now();
            while (gen_.load(std::memory_order_acquire) == seen) {
                if (++spins < 256) {
                    _mm_pause();
                    continue;
                }
                spins = 0;
                if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_idle).count() < spin_us_) continue;
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            }
            seen = gen_.load(std::memory_order_acquire);
            job_(t);
            remaining_.fetch_sub(1, std::memory_order_release);
        }
    }
    int nthreads_ = 1;
    double spin_us_ = 800.0;
    std::function<void(int)> job_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> remaining_{0}, sleepers_{0};
    std::mutex mu_;
    std::condition_variable cv_;
};

double g_prof[4] = {0, 0, 0, 0};   // seconds: dense products, banded products, solve, whole calls (kfsp_padm_profile)
struct ProfTimer {
    double &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit ProfTimer(double &a) : acc(a) {}
    ~ProfTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

void matmul(double alpha, const Dense &A, const Dense &B, Dense &C)
{
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    static const bool wider = wide && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
    size_t nz = 0;
    for (double v : B.a) nz += v != 0.0;
    if (2 * nz > B.a.size()) {                            // mostly non-zero: no point in skipping
        ProfTimer pt(g_prof[0]);
        const int m = A.m;
        auto part = [&](int jbeg, int jend) {
            if (wider) matmul_dense_avx512(alpha, A, B, C, jbeg, jend);
            else if (wide) matmul_dense_avx2(alpha, A, B, C, jbeg, jend);
            else matmul_dense_base(alpha, A, B, C, jbeg, jend);
        };
        PadePool &pool = PadePool::get();
        const int T = m >= 48 ? pool.threads() : 1;      // (a small product is over before a second thread has heard of it)
        if (T <= 1) {
            part(0, m);
            return;
        }
        const int groups = m / 4;                         // whole groups of four columns; the last range also takes the tail
        pool.run([&](int t) {
            const int g0 = (int)((long long)groups * t / T), g1 = (int)((long long)groups * (t + 1) / T);
            const int jb = 4 * g0, je = t == T - 1 ? m : 4 * g1;
            if (je > jb) part(jb, je);
        });
        return;
    }
    ProfTimer pt(g_prof[1]);
    const int m = A.m;
    auto part = [&](int jbeg, int jend) {
        if (wider) matmul_avx512(alpha, A, B, C, jbeg, jend);
        else if (wide) matmul_avx2(alpha, A, B, C, jbeg, jend);
        else matmul_base(alpha, A, B, C, jbeg, jend);
    };
    // (measured with the columns dealt to the pool as the dense products' are: 93 -> 103-120 ms over the resident Goutsias run -
    // a banded product is a few microseconds, less than the threads' hand-shake)
    part(0, m);
}

// X <- Q^{-1} X: LU with row pivoting, then forward and back substitution, all
// written as column operations (the matrices are column major, so every inner
// loop runs over contiguous memory); false if singular.  Zero multipliers are
// skipped: the Pade numerator / denominator of a banded H are banded.
bool solve_in_place(Dense &Q, Dense &X)
{
    const int m = Q.m;
    double *q = Q.a.data();
    double *x = X.a.data();
    // The factorisation touches Q only; what it does to the right-hand sides - the row exchange and the multipliers of
    // every step - is kept (piv, last; the multipliers sit below Q's diagonal) and applied afterwards, column by column:
    // the columns of X are independent, every element sees the same operations in the same order as when they were
    // applied step by step inside the factorisation, and the column ranges can go to the pool's threads.
    static thread_local std::vector<int> piv, lastrow, top;
    piv.resize((size_t)m);
    lastrow.resize((size_t)m);
    top.resize((size_t)m);
    for (int k = 0; k < m; ++k) {
        double *qk = q + (size_t)k * m;
        int p = k;
        for (int i = k + 1; i < m; ++i)
            if (std::fabs(qk[i]) > std::fabs(qk[p])) p = i;
        if (qk[p] == 0.0) return false;
        piv[(size_t)k] = p;
        if (p != k)
            for (int j = 0; j < m; ++j) std::swap(q[(size_t)j * m + k], q[(size_t)j * m + p]);
        const double inv = 1.0 / qk[k];
        int last = k;                                    // last row with a non-zero multiplier
        for (int i = k + 1; i < m; ++i) {
            qk[i] *= inv;
            if (qk[i] != 0.0) last = i;
        }
        lastrow[(size_t)k] = last;
        if (last == k) continue;
        for (int j = k + 1; j < m; ++j) {                // trailing update, column by column
            double *qj = q + (size_t)j * m;
            const double f = qj[k];
            if (f == 0.0) continue;
            for (int i = k + 1; i <= last; ++i) qj[i] -= qk[i] * f;
        }
    }
    // first non-zero row of every column of U: banded factors stay banded
    for (int k = 0; k < m; ++k) {
        const double *qk = q + (size_t)k * m;
        int f = 0;
        while (f < k && qk[f] == 0.0) ++f;
        top[(size_t)k] = f;
    }
    const int *pv = piv.data(), *lr = lastrow.data(), *tp = top.data();
    auto columns = [=](int jbeg, int jend) {
        for (int j = jbeg; j < jend; ++j) {
            double *xj = x + (size_t)j * m;
            for (int k = 0; k < m; ++k) {                // P and L^{-1}, step by step
                if (pv[k] != k) std::swap(xj[k], xj[pv[k]]);
                const int last = lr[k];
                if (last == k) continue;
                const double *qk = q + (size_t)k * m;
                const double f = xj[k];
                if (f == 0.0) continue;
                for (int i = k + 1; i <= last; ++i) xj[i] -= qk[i] * f;
            }
            for (int k = m - 1; k >= 0; --k) {           // U^{-1}, column-oriented back substitution
                const double *qk = q + (size_t)k * m;
                const double v = xj[k] / qk[k];
                xj[k] = v;
                if (v == 0.0) continue;
                for (int i = tp[k]; i < k; ++i) xj[i] -= qk[i] * v;
            }
        }
    };
    PadePool &pool = PadePool::get();
    const int T = m >= 48 ? pool.threads() : 1;
    if (T <= 1) {
        columns(0, m);
        return true;
    }
    pool.run([&](int t) { columns((int)((long long)m * t / T), (int)((long long)m * (t + 1) / T)); });
    return true;
}

}  // namespace

// where the host's share of a step goes: seconds spent so far in {dense products, banded products, the solve, whole calls}
extern "C" void kfsp_padm_profile(double *seconds4, int reset)
{
    for (int i = 0; i < 4; ++i) {
        if (seconds4) seconds4[i] = g_prof[i];
        if (reset) g_prof[i] = 0.0;
    }
}

extern "C" int kfsp_padm(int ideg, int m, double t, const double *H, int ldh, double *E, int *ns_out, double *hnorm_out)
try {
    if (ideg < 1 || ideg > 20) return -1;
    if (m < 1) return -2;
    if (!H) return -4;
    ProfTimer whole(g_prof[3]);
    if (ldh < m) return -5;
    if (!E) return -6;
    // Entries of exp(tH) far from the band underflow during the squaring phase;
    // subnormal operands make x86 arithmetic ~100x slower and carry no
    // information here (|value| < 1e-307), so flush them for the duration of the call.
    struct FlushSubnormals {
        unsigned saved = _mm_getcsr();
        FlushSubnormals() { _mm_setcsr(saved | 0x8040u); }
        ~FlushSubnormals() { _mm_setcsr(saved); }
    } flush_guard;
    static thread_local std::vector<double> workspace;
    if (workspace.size() < 5 * (size_t)m * m) workspace.resize(5 * (size_t)m * m);
    double *ws = workspace.data();
    const size_t mm = (size_t)m * m;
    Dense A(m, ws, false);                               // (every element is assigned below)
    double hnorm = 0.0;
    for (int i = 0; i < m; ++i) {
        double rs = 0.0;
        for (int j = 0; j < m; ++j) {
            A(i, j) = H[(size_t)j * ldh + i];
            rs += std::fabs(A(i, j));
        }
        hnorm = std::max(hnorm, rs);
    }
    hnorm = std::fabs(t * hnorm);
    if (hnorm_out) *hnorm_out = hnorm;
    if (hnorm == 0.0) return -3;   // 'null H', dgpadm.f:84
    const int ns = std::max(0, (int)(std::log(hnorm) / std::log(2.0)) + 2);
    if (ns_out) *ns_out = ns;
    const double scale = t / std::ldexp(1.0, ns);

    std::vector<double> c((size_t)ideg + 1);
    c[0] = 1.0;
    for (int k = 1; k <= ideg; ++k)
        c[(size_t)k] = c[(size_t)k - 1] * (double)(ideg + 1 - k) / (double)(k * (2 * ideg + 1 - k));

    Dense H2(m, ws + mm, false), P(m, ws + 2 * mm, true), Q(m, ws + 3 * mm, true), T(m, ws + 4 * mm, false);   // (products write every element of their result)
    matmul(scale * scale, A, A, H2);
    for (int i = 0; i < m; ++i) {
        P(i, i) = c[(size_t)ideg - 1];
        Q(i, i) = c[(size_t)ideg];
    }
    // Horner in H2, alternately on the even (q) and odd (p) coefficient sets
    bool odd = true;
    for (int k = ideg - 1; k > 0; --k) {
        Dense &U = odd ? Q : P;
        matmul(1.0, U, H2, T);
        for (int i = 0; i < m; ++i) T(i, i) += c[(size_t)k - 1];
        swap_storage(U, T);
        odd = !odd;
    }
    {
        Dense &U = odd ? Q : P;
        matmul(scale, U, A, T);
        swap_storage(U, T);
    }
    for (size_t i = 0; i < Q.a.size(); ++i) Q.a[i] -= P.a[i];
    {
        ProfTimer pt(g_prof[2]);
        if (!solve_in_place(Q, P)) return -7;
    }
    for (double &x : P.a) x *= 2.0;
    for (int i = 0; i < m; ++i) P(i, i) += 1.0;
    if (ns == 0 && odd) {
        for (double &x : P.a) x = -x;
    } else {
        for (int k = 0; k < ns; ++k) {
            matmul(1.0, P, P, T);
            swap_storage(P, T);
        }
    }
    std::memcpy(E, P.a.data(), P.a.size() * sizeof(double));
    return 0;
} catch (...) {
    return 4001;   // out of host memory
}
