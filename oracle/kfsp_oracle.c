/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.  See kfsp_oracle.h.
 *
 * Plain-C, single-thread restatement of the reference hot path.  Compiled
 * with -ffp-contract=off so that a*b+c is two roundings, as in the reference
 * built for baseline x86-64 (no FMA).  Loops are written sequentially on
 * purpose: summation order follows the reference's loops (KrylovSolver.f90)
 * or the textbook BLAS loop where the reference calls BLAS.
 */
#include "kfsp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SpMV */

/* KrylovSolver.f90:593-606: zero y, then for every column i scatter
 * OFFDIAG(j,i)*x(i) into y(ADJ(j,i)) for ADJ>=1 and subtract DIAG(i)*x(i). */
void kfo_spmv_ell(const kfo_ell *A, const double *x, double *y)
{
    const int n = A->n, bw = A->bw, ld = A->ld;
    for (int i = 0; i < n; ++i) y[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        const int32_t *a = A->adj + (size_t)i * ld;
        const double *o = A->offdiag + (size_t)i * ld;
        for (int j = 0; j < bw; ++j) {
            const int k = a[j];
            if (k >= 1) y[k - 1] = y[k - 1] + o[j] * x[i];
        }
        y[i] = y[i] - A->diag[i] * x[i];
    }
}

int64_t kfo_ell_count_nnz(const kfo_ell *A)
{
    int64_t nnz = A->n;
    for (int i = 0; i < A->n; ++i)
        for (int j = 0; j < A->bw; ++j)
            if (A->adj[(size_t)i * A->ld + j] >= 1) ++nnz;
    return nnz;
}

/* Counting-sort transpose.  Visiting source columns i in increasing order and
 * inserting the diagonal when i reaches the row makes every row sorted by
 * column with the diagonal in place = the order in which FMATVEC adds into
 * y(k). */
void kfo_ell_to_csr(const kfo_ell *A, int64_t *rowptr, int32_t *col, double *val)
{
    const int n = A->n, bw = A->bw, ld = A->ld;
    for (int i = 0; i <= n; ++i) rowptr[i] = 0;
    for (int i = 0; i < n; ++i) {
        rowptr[i + 1] += 1;
        for (int j = 0; j < bw; ++j) {
            const int k = A->adj[(size_t)i * ld + j];
            if (k >= 1) rowptr[k] += 1;
        }
    }
    for (int i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) fill[i] = rowptr[i];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < bw; ++j) {
            const int k = A->adj[(size_t)i * ld + j];
            if (k >= 1) {
                const int64_t p = fill[k - 1]++;
                col[p] = i;
                val[p] = A->offdiag[(size_t)i * ld + j];
            }
        }
        const int64_t p = fill[i]++;
        col[p] = i;
        val[p] = -A->diag[i];
    }
    free(fill);
}

void kfo_spmv_csr(int n, const int64_t *rowptr, const int32_t *col,
                  const double *val, const double *x, double *y)
{
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) s = s + val[p] * x[col[p]];
        y[i] = s;
    }
}

/* ---------------------------------------------------------------- BLAS-1 */

double kfo_dot(int n, const double *x, const double *y)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + x[i] * y[i];
    return s;
}

void kfo_axpy(int n, double a, const double *x, double *y)
{
    for (int i = 0; i < n; ++i) y[i] = y[i] + a * x[i];
}

/* The reference calls BLAS DNRM2 (scaled, overflow-safe).  Probability and
 * Krylov vectors are O(1), so the plain form differs only in rounding. */
double kfo_nrm2(int n, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + x[i] * x[i];
    return sqrt(s);
}

void kfo_scal(int n, double a, double *x)
{
    for (int i = 0; i < n; ++i) x[i] = a * x[i];
}

double kfo_asum(int n, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + fabs(x[i]);
    return s;
}

/* ------------------------------------------------------------- dense expm */

/* C(m,m) = alpha * A(m,m) * B(m,m), column major; the DGEMM('n','n',..,beta=0)
 * calls of dgpadm.f:100,121,136,140,163. */
static void gemm_nn(int m, double alpha, const double *A, int lda, const double *B,
                    int ldb, double *C, int ldc)
{
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i < m; ++i) C[(size_t)j * ldc + i] = 0.0;
        for (int l = 0; l < m; ++l) {
            const double b = alpha * B[(size_t)j * ldb + l];
            if (b == 0.0) continue;
            for (int i = 0; i < m; ++i) C[(size_t)j * ldc + i] += b * A[(size_t)l * lda + i];
        }
    }
}

/* Solve Q X = P in place (X overwrites P) by LU with partial pivoting:
 * the DGESV call of dgpadm.f:145. */
static int gesv(int m, double *Q, double *P)
{
    for (int k = 0; k < m; ++k) {
        int piv = k;
        double big = fabs(Q[(size_t)k * m + k]);
        for (int i = k + 1; i < m; ++i) {
            const double a = fabs(Q[(size_t)k * m + i]);
            if (a > big) { big = a; piv = i; }
        }
        if (big == 0.0) return -1;
        if (piv != k) {
            for (int j = 0; j < m; ++j) {
                double t = Q[(size_t)j * m + k]; Q[(size_t)j * m + k] = Q[(size_t)j * m + piv]; Q[(size_t)j * m + piv] = t;
                t = P[(size_t)j * m + k]; P[(size_t)j * m + k] = P[(size_t)j * m + piv]; P[(size_t)j * m + piv] = t;
            }
        }
        const double inv = 1.0 / Q[(size_t)k * m + k];
        for (int i = k + 1; i < m; ++i) Q[(size_t)k * m + i] *= inv;
        for (int j = k + 1; j < m; ++j) {
            const double q = Q[(size_t)j * m + k];
            if (q != 0.0)
                for (int i = k + 1; i < m; ++i) Q[(size_t)j * m + i] -= Q[(size_t)k * m + i] * q;
        }
        for (int j = 0; j < m; ++j) {
            const double p = P[(size_t)j * m + k];
            if (p != 0.0)
                for (int i = k + 1; i < m; ++i) P[(size_t)j * m + i] -= Q[(size_t)k * m + i] * p;
        }
    }
    for (int j = 0; j < m; ++j) {
        for (int k = m - 1; k >= 0; --k) {
            double s = P[(size_t)j * m + k] / Q[(size_t)k * m + k];
            P[(size_t)j * m + k] = s;
            if (s != 0.0)
                for (int i = 0; i < k; ++i) P[(size_t)j * m + i] -= Q[(size_t)k * m + i] * s;
        }
    }
    return 0;
}

/* dgpadm.f:2-169 (and the hnorm output of DGPADMnorm :171-339). */
int kfo_padm(int ideg, int m, double t, const double *H, int ldh, double *E,
             int *ns_out, double *hnorm_out)
{
    const size_t mm = (size_t)m * m;
    /* scaling: ns such that ||t*H/2^ns|| < 1/2          dgpadm.f:68-85 */
    double *rows = (double *)calloc((size_t)m, sizeof(double));
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) rows[i] += fabs(H[(size_t)j * ldh + i]);
    double hnorm = 0.0;
    for (int i = 0; i < m; ++i) hnorm = fmax(hnorm, rows[i]);
    free(rows);
    hnorm = fabs(t * hnorm);
    if (hnorm_out) *hnorm_out = hnorm;
    if (hnorm == 0.0) return -3;                         /* :84 'null H' */
    int ns = (int)(log(hnorm) / log(2.0)) + 2;           /* INT() truncates */
    if (ns < 0) ns = 0;
    const double scale = t / ldexp(1.0, ns);
    const double scale2 = scale * scale;
    if (ns_out) *ns_out = ns;

    /* Pade coefficients                                   dgpadm.f:89-96 */
    double coef[32];
    {
        const int i = ideg + 1, j = 2 * ideg + 1;
        coef[0] = 1.0;
        for (int k = 1; k <= ideg; ++k)
            coef[k] = (coef[k - 1] * (double)(i - k)) / (double)(k * (j - k));
    }
    double *buf = (double *)malloc(sizeof(double) * mm * 4);
    double *H2 = buf, *P = buf + mm, *Q = buf + 2 * mm, *F = buf + 3 * mm;
    gemm_nn(m, scale2, H, ldh, H, ldh, H2, m);            /* :100 */
    for (size_t i = 0; i < mm; ++i) { P[i] = 0.0; Q[i] = 0.0; }
    for (int j = 0; j < m; ++j) {                         /* :104-113 */
        P[(size_t)j * (m + 1)] = coef[ideg - 1];
        Q[(size_t)j * (m + 1)] = coef[ideg];
    }
    /* Horner, alternating between q (odd) and p           :117-132 */
    int iodd = 1;
    for (int k = ideg - 1; k > 0; --k) {
        double *used = iodd ? Q : P;
        gemm_nn(m, 1.0, used, m, H2, m, F, m);
        for (int j = 0; j < m; ++j) F[(size_t)j * (m + 1)] += coef[k - 1];
        if (iodd) { double *tq = Q; Q = F; F = tq; } else { double *tp = P; P = F; F = tp; }
        iodd = 1 - iodd;
    }
    /* (+/-)(I + 2*(p\q))                                  :136-150 */
    if (iodd == 1) {
        gemm_nn(m, scale, Q, m, H, ldh, F, m);
        double *tq = Q; Q = F; F = tq;
    } else {
        gemm_nn(m, scale, P, m, H, ldh, F, m);
        double *tp = P; P = F; F = tp;
    }
    for (size_t i = 0; i < mm; ++i) Q[i] = Q[i] - P[i];   /* DAXPY(-1,p,q) */
    if (gesv(m, Q, P) != 0) { free(buf); return -4; }
    for (size_t i = 0; i < mm; ++i) P[i] = 2.0 * P[i];
    for (int j = 0; j < m; ++j) P[(size_t)j * (m + 1)] += 1.0;
    double *put = P;
    if (ns == 0 && iodd == 1) {
        for (size_t i = 0; i < mm; ++i) P[i] = -P[i];
    } else {
        /* squaring                                        :159-166 */
        int odd = 1;
        for (int k = 0; k < ns; ++k) {
            double *get = odd ? P : Q;
            put = odd ? Q : P;
            gemm_nn(m, 1.0, get, m, get, m, put, m);
            odd = 1 - odd;
        }
    }
    memcpy(E, put, sizeof(double) * mm);
    free(buf);
    return 0;
}

/* --------------------------------------------------------------- Arnoldi */

/* KrylovSolver.f90:236-266.  1-based j like the reference; V(:,j) is
 * V + (j-1)*n. */
int kfo_arnoldi(const kfo_ell *A, int m, int jold, int qiop, double break_tol,
                double *V, double *H, int mh, double *avnorm, int *k1, int *nmult)
{
    const int n = A->n;
    int istart = 1;
    *k1 = 2;
    for (int j = jold; j <= m; ++j) {
        double *vj = V + (size_t)(j - 1) * n, *w = V + (size_t)j * n;
        ++*nmult;
        kfo_spmv_ell(A, vj, w);                                     /* :240 */
        if (qiop > 0) istart = (j - qiop + 1 > 1) ? j - qiop + 1 : 1;
        for (int i = istart; i <= j; ++i) {                         /* :242-246 */
            const double *vi = V + (size_t)(i - 1) * n;
            const double hij = kfo_dot(n, vi, w);
            kfo_axpy(n, -hij, vi, w);
            H[(size_t)(j - 1) * mh + (i - 1)] = hij;
        }
        const double hj1j = kfo_nrm2(n, w);                         /* :247 */
        if (hj1j <= break_tol) {                                    /* :249-256 */
            *k1 = 0;
            H[(size_t)m * mh + m + 1] = 1.0;                        /* :266 */
            return j;
        }
        H[(size_t)(j - 1) * mh + j] = hj1j;
        kfo_scal(n, 1.0 / hj1j, w);                                 /* :258 */
    }
    /* :261-263.  J1V is only advanced inside the loop (:259), so when a
     * dimension change SHRANK m below jold (:404, :426) the loop body never
     * runs and the extra product is taken from column jold, not m+1. */
    const int jl = (jold > m) ? jold : m + 1;
    ++*nmult;
    kfo_spmv_ell(A, V + (size_t)(jl - 1) * n, V + (size_t)jl * n);
    *avnorm = kfo_nrm2(n, V + (size_t)jl * n);
    H[(size_t)m * mh + m + 1] = 1.0;                                /* :266 */
    return m;
}

/* w = beta * V(:,1:mx) * y, clamp negatives, return ||w||_1.
 * KrylovSolver.f90:444-450 (DGEMV 'N' as the reference BLAS loop: column
 * sweep, y(i) += temp*A(i,j)). */
static double combine(int n, int mx, double beta, const double *V, const double *y, double *w)
{
    for (int i = 0; i < n; ++i) w[i] = 0.0;
    for (int j = 0; j < mx; ++j) {
        const double temp = beta * y[j];
        const double *vj = V + (size_t)j * n;
        for (int i = 0; i < n; ++i) w[i] = w[i] + temp * vj[i];
    }
    for (int i = 0; i < n; ++i)
        if (w[i] < 0.0) w[i] = 0.0;
    return kfo_asum(n, w);
}

int kfo_expv_fixed(const kfo_ell *A, int m, double tau, int nsteps, double *w,
                   double *wsums)
{
    const int n = A->n, mh = m + 2;
    if (m < 1 || m >= n) return -3;
    double *V = (double *)malloc(sizeof(double) * (size_t)n * (m + 2));
    double *H = (double *)malloc(sizeof(double) * (size_t)mh * mh);
    double *E = (double *)malloc(sizeof(double) * (size_t)mh * mh);
    int rc = 0;
    for (int s = 0; s < nsteps; ++s) {
        const double beta = kfo_nrm2(n, w);
        const double p1 = 1.0 / beta;
        for (int i = 0; i < n; ++i) V[i] = p1 * w[i];               /* :223-226 */
        memset(H, 0, sizeof(double) * (size_t)mh * mh);
        double avnorm = 0.0;
        int k1, nmult = 0;
        const int mb = kfo_arnoldi(A, m, 1, 2, 1.0e-7, V, H, mh, &avnorm, &k1, &nmult);
        int mx = mb + k1, ns;
        double hn;
        rc = kfo_padm(6, mx, tau, H, mh, E, &ns, &hn);              /* :274 */
        if (rc) break;
        mx = mb + (k1 - 1 > 0 ? k1 - 1 : 0);                        /* :438 */
        const double ws = combine(n, mx, beta, V, E, w);
        if (wsums) wsums[s] = ws;
    }
    free(V); free(H); free(E);
    return rc;
}

/* ------------------------------------------------------ adaptive DGEXPV_FSP */

static double ipow(double x, int e)
{
    /* real**integer as compilers expand it: repeated squaring, reciprocal
     * for negative exponents */
    int neg = e < 0;
    unsigned u = (unsigned)(neg ? -e : e);
    double r = 1.0, b = x;
    while (u) { if (u & 1u) r *= b; b *= b; u >>= 1; }
    return neg ? 1.0 / r : r;
}

static double nintd(double x) { return round(x); }   /* NINT: half away from 0 */

/* round to 2 significant digits, KrylovSolver.f90:186-187 (+0.55) and
 * :344-345 (+0) */
static double round2(double t, double add, double sqr1)
{
    const double p1 = ipow(10.0, (int)nintd(log10(t) - sqr1) - 1);
    return trunc(t / p1 + add) * p1;
}

/* KrylovSolver.f90:618-639.  The integer sub-expressions are default INTEGER
 * in the reference and wrap at 32 bits for N >~ 10^6; restated with explicit
 * wrap-around so the cost comparison (:362) takes the same branch. */
static double krylov_cost(double t_now, double t_out, double tau, int m, int n,
                          double hnorm, int nnz, int qiop)
{
    const double nom = 25.0 / 3.0 + (double)((2 + (int)(log(tau * hnorm) / log(2.0))) > 0
                                                 ? (2 + (int)(log(tau * hnorm) / log(2.0))) : 0);
    const uint32_t a = 2u * (uint32_t)(m + 1) * (uint32_t)nnz;
    const uint32_t b = (uint32_t)(5 * m + 4 * qiop * m + 2 * qiop - 2 * qiop * qiop + 7) * (uint32_t)n;
    const int32_t ab = (int32_t)(a + b);
    const double per = (double)ab + 2.0 * nom * (m + 2) * (m + 2) * (m + 2);
    return nintd((t_out - t_now) / tau) * per;
}

/* would DROP_STATES (StateSpace.f90:431-548) compact the FSP?  FIND_DROPTOL
 * :398-427, marking :475-495, the 10% rule :497. */
static int would_drop(const kfo_ell *A, const double *w, double dsum, double *tmp)
{
    const int n = A->n;
    double droptol = 1.0e-8;
    for (int it = 0; it < 400; ++it) {
        double s = 0.0;
        for (int i = 0; i < n; ++i)
            if (w[i] < droptol && w[i] > 0) s = s + w[i];
        if (s < dsum) break;
        droptol = droptol / 10.0;
    }
    int cnt = 0;
    for (int i = 0; i < n; ++i)
        if (w[i] < droptol) ++cnt;
    kfo_spmv_ell(A, w, tmp);
    for (int i = 0; i < n; ++i)
        if (tmp[i] > 1.0e-8) --cnt;          /* :491-494, decremented even if unmarked */
    return (cnt * 1.0) / (n * 1.0) > 0.1;
}

int kfo_dgexpv_fixed_fsp(const kfo_ell *A, double t, const double *v, double *w,
                         double fsptol, double krytol, kfo_stats *st,
                         int max_log, double *log_tau, int *log_m, double *log_wsum)
{
    enum { M_MAX = 100, M_MIN = 10, IDEG = 6 };
    const double DELTA = 1.2, GAMMA = 0.9;
    const int n_fsp = A->n;
    const int qiop = 2;
    const int trace = getenv("KFO_TRACE") != NULL;
    const double anorm = 1.0;                                        /* :129 */
    int n = n_fsp, m = M_MIN;

    double *V = (double *)malloc(sizeof(double) * (size_t)n_fsp * (M_MAX + 2));
    double *H = (double *)calloc((size_t)(M_MAX + 2) * (M_MAX + 2), sizeof(double));
    double *Htmp = (double *)calloc((size_t)(M_MAX + 2) * (M_MAX + 2), sizeof(double));
    double *E = (double *)calloc((size_t)(M_MAX + 2) * (M_MAX + 2), sizeof(double));
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n_fsp);

    int ibrkflag = 0, nmult = 0, nreject = 0, nexph = 0, nscale = 0, nstep = 0, nlog = 0, nws = 0;
    int status = 0;
    const double t_out = fabs(t);
    double tbrkdwn = 0.0, t_now = 0.0, t_new = 0.0, t_step = 0.0;
    (void)tbrkdwn;
    /* machine epsilon by the 4/3 trick                              :166-170 */
    double eps;
    {
        volatile double p1 = 4.0 / 3.0, p2, p3;
        do { p2 = p1 - 1.0; p3 = p2 + p2 + p2; eps = fabs(p3 - 1.0); } while (eps == 0.0);
    }
    if (krytol <= eps) krytol = sqrt(eps);                           /* :171 */
    const double rndoff = eps * anorm;
    const double break_tol = 1.0e-7;
    const double sgn = (t < 0) ? -1.0 : 1.0;

    memcpy(w, v, sizeof(double) * (size_t)n);                        /* :176 */
    double beta = kfo_nrm2(n, w);
    const double sqr1 = sqrt(0.1);
    double xm = 1.0 / (double)m;
    {
        double p1 = krytol * ipow((m + 1) / 2.72, m + 1) * sqrt(2.0 * 3.14 * (m + 1));
        t_new = (1.0 / anorm) * pow(p1 / (4.0 * beta * anorm), xm);
        t_new = round2(t_new, 0.55, sqr1);                           /* :186-187 */
    }
    int n_now = n, iexpand = 0, irejectfsp = 0;
    double wsum_old = 1.0, wsum = 0.0;
    int nnz = (A->bw + 1) * n_fsp;                                   /* :196 */
    int imreject = 0, jold = 1, m_new = m, orderold = 1, kestold = 1;
    /* used before set in the reference (:313,316,474); zero here */
    double omega = 0.0, omega_old = 0.0, t_old = 0.0, errorold = 0.0, tau_old = 0.0;
    int m_old = 0;
    double order = 0.0, k_factor = 0.0, hnorm = 0.0, err_loc = 0.0, avnorm = 0.0;
    double fsporder = 2.0, error = 0.0;
    int m_changed = 0, k1 = 2, mh = m + 2, mx = 0, mbrkdwn = m, ireject = 0, ns = 0;

    while (t_now < t_out) {                                          /* label 100 */
        t_step = fmin(t_out - t_now, t_new);
        n = n_now;
        m = (n - 1 < m_new) ? n - 1 : m_new;
        mbrkdwn = m;
        k1 = 2;
        mh = m + 2;
        ++nstep;
        {
            const double p1 = 1.0 / beta;
            for (int i = 0; i < n; ++i) V[i] = p1 * w[i];            /* :223-226 */
        }
        memset(H, 0, sizeof(double) * (size_t)mh * mh);
        ireject = 0;

    arnoldi:                                                          /* label 101 */
        mbrkdwn = kfo_arnoldi(A, m, jold, qiop, break_tol, V, H, mh, &avnorm, &k1, &nmult);
        if (k1 == 0) {                                               /* :250-254 */
            ibrkflag = 1;
            tbrkdwn = t_now;
            t_step = t_out - t_now;
        }

    pade:                                                             /* label 401 */
        ++nexph;
        mx = mbrkdwn + k1;
        if (kfo_padm(IDEG, mx, sgn * t_step, H, mh, E, &ns, &hnorm) != 0) { status = -4; goto done; }
        nscale += ns;
        /* local error estimate                                      :290-305 */
        if (k1 == 0) {
            err_loc = krytol;
        } else {
            const double p1 = fabs(E[m]) * beta;
            const double p2 = fabs(E[m + 1]) * beta * avnorm;
            if (p1 > 10.0 * p2) { err_loc = p2; xm = 1.0 / (double)m; }
            else if (p1 > p2) { err_loc = (p1 * p2) / (p1 - p2); xm = 1.0 / (double)m; }
            else { err_loc = p1; xm = 1.0 / (double)(m - 1); }
        }
        if (isnan(err_loc)) { t_step = t_step / 5.0; goto pade; }    /* :307-310 */

        omega_old = omega;
        omega = err_loc / (krytol * t_step);                         /* :314 */
        if (trace) fprintf(stderr, "kfo: step %d m=%d mx=%d k1=%d mbrk=%d t_step=%.17g err_loc=%.17g omega=%.6g avnorm=%.6g hnorm=%.6g\n",
                           nstep, m, mx, k1, mbrkdwn, t_step, err_loc, omega, avnorm, hnorm);
        if (m == m_old && t_step != t_old && ireject >= 1) {         /* :316-324 */
            order = fmax(1.0, log(omega / omega_old) / log(t_step / t_old));
            orderold = 0;
        } else if (orderold || ireject == 0) {
            order = (double)m / 4.0;
            orderold = 1;
        } else {
            orderold = 1;
        }
        if (m != m_old && t_step == t_old && ireject >= 1) {         /* :326-334 */
            k_factor = fmax(1.1, pow(omega / omega_old, 1.0 / (double)(m_old - m)));
            kestold = 0;
        } else if (kestold || ireject == 0) {
            kestold = 1;
            k_factor = 2.0;
        } else {
            kestold = 1;
        }
        t_old = t_step;
        m_old = m;
        if ((m == M_MAX && omega > DELTA) || imreject > 4) {         /* :339-346 */
            t_new = fmin(t_out - t_now,
                         fmax(t_step / 5.0, fmin(5.0 * t_step, GAMMA * t_step * pow(omega, -1.0 / order))));
            t_new = round2(t_new, 0.0, sqr1);
            m_changed = 0;
        } else {                                                     /* :348-372 */
            const double t_opt = fmin(t_out - t_now,
                                      fmax(t_step / 5.0, fmin(5.0 * t_step, GAMMA * t_step * pow(omega, -1.0 / order))));
            int a = M_MIN;
            if (3 * m / 4 > a) a = 3 * m / 4;
            {
                const int c = m + (int)ceil(log(omega) / log(k_factor));
                if (c > a) a = c;
            }
            int m_opt = a;
            if (M_MAX < m_opt) m_opt = M_MAX;
            {
                const int c = (int)ceil(4.0 * m / 3.0) + 1;
                if (c < m_opt) m_opt = c;
            }
            /* COST1/COST2 are default REAL in the reference (:109) */
            const float cost1 = (float)krylov_cost(t_now, t_out, t_opt, m, n, hnorm, nnz, qiop);
            const float cost2 = (float)krylov_cost(t_now, t_out, t_step, m_opt, n, hnorm, nnz, qiop);
            if (trace) fprintf(stderr, "kfo:   t_opt=%.17g m_opt=%d order=%.6g kf=%.6g cost1=%.9g cost2=%.9g\n",
                               t_opt, m_opt, order, k_factor, (double)cost1, (double)cost2);
            if (cost1 <= cost2) {
                t_new = round2(t_opt, 0.0, sqr1);
                m_new = m;
                m_changed = 0;
            } else {
                m_new = m_opt;
                t_new = t_step;
                m_changed = 1;
            }
        }
        if (k1 != 0 && omega > DELTA) {                              /* :375 (MXREJECT = 0) */
            if (!m_changed) {                                        /* :377-399 */
                t_step = fmin(t_out - t_now, fmax(t_step / 5.0, fmin(5.0 * t_step, t_new)));
                t_step = round2(t_step, 0.55, sqr1);
                ++ireject;
                ++nreject;
                goto pade;
            } else {                                                 /* :400-433 */
                ++nreject;
                ++imreject;
                m = m_new;
                memcpy(Htmp, H, sizeof(double) * (size_t)mh * mh);
                mbrkdwn = m;
                k1 = 2;
                mh = m + 2;
                t_step = fmin(t_out - t_now, t_new);
                memset(H, 0, sizeof(double) * (size_t)mh * mh);
                for (int j = 1; j <= m_old; ++j)
                    for (int i = 1; i <= j + 1; ++i)
                        H[(size_t)(j - 1) * (m + 2) + i - 1] = Htmp[(size_t)(j - 1) * (m_old + 2) + i - 1];
                jold = m_old;
                goto arnoldi;
            }
        }
        imreject = 0;                                                /* :435-439 */
        jold = 1;
        if (err_loc < 1.0e-16) t_new = fmax(t_new, 2.0 * t_step);
        mx = mbrkdwn + ((k1 - 1 > 0) ? k1 - 1 : 0);
        irejectfsp = 0;

        int to_ssa = 0;
        for (;;) {                                                   /* :442-495 */
            wsum = combine(n, mx, beta, V, E, w);
            if (log_wsum && nws < max_log) log_wsum[nws] = wsum;
            ++nws;
            error = wsum_old - wsum;
            if (wsum >= 1.0 - (t_now + t_step) * fsptol / t_out) break;   /* :458, :615 */
            iexpand = 1;
            ++irejectfsp;
            if (irejectfsp >= 5) {                                   /* :466-470 */
                for (int i = 0; i < n; ++i) w[i] = beta * V[i];
                --nstep;
                to_ssa = 1;
                break;
            } else if (irejectfsp == 1) {
                fsporder = 2.0;
            } else {
                fsporder = log(error / errorold) / log(t_step / tau_old) - 1.0;
            }
            const double tfsp = GAMMA * t_step * pow(fsptol * t_step / (error * t_out), 1.0 / fsporder);
            errorold = error;
            tau_old = t_step;
            t_step = fmin(t_out - t_now, fmax(t_step / 5.0, fmin(0.9 * t_step, tfsp)));
            t_step = round2(t_step, 0.55, sqr1);
            ++nexph;                                                 /* :489-493 */
            if (kfo_padm(IDEG, mx, sgn * t_step, H, mh, E, &ns, NULL) != 0) { status = -4; goto done; }
            nscale += ns;
        }
        if (!to_ssa) {
            t_now = t_now + t_step;                                  /* :498-499 */
            wsum_old = wsum;
            if (nlog < max_log) {
                if (log_tau) log_tau[nlog] = t_step;
                if (log_m) log_m[nlog] = m;
            }
            ++nlog;
            if (t_now >= t_out) break;                               /* :506 */
            if (nstep > 1 && iexpand != 1) {                         /* :509-512 */
                const double dsum = wsum - (1.0 - t_now * fsptol / t_out);
                if (dsum > 0.0 && would_drop(A, w, dsum, tmp)) { status = 11; goto done; }
            }
        }
        if (iexpand == 1 && t_now < t_out) { status = 10; goto done; }   /* :518-534 */

        nnz = (A->bw + 1) * n_fsp;                                   /* :537-548 */
        n_now = n_fsp;
        beta = kfo_nrm2(n_now, w);
        err_loc = fmax(err_loc, rndoff);
        t_new = round2(t_new, 0.55, sqr1);
    }

done:
    if (st) {
        st->nmult = nmult; st->nexph = nexph; st->nscale = nscale; st->nstep = nstep;
        st->nreject = nreject; st->ibrkflag = ibrkflag; st->mbrkdwn = mbrkdwn;
        st->n_wsum = nws; st->status = status; st->t_now = t_now;
    }
    free(V); free(H); free(Htmp); free(E); free(tmp);
    return status;
}
