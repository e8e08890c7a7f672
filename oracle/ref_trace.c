/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * BLAS observers for oracle/_ref/ref_trace: the UNMODIFIED reference calls
 * BLAS/LAPACK by their F77 names (KrylovSolver.f90:177,444,450,540; dgpadm.f:100-163);
 * the definitions below sit in the executable, so those calls land here, are
 * written to a trace file and are then passed on - arguments untouched - to the
 * routine they would have reached without us (dlsym(RTLD_NEXT): MKL).  Nothing in
 * the reference is changed and no arithmetic is done here; what the solver computes
 * is bit-identical with and without the observers (oracle/make_golden.py checks
 * that against the plain ref_dump run).
 *
 * What is recorded, in call order (native endian; one tag byte, then the fields):
 *   'B' DNRM2 on the solution vector W (:177, :540)   int32 n; f64 beta; f64 w[n]
 *       followed, when the state list changed, by
 *   'F' the FSP as the driver's hook hands it over     int32 ns, nr, n; int32 state[ns*n];
 *                                                      int32 adj[nr*n]; f64 offdiag[nr*n]; f64 diag[n]
 *   'P' first DGEMM of a DGPADM(norm) call (H*H)       int32 m, lda; f64 alpha (= (t/2^ns)^2); f64 H[m*m]
 *   'G' any other DGEMM                                int32 m; f64 alpha; int32 a_is_b
 *   'V' DGESV (one per DGPADM call)                    int32 n
 *   'C' DGEMV (:444)                                   int32 n, mx; f64 beta; f64 y[mx]
 *   'S' DASUM (:450)                                   int32 n; f64 wsum; f64 w[n]
 *   'N' any other DNRM2 (:247 h(j+1,j), :263 AVNORM)   int32 n; f64 value
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static FILE *tf = NULL;
static const double *w_base = NULL;
static void (*fsp_hook)(void) = NULL;
static int32_t *last_state = NULL;
static size_t last_state_len = 0;

static void put(const void *p, size_t bytes) { if (tf) fwrite(p, 1, bytes, tf); }
static void put_tag(char c) { if (tf) fputc(c, tf); }
static void put_i32(int32_t v) { put(&v, 4); }
static void put_f64(double v) { put(&v, 8); }

static void *next_sym(const char *name)
{
    void *p = dlsym(RTLD_NEXT, name);
    if (!p) {
        fprintf(stderr, "ref_trace: no BLAS provides %s\n", name);
        abort();
    }
    return p;
}

void ref_trace_begin(const char *path, const double *w, void (*hook)(void))
{
    tf = fopen(path, "wb");
    if (!tf) {
        perror("ref_trace: cannot open the trace file");
        abort();
    }
    w_base = w;
    fsp_hook = hook;
}

void ref_trace_end(void)
{
    if (tf) fclose(tf);
    tf = NULL;
    w_base = NULL;
    fsp_hook = NULL;
}

/* called by the driver's hook (Fortran) with the current FSP */
void ref_trace_put_fsp(const int32_t *ns, const int32_t *nr, const int32_t *n, const int32_t *state,
                       const int32_t *adj, const double *offdiag, const double *diag)
{
    const size_t len = (size_t)(*ns) * (size_t)(*n);
    if (last_state && last_state_len == len && memcmp(last_state, state, len * 4) == 0) return;
    free(last_state);
    last_state = (int32_t *)malloc(len * 4 + 4);
    memcpy(last_state, state, len * 4);
    last_state_len = len;
    put_tag('F');
    put_i32(*ns);
    put_i32(*nr);
    put_i32(*n);
    put(state, len * 4);
    put(adj, (size_t)(*nr) * (size_t)(*n) * 4);
    put(offdiag, (size_t)(*nr) * (size_t)(*n) * 8);
    put(diag, (size_t)(*n) * 8);
}

double dnrm2_(const int *n, const double *x, const int *incx)
{
    static double (*real)(const int *, const double *, const int *) = NULL;
    if (!real) real = (double (*)(const int *, const double *, const int *))next_sym("dnrm2_");
    const double r = real(n, x, incx);
    if (tf && x == w_base && *incx == 1) {
        put_tag('B');
        put_i32(*n);
        put_f64(r);
        put(x, (size_t)(*n) * 8);
        if (fsp_hook) fsp_hook();
    } else if (tf) {
        put_tag('N');
        put_i32(*n);
        put_f64(r);
    }
    return r;
}

double dasum_(const int *n, const double *x, const int *incx)
{
    static double (*real)(const int *, const double *, const int *) = NULL;
    if (!real) real = (double (*)(const int *, const double *, const int *))next_sym("dasum_");
    const double r = real(n, x, incx);
    if (tf && *incx == 1) {
        put_tag('S');
        put_i32(*n);
        put_f64(r);
        put(x, (size_t)(*n) * 8);
    }
    return r;
}

void dgemv_(const char *trans, const int *m, const int *n, const double *alpha, const double *a, const int *lda,
            const double *x, const int *incx, const double *beta, double *y, const int *incy, size_t ltrans)
{
    static void (*real)(const char *, const int *, const int *, const double *, const double *, const int *,
                        const double *, const int *, const double *, double *, const int *, size_t) = NULL;
    if (!real) real = next_sym("dgemv_");
    if (tf) {
        put_tag('C');
        put_i32(*m);
        put_i32(*n);
        put_f64(*alpha);
        put(x, (size_t)(*n) * 8);
    }
    real(trans, m, n, alpha, a, lda, x, incx, beta, y, incy, ltrans);
}

void dgemm_(const char *ta, const char *tb, const int *m, const int *n, const int *k, const double *alpha,
            const double *a, const int *lda, const double *b, const int *ldb, const double *beta, double *c,
            const int *ldc, size_t lta, size_t ltb)
{
    static void (*real)(const char *, const char *, const int *, const int *, const int *, const double *,
                        const double *, const int *, const double *, const int *, const double *, double *,
                        const int *, size_t, size_t) = NULL;
    if (!real) real = next_sym("dgemm_");
    if (tf) {
        if (a == b && *alpha != 1.0) {
            put_tag('P');
            put_i32(*m);
            put_i32(*lda);
            put_f64(*alpha);
            for (int j = 0; j < *m; ++j) put(a + (size_t)j * (size_t)(*lda), (size_t)(*m) * 8);
        } else {
            put_tag('G');
            put_i32(*m);
            put_f64(*alpha);
            put_i32(a == b ? 1 : 0);
        }
    }
    real(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lta, ltb);
}

void dgesv_(const int *n, const int *nrhs, double *a, const int *lda, int *ipiv, double *b, const int *ldb, int *info)
{
    static void (*real)(const int *, const int *, double *, const int *, int *, double *, const int *, int *) = NULL;
    if (!real) real = next_sym("dgesv_");
    if (tf) {
        put_tag('V');
        put_i32(*n);
    }
    real(n, nrhs, a, lda, ipiv, b, ldb, info);
}
