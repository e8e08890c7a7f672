/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the reference hot path of
 * voduchuy/KrylovFspSsa: exp(tA)v on the FSP-restricted CME generator.
 * Every function cites the reference lines it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the
 * product (krylovfspssa_amd/) never links or imports it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 * against fixtures captured from the unmodified reference compiled in the
 * build container (oracle/Makefile -> oracle/_ref/ref_dump, fixtures in
 * tests/golden/, generator oracle/make_golden.py):
 *   - kfo_padm          vs DGPADM outputs                (padm.npz)
 *   - kfo_dgexpv_fixed_fsp vs CME_SOLVE on closed systems (solve_ring4/6.npz):
 *     final probability vector, every step size, Krylov dimension and WSUM.
 *   - kfo_spmv_ell is pinned indirectly through those runs (the reference's
 *     FMATVEC is an internal procedure, KrylovSolver.f90:574-577, and cannot
 *     be called from outside) and directly against an independent scipy
 *     product in the tests.
 */
#ifndef KFSP_ORACLE_H
#define KFSP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The reference's generator storage, TYPE FSP_MATRIX (StateSpace.f90:13-17):
 * column-oriented ELL.  adj/offdiag are [state i][slot j] with leading
 * dimension ld (= Fortran ADJ(j,i), ld = SIZE(ADJ,1)); adj is 1-based, 0 =
 * successor not in the FSP, -1 = illegal; diag is stored positive. */
typedef struct {
    int32_t n, bw, ld;
    const int32_t *adj;
    const double *offdiag;
    const double *diag;
} kfo_ell;

/* y = A x, scatter form.  KrylovSolver.f90:577-607. */
void kfo_spmv_ell(const kfo_ell *A, const double *x, double *y);

/* BLAS level-1/2 call sites of the hot path (KrylovSolver.f90:176-177,
 * 243-258, 444, 450), sequential loops. */
double kfo_dot(int n, const double *x, const double *y);
void kfo_axpy(int n, double a, const double *x, double *y);
double kfo_nrm2(int n, const double *x);
void kfo_scal(int n, double a, double *x);
double kfo_asum(int n, const double *x);

/* exp(t*H) for an m x m matrix (column-major, leading dimension ldh) by the
 * (ideg,ideg) Pade approximant with scaling and squaring.  dgpadm.f:2-169.
 * E is m*m column-major (ld m).  *ns = number of squarings, *hnorm =
 * |t|*||H||_inf (the extra output of DGPADMnorm, dgpadm.f:171-339).
 * returns 0, or <0 on a null H / singular solve. */
int kfo_padm(int ideg, int m, double t, const double *H, int ldh, double *E,
             int *ns, double *hnorm);

/* One IOP Arnoldi pass, columns jold..m (1-based like the reference), plus
 * the extra matvec for AVNORM.  KrylovSolver.f90:236-266.
 * V: n x (m+2) column-major, ld n, V(:,1..jold) given.  H: mh x mh, mh = m+2.
 * returns mbrkdwn (= m when no happy breakdown); *k1 = 2 or 0. */
int kfo_arnoldi(const kfo_ell *A, int m, int jold, int qiop, double break_tol,
                double *V, double *H, int mh, double *avnorm, int *k1, int *nmult);

/* Benchmark mode (BASELINE config 2: fixed Krylov dimension m, fixed step tau,
 * nsteps steps, no adaptivity, no FSP change): per step
 *   beta=||w||, v1=w/beta, Arnoldi (above), E=exp(tau*H) of order m+2,
 *   w = beta*V(:,1:m+1)*E(1:m+1,1), clamp negatives, wsum=||w||_1
 * i.e. KrylovSolver.f90:223-266, 270-277, 438, 444-450 with the accept/reject
 * logic removed.  wsums[nsteps] receives the mass after every step. */
int kfo_expv_fixed(const kfo_ell *A, int m, double tau, int nsteps, double *w,
                   double *wsums);

typedef struct {
    int nmult, nexph, nscale, nstep, nreject, ibrkflag, mbrkdwn;
    int n_wsum;          /* number of FSP acceptance evaluations (:452) */
    int status;          /* 0 ok, 10 = FSP needs expansion, 11 = states would be dropped */
    double t_now;
} kfo_stats;

/* The adaptive solver DGEXPV_FSP (KrylovSolver.f90:40-653) restated for a
 * FIXED state space: MATRIX_STARTER/ONESTEP_EXTENDER (:130-134) are the
 * caller's business, and the two places where the reference would change the
 * FSP (DROP_STATES :509-512 actually compacting, SSA expansion :518-534) end
 * the run with status 11 / 10.  Step log arrays (length max_log) may be NULL.
 * v: start vector, w: result (length n). */
int kfo_dgexpv_fixed_fsp(const kfo_ell *A, double t, const double *v, double *w,
                         double fsptol, double krytol, kfo_stats *st,
                         int max_log, double *log_tau, int *log_m,
                         double *log_wsum);

/* Reference-layout (column ELL) -> row CSR with the diagonal -DIAG stored in
 * place, rows sorted by column.  A sequential row gather over this CSR adds in
 * the same order as FMATVEC's scatter.  rowptr[n+1], col/val
 * sized by kfo_ell_count_nnz. */
int64_t kfo_ell_count_nnz(const kfo_ell *A);
void kfo_ell_to_csr(const kfo_ell *A, int64_t *rowptr, int32_t *col, double *val);
void kfo_spmv_csr(int n, const int64_t *rowptr, const int32_t *col,
                  const double *val, const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
