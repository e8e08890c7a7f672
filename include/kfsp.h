/*
 * kfsp.h - C ABI of libkfsp_hip: the exp(tA)v hot path of the Krylov-FSP CME
 * solver on one MI355X (gfx950) per process.
 *
 * The reference (voduchuy/KrylovFspSsa) has no FFI; its seams are Fortran
 * internal.  Each entry point below names the reference code it stands in for
 * (paths relative to the reference tree).  The host side that drives these
 * (the Fortran modules under krylovfspssa_amd/fortran through ISO_C_BINDING, or the ctypes mirror
 * krylovfspssa_amd/host.py) keeps the reference's own interface.
 *
 * Conventions
 *   - plain pointers and sizes only; every array argument is HOST memory unless
 *     its name ends in _dev.  The library copies on set_* and owns all device
 *     memory until kfsp_destroy.  No pointer is retained past a call.
 *   - return value: 0 ok; <0 = index of the bad argument (like IFLAG,
 *     KrylovSolver.f90:142-149); >0 = 1000+hipError_t, 2000+ncclResult_t,
 *     3000+ host Pade failure, 4000 exception inside the library or a
 *     callback, 4001 out of host memory.  Never exits, never throws (C++
 *     exceptions are caught at the boundary).  kfsp_last_error() gives the text.
 *   - indices in reference-layout arrays are 1-based exactly as the reference
 *     stores them; CSR entry points are 0-based.
 *   - one host thread per context; calls return when their host-visible
 *     outputs are valid (device work is synchronised inside).
 *   - there is NO CPU fallback: without a usable GPU kfsp_create fails.
 */
#ifndef KFSP_H
#define KFSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kfsp_ctx kfsp_ctx;

#define KFSP_M_MAX 100            /* KrylovSolver.f90:47 */
#define KFSP_UNIQUE_ID_BYTES 128  /* sizeof(ncclUniqueId) */

/* ---- context --------------------------------------------------------- */
/* Replaces the solver's static workspace WSP/IWSP (KrylovSolver.f90:50-52). */
int kfsp_create(int device, kfsp_ctx **out);
int kfsp_destroy(kfsp_ctx *ctx);
const char *kfsp_last_error(const kfsp_ctx *ctx);
/* library/ABI version, bumped on any signature change */
int kfsp_abi_version(void);

/* ---- multi-GPU (row partition; no counterpart in the serial reference) -- */
/* rank 0 creates the id, the host launcher broadcasts the bytes, every rank
 * calls kfsp_comm_init.  After it, n-arguments below stay GLOBAL sizes and
 * each context owns the contiguous row block
 *   [rank*L, min((rank+1)*L, n)),  L = ceil(n / nranks) rounded up to 64. */
int kfsp_comm_unique_id(void *id_bytes /* [KFSP_UNIQUE_ID_BYTES] */);
int kfsp_comm_init(kfsp_ctx *ctx, int nranks, int rank, const void *id_bytes);
/* Loop-back transport for boxes with ONE GPU: nranks contexts of one process (all of them may sit
 * on the same device), each driven by its own host thread, form a group whose collectives are host
 * barriers around device copies.  Everything around the collective - strip packing, halo margins,
 * split interior/boundary launches, staged scalars, the row-block arithmetic for rank > 0 - is the
 * code that runs over RCCL.  For tests; not a performance path.  Calls on the contexts of a group
 * must be made by all ranks in the same order (as with RCCL); a rank that never arrives makes the
 * others fail with code 2999 after 120 s. */
int kfsp_loopback_create(int nranks, void **group);
int kfsp_loopback_destroy(void *group);   /* after the contexts that used it */
int kfsp_comm_init_loopback(kfsp_ctx *ctx, void *group, int rank);
int kfsp_row_block(const kfsp_ctx *ctx, int64_t n, int64_t *row0, int64_t *nrows);
/* the same arithmetic without a context (host only; usable without a GPU):
 * block of `rank` out of `nranks` for n states, and the padded block length L */
int kfsp_partition(int64_t n, int nranks, int rank, int64_t *row0, int64_t *nrows, int64_t *block_len);

/* ONE host thread driving the whole partition: a GROUP context is a head handle over nranks ordinary contexts,
 * one worker thread each, on devices[0..nranks) - distinct devices talk RCCL, a device named more than once (or
 * devices == NULL: all on device 0) makes the group a loop-back group on that device (one-GPU rehearsal).  The
 * head behaves like a ONE-RANK context: whole vectors, whole arrays, GLOBAL sizes in and out, while generator
 * rows, Krylov basis and w are partitioned over the ranks.  This is how a serial caller - DGEXPV_FSP's Fortran
 * host (KrylovSolver.f90:40) is one - uses several GPUs: every entry point below accepts a head; kfsp_dgexpv and
 * its drop / expand callbacks (KrylovSolver.f90:509-534) run once, on the caller's thread and the caller's one copy
 * of the state space.  Scalars all ranks must agree on (beta, H, AVNORM, WSUM, the drop plan) are compared bit
 * for bit on the way back: 4002 if they ever differ.  Not on a head: kfsp_comm_init*, kfsp_selftest_stream.
 * WATCHDOG: the fan-out of a call never waits for ever.  Once a rank has returned an error its peers get "group_grace_ms"
 * (option on the head; default 15 s, environment KFSP_GROUP_GRACE_S) to come back, and every call has "group_timeout_ms"
 * (default 1800 s, KFSP_GROUP_TIMEOUT_S) in all; when either expires the communicators of all ranks are aborted
 * (ncclCommAbort / the loop-back transport's release), which brings back the ranks that sat in a collective their peer never
 * entered; the call returns the failing rank's code with "rank p: ..." as the error text (2999 when no rank failed and only
 * the deadline expired), and the group is BROKEN: every later call returns 2999 at once - destroy it and create a new one.
 * 2998: ranks did not come back even after the abort ("group_settle_ms", default 30 s); their threads are abandoned, the
 * caller still gets control back.  The process is never ended.  ("group_inject_failure" = p is a test hook: rank p fails the
 * next fan-out with -77 before doing anything.)  (KFSP_NRANKS with distinct KFSP_DEVICES - RCCL between real
 * devices - has not run on hardware yet: no multi-GPU box was available to the builder; tests/test_gpu_two_ranks.py is the
 * first thing to run on one.) */
int kfsp_create_group(int nranks, const int *devices, kfsp_ctx **out);
/* The watchdog alone, without any device (tests/test_group_watchdog.py): nranks worker threads over a loop-back transport;
 * every rank works work_ms, then rank failing_rank (-1: none) returns -77 before the collective, rank hanging_rank (-1:
 * none) sleeps hang_ms instead of entering it, the others wait in one collective.  Out: the code the fan-out returned,
 * the rank it blamed (-1: none), the seconds it took, whether the group ended broken / stuck. */
int kfsp_group_selftest(int nranks, int failing_rank, int hanging_rank, int work_ms, int hang_ms, int timeout_ms, int grace_ms,
                        int settle_ms, int *rc_out, int *who_out, double *seconds, int *broken, int *stuck);
/* ranks behind a context: 1 for an ordinary one */
int kfsp_group_size(const kfsp_ctx *ctx, int *nranks);

/* ---- generator ------------------------------------------------------- */
/* TYPE FSP_MATRIX verbatim (StateSpace.f90:13-17): ADJ(bw,n) int32 1-based
 * (0 = successor outside the FSP, -1 = illegal), OFFDIAG(bw,n), DIAG(n)
 * (positive), leading dimension ld >= bw.  Builds the row-gather form on the
 * device.  Call again whenever the FSP changed (after MATRIX_STARTER /
 * ONESTEP_EXTENDER / SSA_EXTENDER / DROP_STATES: KrylovSolver.f90:130-134,
 * 511, 528-529). */
int kfsp_set_matrix_ell(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld,
                        const int32_t *adj, const double *offdiag, const double *diag);
/* The same after the FSP GREW (ONESTEP_EXTENDER / SSA_EXTENDER, KrylovSolver.f90:528-529): the propensities
 * of a listed state never change while it stays listed (StateSpace.f90:207-212), so the OFFDIAG / DIAG
 * columns of the first n_unchanged states - the FSP the last kfsp_set_matrix_ell / kfsp_update_matrix_ell
 * was given, or a leading part of it - are taken from the device's copy and only the columns behind them
 * travel (their arguments are still the WHOLE arrays); ADJ is uploaded in full (links of old states do
 * change).  n_unchanged is ignored (treated as 0) when the device does not hold those columns any more
 * (another kind of generator was set in between, kfsp_drop_compact ran, ld changed). */
int kfsp_update_matrix_ell(kfsp_ctx *ctx, int32_t n, int32_t bw, int32_t ld, const int32_t *adj,
                           const double *offdiag, const double *diag, int32_t n_unchanged);
/* Optional, before kfsp_set_matrix_ell: the species counts of the n states of
 * that generator, FSP%STATE(1:ns, 1:n) (StateSpace.f90:22), leading dimension
 * ld >= ns.  The reference lists states in the order SSA_EXTENDER /
 * ONESTEP_EXTENDER discovered them (:347-396, :550-630), which scatters the
 * neighbours of a state over the whole vector; with the coordinates the library
 * can keep generator and vectors in lexicographic state order INTERNALLY
 * (species 1 fastest), so that the gathers of a product coalesce (2x on the
 * product at 10^6 discovery-ordered states).  Nothing changes at the boundary:
 * every array handed in or out stays in the caller's order.
 * Every row is still summed in FMATVEC's order (KrylovSolver.f90:598-604: its entries are
 * kept sorted by the CALLER's column index), so products are bit-identical to the plain path;
 * only the reductions over states (dot products, norms, WSUM) add their terms in another
 * order.  ON by default (option state_order = 0 switches it off); it applies to the next
 * kfsp_set_matrix_ell with the same n only, and not below option state_order_min states
 * (default 32768) or while generators are short-lived (the one being
 * replaced saw fewer than option state_order_products products, default 48: reordering costs
 * about 20 of them at 10^6 states).
 * With a communicator the order is GLOBAL: every rank sorts all n keys (same input, same result), owns the
 * block [row0, row0 + nloc) of the internal order, and a vector handed in or out as "this rank's block of the
 * caller's order" passes through one all-gather of the whole vector (kfsp_set_vector / kfsp_get_vector /
 * kfsp_spmv / kfsp_get_basis; a few times per FSP change, never per product). */
int kfsp_set_state_coords(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state);
/* The same after the FSP GREW: the coordinates of the first n_unchanged states - a leading part of what the last
 * kfsp_set_state_coords / kfsp_update_state_coords / kfsp_drop_rebuild left on the device - are not sent again (a listed
 * state never changes; `state` is still the whole array).  n_unchanged is treated as 0 when those are not resident. */
int kfsp_update_state_coords(kfsp_ctx *ctx, int32_t n, int32_t ns, int32_t ld, const int32_t *state, int32_t n_unchanged);
/* 1 if the generator last set is held in the internal state order */
int kfsp_state_order_active(const kfsp_ctx *ctx, int *active);
/* Synthetic / pre-transposed input: gather rows [row0, row0+nrows) of an
 * n x n generator in CSR, 0-based GLOBAL column indices, the diagonal stored
 * as an ordinary (negative) entry.  rowptr has nrows+1 entries starting at 0. */
int kfsp_set_matrix_csr(kfsp_ctx *ctx, int64_t n, int64_t row0, int64_t nrows,
                        const int64_t *rowptr, const int32_t *col, const double *val);
/* Matrix-free generator of a lexicographic BOX [0,dims[0]) x ... x [0,dims[ns-1]) (species 1
 * fastest; n = prod dims states, state index = sum x_s stride_s) for propensities that are products
 * of one-species factors, a_k(x) = prod_{i < ndep[k]} T_{k,i}[x_{dep_species[k][i]}] - mass action,
 * Hill functions of one species, ...: the kernel stores NO generator entries, it rebuilds every row
 * of FMATVEC (KrylovSolver.f90:577-607) on the FSP = the box from the row index and the factor
 * tables (DIAG = the sum of ALL propensities, StateSpace.f90:207-212), moving 16 B per state
 * (x once, y once) instead of 8 B per stored entry + 24 B.  The tables are made on the host with
 * the model's own propensity code (ModelModule.f90:163-199 evaluated at every population count),
 * so a propensity with one factor has exactly the stored value.
 *   stoich       [nr][ns] state change of each reaction
 *   ndep         [nr]     factors of each propensity, 1..3
 *   dep_species  [nr][3]  0-based species of each factor (unused entries ignored)
 *   tables       the factors, concatenated reaction by reaction, factor by factor; factor (k, i) has
 *                dims[dep_species[k][i]] entries (a constant propensity is one table of equal entries)
 * Works with a communicator like a banded generator (halo strips).  ns <= 8, nr <= 16, every reaction
 * changes <= 3 species, all tables together <= 6000 entries.
 * Option "box_store" = 1: the same call WRITES THE GENERATOR OUT on the device as stored diagonals (the banded
 * form kfsp_set_matrix_csr would build from the gather rows of this box - with one factor per propensity the
 * very same entries) - a stored generator of any size without host arrays of that size. */
int kfsp_set_matrix_box(kfsp_ctx *ctx, int32_t ns, const int32_t *dims, int32_t nr, const int32_t *stoich,
                        const int32_t *ndep, const int32_t *dep_species, const double *tables);
/* what the device holds: rows (local), stored off-diagonal slots incl. padding,
 * true nonzeros incl. diagonal (local rows) */
int kfsp_matrix_info(const kfsp_ctx *ctx, int64_t *nrows, int64_t *slots, int64_t *nnz);
/* HBM bytes ONE product launch has to move for the layout the device holds (local rows): the
 * stored generator (banded: 8 B per stored diagonal entry, minus the empty 128-row segments the
 * masked kernel skips, plus its mask words; SELL-64: 12 B per slot incl. padding plus chunk
 * offsets) + 24 B per row (DIAG, x once, y).  x re-fetches are not in it; the rocprofv3 counters
 * are.  force_sell = 1: the figure for kfsp_spmv_bench variant 2; 3: for variant 3 (plain columns).
 * SELL-64 with dictionary-coded columns (format 5): coded chunks count 8 B per entry + their code words + their
 * offset tables (in 64-byte lines) instead of 12 B per entry. */
int kfsp_matrix_bytes(const kfsp_ctx *ctx, int force_sell, int64_t *bytes);
/* what the device holds and how a partitioned product exchanges its source vector, v[8]:
 *   v[0] kernel format: 0 SELL-64, 1 banded, 2 banded with group masks, 3 / 4 matrix-free box (interpreted / fast path),
 *        5 SELL-64 with dictionary-coded columns (option "sell_code"), 6 matrix-free fast path with the near part of x in LDS,
 *        7 matrix-free fast path in pencils along the slowest species (option "box_pencil"), 8 the same in slabs (box_pencil = 2)
 *   v[1] exchange: 0 none (no communicator), 1 halo strips, 2 all-gather of the whole vector;  v[2] halo rows
 *   v[3] reach max |col - row| of the local SELL rows (-1: not a SELL generator)
 *   v[4] chunks with coded columns, v[5] chunks, v[6] 64-bit code words, v[7] internal state order active */
int kfsp_layout_info(const kfsp_ctx *ctx, int64_t *v);
/* How the rebuilds of a RESIDENT FSP went (kfsp_expand_resident, kfsp_drop_rebuild; option "build_speculate"), v[6]:
 *   v[0] generators built without stopping for the link statistics and the slot count, v[1] of those (or of the speculative
 *   state orders before them) repeated the slow way because a check at the end failed - a coordinate range crossed a power of
 *   two, or the slow path would have stored diagonals, v[2] the last generator built from resident arrays is a SELL one,
 *   v[3] a key layout is cached for the next state order, v[4] state orders made FROM the previous one - the appended states'
 *   keys sorted and merged into the kept list, or the list compacted after a drop - instead of sorting every key, v[5] 0.
 *   A group context answers for its first rank. */
int kfsp_build_info(const kfsp_ctx *ctx, int64_t *v);
/* The ORDER in which a product takes its wavefront trips (one trip = 128 consecutive rows of a banded or matrix-free
 * generator, 64 of a SELL one; ntrips = ceil(local rows / that)): position t of the sweep computes trip order[t].  Rows
 * are independent (FMATVEC as a row gather), so y does not depend on the order - bits included; what changes is which rows
 * of x are in an XCD's L2 when their far neighbours are computed.  kfsp_set_matrix_box chooses a tiled order itself for
 * boxes with strides beyond the L2 (option "box_tile"); this entry point is for experiments.  Belongs to the current
 * generator (the next kfsp_set_matrix_* drops it); ntrips = 0 restores the ascending order.  Single rank, whole-product
 * launches. */
int kfsp_set_trip_order(kfsp_ctx *ctx, int64_t ntrips, const int32_t *order);
/* global number of states of the generator last set (FSP%SIZE) */
int kfsp_num_states(const kfsp_ctx *ctx, int64_t *n);

/* ---- probability vector w (the solver's W == FSP%VECTOR, KrylovSolver.f90:33-34) */
/* local row block of the context (the whole vector when nranks == 1) */
int kfsp_set_vector(kfsp_ctx *ctx, int64_t nlocal, const double *w);
int kfsp_get_vector(kfsp_ctx *ctx, int64_t nlocal, double *w);

/* ---- the hot path ---------------------------------------------------- */
/* beta = ||w||_2 and v_1 = w / beta.  KrylovSolver.f90:177/540 and :223-226. */
int kfsp_begin_step(kfsp_ctx *ctx, double *beta);

/* IOP Arnoldi columns jold..m plus the extra product for AVNORM.
 * KrylovSolver.f90:236-266: FMATVEC, DDOT/DAXPY over the last qiop vectors,
 * DNRM2, happy-breakdown test, DSCAL; then H(m+2,m+1) = 1.
 * H is the host Hessenberg image, column-major, leading dimension ldh >= m+2;
 * columns jold..m (and the corner) are written, others left alone.
 * *mbrkdwn = m or the breakdown column; *k1 = 2 or 0 (:250); *avnorm as :263.
 * The basis stays on the device; a later call with jold > 1 extends it
 * (dimension change, :400-432), including the reference's behaviour when the
 * dimension shrank below jold (extra product taken from column jold). */
int kfsp_arnoldi(kfsp_ctx *ctx, int m, int jold, int qiop, double break_tol,
                 double *H, int ldh, int *mbrkdwn, int *k1, double *avnorm);

/* w = beta * V(:,1:mx) * y ; w = max(w,0) ; *wsum = ||w||_1.
 * KrylovSolver.f90:444 (DGEMV), :447-449 (clamp), :450 (DASUM) in one pass. */
int kfsp_combine(kfsp_ctx *ctx, int mx, double beta, const double *y, double *wsum);

/* w = beta * v_1 : the FSP-rejection path, KrylovSolver.f90:467. */
int kfsp_restore_w(kfsp_ctx *ctx, double beta);

/* y = A x through the device kernel (the FMATVEC seam, KrylovSolver.f90:577;
 * also what DROP_STATES receives as its matvec, StateSpace.f90:486).
 * x: full length-n vector on every rank, y: local row block. */
int kfsp_spmv(kfsp_ctx *ctx, const double *x, double *y);
/* y = A w for the resident w (DROP_STATES, StateSpace.f90:486), local rows */
int kfsp_spmv_w(kfsp_ctx *ctx, double *y);

/* ---- DROP_STATES on the device (StateSpace.f90:398-427, :470-546) ---------------- */
/* The decision on the resident w and generator, without moving either to the host:
 *   droptol    FIND_DROPTOL's threshold (:416-426): the first of 1e-8, 1e-9, ... (the reference's
 *              own divisions by 10) whose sum of the entries 0 < w < threshold is below dsum; all
 *              thresholds of a pass are summed in ONE sweep over w (same entries per threshold as the
 *              reference's sweeps; fixed reduction order, reproducible run to run)
 *   drop_count the reference's DROP_COUNT (:476-495) = #(w < droptol) - #((A w) > 1e-8), its
 *              counting quirk included: the caller compacts iff drop_count / n > 0.1 (:497)
 *   n_flagged  states actually flagged: w < droptol and not (A w) > 1e-8
 * The flags stay on the device for the two calls below.
 * With a communicator (all ranks call): every rank sums the thresholds over its block, ONE all-reduce carries
 * the sixteen sums of a pass; marks and counts per block, counts all-reduced, the flag bytes of all blocks
 * all-gathered - every rank then holds the plan and the flags of ALL states (its host copy of the state
 * lists needs them all). */
int kfsp_drop_plan(kfsp_ctx *ctx, double dsum, double *droptol, int64_t *drop_count, int64_t *n_flagged);
/* the flags of the last plan, one byte per state in the caller's order (1 = dropped) */
int kfsp_drop_flags(kfsp_ctx *ctx, int64_t n, uint8_t *dropped);
/* Compacts the resident w by those flags (kept entries keep their order, :500-546 for W); *n_new =
 * n - n_flagged.  The context then WAITS for the generator of the compacted FSP: the next
 * kfsp_set_matrix_ell / _csr must have n_new states and makes the compacted vector its w (do not
 * call kfsp_set_vector in between, it would replace it).
 * With a communicator the compacted FSP has another partition: every rank assembles the whole vector (one
 * all-gather), compacts it with the flags it holds, and takes its new block when the generator arrives. */
int kfsp_drop_compact(kfsp_ctx *ctx, int64_t *n_new);

/* After kfsp_drop_compact: the generator of the compacted FSP from the device's OWN copy of the reference arrays - the
 * columns of the kept states move up in list order, ADJ is renumbered through the keep-prefix-sum, dropped targets become 0
 * (StateSpace.f90:500-546 for STATE / ADJ / OFFDIAG / DIAG), the gather form is rebuilt and the compacted w adopted: no
 * array travels.  Stands in for the kfsp_set_state_coords + kfsp_set_matrix_ell the host would otherwise send for the
 * compacted FSP (the host still compacts ITS copy of the lists with the flags).  -9 when the arrays of this FSP are not
 * resident (a CSR / box generator, option host_build): upload the compacted generator as before. */
int kfsp_drop_rebuild(kfsp_ctx *ctx);

/* ---- ONESTEP_EXTENDER on the device (StateSpace.f90:347-396 with ADD_STATE :136-246) ---- */
/* The integer work of one reachability sweep over the listed states state[0..n) (ns counts each,
 * leading dimension ld_state) with link array adj (nr links each, leading dimension ld_adj; the
 * reference's encoding: 1-based index, 0 = target not listed, -1 = negative target): every open
 * link is followed; the DISTINCT unlisted targets become new states n+1.. in the order in which
 * the reference's double loop (state by state, reaction by reaction) meets them first; all links
 * of old and new states are completed.  Out: *n_new = new number of states; state_new = the
 * appended states (leading dimension ld_state, room for capacity - n of them); adj_out = the
 * complete link array of all *n_new states (leading dimension ld_adj, room for capacity states).
 * stoich is [nr][ns]; max_count = MAXNUMBERMOLECULES (StateSpace.f90:11): targets above it are not
 * states.  Propensities (OFFDIAG, DIAG) of the new states and the caller's own look-up structures
 * stay with the caller.  Everything is host memory.  Deterministic: the listed states and the candidates' targets
 * go through two open-addressing tables, a slot keeps the SMALLEST candidate ordinal that named its target (atomic
 * min - the survivor does not depend on the order of the insertions), and the heads are numbered by a scan over the
 * states (csrc/kfsp_expand.hip; round 2's two radix sorts and its 63-bit key limit are gone).  A column that arrives
 * as zeros (a state appended but not linked yet) is completed like any other.
 * returns -9 for more than 2^31 (state, reaction) pairs, -11 when capacity is too small (the reference STOPs with
 * 'FSP SIZE EXCEEDS MEMORY LIMIT'). */
int kfsp_onestep(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state,
                 int32_t ld_state, const int32_t *adj, int32_t ld_adj, int32_t max_count, int32_t capacity, int32_t *n_new,
                 int32_t *state_new, int32_t *adj_out);

/* ---- propensities on the device (ModelModule.f90:163-199 through the stack code of FortranParser.f90:187-302) ---- */
/* The model's propensity program, once per model (and again after RESET_PARAMETERS): reaction k is the postfix code
 * code[code_off[k] .. code_off[k+1]) with immediates imm[imm_off[k] ..) in order of use.  Opcodes (those of the host's
 * expression type, krylovfspssa_amd/fortran/kfsp_expr.f90, which tests/golden/exprtable.npz pins against the
 * reference's parser): 1 IMM, 2 NEG, 3 ADD, 4 SUB, 5 MUL, 6 DIV, 7 POW, 10 + f for the functions f = 1..14 abs exp
 * log10 log sqrt sinh cosh tanh sin cos tan asin acos atan, 100 + i for variable i = 1..ns (species counts) and
 * ns+1..ns+nparams (parameters); x / 0, log / log10 of x <= 0, sqrt of x < 0, asin / acos outside [-1, 1] make the
 * whole expression 0, as in the reference.
 * + - * / NEG are evaluated one IEEE operation at a time: the same bits as the host.  pow and the functions come
 * from the device's math library (<= 2 ulp from the host's).  So that the columns stay BIT-IDENTICAL to the host's
 * wherever that is possible, a propensity that depends on ONE species only may be handed over as a table made with
 * the host's own evaluator: tab_species[k] = that species (0-based) or -1, tab[k * tab_len + x] = a_k at population x
 * (tab_len may be 0: no tables; populations >= tab_len are interpreted).  Stack depth <= 32. */
int kfsp_set_propensity_program(kfsp_ctx *ctx, int32_t ns, int32_t nr, int32_t nparams, const double *params,
                                const int32_t *code_off, const int32_t *code, const int32_t *imm_off, const double *imm,
                                const int32_t *tab_species, int32_t tab_len, const double *tab);
/* Tables over TWO species, for the program just set - how a COMPILED-IN propensity function (MODEL%CUSTOMPROP,
 * ModelModule.f90:163-199, examples/transcr6d.f90:63-90) reaches the device: there is no code to hand over, so the host
 * tabulates the user's own function (exact by construction).  Reaction k with s1[k] >= 0 depends on species s1[k] and s2[k]
 * (0-based) and reads tab2[off[k] + x[s2] * n1[k] + x[s1]] for populations x[s1] < n1[k], x[s2] < n2[k]; s1[k] < 0: no such
 * table (code / product chain / one-species table of kfsp_set_propensity_program apply).  len = doubles in tab2.
 * A population BEYOND a table cannot be served: the operation that met it - kfsp_propensities, kfsp_onestep_columns,
 * kfsp_ssa_streams, kfsp_expand_resident - returns -16 and has changed nothing the caller can see (host outputs are to be
 * discarded; the resident lists still describe the FSP as it was); kfsp_propensity_overflow says which species went how
 * far; the caller sets larger tables (kfsp_set_propensity_program + this call) and repeats the operation - deterministic,
 * so the repeat is what a larger table would have given at once. */
int kfsp_set_propensity_tables2(kfsp_ctx *ctx, int32_t nr, const int32_t *s1, const int32_t *s2, const int32_t *n1, const int32_t *n2,
                                const int64_t *off, int64_t len, const double *tab2);
/* after a -16: max_missed[s] = the largest population of species s that lay beyond a table (0: none) */
int kfsp_propensity_overflow(kfsp_ctx *ctx, int32_t ns, int32_t *max_missed);
/* OFFDIAG(1:nr, i) = a_k(x_i) and DIAG(i) = their sum in reaction order (ADD_STATE, StateSpace.f90:207-212) for the
 * n states state[ld_state][n] (host arrays in and out) */
int kfsp_propensities(kfsp_ctx *ctx, int32_t n, const int32_t *state, int32_t ld_state, double *offdiag, int32_t ld_off,
                      double *diag);
/* kfsp_onestep that also returns the COMPLETE columns of the appended states: offdiag_new[ld_off][*n_new - n] and
 * diag_new[*n_new - n] from the propensity program - the host then only enters the new states into its look-up
 * table.  -15 without a program for (ns, nr). */
int kfsp_onestep_columns(kfsp_ctx *ctx, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n, const int32_t *state,
                         int32_t ld_state, const int32_t *adj, int32_t ld_adj, int32_t max_count, int32_t capacity,
                         int32_t *n_new, int32_t *state_new, int32_t *adj_out, double *offdiag_new, int32_t ld_off,
                         double *diag_new);

/* ---- SSA paths on the device: the INDEPENDENT-STREAM expansion ---------------------------------------------- */
/* The reference's SSA_EXTENDER (StateSpace.f90:550-630) draws all paths from ONE random stream and lets each see the
 * states of the earlier ones: sequential by definition, it stays on the host.  What runs here is the host's opt-in
 * variant (SSA_EXTENDER_STREAMS of krylovfspssa_amd/fortran/kfsp_statespace.f90, KFSP_SSA_STREAMS=1): one path of
 * length `timestep` from EVERY listed state, each on a Lehmer stream of its own seeded from (seedmix, index of the
 * seed state), walking the FSP as it stands (state / adj / offdiag / diag: the reference's arrays, host memory),
 * passing through unlisted states on propensities of the model's program (kfsp_set_propensity_program), stopping at
 * a negative or > max_count population, at an absorbing state, or when it falls back onto an earlier seed.  Out: the
 * DISTINCT unlisted states the paths met, in (seed state, position on the path) order of their first occurrence -
 * state_new[ld_state][*n_found] - with their propensity columns offdiag_new[ld_off][*n_found], diag_new[*n_found];
 * the caller appends and links them.  The arithmetic (generator, reaction choice, the waiting time through a
 * fixed-sequence logarithm) is defined so that the host variant and this one give the same states in the same
 * order, bit for bit, whenever the program's propensities are bit-exact (tables / + - * /, i.e. all shipped models).
 * Returns -11 when more than capacity_new states were found (nothing is returned then). */
int kfsp_ssa_streams(kfsp_ctx *ctx, double timestep, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t n,
                     const int32_t *state, int32_t ld_state, const int32_t *adj, const double *offdiag, int32_t ld_adj,
                     const double *diag, int32_t max_count, int32_t capacity_new, int32_t *n_found, int32_t *state_new,
                     double *offdiag_new, int32_t ld_off, double *diag_new);

/* ---- the expansion step on the RESIDENT lists (KrylovSolver.f90:518-534) ---------------------------------------- */
/* SSA_EXTENDER (the independent-stream walk of kfsp_ssa_streams; skipped when t_ssa <= 0) followed by ONESTEP_EXTENDER
 * (kfsp_onestep_columns) on the device's OWN copy of the FSP - the coordinates kfsp_set_state_coords left there (option
 * keep_coords = 1 makes them stay whatever the state order decides) and the reference arrays of kfsp_set_matrix_ell /
 * kfsp_update_matrix_ell / kfsp_drop_rebuild.  The states the walk met are appended with their propensity columns and
 * no links, the sweep completes every link and appends its own states, the gather form (and the state order, by the
 * rule of kfsp_set_state_coords) is rebuilt from the grown arrays, the resident w gets zeros for the appended states
 * (:530-533).  Nothing crosses the bus but counters; the same states in the same order, the same links and columns as
 * the two calls above give on host copies of the lists (tests/test_gpu_expand.py).  *n_new = states afterwards,
 * *n_from_ssa (may be null) = those the walk appended.  -9 when the arrays or coordinates of the current FSP are not
 * resident; -11 when more than `capacity` states would be listed.  Under a row partition (all ranks call; a group
 * context does it for them) every rank holds the whole lists and expands them redundantly - deterministic kernels,
 * identical results, nothing to exchange but the vector, which is assembled in the caller's order (one all-gather) and
 * dealt out again in the new partition - and rebuilds its own row block.
 * A caller that keeps its own copy of the lists refreshes it with kfsp_download_fsp when it needs it. */
int kfsp_expand_resident(kfsp_ctx *ctx, double t_ssa, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich,
                         int32_t max_count, int32_t capacity, int64_t *n_new, int64_t *n_from_ssa);
/* the resident lists -> host arrays (any pointer may be null): state[ld_state][n], adj / offdiag[ld_adj][n], diag[n] in
 * the caller's order and the reference's encoding; n must be the current number of states */
int kfsp_download_fsp(kfsp_ctx *ctx, int32_t n, int32_t *state, int32_t ld_state, int32_t *adj, double *offdiag, int32_t ld_adj,
                      double *diag);

/* single reductions over the resident w (tests; FIND_DROPTOL-style sums) */
int kfsp_nrm2_w(kfsp_ctx *ctx, double *out);
int kfsp_asum_w(kfsp_ctx *ctx, double *out);
/* copy basis column j (1-based, normalised) of the local block to the host */
int kfsp_get_basis(kfsp_ctx *ctx, int j, int64_t nlocal, double *v);

/* ---- host dense kernel (stays on the host by design) ------------------ */
/* exp(t*H), (ideg,ideg) Pade + scaling/squaring = DGPADM / DGPADMnorm
 * (src/expokit/dgpadm.f:2-169, :171-339).  E: m*m column-major (ld m). */
int kfsp_padm(int ideg, int m, double t, const double *H, int ldh, double *E,
              int *ns, double *hnorm);
/* diagnostic: seconds spent so far inside kfsp_padm in {its dense products (the squaring phase; dealt to KFSP_PADE_THREADS
 * threads), its banded products, the LU solve, whole calls}; reset != 0 clears the counters */
void kfsp_padm_profile(double *seconds4, int reset);

/* ---- the adaptive solver ---------------------------------------------- */
/* Seams of DGEXPV_FSP through which the host's state-space code is reached.
 * A callback may call back into this library (kfsp_get_vector, kfsp_spmv_w,
 * kfsp_set_matrix_ell, kfsp_set_vector ...) and must leave the context holding
 * the generator and the probability vector of the (possibly changed) FSP.
 *   drop   : DROP_STATES(W, FSP, MODEL, DSUM, FMATVEC), KrylovSolver.f90:509-512
 *   expand : SSA_EXTENDER(T_SSA, ..) + ONESTEP_EXTENDER, :518-534
 *   log    : the reference's VERBOSITY prints and the unconditional WSUM print
 *            (:233,384-389,427-431,452,502,523); vals layout per event below.
 * Any of them may be NULL: no drop = nothing is ever dropped, no expand =
 * kfsp_dgexpv returns 10 when the FSP criterion calls for an expansion. */
enum {
    KFSP_EV_BEGIN_IOP = 1,   /* (none) */
    KFSP_EV_WSUM = 2,        /* wsum */
    KFSP_EV_STEP = 3,        /* nstep, n, t_step, t_new, t_now, m */
    KFSP_EV_REJECT_STEP = 4, /* t_old, err_loc, err_required, t_step_new */
    KFSP_EV_DIM_CHANGE = 5,  /* err_loc, err_required, m_new */
    KFSP_EV_CALL_SSA = 6,    /* t_ssa */
    KFSP_EV_READY = 7        /* nstep, t_now, beta, n: w and the FSP are final for the step that
                                begins next (after :540; also once before the first step, :177) */
};
typedef struct {
    void *user;
    int (*drop)(void *user, double dsum, int64_t *n_new);
    int (*expand)(void *user, double t_ssa, int64_t *n_new);
    void (*log)(void *user, int event, const double *vals, int nvals);
} kfsp_fsp_ops;

/* What the reference gathers in IWSP(1:7)/WSP(1:10) and then discards
 * (KrylovSolver.f90:554-573), plus counts of the FSP events. */
typedef struct {
    int32_t nmult, nexph, nscale, nstep, nreject, ibrkflag, mbrkdwn;
    int32_t n_wsum, n_expand, n_drop_calls;
    double step_min, step_max, x_error, s_error, tbrkdwn, t_now, hump, beta;
} kfsp_stats;

/* DGEXPV_FSP (KrylovSolver.f90:40-653) from :151 on: the adaptive time loop on
 * the context's resident generator and vector w (= V on entry, = W on exit).
 * MATRIX_STARTER and the five initial ONESTEP_EXTENDER calls (:130-134) are
 * the caller's (they build what is uploaded first).  n_reactions is
 * MODEL%NREACTIONS (enters the cost model through NNZ, :196,:537).
 * returns 0; 10 = expansion needed but ops->expand == NULL; <0 bad argument;
 * >0 device/host failure as everywhere. */
int kfsp_dgexpv(kfsp_ctx *ctx, double t, double fsptol, double krytol, int n_reactions,
                const kfsp_fsp_ops *ops, kfsp_stats *stats);

/* ---- lock-step diagnostics ---------------------------------------------- */
/* The accept/reject decisions of DGEXPV_FSP hinge on quantities that amplify
 * rounding differences (the tail of exp(tau*H) after ~75 IOP columns), so two
 * correct implementations eventually take different - equally valid - steps.
 * To compare the ARITHMETIC of this library with a recorded run of another
 * implementation step by step, kfsp_dgexpv_replay runs the loop of kfsp_dgexpv
 * but lets a record of that run make every choice.  Each choice is still
 * computed here first; where it differs from the record a kfsp_fork entry says
 * which comparison went the other way and by how much.
 *
 * script: n_rows rows of 4 doubles (kind, a, b, c), in the order the loop
 * consumes them:
 *   1 BEGIN   (t_step, m, -)        a time step starts with T_STEP (:208; 0 = not recorded,
 *                                   nothing is compared or forced) and M (:211)
 *   2 KRYLOV  (code, t_step, m)     outcome of the error test (:375) after each Pade
 *                                   evaluation: 0 accept; 1 redo with step size t_step on the
 *                                   same basis (:377-399); 2 extend the basis to dimension m and
 *                                   use step size t_step (:400-432)
 *   3 FSP     (code, t_step, wsum)  outcome of the mass test (:458) after each solution
 *                                   update; wsum = the recorded WSUM (compared, see
 *                                   max_wsum_diff): 0 accept; 1 retry with step size t_step
 *                                   (:471-493); 2 fifth failure, restore w and expand (:466-470)
 *   4 END     (n, t_new, -)         the step is over: n = size of the FSP the next step runs
 *                                   on (compared after the drop / expand callbacks); t_new =
 *                                   T_NEW as known from the recorded run at :502 (0 = unknown)
 * returns like kfsp_dgexpv; 20 = the script ended early or its next row is not
 * of the kind the loop needs (the runs are no longer comparable); 21 = a happy
 * breakdown on one side only; -8 bad replay. */
enum {
    KFSP_FORK_BEGIN_TAU = 1,     /* own[0]/forced[0] = T_STEP; lhs = T_NEW, rhs = T_OUT - T_NOW */
    KFSP_FORK_BEGIN_M = 2,       /* own[0]/forced[0] = M; lhs = M_NEW, rhs = N - 1 */
    KFSP_FORK_KRYLOV_TEST = 3,   /* accept vs reject: own/forced = (code, value); lhs = OMEGA, rhs = DELTA */
    KFSP_FORK_KRYLOV_CHOICE = 4, /* step size vs dimension: lhs = COST1, rhs = COST2 (:359-362) */
    KFSP_FORK_KRYLOV_VALUE = 5,  /* same branch, other value: own/forced = (t_step, m) */
    KFSP_FORK_FSP_TEST = 6,      /* own/forced = (code, wsum); lhs = WSUM, rhs = 1 - FERRORBOUND */
    KFSP_FORK_FSP_TAU = 7,       /* own[0]/forced[0] = T_STEP of the retry; lhs = ERROR, rhs = FSPORDER */
    KFSP_FORK_T_NEW = 8,         /* own[0]/forced[0] = T_NEW after the step; lhs = OMEGA, rhs = ORDER */
    KFSP_FORK_FSP_SIZE = 9,      /* own[0]/forced[0] = FSP size after the callbacks */
    KFSP_FORK_UNSAFE_ACCEPT = 10,/* safe mode: the recorded accept fails this run's test even at M_MAX
                                    and is carried out all the same; lhs = OMEGA, rhs = DELTA */
    KFSP_FORK_BREAKDOWN = 11     /* happy breakdown (:249) on one side only: own[0] = column here (0 = none),
                                    forced[0] = 1 if recorded; lhs = h(j+1,j), rhs = BREAK_TOL; returns 21 */
};
typedef struct {
    int32_t step;        /* NSTEP at the decision */
    int32_t kind;        /* KFSP_FORK_* */
    double own[2];       /* what this run computed */
    double forced[2];    /* what the record made it do */
    double lhs, rhs;     /* the two sides of the comparison behind the choice */
} kfsp_fork;
typedef struct {
    const double *script;   /* in: rows of 4 doubles */
    int64_t n_rows;
    kfsp_fork *forks;       /* in: room for max_forks entries (may be NULL) */
    int32_t max_forks;
    int32_t n_forks;        /* out: differences found (may exceed max_forks; the first ones are kept) */
    int64_t rows_used;      /* out */
    double max_wsum_diff;   /* out: max |WSUM here - recorded WSUM| over all solution updates */
    int32_t safe;           /* in: 0 = carry out every recorded choice; 1 = when the record accepts a
                               Krylov step that fails THIS run's error test (OMEGA > DELTA), keep the
                               recorded step size but enlarge the basis by the run's own rule until the
                               test passes (m < M_MAX), so that no error the record does not have is
                               injected and the two runs stay comparable step by step; 2 = the same, and
                               return 21 at the first step that leaves a state list of another size than
                               the record's (KFSP_FORK_FSP_SIZE): beyond it the record describes another problem */
    int32_t n_safe_extensions;   /* out: how often that happened */
} kfsp_replay;
int kfsp_dgexpv_replay(kfsp_ctx *ctx, double t, double fsptol, double krytol, int n_reactions,
                       const kfsp_fsp_ops *ops, kfsp_stats *stats, kfsp_replay *replay);

/* ---- benchmark mode --------------------------------------------------- */
/* nsteps steps of fixed Krylov dimension m and fixed step tau on the resident
 * w (BASELINE config 2, "m=30"): begin_step, arnoldi(1..m), Pade of order m+2,
 * combine with mx = m+1.  wsums[nsteps] = mass after each step. */
int kfsp_expv_fixed(kfsp_ctx *ctx, int m, double tau, int nsteps, double *wsums);

/* reps back-to-back launches of the SpMV kernel y = A v_1 on the context's
 * stream, bracketed by HIP events: *ms_total = elapsed GPU time.  With
 * nranks > 1 every launch is preceded by the all-gather of the source slab,
 * as in the solver.  variant: 0 = the format the library chose (banded DIA when
 * the rows allow it, else SELL-64), 2 = SELL-64 even when DIA is active, 3 = SELL-64 reading its plain
 * 4-byte columns even when the dictionary-coded form is active (A/B of format 5 against format 0). */
int kfsp_spmv_bench(kfsp_ctx *ctx, int reps, int variant, float *ms_total);

/* reps exchanges of the source vector ALONE (what precedes every product of a partitioned generator: halo strips
 * or the all-gather of the whole vector), bracketed by HIP events; *bytes_in = bytes this rank receives per
 * exchange (0 and 0 ms without a communicator).  bench.py reports exchange time and link GB/s from it. */
int kfsp_exchange_bench(kfsp_ctx *ctx, int reps, float *ms_total, int64_t *bytes_in);

/* diagnostics: reps launches that read exactly nbytes from a scratch buffer
 * with elem_bytes (4, 8, 16) per lane in the SpMV's access shape; used to
 * calibrate the rocprofv3 FETCH_SIZE counter on a known byte count. */
int kfsp_selftest_stream(kfsp_ctx *ctx, int64_t nbytes, int elem_bytes, int reps, float *ms_total);

/* accumulated wall time (ms) of the synchronous entry points by phase since
 * the last reset: Arnoldi passes (product + orthogonalisation kernels and the
 * copy of H), combine calls, begin_step calls, host Pade, generator uploads
 * (host->device copy + device build), time spent inside the drop / expand
 * callbacks of kfsp_dgexpv, and - a part of the callbacks' time when they use it -
 * the device work of kfsp_onestep incl. its copies. */
enum { KFSP_T_ARNOLDI = 0, KFSP_T_COMBINE = 1, KFSP_T_BEGIN = 2, KFSP_T_CALLBACKS = 3,
       KFSP_T_HOST_PADE = 4, KFSP_T_UPLOAD = 5, KFSP_T_ONESTEP = 6, KFSP_T_COUNT = 7 };
int kfsp_get_timers(kfsp_ctx *ctx, double *ms /* [KFSP_T_COUNT] */, int reset);
/* add ms to a phase (used by kfsp_dgexpv, which is a client of this ABI) */
int kfsp_add_timer(kfsp_ctx *ctx, int phase, double ms);

/* tuning knobs (name/value); unknown name -> -2: "grid_blocks",
 * "vec_grid_blocks", "nt_loads", "format", "fused_ortho",
 * "host_build", "halo", "halo_p2p" (1: halo strips travel between neighbouring ranks only, ncclSend/ncclRecv straight into the
 * column margins; 0, default: one all-gather of every rank's strips), "halo_sell" (0: SELL generators always all-gather the whole source
 * vector; 1, default: a SELL generator whose reach max |col - row| is at most one block - bounded under the internal state order -
 * exchanges halo strips like a banded one), "overlap", "small_kernel", "small_lds", "dia_mask", "box_lds" (1: the single-factor matrix-free product stages the part of x within "box_reach" rows - default
 * 512 - of a workgroup's rows in LDS and serves the near entries from there, kernel format 6; 0, default: every entry gathers from
 * global memory, format 4, which measured faster on every box; bit-identical products), "ssa_partition" (1, default: kfsp_expand_resident under a communicator has every rank walk
 * only ITS share of the SSA seeds and all-gathers the paths' records - the same states in the same order as the unpartitioned walk;
 * 0: every rank walks all seeds), "ssa_general" (1: kfsp_ssa_streams / kfsp_expand_resident walk with the general kernel even for models of <= 8 species and
 * <= 16 reactions, which otherwise take the register-resident one; same paths; default 0), "ssa_regs" (1, default: when every propensity is a product chain of at
 * most three operands or sits behind a one-species table, the register-resident walk evaluates an unlisted state from descriptors it keeps in registers
 * instead of interpreting the program - the paths a launch waits for are the ones that walk outside the FSP; same paths; 0: always the program), "ssa_filter" (1, default:
 * a bit map in front of the walk's table of listed states answers "not listed" for most unlisted targets without a trip to the table; 0: every look-up probes the table), "keep_coords" (1: the coordinates of kfsp_set_state_coords / kfsp_update_state_coords stay on the device
 * even when no state order is derived from them - kfsp_expand_resident needs them; default 0), "ssa_resident" (1: the caller vouches that the FSP arrays handed to kfsp_ssa_streams are the ones of its last
 * kfsp_update_matrix_ell / kfsp_set_state_coords: they are taken from the device's copies instead of being uploaded again; default 0),
 * "sell_code" (dictionary-coded SELL columns, DESIGN.md 4.1c: -1 auto = under the internal state order,
 * 0 never, 1 always try), "m_max" (largest Krylov dimension the basis is allocated for, default and maximum 100 = M_MAX of
 * KrylovSolver.f90:47; a smaller value saves 8 * rows bytes per column - 90 GB at 10^8 states - and makes kfsp_arnoldi refuse
 * a larger m; it takes effect when the NEXT generator is set - until then every bound follows the basis that is allocated, so
 * raising it and calling kfsp_arnoldi with the larger m before a new generator returns -2; kfsp_dgexpv needs the default), "box_store" (1: kfsp_set_matrix_box stores the generator as diagonals), "box_pencil" (-1, default: a matrix-free box whose slowest species is
 * coupled only through its own +-1 entries is multiplied PENCIL by PENCIL - a wavefront owns 128 rows of one plane of that species and
 * walks the planes, so that those entries' sources are the lane's own previous / next elements and everything that depends on the other
 * coordinates is worked out once per pencil; kernel format 7, products bit-identical to format 4; 0: never; 1: also on small boxes; 2: pencils in SLABS, format 8 - a workgroup's
 * wavefronts walk the lines of the second-slowest species in step and exchange their pairs through LDS: less traffic, bit-identical, but
 * measured slower than format 7 - one workgroup barrier per step at 11 wavefronts per CU - and therefore never chosen automatically; "box_slab_waves" = most lines a workgroup takes, default and maximum 12), "box_tile" (the order in which
 * products over a box take their 128-row trips: -1, default: tiled - blocks of 1024 rows below a stride of at most 16 K rows, per
 * block every slower line back to back - when neither the vector nor the windows of its far strides fit the 256 MiB Infinity
 * Cache (22^6: 14 % faster), ascending otherwise; 0 always ascending; 1 always tiled; same bits either way), "box_generic" (1: matrix-free boxes take
 * the run-time interpreted kernel even when the single-factor fast path applies), "state_order" (1: use
 * kfsp_set_state_coords, the default; 0: never), "state_order_min" (smallest generator that is
 * reordered, default 32768), "state_order_products" (products the previous
 * generator must have seen, default 48), "sell_sigma" (rows per window in which the internal order puts the longest
 * rows first, >= 128; default 0: plain lexicographic order), "build_speculate" (1, default: a resident FSP is re-ordered and its
 * generator rebuilt with ONE host synchronisation instead of five - the key layout of the last order and the SELL form of the last
 * generator are assumed, the entry arrays reserved for their bound, everything checked at the end and repeated the slow way when
 * it did not hold (kfsp_build_info); the order itself is carried over - appended keys merged in, dropped ones compacted out -
 * instead of sorted anew; and rows of at most 16 entries are put into FMATVEC's order by ranking them in registers;
 * 0: every number is waited for, rows are insertion-sorted in memory; same generator, bit for bit, either way) ... see DESIGN.md */
int kfsp_set_option(kfsp_ctx *ctx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
