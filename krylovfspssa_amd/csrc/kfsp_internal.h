// Internal declarations shared by the host side (kfsp_api.cpp) and the gfx950
// kernels (kfsp_kernels.hip) of libkfsp_hip.  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace kfsp {

constexpr int kBlock = 256;        // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kChunk = 64;         // SELL chunk height = one wavefront
constexpr int kMaxGrid = 2048;     // 256 CUs x 8 resident 256-thread blocks
constexpr int kMMax = 100;         // KrylovSolver.f90:47
constexpr int kMH = kMMax + 2;     // leading dimension of the device H image

// Generator rows of this rank in SELL-64-1: chunk c holds rows [64c, 64c+64),
// its entries live at off[c] + slot*64 + lane (lane = row within chunk), so a
// wavefront reads 64 consecutive int32 / f64 per slot.  The diagonal is kept
// apart (positive, as DIAG in StateSpace.f90:16) and padded slots carry
// val = 0 with a valid column.
// SELL-64 slot layout: the slots of a chunk (width w) go in PAIRS, the two slots of a pair adjacent per
// lane - slot k of lane l sits at off[c] + (k / 2) * 128 + 2 l + (k & 1) - so that a lane reads the columns
// of two slots as one 8-byte and their values as one 16-byte load (a CU issues vector-memory instructions
// at a fixed rate whatever their width); the last slot of an odd width stands alone, 64 entries.  A row's
// slots stay in FMATVEC's order.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int64_t sell_pos(int64_t chunk_off, int w, int k, int lane)
{
    if ((w & 1) && k == w - 1) return chunk_off + (int64_t)(w >> 1) * 128 + lane;
    return chunk_off + (int64_t)(k >> 1) * 128 + 2 * lane + (k & 1);
}

struct SellDev {
    int64_t nrows;       // local rows
    int64_t nchunks;     // ceil(nrows / 64)
    const int64_t *off;  // [nchunks + 1] entry offsets
    const int32_t *col;  // global 0-based column
    const double *val;
    const double *diag;  // [nchunks * 64]
    // dictionary-coded columns (kernel format 5), null otherwise: see kSellCode* below
    const int32_t *dtab;               // [nchunks][64] column offsets (col - row) that occur in the chunk
    const int32_t *dtlen;              // [nchunks] entries of the table in use; 0 = the chunk keeps plain columns
    const unsigned long long *code;    // code words: word j of lane l of chunk c at codeoff[c] + 64 j + l
    const int64_t *codeoff;            // [nchunks + 1]
};

// SELL-64 with dictionary-coded columns.  In a locality-preserving state order (the internal lexicographic order,
// or a box) the 64 x (width) entries of a chunk use only a handful of distinct column OFFSETS col - row: the
// reaction shifts of the lattice, a few variants of each where the chunk crosses a line end of the state set.
// Instead of 4 bytes of column per entry the chunk stores that handful once (<= 64 offsets, lane j of the wave
// holds entry j) and every entry a 6-bit index into it, ten to a 64-bit word per row: 8 w + 8 ceil(w / 10)
// bytes per row and ~2 for the table, against 12 w.  The kernel turns an index back into the offset with one
// ds_bpermute (a cross-lane read of the table register).  Values, their order in a row (FMATVEC's,
// KrylovSolver.f90:598-604) and the addresses gathered are those of plain SELL: products are bit-identical.  A
// chunk with more than 64 distinct offsets (discovery-ordered states) keeps its plain columns (dtlen = 0).
constexpr int kSellCodeBits = 6, kSellCodePerWord = 10;

// Banded form for state sets whose ordering makes every reaction a constant
// index shift (lattice boxes in lexicographic order): diagonal d holds
// val[d*ld + r] = A(r, r + delta[d]); no column indices at all.  Entries that
// would leave [0, n) are stored as 0 and their x index is clamped.
constexpr int kMaxDiag = 16;
struct DiaDev {
    int nd;
    int32_t delta[kMaxDiag];
    const double *val;   // nd x ld
    int64_t ld;          // >= nchunks * 64
    const double *diag;
    int64_t nchunks;
    int64_t n;           // global number of states (clamp bound)
    // optional: bit d of gmask[c] tells whether diagonal d has any entry in the
    // 128-row group c; empty segments are then read from `zero` (128 zeros that
    // stay in cache) instead of from the stored diagonal
    const uint32_t *gmask;
    const double *zero;
};

// Matrix-free generator of a lexicographic box [0,d_1) x ... x [0,d_ns) (species 1 fastest): no
// stored entries at all.  Propensities are products of one-species factors,
//   a_k(x) = prod_i T_{k,i}[x_{s(k,i)}]   (mass action, Hill functions of one species, ...),
// tabulated on the host with the model's own propensity code; the tables (a few KB) live in LDS and
// the kernel rebuilds every row from the row index: y_r = -(sum_k a_k(x)) x_r + sum_k a_k(x - nu_k) x_{r + delta_k}
// for the sources x - nu_k inside the box (FMATVEC on the FSP = the box, KrylovSolver.f90:577-607, with
// DIAG = the sum of ALL propensities, StateSpace.f90:207-212).  HBM traffic: x once, y once.
constexpr int kBoxMaxS = 8, kBoxMaxR = 16, kBoxMaxDep = 3;
struct BoxDev {
    int32_t ns, nr, ntab, pad;
    int32_t dims[kBoxMaxS];
    double inv_dim[kBoxMaxS];
    int32_t delta[kBoxMaxR];                  // x index of reaction k's source state relative to the row (ascending)
    int32_t dorder[kBoxMaxR];                 // reactions in the model's own order (the order DIAG is summed in)
    int8_t ndep[kBoxMaxR], nmov[kBoxMaxR];
    int8_t dep_s[kBoxMaxR][kBoxMaxDep];       // species a_k depends on
    int8_t dep_nu[kBoxMaxR][kBoxMaxDep];      // stoichiometry of that species (source coordinate = x - nu)
    int32_t dep_off[kBoxMaxR][kBoxMaxDep];    // offset of its factor table in the table image
    int8_t mov_s[kBoxMaxR][kBoxMaxDep];       // species the reaction changes
    int8_t mov_nu[kBoxMaxR][kBoxMaxDep];
    int32_t mov_dim[kBoxMaxR][kBoxMaxDep];    // dims[mov_s]: keeps every index into this struct a compile-time constant
};

// The same generator when every propensity has ONE factor and no reaction changes a species by
// more than 2 (all benchmark boxes): the entries are grouped by the species their factor depends
// on, `per` slots per species (unused slots are never valid), so that in the kernel both the
// species and the slot of an entry are compile-time constants: straight-line code on coordinate
// registers, all gathers of a row in flight together.  The LDS image carries, behind the factor
// tables, one table per species of {sum of the species' propensities at this count, valid bits}
// (BoxDF in the kernel): bit e says that entry e's source coordinate of this species lies inside
// the box; the AND over the species is the set of entries the row has.  Element 0 of the image is
// 0.0 - what an invalid entry reads.  Lives in device memory behind the image; uniform.
constexpr int kBoxFastS = 6, kBoxFastPer = 4;
constexpr int kBoxImageHead = 2;              // doubles in front of the factor tables (the 0.0; keeps 16-byte alignment)
struct BoxFast {
    int32_t ns, per, bias8, pad1;             // species (padded with dimension 1), slots per species, see BoxRegs
    int32_t dims[kBoxFastS];
    double inv_dim[kBoxFastS];
    int32_t koff8[kBoxFastS][kBoxFastPer];    // 8 * (offset of the entry's factor table in the image - nu of its own species)
    int32_t delta8[kBoxFastS][kBoxFastPer];   // 8 * (x index of the source state relative to the row)
    int32_t df8[kBoxFastS];                   // byte offset of the species' {sum, valid} table in the image
};

// A scalar that is the sum of n doubles at p (block partials of the producing
// kernel, or one finished / all-reduced value).  Consumers sum it themselves in
// a fixed order: no atomics, no extra launch, bit-reproducible.
struct Pending {
    const double *p;
    int n;
};

struct SpmvArgs {
    SellDev A;
    DiaDev D;             // used instead of A by the banded kernels
    BoxDev B;             // matrix-free box generator (format 3): the descriptor, by value (uniform kernel argument)
    const double *box_tab;   // ... and its factor tables (device memory, staged to LDS by the kernel)
    const BoxFast *box_fast; // format 4: the single-factor form of the same box (device memory)
    const double *xg;     // gather source, global indexing
    int64_t row0;         // global index of local row 0
    double *y;            // local rows
    Pending sq;           // squared norm of the source column (modes 1,2)
    double *sq_final;     // where block 0 stores the finished squared norm
    double *h_sub;        // where block 0 stores sqrt of it (H(j,j-1)), may be null
    const double *udot;   // modes 1,3: vector of the first dot product
    double *partial;      // [gridDim.x] block partials of the fused reduction
    const double *udot2;  // mode 3: second dot vector
    double *partial2;
    double break_tol;     // <0: no breakdown test
    int *brk_flag;
    // wavefront trips covered by this launch (a trip = one 64-row chunk, or one
    // 128-row group of the banded form); lets a product be split into an interior
    // launch and boundary launches that wait for the halo exchange
    int64_t trip_begin, trip_end;
    // linear trip t stands for trip (t < trip_split ? t : t + trip_jump): one launch
    // can cover the two boundary ranges [0, lo) and [hi, trips)
    int64_t trip_split, trip_jump;
    // optional: the ORDER in which the wavefront trips are taken - linear position t stands for trip trip_order[t].
    // Rows are independent, so any order gives the same y; a tiled order keeps the far neighbours of a box's rows in
    // the XCD's L2 (kfsp_set_trip_order / box tiling).  Null: ascending.  Whole-product launches only.
    const int32_t *trip_order;
};

// Both Gram-Schmidt updates of an IOP(2) column in one pass.  With
//   a = u1.w, b = u2.w (from the product kernel), g = u2.u1 (from the previous
// column's update) the modified Gram-Schmidt coefficients are
//   h1 = a s1,  h2 = (b - h1 s1 g) s2       [ = v2.(w - h1 v1) ]
// and w -= h1 s1 u1 + h2 s2 u2; partials: w.w (next norm) and w.u2 (next g).
struct Ortho2Args {
    int64_t npairs;
    double *w;
    const double *u1;     // u_{j-1}, null for the first column
    const double *u2;     // u_j
    Pending a, b, g;
    const double *sq1, *sq2;
    double *partial_sq, *partial_g;
    double *h1_out, *h2_out;
    double *g_final;      // block 0 stores the finished g (restart needs it)
    const int *brk_flag;
};

struct OrthoArgs {
    int64_t npairs;       // padded local length / 2
    double *w;            // column being orthogonalised (in/out)
    const double *ui;     // basis vector subtracted (unnormalised)
    Pending dot;          // sum = u_i . w
    const double *sq_i;   // finished squared norm of u_i
    const double *unext;  // next dot vector, or null -> accumulate w.w
    double *partial;
    double *h_out;        // H(i,j)
    const int *brk_flag;
};

struct CombineArgs {
    int64_t npairs;
    int mx;
    double beta;
    const double *V;      // unnormalised basis, column stride ldv
    int64_t ldv;
    const double *sq;     // finished squared norms, sq[j] for column j (1-based)
    const double *y;      // device copy of the coefficient vector
    double *w;
    double *partial;
};

// A whole IOP(2) Arnoldi pass in ONE launch of ONE workgroup, for state spaces
// small enough (<= kSmallRows rows, source vector resident in LDS) that kernel boundaries, not bytes, set the
// time: BASELINE config 1 (toggle, ~10^3 states) spends ~10 us per column in two
// launches of which < 1 us is work.
constexpr int kSmallRows = 4096;     // 4 rows per lane in registers, source column in 32 KB of LDS
struct SmallArnoldiArgs {
    SellDev A;
    DiaDev D;
    double *V;            // column 0 of the basis (halo margin already applied)
    int64_t ldv;
    int64_t nact;         // rows each column carries (multiple of 64)
    int m, jold;
    double *sq;           // squared norms, index = column (1-based)
    double *gfin;         // u_j . u_{j-1}, index = j
    double *Hd;           // kMH x kMH image + avnorm^2, avnorm behind it
    double break_tol;
    int *brk_flag;
    int64_t slots;        // stored SELL slots (the generator goes to LDS when they fit beside the source column)
};
// lds_limit: bytes of LDS one workgroup may use; returns a hipError_t
int launch_arnoldi_small(const SmallArnoldiArgs &a, bool dia, int64_t lds_limit, hipStream_t s);

// kernel launchers (kfsp_kernels.hip)
// fmt: 0 SELL-64, 1 banded, 2 banded with group masks, 3 matrix-free box (lds_bytes = size of the factor tables),
// 4 matrix-free box, single-factor fast path, 5 SELL-64 with dictionary-coded columns
void launch_spmv(int mode, int grid, const SpmvArgs &a, bool nontemporal, int fmt, hipStream_t s, size_t lds_bytes = 0);
// format 6: the single-factor matrix-free product with the near part of x staged in LDS (reach rows on either side of a
// workgroup's 512; lds_bytes = table image + two windows)
void launch_spmv_boxlds(int mode, int grid, const SpmvArgs &a, hipStream_t s, size_t lds_bytes, int reach);
// format 8: a workgroup's wavefronts walk the pencils of consecutive lines of the second-slowest species in step (k_spmv_slab)
void launch_spmv_slab(int mode, int grid, int waves, const SpmvArgs &a, hipStream_t s, size_t lds_bytes, int64_t plane_rows, int planes,
                      int64_t line_rows, int lines, int groups, int64_t lo_trips, bool simple);
// format 7: a wavefront walks the planes of the slowest species (kfsp_kernels.hip, k_spmv_pencil)
void launch_spmv_pencil(int mode, int grid, const SpmvArgs &a, hipStream_t s, size_t lds_bytes, int64_t plane_rows, int planes,
                        int64_t base_trips, const int32_t *order, bool simple);
void launch_ortho2(int grid, const Ortho2Args &a, hipStream_t s);
// H image and breakdown flag of an Arnoldi pass cleared by ONE launch (two memsets before: 6 400 of them in the Goutsias run)
void launch_pass_reset(double *H, int count, int *flag, hipStream_t s);
void launch_ortho(int grid, const OrthoArgs &a, hipStream_t s);
void launch_combine(int grid, const CombineArgs &a, hipStream_t s);
// u1 = w, partial = sum w^2
void launch_copy_nrm2(int grid, int64_t npairs, const double *w, double *u1, double *partial, hipStream_t s);
// partial = sum |w| or sum w^2
void launch_reduce(int grid, int64_t npairs, const double *w, int squared, double *partial, hipStream_t s);
// out[0] = sum(p) ; out_sqrt (may be null) = sqrt of it
void launch_finalize(Pending p, double *out, double *out_sqrt, hipStream_t s);
// w = scale * u
void launch_scale_copy(int grid, int64_t npairs, const double *u, const double *sq_u, double beta, double *w, hipStream_t s);

// counter calibration: read nbytes with elem_bytes per lane
void launch_stream_read(int grid, int elem_bytes, int64_t nbytes, const void *p, double *sink, hipStream_t s);

}  // namespace kfsp
