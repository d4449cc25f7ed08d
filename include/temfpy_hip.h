/* temfpy_hip.h -- C ABI of the MI355X (gfx950) backend for TeMFpy's Slater -> MPS sweep.
 *
 * The reference (/root/reference/src/temfpy) is pure Python and has no FFI seam on
 * this path; its only native calls are LAPACK through NumPy.  The seam is therefore
 * introduced *below* the reference's per-cut / per-site numerics, and every entry
 * point names the reference code it replaces (file:line in /root/reference/src/temfpy).
 * INTEGRATION.md shows the ctypes binding a TeMFpy maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Device entry
 *    points only enqueue work; they never synchronise and never allocate.
 *  - dtype: TMF_F64 (real double) or TMF_C128 (interleaved re,im doubles).
 *  - All matrices are column-major with an explicit leading dimension (in elements).
 *  - Batched entry points take a device array of descriptors (one per problem) so that
 *    problems of different sizes run in one launch.  Descriptor pointers are device
 *    addresses stored as uint64_t.
 *  - Return value: 0 on success, negative TMF_E_* otherwise; tmf_last_error() gives text.
 */
#ifndef TEMFPY_HIP_H
#define TEMFPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMF_F64 0
#define TMF_C128 1

#define TMF_OK 0
#define TMF_E_ARG (-1)     /* bad argument (maps to ValueError / AssertionError) */
#define TMF_E_HIP (-2)     /* HIP runtime error (maps to RuntimeError)           */
#define TMF_E_LIMIT (-3)   /* problem exceeds a compiled-in limit                 */

const char* tmf_last_error(void);
int tmf_version(void);
/* number of visible HIP devices, or a negative TMF_E_HIP */
int tmf_device_count(void);

/* ------------------------------------------------------------------------------------
 * Device: batched dense kernels (the arithmetic behind numpy.linalg.eigh / @ / det / inv
 * at slater.py:347, :1071, :1079-1088, :869)
 * ---------------------------------------------------------------------------------- */

/* C = alpha * op(A) * B + beta * C ; op(A) = A (opA=0) or A^H (opA=1).
 * fp64 MFMA (v_mfma_f64_16x16x4_f64).  Replaces the `@` products of slater.py:1071,
 * :1080, :1087 and the GEMM-shaped part of the block diagonalisation (slater.py:347). */
typedef struct {
  uint64_t A, B, C;          /* device addresses                                    */
  int32_t M, N, K;           /* C is M x N, contraction length K                    */
  int32_t lda, ldb, ldc;
} tmf_gemm_desc;             /* 48 bytes */

int tmf_gemm_batched(int dtype, int opA, double alpha, double beta, const tmf_gemm_desc* d_desc,
                     const int32_t* d_tiles /* [ntiles][4]: problem, tile_m, tile_n, 0 */, int ntiles,
                     int tile_n /* 64 or 16 */, void* stream);
/* A/B switch (tests): complex products by four real MFMAs per multiply-add (componentwise error bound) instead of the
 * default three (3M scheme, normwise bound); process-wide, applies to the launches that follow. */
void tmf_gemm_set_4m(int on);

/* Tall-skinny form of the above for op(A) = A^H, N <= 16 and a long contraction (the coefficient
 * products Q^H P of the blocked Gram-Schmidt): 16 x 16 output tiles, 64 rows of A and B per step.
 * d_tiles: [ntiles][4] = problem, tile_m (units of 16 rows of C), 0, 0. */
int tmf_gemm_tall_batched(int dtype, double alpha, double beta, const tmf_gemm_desc* d_desc, const int32_t* d_tiles,
                          int ntiles, void* stream);

/* Orthonormalise the columns [c0, c0+w) of A (n x *) in place against themselves
 * (classical Gram-Schmidt, twice), one workgroup per problem, panel staged in LDS.
 * With `norms` given, a column whose residual is below 1e-14 of its original norm (pure
 * rounding noise: the column is numerically dependent on the previous ones) is set to ZERO
 * instead of being normalised; downstream such a column carries singular value 0.
 * Building block of the blocked QR that orthonormalises orbital slabs (the
 * orthonormal eigenvector blocks of slater.py:347). */
typedef struct {
  uint64_t A;                /* device address of the first panel column            */
  uint64_t norms;            /* 0, or device address of w doubles: the norms the panel's columns
                                had BEFORE any projection (tmf_column_norms_batched)  */
  int32_t n, w, lda, pad;
} tmf_panel_desc;            /* 32 bytes */

int tmf_orth_panel_batched(int dtype, const tmf_panel_desc* d_desc, int nprob, int max_n, int max_w,
                           void* stream);

/* Blocked Gram-Schmidt QR of many slabs in one call (the loop over panels runs in C++; the GEMM and
 * panel descriptors of every step are built on the device from these records): orthonormalises the
 * columns [c_begin, c_end) of each slab against all columns before them, `passes` projections per
 * panel (2: well-conditioned slabs, 3: numerically rank-deficient range-finder slabs).
 * scratch: c_end x 16 elements per slab; norms: 0 or the raw column norms of [c_begin, c_end)
 * (tmf_column_norms_batched), see tmf_orth_panel_batched.  h_desc is a host copy of d_desc.
 * d_work: device scratch of at least tmf_bcgs_work_bytes(h_desc, nprob) bytes. */
typedef struct {
  uint64_t base, scratch, norms;
  int32_t rows, ld, c_begin, c_end;
} tmf_bcgs_desc;             /* 40 bytes */
int64_t tmf_bcgs_work_bytes(const tmf_bcgs_desc* h_desc, int nprob);
/* One launch for the inner part of tmf_bcgs_batched's Cholesky-QR mode: the columns [c_begin + t, c_begin + t + 64) of
 * every slab (already projected against all earlier columns) are orthonormalised in themselves - 16-column panels, each
 * projected against the earlier panels of the block and put through Cholesky-QR twice - by one 1024-thread workgroup per
 * slab with the panel in registers (csrc/block_orth.hip).  rows <= max_rows <= 960.  Part of the replacement of
 * numpy.linalg.eigh's orthonormal eigenvector blocks (slater.py:347). */
int tmf_block_orth_batched(int dtype, const tmf_bcgs_desc* d_desc, int nprob, int t, int max_rows, void* stream);
/* diagnostics (TMF_BORTH_STAMPS=1): cycles per phase of the first wavefront [0..7], workgroups, panels, rows [8..10], helper wavefront: waiting, factorising [11, 12]; clears */
int tmf_block_orth_stamps(uint64_t* out16);
int tmf_bcgs_batched(int dtype, const tmf_bcgs_desc* d_desc, const tmf_bcgs_desc* h_desc, int nprob, int passes,
                     int flags /* bit 0: Cholesky-QR (twice) inside the panels instead of the LDS Gram-Schmidt panel
                                  kernel - for well-conditioned slabs (filled-orbital bases) only;
                                  bit 1: outer blocks of 64 instead of 16 columns (scratch: c_end x 64 per slab) */,
                     void* d_work, int64_t work_bytes, void* stream);

/* One-sided (Hestenes) Jacobi on a p x p matrix X held in LDS: X V = U diag(s).
 * Outputs s (descending) and V (columns permuted accordingly); optionally U.
 * p <= 64.  Used for the small SVD / Hermitian eigenproblems that replace
 * numpy.linalg.eigh (slater.py:347) and numpy.linalg.svd (utils.py:90). */
typedef struct {
  uint64_t X;                /* in: p x p, ldx                                      */
  uint64_t V;                /* out: p x p, ldv                                     */
  uint64_t U;                /* out (may be 0): p x p normalised columns of X V      */
  uint64_t s;                /* out: p doubles, descending                          */
  uint64_t count;            /* out (may be 0): int32, #columns with s^2 >= thresh2 */
  double thresh2;            /* > 0: columns of V (and U) with s^2 < thresh2 are zeroed */
  int32_t p, ldx, ldv, ldu;
} tmf_jacobi_desc;           /* 64 bytes */

int tmf_jacobi_batched(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p, int32_t* d_sweeps,
                       void* stream);

/* Same decomposition, left singular vectors only: desc.V is ignored, desc.U is required.  No
 * rotation accumulator -> half the LDS (two workgroups per CU) and 40 % fewer flops.  Fed with
 * the conjugate transpose of the triangular factor (columns graded by the singular values),
 * which one-sided Jacobi diagonalises in far fewer sweeps than the factor itself. */
int tmf_svd_left_batched(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p, int32_t* d_sweeps,
                         void* stream);

/* Block variant of the two entry points above for 64 < p <= 512 (a cut whose entanglement rank
 * exceeds the 64-column range finder): X and the rotation accumulator stay in global memory, pairs of
 * column blocks are staged in LDS.  X is DESTROYED.  with_v = 0: desc.U receives the sorted,
 * normalised left singular vectors (desc.V unused).  with_v = 1: desc.V is a p x p workspace and
 * desc.U receives the sorted right singular vectors / eigenvectors. */
int tmf_jacobi_block_batched(int dtype, int with_v, const tmf_jacobi_desc* d_desc, int nprob, int max_p,
                             int32_t* d_sweeps, void* stream);

/* Compacting variant for rank-deficient factors, p <= 512, thresh2 > 0 required: columns of X with squared norm
 * below thresh2 * 1e-4 / p take no part (they cannot lift a singular value over the threshold: all of them
 * together move the spectrum by less than 1e-2 sqrt(thresh2)); the active
 * columns of X and of the accumulated rotations are held p x nact in LDS when they fit, else in global memory
 * (desc.X in place, desc.V = p x p workspace).  desc.U receives the right singular vectors sorted by
 * descending singular value, zero columns for everything below the threshold; desc.s, desc.count as above.
 * desc.V == 0: no accumulator, desc.U receives the sorted, normalised LEFT singular vectors instead (for the
 * conjugate transpose of a QR-preconditioned factor these are the wanted right vectors, at half the work).
 * The `npc.svd` of the leftward sweep of TeNPy's MPS.canonical_form_finite (gutzwiller.py:266 / :471). */
int tmf_jacobi_compact_batched(int dtype, const tmf_jacobi_desc* d_desc, int nprob, int max_p, int32_t* d_sweeps,
                               void* stream);

/* Blocked LU with partial pivoting restricted to the leading k x k "always" block of
 * W (mb x mk).  Returns det(W[:k,:k]) and leaves the Schur
 * complement W[k:,k:] - W[k:,:k] W[:k,:k]^-1 W[:k,k:] in S.  Replaces det/inv and the two
 * products of slater.py:1077-1090. */
typedef struct {
  uint64_t W;                /* in/out workspace mb x mk, ldw                       */
  uint64_t S;                /* out: (mb-k) x (mk-k), lds                           */
  uint64_t det;              /* out: one element (re[, im])                          */
  int32_t mb, mk, k, ldw, lds, pad;
} tmf_schur_desc;            /* 48 bytes */

/* max_mb = largest mb of the batch (sizes the LDS panel). S may be 0: the Schur complement is then
 * read in place at W + k + k*ldw. */
int tmf_lu_schur_batched(int dtype, const tmf_schur_desc* d_desc, int nprob, int max_mb, void* stream);

/* The same factorisation as tmf_lu_schur_batched, blocked over several launches so that the rank-64 trailing
 * update of all sites is one batched MFMA GEMM (tmf_gemm_batched with alpha = -1, beta = 1) instead of VALU code
 * inside a one-workgroup-per-site kernel.  For j0 = 0, 64, 128, ... < max k:
 *   tmf_lu_block_batched   pivoted LU (pivots among the always rows) of the columns [j0, min(k, j0 + 64)) of every
 *                          site with k > j0, rows j0 .. mb-1; L stays in W, the pivot rows go to `piv`, det(A) is
 *                          accumulated in `det` (j0 = 0 initialises it; sites with k = 0 get det = 1)
 *   tmf_lu_trsm_batched    the block's row interchanges on the trailing columns, then U12 = L11^-1 A12 into `T`
 *   tmf_gemm_batched       A22 -= L21 U12   (A = W + cend + j0 ldw, B = T, C = W + cend + cend ldw, K = cend - j0)
 * Afterwards det holds det(W[:k,:k]) and W[k:,k:] the Schur complement (slater.py:1077-1090). */
typedef struct {
  uint64_t W, det;           /* mb x mk workspace (ldw); one element                                 */
  uint64_t piv;              /* int32[k]: absolute pivot row of every always column                  */
  uint64_t T;                /* scratch 64 x mk elements (leading dimension 64)                      */
  int32_t mb, mk, k, ldw;
} tmf_lublock_desc;          /* 48 bytes */
int tmf_lu_block_batched(int dtype, const tmf_lublock_desc* d_desc, int nprob, int j0, int wb, int max_mb, void* stream);
int tmf_lu_trsm_batched(int dtype, const tmf_lublock_desc* d_desc, int nprob, int j0, int wb, int max_cols, void* stream);

/* The same Schur complement by block Gaussian elimination with the row search confined to the 64 x 64 diagonal block D of
 * every outer step; everything outside the block is GEMM work with the explicit inverse of D:
 *   tmf_diag_inverse_batched   D^-1 of W[j0:cend, j0:cend] (Gauss-Jordan with row pivoting, in registers) into `inv` (dense,
 *                              leading dimension 64), det(A) accumulated in `det`; d_stats[2 p] = smallest |pivot|^2 of
 *                              matrix p so far, d_stats[2 p + 1] = largest |entry|^2 of D^-1 over the blocks with
 *                              always-rows below them (reset at step 0): the caller repeats the LU with
 *                              tmf_lu_block_batched when it exceeds its threshold
 *   tmf_gemm_batched           X = D^-1 W[j0:cend, cend:]
 *   tmf_gemm_batched           W[cend:, cend:] -= W[cend:, j0:cend] X
 * Outer step s works on columns [j0, cend) of the k always-columns, counted from the end: block 0 = [0, (k-1) % 64 + 1),
 * block s = the next 64.  Valid because the always-occupied orbitals of neighbouring cuts are nearly aligned (same random
 * block behind every filled basis, tmf_site_prepare pairs them): their overlap matrix is close to block diagonal and
 * ||D^-1 A12|| stays O(1) (see lu_schur.hip). */
typedef struct {
  uint64_t W, det;           /* mb x mk workspace (ldw); one element                                 */
  uint64_t inv;              /* out: D^-1, 64 x 64 elements (leading dimension 64)                   */
  int32_t mb, mk, k, ldw;
} tmf_diaginv_desc;          /* 40 bytes */
int tmf_diag_inverse_batched(int dtype, const tmf_diaginv_desc* d_desc, int nprob, int step, void* d_stats, void* stream);
/* *d_flag = 1 when any of the nprob matrices has a block-inverse entry above `cap` (or a NaN) or when `force` is set, else 0;
 * summary[0..2] = smallest |pivot|, largest |D^-1 entry|, flag (device memory or page-locked host memory). */
int tmf_diag_inverse_verdict(const void* d_stats, int nprob, double cap, int force, int32_t* d_flag, double* summary, void* stream);
/* Conditional-launch scope of the calling thread: while d_flag != NULL, the kernels launched by tmf_gather_signed_batched,
 * tmf_lu_block_batched, tmf_lu_trsm_batched and tmf_gemm_batched return at once on the device when *d_flag == 0 (read when
 * the kernel runs, not when it is enqueued).  The fallback of the block-local elimination is enqueued this way, so the
 * host never waits for the verdict.  tmf_launch_condition(NULL) ends the scope. */
void tmf_launch_condition(const int32_t* d_flag);

/* The hot kernel: batched gathered determinants (slater.py:828-869, `_tensor_block`,
 * 90 % of the reference's wall time).  For every tile, for every pair (a, b) of a bra
 * row and a ket row of one charge sector, out[a, b] = scale * det(S[rows(a)][:, cols(b)]).
 * S (<= 4096 elements) is staged in LDS once per workgroup. */
typedef struct {
  uint64_t S;                /* sb x sk, lds                                        */
  uint64_t scale;            /* device address of det_always (2 doubles re,im)      */
  uint64_t bra_idx;          /* uint8 [nsb][n]: row positions, ascending            */
  uint64_t ket_idx;          /* uint8 [nsk][n]: column positions, ascending         */
  uint64_t out;              /* nsb x nsk row-major block                           */
  int32_t sb, sk, lds, n;    /* n = order of every determinant of the sector        */
  int32_t nsb, nsk, a0, a1;  /* this tile handles bra rows [a0, a1), all ket rows   */
} tmf_det_desc;              /* 72 bytes */

/* `order` = the exact order n of every minor of the launch for 0 <= n <= 32 (one straight-line
 * kernel per order: 8 / 16 / 32 lanes own one determinant), or 64 = generic LDS-resident kernel
 * for 33 <= n <= 64 (n read from the descriptor).  lds_bytes = dynamic LDS per workgroup:
 * align16(sb*sk*elem) + align16(nsk*n) + align16((a1-a0)*n)
 *   + 4 * ((n|1)*sk + gpw*(n+1)) * elem      (gpw = 8, 4, 2 groups per wavefront for n <= 8, 16, 32)
 * or + n*n*elem for order 64; max over the tiles of the launch. */
int tmf_det_gather_batched(int dtype, int order, const tmf_det_desc* d_desc, int ntiles, int lds_bytes,
                           void* stream);

/* Same result as tmf_det_gather_batched through one shared reduction per bra row-set: a
 * Gauss-Jordan elimination of S[rows(a), :] with full pivoting (once per a, lane = column) turns
 * every minor into a determinant of order d = |cols(b) \ pivot columns| (d <= 2 for ~96 % of
 * the minors of a sweep).  Requires sk <= 64 and 1 <= order <= 32.  lds_bytes per workgroup:
 * align16(sb*sk*elem) + align16(nsk*n) + align16(nsk*8) + align16((a1-a0)*n)
 *   + 4 * (((n|1)*sk + 264) * elem + 64). */
int tmf_det_reduced_batched(int dtype, int order, const tmf_det_desc* d_desc, int ntiles, int lds_bytes,
                            void* stream);

/* Same result through ONE pivoted exchange (principal pivot transform) of the sector matrix per
 * workgroup: every minor is +- det(M*) times a determinant of order d = number of orbitals in which
 * the pair (a, b) differs from the pivot configuration (0..4 for almost all pairs), one lane per pair.
 * Not specialised on the order: tiles of all orders go into one launch.  Requires sb, sk <= 64 and
 * n <= 32.  lds_bytes per workgroup: align16(sb*sk*elem) + (nsk + (a1-a0)) * 8 (+ pad to 16)
 *   + 4 * (max(264, n*n) * elem + 288). */
int tmf_det_ppt_batched(int dtype, const tmf_det_desc* d_desc, int ntiles, int lds_bytes, void* stream);
/* mask_bits = 32: every tile of the launch has sb, sk <= 32 (the occupation masks and the sign bookkeeping then run on 32-bit
 * words and the ket-only part of the sign is computed once per ket set: 2.72 instead of 3.26 ms at the benchmark size); 64: as tmf_det_ppt_batched. */
int tmf_det_ppt_batched_w(int dtype, const tmf_det_desc* d_desc, int ntiles, int lds_bytes, int mask_bits, void* stream);
/* Diagnostics (process started with TMF_PPT_STAMPS=1; tools/ppt_probe.py): shader cycles of the four phases of a workgroup
 * (load, exchange, tables, pairs) per wavefront slot [4 w + phase], summed over all launches since the last call; [16]
 * workgroups, [17] pairs, [18] sum of n.  Synchronises the device and clears the counters. */
int tmf_det_ppt_stamps(uint64_t* out32);

/* Batched gathered Pfaffians (pfaffian.py:1429-1479 `_tensor_block` + :1413-1426 `_many_pfaffian`,
 * i.e. pfapack.ctypes.pfaffian in a Python loop).  For every pair (bra row a, ket row b):
 * out[a, b] = scale * Pf(N[idx, idx]),  idx = [ket positions of b (n2), bra positions of a (n1)].
 * `order` = n1 + n2 of every tile of the launch (even, <= 32).  lds_bytes per workgroup:
 * align16(nn*nn*elem) + align16(nsk*n2) + align16((a1-a0)*n1) + (256/G) * 2*order*elem,
 * G = 8, 16, 32 for order <= 8, 16, 32. */
typedef struct {
  uint64_t N;                /* nn x nn skew-symmetric Pfaffian matrix, ldn          */
  uint64_t scale;            /* device address of the Onishi norm factor (one element) */
  uint64_t bra_idx;          /* uint8 [nsb][n1]                                      */
  uint64_t ket_idx;          /* uint8 [nsk][n2]                                      */
  uint64_t out;              /* nsb x nsk row-major block                            */
  int32_t nn, ldn, n1, n2;
  int32_t nsb, nsk, a0, a1;
} tmf_pf_desc;               /* 72 bytes */
int tmf_pf_gather_batched(int dtype, int order, const tmf_pf_desc* d_desc, int ntiles, int lds_bytes,
                          void* stream);

/* ---- assembly kernels of the Pfaffian path (complex128 only) ------------------------------- */
/* dst[:, j] = M2C( [conj] src[:, col_src[j]] ): Bogoliubov matrix [a | a^dag] of a cut side in the
 * complex-fermion basis from mode blocks in the Majorana basis (pfaffian.py:880-895, :104-124). */
typedef struct {
  uint64_t src, dst;         /* n2 x * (lds_) and n2 x n2 (ldd), column-major        */
  uint64_t col_src;          /* int32[n2] source column of every result column       */
  uint64_t col_conj;         /* int8[n2]  1: take the complex conjugate (Majorana basis) */
  int32_t n2, lds_, ldd, pad;
} tmf_nambu_asm_desc;        /* 48 bytes */
int tmf_nambu_assemble_batched(const tmf_nambu_asm_desc* d_desc, int nprob, void* stream);

/* W = [[Vr[L:,L:], Q],[P, 0]], Q = [I[:,idx1] | Vr[L:,idx2]], P = [Vr[idx1,L:] ; I[idx2,:]]
 * (the operand of tmf_lu_schur_batched that yields det(U^*) and the AA, BA, BB blocks of
 * pfaffian.py:1384-1391 in one pass). */
typedef struct {
  uint64_t Vr, W;            /* 2L x 2L (ldv) ; (L+na+nb)^2 (ldw)                     */
  uint64_t idx1, idx2;       /* int32[na], int32[nb]                                 */
  int32_t L, na, nb, ldv, ldw, pad;
} tmf_nambu_w_desc;          /* 56 bytes */
int tmf_nambu_w_batched(const tmf_nambu_w_desc* d_desc, int nprob, void* stream);

/* N = [[BB, BA],[-BA^T, AA]] (antisymmetrised) from S = -P (U^*)^-1 Q  (pfaffian.py:1394-1400) */
typedef struct {
  uint64_t S, N;             /* (na+nb)^2 each, lds_ / ldn                           */
  int32_t na, nb, lds_, ldn;
} tmf_pf_matrix_desc;        /* 32 bytes */
int tmf_pf_matrix_batched(const tmf_pf_matrix_desc* d_desc, int nprob, void* stream);
/* norm[i] = sqrt(|det[i]|): the Onishi norm sqrt(prod sv(U)) of pfaffian.py:1352-1359 from det(U^*) (complex128 elements) */
int tmf_onishi_norms(const void* d_det, void* d_norm, int n, void* stream);

/* ---- result path: page-locked host memory ----------------------------------------- */
/* The reference's tensors are NumPy arrays in host memory (slater.py:1137-1141 fills them in
 * place); here they leave the GPU by asynchronous copies into page-locked memory.
 * tmf_host_register page-locks a caller-owned range (e.g. a POSIX shared-memory segment that
 * the rank assembling the MPS maps too: every GPU of a node then writes its site shard through
 * its own PCIe link, no collective - slater.py:1301-1346, independent sites).
 * tmf_memcpy_async: to_host = 1 device -> host, 0 host -> device; only enqueues.          */
/* bytes (a multiple of 4) from device memory into page-locked host memory by a kernel instead of the copy engine: small
 * results needed by the host in the middle of a conversion do not wait behind a tensor download in flight */
int tmf_export_words(void* host_mapped_dst, const void* d_src, int64_t bytes, void* stream);
int tmf_host_register(void* ptr, int64_t bytes);
int tmf_host_unregister(void* ptr);
int tmf_memcpy_async(void* dst, const void* src, int64_t bytes, int to_host, void* stream);

/* ---- small device utilities ------------------------------------------------------ */
/* out (n x n col-major) = transpose of the row-major host layout already on device   */
int tmf_transpose(int dtype, const void* d_in, void* d_out, int n, void* stream);
/* fill with a counter-based standard-normal stream (deterministic in seed, index)    */
int tmf_fill_normal(int dtype, void* d_out, int64_t count, uint64_t seed, void* stream);
/* column norms / scaling, W assembly: generic "gather with signs" kernel              */
typedef struct {
  uint64_t src;              /* col-major, lds_                                     */
  uint64_t dst;              /* col-major, ldd                                      */
  uint64_t row_sel;          /* int32[rows]: source row, or -(1+j) -> row j of `phys` */
  uint64_t col_sel;          /* int32[cols]: source col                              */
  uint64_t row_sign;         /* int8[rows]                                          */
  uint64_t col_sign;         /* int8[cols]                                          */
  uint64_t phys;             /* col-major base of the physical rows: element j + col_sel[c] * ldp */
  int32_t rows, cols, lds_, ldd, ldp, pad;
} tmf_gather_desc;           /* 80 bytes */
int tmf_gather_signed_batched(int dtype, const tmf_gather_desc* d_desc, int nprob, void* stream);

/* Strided block copies, optionally transposing: dst = src (flags 0), dst = src^T (flags 1) or
 * dst = src^H (flags 3); rows x cols is the shape of the SOURCE block, both column-major.  Regroups the
 * charge blocks of MPS tensors between the (p vL) x vR and vL x (p vR) matrix forms of the
 * canonicalisation sweeps behind gutzwiller.py:266 / :471 (TeNPy `combine_legs` / `split_legs`).
 * max_tiles: largest number of 32 x 32 tiles of one block (sizes the grid; more tiles are looped). */
typedef struct {
  uint64_t src, dst;
  int32_t rows, cols, lds_, ldd, flags, pad;
} tmf_copy_desc;             /* 40 bytes */
int tmf_copy_blocks_batched(int dtype, const tmf_copy_desc* d_desc, int nprob, int max_tiles, void* stream);

/* Householder QR of many matrices, one workgroup each: A (m x n, column-major) is replaced by the thin
 * orthonormal factor Q and R (n x n upper triangular; flags & 1: its conjugate transpose R^H instead) is
 * written to `R` (may be 0).  flags & 2: only R is wanted, Q is not formed (A is left holding the reflectors).  For m < n, Q is m x m followed by zero columns and R has zero rows beyond m,
 * so shapes stay fixed.  Orthogonal to machine precision for any rank (no rank decision; a column whose squared length
 * underflows counts as zero): the `npc.qr` / first half of `npc.svd` of TeNPy's MPS.canonical_form_finite behind
 * gutzwiller.py:266 / :471, and the orthonormal bases of range finders wider than 64 columns inside the replacement of
 * numpy.linalg.eigh (slater.py:347) for cuts with more than 64 entangled orbitals. */
typedef struct {
  uint64_t A, R;
  int32_t m, n, lda, ldr, flags, pad;
} tmf_qr_desc;               /* 40 bytes */
int tmf_house_qr_batched(int dtype, const tmf_qr_desc* d_desc, int nprob, int max_m, int max_n, void* stream);

/* Householder QR of tall slabs with the working panel in LDS (n x c, c <= 64, n <= 4096): A is replaced by the
 * thin orthonormal factor Q (flags & 2: Q stays in the scratch, A is left holding the reflectors; flags & 4: no Q at
 * all, only R), R or R^H (flags & 1) goes to `R` (may be 0); `Q` is a caller-provided scratch of n x c elements
 * (leading dimension ldq; unused with flags & 4).  flags & 8: Q later - A is left holding the reflectors and `Q` points to
 * a buffer of c elements that receives their scalars; tmf_house_form_q_batched then turns A into Q.  One workgroup per
 * slab.  The two range-finder QRs of every cut (orthonormal bases inside the replacement of
 * numpy.linalg.eigh, slater.py:347): orthogonal for any numerical rank, one launch instead of ~60. */
typedef struct {
  uint64_t A, Q, R;
  int32_t n, c, lda, ldq, ldr, flags;
} tmf_slab_desc;             /* 48 bytes */
int tmf_house_slab_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream);
/* Thin Q, in place, of slabs factored with flags & 8 (same descriptors: A = the reflectors, Q = their scalars; R, ldq, ldr
 * and flags are not read).  For chains of dependent factorisations in which only R feeds the next link (the canonicalisation
 * sweeps, gutzwiller.py:266 / :471): every Q of the chain in one launch afterwards. */
int tmf_house_form_q_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream);
/* The same, with every column of a block in registers when all blocks of the launch are real and at most 256 x 128
 * (else the panel kernel): which kernel runs then depends on the launch, so not for callers that need a slab's result to be
 * independent of its batch (the sharded sweep).  Used by the canonicalisation sweeps of gutzwiller.py:266 / :471. */
int tmf_house_qr_regs_batched(int dtype, const tmf_slab_desc* d_desc, int nprob, int max_n, int max_c, void* stream);
/* diagnostics (TMF_SLAB_STAMPS=1): cycles per phase of one wavefront [0..7], workgroups, rows [8, 9]; clears */
int tmf_house_slab_stamps(uint64_t* out16);

/* Products of nested blocks of one D x D matrix C (column-major) with a shared block Omega whose
 * rows are indexed by the GLOBAL orbital index: for every cut position x with dest[x] != 0
 *
 *     out_x[lr, c] = sum_{j in J(x)} C[r, j] * Omega[j, c],      c < ncol[x],
 *     J(x) = { j < x } (suffix = 0) or { j >= x } (suffix = 1),
 *     rows r < x, lr = r (rows_ge = 0)   or   rows r >= x, lr = r - x (rows_ge = 1),
 *
 * written to the column-major slab dest[x] with leading dimension ld[x].  One running sum over j
 * serves all cuts (O(D^2 c) flops); it replaces the per-cut products A_x Omega / F_x Omega in the
 * randomised replacement of numpy.linalg.eigh (slater.py:347):
 *     A_x = C[:x,:x]: suffix 0, rows_ge 0        A_x = C[x:,x:]: suffix 1, rows_ge 1
 *     F_x = C[:x,x:]: suffix 1, rows_ge 0        F_x = C[x:,:x]: suffix 0, rows_ge 1          */
typedef struct {
  uint64_t C, Omega;         /* device addresses, leading dimensions ldc / ldo        */
  uint64_t dest;             /* uint64[D + 1]: slab of cut x, 0 = no cut there        */
  uint64_t ncol;             /* int32[D + 1]                                          */
  uint64_t ld;               /* int32[D + 1]                                          */
  int32_t D, ldc, ldo, suffix, rows_ge, x_lo, x_hi, maxc;   /* x_lo..x_hi: cuts present; maxc = max ncol */
} tmf_nested_desc;           /* 72 bytes */
int tmf_nested_products_batched(int dtype, const tmf_nested_desc* d_desc, int ndesc, int D, int maxc, void* stream);

/* Self-check of the Schmidt decomposition (testing.check_schmidt_decomposition, testing.py:131-177):
 * largest absolute deviation between a block T of the correlation matrix and its reconstruction,
 * folded into *out (double, must be zeroed by the caller) with an atomic max; NaN is reported as +inf.
 *   mode 0: | T[r,c] - sum_{j<q} X[r,j] w[j] conj(Y[c,j']) |,  j' = j, or q-1-j when y_reverse
 *           (testing.py:158-159 "vL does not diagonalise C_LL", :170-171, :176-177 "do not SVD C_LR")
 *   mode 1: | T[r,c] - sum_{i<inner} conj(X[i,r]) Y[i,c] |,  T = identity when T == 0
 *           (testing.py:155-156 "vL is not unitary", evaluated on the kept columns as V^H V = 1)
 * `tiles`: int32[ntiles][3] = (problem, tile_row, tile_col) over 64 x 64 tiles of the rows x cols output. */
typedef struct {
  uint64_t T, X, Y, w, out;  /* w: q doubles (0 = all ones); out: one double per problem           */
  int32_t rows, cols, q, inner, ldt, ldx, ldy, mode, y_reverse, pad;
} tmf_recon_desc;            /* 80 bytes */
int tmf_recon_error_batched(int dtype, const tmf_recon_desc* d_desc, const int32_t* d_tiles, int ntiles, void* stream);

/* out[j] = 2-norm of column j of src (n x c) */
typedef struct {
  uint64_t src, out;
  int32_t n, c, lds_, pad;
} tmf_norms_desc;            /* 32 bytes */
int tmf_column_norms_batched(int dtype, const tmf_norms_desc* d_desc, int nprob, void* stream);

/* normalise each column of A (n x c) by its 2-norm; optionally reverse column order
 * and flip the sign of odd columns (slater.py:410) while copying into dst;
 * reverse & 2: also take the complex conjugate;  reverse & 4: keep the real part only
 * (before normalising: the real basis of the eigenvalue-1/2 modes, pfaffian.py:807-816)      */
typedef struct {
  uint64_t src, dst;
  int32_t n, c, lds_, ldd, reverse, flip_odd;
} tmf_colnorm_desc;          /* 40 bytes */
int tmf_normalise_columns_batched(int dtype, const tmf_colnorm_desc* d_desc, int nprob, void* stream);

/* Rescaling inside a chain of dependent factorisations (the QR sweeps of gutzwiller.py:266 / :471 through TeNPy's
 * canonical_form_finite, which renormalises every step): all `nblk` blocks are multiplied by the SAME power of two 2^-e,
 * e = exponent of their largest entry, so that the products of a long chain neither underflow nor overflow; e is added to
 * the running sum *d_acc, whose new value is also written to *d_out (one slot per step).  Exact: powers of two. */
typedef struct {
  uint64_t A;
  int32_t rows, cols, ld, pad;
} tmf_rescale_desc;                /* 24 bytes */
int tmf_rescale_pow2_batched(int dtype, const tmf_rescale_desc* d_desc, int nblk, int64_t* d_acc, int64_t* d_out, void* stream);

/* Canonical gauge of the entangled orbitals of a cut side.  The reference takes whatever phases (and, inside an exactly
 * degenerate group, whatever basis) LAPACK's eigh returns, slater.py:347; its own example (src/examples/iMPS.py:27-38)
 * relies on two calls with the same matrix giving the same vectors.  Here the Ritz vectors of two sweeps differ by such a
 * choice, so it is fixed: inside every group of columns (`start[c]` = first column of the group of column c, at most 8
 * columns; larger groups are left alone) the basis is rotated until its overlaps with fixed pseudo-random vectors w_0,
 * w_1, ... form a lower triangular matrix with a positive diagonal; a single column gets the phase that makes <w_0 | v>
 * positive.  The weights are indexed by the distance of a row from the cut (from_top: row r is r sites away, else row
 * n - 1 - r is), so the result does not depend on where the chain sits inside a larger matrix. */
typedef struct {
  uint64_t V, start;               /* V: n x k, leading dimension ld; start: int32[k] on the device */
  int32_t n, k, ld, from_top;
} tmf_gauge_desc;                  /* 32 bytes */
int tmf_canonical_gauge_batched(int dtype, const tmf_gauge_desc* d_desc, int nprob, int max_n, int max_k, void* stream);

/* ------------------------------------------------------------------------------------
 * Host: integer / combinatorial part of the sweep (no GPU needed)
 * ---------------------------------------------------------------------------------- */

/* Best-first enumeration of the most probable occupation patterns
 * (schmidt_utils.py:211-324 incl. StoppingCondition.__call__/truncate :99-185), followed
 * by the stable sort by left charge and the Schmidt values of slater.py:672-689.
 *   e[k]         entangled eigenvalues of C_LL, descending (k <= 128)
 *   sectors      NULL (all) or an array of `n_sectors` allowed left charges
 * Outputs (caller-allocated, capacity chi_cap rows; chi_cap >= chi_max + 1 when chi_max > 0):
 *   sets[chi][2]   128-bit occupation masks (bit i: orbital i occupied on the left)
 *   lam_raw[chi]   unnormalised Schmidt values, q_left[chi] left charges (ascending)
 *   *chi           number of kept vectors, *n_checked  subsets popped (for logging)
 * chi_max <= 0 means "no limit".  Returns TMF_E_LIMIT if chi_cap is too small. */
int tmf_cut_vectors(const double* e, int k, int filled_left, int64_t chi_max, double svd_min,
                    double degeneracy_tol, const int64_t* sectors, int n_sectors, int64_t chi_cap,
                    uint64_t* sets, double* lam_raw, int32_t* q_left, int64_t* chi, int64_t* n_checked);

/* Everything integer that slater.py:1023-1067 (+ _select_orbitals :760-825, the sector
 * loop :1132-1141 and the index lists of _tensor_block :857-864) derives from the
 * occupation patterns of the two cuts next to a site.
 *
 * Orbital numbering of a cut side (matches the device layout V = [entangled | filled]):
 * columns 0..k-1 are the entangled orbitals in that side's order, k..k+nf-1 the filled ones.
 * For the left side entangled column j is occupied iff bit j of the mask is set; for the
 * right side entangled column j is occupied iff bit (k-1-j) is NOT set (slater.py:465).
 *
 * Outputs (caller-allocated with the capacities given in `cap`):
 *   row_sel/row_sign [mb], col_sel/col_sign [mk]  rows/cols of W = [always block | S order];
 *        row_sel == -1 marks the physical orbital
 *   k_always, sb, sk
 *   bra_p/bra_alpha [2 chi_b]  physical occupation and bra Schmidt index of every merged row
 *   sectors: for each ket charge sector with matching bra rows: q, r0, r1, c0, c1, n, and the
 *        offsets of its uint8 index lists in idx_pool and of its block in the site output.
 */
typedef struct {
  int32_t mode;              /* 0 = left (A tensor), 1 = right (B tensor)           */
  int32_t k_b, nf_b, chi_b;  /* bra cut side: entangled, filled, kept vectors        */
  int32_t k_k, nf_k, chi_k;  /* ket cut side                                        */
  int32_t pad;
} tmf_site_in;

typedef struct {
  int32_t mb, mk, k_always, sb, sk, n_sectors;
  int64_t idx_bytes;         /* bytes used in idx_pool                              */
  int64_t out_elems;         /* elements of the site's concatenated blocks          */
} tmf_site_out;

typedef struct {
  int32_t q, r0, r1, c0, c1, n;
  int64_t bra_off, ket_off;  /* byte offsets into idx_pool                          */
  int64_t out_off;           /* element offset into the site output                 */
} tmf_sector;

int tmf_site_prepare(const tmf_site_in* in, const uint64_t* sets_b, const int32_t* q_b,
                     const uint64_t* sets_k, const int32_t* q_k, int32_t* row_sel, int8_t* row_sign,
                     int32_t* col_sel, int8_t* col_sign, int32_t* bra_p, int32_t* bra_alpha,
                     tmf_sector* sectors, int32_t sector_cap, uint8_t* idx_pool, int64_t idx_cap,
                     tmf_site_out* out);

/* Batched, multi-threaded forms (one call per sweep).  Per-cut outputs live at stride `cap`
 * (sets: cap*2 words) in the caller's arrays; `chi[i]` receives the number of kept vectors. */
/* fn(i, arg) for i in [0, n) on the library's persistent worker threads (first non-zero status is returned) */
int tmf_host_parallel_for(int n, int nthreads, int (*fn)(int, void*), void* arg);
int tmf_cut_vectors_batch(int ncuts, const double* e_pool, const int64_t* e_off, const int32_t* k,
                          const int32_t* filled_left, int64_t chi_max, double svd_min, double degeneracy_tol,
                          const int64_t* sectors, int n_sectors, int64_t cap, uint64_t* sets, double* lam_raw,
                          int32_t* q_left, int64_t* chi, int64_t* n_checked, int nthreads);

typedef struct {
  int32_t mode, cut_b, cut_k;        /* cut indices into the arrays of tmf_cut_vectors_batch */
  int32_t k_b, nf_b, k_k, nf_k, sec_cap;
  int64_t row_off, col_off, bra_off, sec_off, idx_off, idx_cap;  /* where this site's outputs go */
} tmf_site_job;                      /* 80 bytes */

int tmf_site_prepare_batch(int nsites, const tmf_site_job* jobs, const uint64_t* sets, const int32_t* q_left,
                           const int64_t* chi, int64_t cap, int32_t* row_sel, int8_t* row_sign, int32_t* col_sel,
                           int8_t* col_sign, int32_t* bra_p, int32_t* bra_alpha, tmf_sector* sectors,
                           uint8_t* idx_pool, tmf_site_out* outs, int nthreads);

/* Host: tile descriptors of tmf_det_ppt_batched for all sites of a sweep from the outputs of
 * tmf_site_prepare_batch (one tile = one charge sector, or a range of its bra rows beyond
 * `pairs_per_tile` pairs; largest tiles first).  Sectors that kernel does not take are returned in
 * `rest` as (site << 32 | sector).  Call with tiles = NULL to count; returns the number of tiles.
 * flops_n3 = sum over sectors of pairs * n^3 (the caller multiplies by 8/3 or 2/3). */
typedef struct {
  uint64_t S, scale, idx_base, out_base;   /* device addresses of the site's Schur complement, det_always,
                                              index pool and output block */
  int32_t lds, pad;                        /* leading dimension of S */
} tmf_det_site;                            /* 40 bytes */
int64_t tmf_det_tiles_build(int nsites, const tmf_site_job* jobs, const tmf_site_out* outs, const tmf_sector* sectors,
                            const tmf_det_site* sites, int elem_bytes, int64_t pairs_per_tile, tmf_det_desc* tiles,
                            int64_t tile_cap, int64_t* rest, int64_t rest_cap, int64_t* n_rest, int32_t* lds_max,
                            double* flops_n3, int64_t* n_pairs);

/* ------------------------------------------------------------------------------------
 * Sweep level: one conversion per call.  Replaces the driver loop of slater.C_to_MPS
 * (slater.py:1293-1346: centre cut, right sweep, left sweep) together with
 * SchmidtVectors.from_correlation_matrix (:702-755) and MPSTensorData.from_schmidt_vectors
 * (:975-1104) for all cuts / sites at once: every stage is one batched launch over all of them, the
 * descriptors are built here in C++, two host round trips (eigenvalues down, index lists up).
 *
 * A context owns one device: its streams (launch, download, upload), a device arena for the
 * temporaries of a sweep, page-locked staging memory and the double-buffered output blocks.
 * Calls on one context are not re-entrant (one conversion at a time; the DOWNLOAD of a finished
 * conversion may still be running while the next one computes).
 *
 * Staged form (what the Python host code drives; the stages are separate calls because a
 * multi-GPU run reduces the range-finder decisions over the ranks between them):
 *   tmf_sweep_begin      C up (row-major, as NumPy holds it), cut-side problems of the site range
 *   tmf_sweep_entangled  entangled orbitals + eigenvalues of every cut with a P-column range finder;
 *                        blocks until the eigenvalues are in host memory; reports the quantities the
 *                        adaptive rule needs.  Repeat with iterations = 1 or a larger P as needed.
 *   tmf_sweep_sites      classification, enumeration (schmidt_utils.py:211-324) and site preparation
 *                        (host threads) overlapped with the filled-orbital bases; overlaps (slater.py:1071),
 *                        Schur complements (:1077-1090), all determinants (:828-869) enqueued; reports
 *                        the sizes of the result arrays
 *   tmf_sweep_download   bookkeeping arrays into caller memory, tensors by asynchronous DMA into caller
 *                        (page-locked) memory on the download stream; returns a ticket
 *   tmf_sweep_wait       blocks until the ticket's tensors have landed; returns the deviations of
 *                        testing.check_schmidt_decomposition (testing.py:131-177) for the centre cut
 * tmf_slater_sweep runs all of them for one rank and owns the result (tmf_result_* accessors).
 * ---------------------------------------------------------------------------------- */
typedef struct tmf_ctx tmf_ctx;
typedef struct tmf_result tmf_result;

int tmf_ctx_create(int device, tmf_ctx** out);
void tmf_ctx_destroy(tmf_ctx* ctx);

#define TMF_SWEEP_CHECKS 1u        /* evaluate the self-check of the centre cut (TEST_ACTION != "pass")   */
#define TMF_SWEEP_TIME_KERNELS 2u  /* HIP events around every MFMA GEMM and determinant launch (bench.py)  */
#define TMF_SWEEP_RANGE_BCGS 4u    /* range-finder QRs by blocked Gram-Schmidt instead of the slab kernel  */
#define TMF_SWEEP_NO_CHOLQR 8u     /* filled-basis panels by the LDS Gram-Schmidt kernel                  */
#define TMF_SWEEP_DET_REDUCED 16u  /* all minors through tmf_det_reduced_batched (A/B switch)              */
#define TMF_SWEEP_DET_DIRECT 32u   /* ... through tmf_det_gather_batched                                  */
#define TMF_SWEEP_C_ON_DEVICE 64u  /* C is a device pointer (row-major) instead of host memory            */
#define TMF_SWEEP_TWO_PASSES 128u  /* two projection passes in the filled-basis Gram-Schmidt              */
#define TMF_SWEEP_ONE_STREAM 1024u  /* filled-basis Gram-Schmidt of the left and right blocks on one stream (A/B)   */
#define TMF_SWEEP_NARROW_BCGS 512u /* 16- instead of 64-column outer blocks in the filled-basis Gram-Schmidt (A/B) */
#define TMF_SWEEP_LU_SINGLE 256u   /* Schur complements by tmf_lu_schur_batched (one workgroup per site; A/B)  */
#define TMF_SWEEP_LU_PIVOTED 2048u /* ... by the fully pivoted blocked LU (tmf_lu_block_batched; A/B).  Default: pivoting inside the
                                    * 64 x 64 diagonal blocks (tmf_diag_inverse_batched), fully pivoted only when a block inverse grows */
#define TMF_SWEEP_UNFUSED_BCGS 8192u /* filled-basis blocks orthonormalised by the chain of per-panel launches instead of tmf_block_orth_batched (A/B) */
#define TMF_SWEEP_LU_FORCE_FALLBACK 4096u /* tests: treat every pivot as too small, i.e. run the default and then the fallback */

typedef struct {
  int64_t L;                 /* C is L x L, row-major, real (double) or complex (re, im doubles)          */
  int64_t chi_max;           /* <= 0: no limit (StoppingCondition.chi_max = None)                         */
  double svd_min, degeneracy_tol;
  const int64_t* sectors;    /* NULL, or n_sectors allowed left charges (StoppingCondition.sectors)       */
  int64_t ortho_center;      /* already resolved: slater.py:1291                                          */
  int64_t site_lo, site_hi;  /* sites [site_lo, site_hi) of this rank; 0, L for a whole chain             */
  int32_t n_sectors, is_complex, host_threads;
  uint32_t flags;
} tmf_sweep_params;          /* 80 bytes */

typedef struct {
  int64_t ncut, cap, ns, sec_tot, bra_tot, e_tot, out_elems, elem_bytes;
} tmf_sweep_dims;

/* Destination of the result arrays (host memory of the caller).  Shapes from tmf_sweep_dims:
 * my_cuts[ncut] i64; c_sets[ncut][cap][2] u64; c_lam[ncut][cap] f64; c_q[ncut][cap] i32; c_chi, c_chk[ncut] i64;
 * e_pool[e_tot + 1] f64; e_off[ncut] i64; kk_cut, nfl, nfr[ncut] i32; mode[ns] i32; sec_off, nsec[ns] i64;
 * sectors[sec_tot + 1] tmf_sector; out_off, bra_off, chi_b, chi_k[ns] i64; bra_p, bra_alpha[bra_tot + 1] i32;
 * det[ns] elements; out[out_elems] elements (page-locked memory, else the copy is staged and slow). */
typedef struct {
  void *my_cuts, *c_sets, *c_lam, *c_q, *c_chi, *c_chk, *e_pool, *e_off, *kk_cut, *nfl, *nfr;
  void *mode, *sec_off, *nsec, *sectors, *out_off, *bra_off, *chi_b, *chi_k, *bra_p, *bra_alpha, *det, *out;
} tmf_sweep_ptrs;

typedef struct {
  double stage_ms[16];       /* host wall time per stage, see tmf_sweep_stage_name                        */
  double gemm_ms, gemm_flops;            /* with TMF_SWEEP_TIME_KERNELS: all MFMA GEMM launches of the sweep */
  double det_ms, det_flops, det_all_ms;  /* dominant determinant launch (by time); all determinant launches  */
  int64_t n_det, n_gemm_launches;
  int32_t det_kind;          /* 0 pivoted exchange, 1 reduced, 2 direct */
  int32_t det_order;
  int32_t range_width, range_iterations;
  double range_floor;
  int64_t n_fermion, device_bytes;
  double lu_min_pivot;       /* smallest |pivot| of the block-local elimination of the last sweep (0: method not used) */
  double lu_max_inverse;     /* its largest |entry| of a diagonal-block inverse (fallback above the cap)               */
  int64_t lu_fallbacks;      /* sweeps of this context that had to repeat the LU fully pivoted                */
  /* the GEMM figures above split by kernel instantiation, so that they can be recomputed from a rocprofv3 kernel
   * statistic alone: [0] gemm_kernel<T, 0, 64> (A B), [1] gemm_kernel<T, 1, 64> (A^H B), [2] gemm_kernel<T, *, 16> */
  double gemm_split_ms[3], gemm_split_flops[3];
  int64_t gemm_split_launches[3];
} tmf_sweep_info;

int tmf_sweep_begin(tmf_ctx* ctx, const void* C, const tmf_sweep_params* par);
/* smallest_sigma: largest over the truncated cuts of the smallest singular value the P columns captured;
 * saturated: a truncated cut has P or more directions above the threshold; weak: (iterations > 0) a cut whose
 * smallest captured value still exceeds 4.6e-4 sqrt(threshold); max_sweeps: largest Jacobi sweep count
 * (60 = not converged).  bad_cut: a cut the flags refer to (for messages). */
int tmf_sweep_entangled(tmf_ctx* ctx, int P, int iterations, double* smallest_sigma, int32_t* saturated, int32_t* weak,
                        int32_t* max_sweeps, int64_t* bad_cut);
int tmf_sweep_sites(tmf_ctx* ctx, tmf_sweep_dims* dims);
int tmf_sweep_download(tmf_ctx* ctx, const tmf_sweep_ptrs* dst, int want_tensors, int64_t* ticket);
int tmf_sweep_query(tmf_ctx* ctx, int64_t ticket);     /* 1: landed, 0: still copying, < 0: error */
int tmf_sweep_wait(tmf_ctx* ctx, int64_t ticket, double* checks /* 5 doubles */, int32_t* n_checks);
int tmf_sweep_info_get(tmf_ctx* ctx, tmf_sweep_info* out);
const char* tmf_sweep_stage_name(int i);
/* device address / element count of the tensors of the last tmf_sweep_sites (valid until the next one) */
int tmf_sweep_device_out(tmf_ctx* ctx, uint64_t* d_out, int64_t* elems);

/* ---- Pfaffian (BCS / Nambu mean-field) -> MPS: pfaffian.C_to_MPS (pfaffian.py:1785-1921) in one call ---------------
 * C: the 2L x 2L Nambu correlation matrix in the MAJORANA basis, row-major complex128 in host memory (pfaffian.py:750-751
 * converts from the complex-fermion basis); par->L = number of sites L, par->is_complex = 1, par->ortho_center resolved
 * (pfaffian.py:1843), site_lo / site_hi ignored (whole chain).  Runs the cut decomposition of every bond
 * (SchmidtModes / SchmidtVectors.from_correlation_matrix, pfaffian.py:685-920, :1008-1248: entangled pairs, vacuum
 * parities, best-first enumeration, (parity, number) order), `_pfaffian_matrix` of every site (pfaffian.py:1258-1410) and all
 * sub-Pfaffians (`_tensor_block`, pfaffian.py:1429-1479); with TMF_SWEEP_CHECKS the deviations of
 * testing.check_schmidt_decomposition for both sides of every cut (pfaffian.py:919).  range_floor_tol <= 0 -> 3e-15.
 * Returns TMF_E_HALF_MODES when a cut carries eigenvalue-1/2 modes (pfaffian.py:803-816): following the reference there
 * needs SciPy's seeded ortho_group stream, which the Python driver (temfpy_amd/engine_pf.py, same kernels) has. */
#define TMF_E_HALF_MODES (-4)
typedef struct tmf_pf_result tmf_pf_result;
int tmf_pfaffian_sweep(tmf_ctx* ctx, const void* C, const tmf_sweep_params* par, double range_floor_tol, tmf_pf_result** out);
typedef struct {
  int64_t x;
  int32_t k, chi, p_left, p_right;   /* entangled pairs, kept Schmidt vectors, vacuum parities pL / pR (SchmidtModes) */
  const double* e;           /* k lower eigenvalues, ascending (pfaffian.py:839)                                    */
  const uint8_t* sets;       /* chi x k row-major: excitation pattern of every Schmidt vector, rows in (parity, number) order */
  const double* lam_raw;     /* unnormalised Schmidt values, same order                                            */
} tmf_pf_bond_view;
typedef struct {
  int64_t site;
  int32_t mode, qtotal;      /* 0: A tensor, 1: B tensor; total parity charge of the tensor (pfaffian.py:1733)      */
  int32_t chi_bra, chi_ket, n_blocks, pad;
  double norm;               /* Onishi norm sqrt(prod sv(U)) (pfaffian.py:1352-1359), already multiplied into the blocks */
  const int32_t* leg_idx_bra; /* 2 chi_bra entries: row of the unsorted (p, bra) pipe for every row of the sorted leg  */
} tmf_pf_site_view;
typedef struct {
  int32_t n_bra, n_ket, r0, r1, c0, c1;   /* excitation numbers of the bra / ket sector; rows of the sorted bra leg, ket columns */
  const void* data;          /* (r1 - r0) x (c1 - c0) complex128, row-major                                         */
} tmf_pf_block_view;
int tmf_pf_result_dims(const tmf_pf_result* r, int64_t* L, int64_t* ortho_center, int64_t* out_elems, int32_t* n_checks,
                       tmf_sweep_info* info);
int tmf_pf_result_bond(const tmf_pf_result* r, int64_t b, tmf_pf_bond_view* out);
int tmf_pf_result_site(const tmf_pf_result* r, int64_t i, tmf_pf_site_view* out);
int tmf_pf_result_block(const tmf_pf_result* r, int64_t i, int64_t j, tmf_pf_block_view* out);
/* n_checks entries: deviation, cut, kind (0 "vL is not unitary", 1 "vL does not diagonalise C_LL", 2 / 3 the same for R) */
int tmf_pf_result_checks(const tmf_pf_result* r, double* values, int32_t* cuts, int32_t* kinds);
/* The whole result as flat tables (valid until tmf_pf_result_free): what the per-bond / per-site / per-block accessors
 * return, concatenated.  bond: (k, chi, pL, pR) per bond; e / sets / lam_raw of bond b at [x_off[b], x_off[b + 1]);
 * site: (mode, qtotal, chi_bra, chi_ket, n_blocks) per site; blk: (n_bra, n_ket, r0, r1, c0, c1, element offset into out)
 * per block, the blocks of site i at rows [blk_off[i], blk_off[i + 1]). */
typedef struct {
  int64_t n_bonds, n_sites, n_blocks, out_elems;
  const int32_t* bond;
  const int64_t* e_off;
  const double* e;
  const int64_t* sets_off;
  const uint8_t* sets;
  const int64_t* lam_off;
  const double* lam_raw;
  const int32_t* site;
  const double* norm;
  const int64_t* leg_off;
  const int32_t* leg_idx_bra;
  const int64_t* blk_off;
  const int64_t* blk;
  const void* out;
} tmf_pf_flat;
int tmf_pf_result_flat(const tmf_pf_result* r, tmf_pf_flat* out);
/* Copies the tensors (out_elems complex128 elements) from the context's device memory into `dst` (page-locked memory for
 * full PCIe speed) and blocks until they are there; call before the context's next sweep.  Afterwards the `data` / `out`
 * pointers of the accessors point into `dst` (NULL before). */
int tmf_pf_result_download(tmf_pf_result* r, void* dst);
void tmf_pf_result_free(tmf_pf_result* r);

/* One call: the whole conversion of sites [site_lo, site_hi) with the adaptive range finder (64, 128, 256
 * columns; one subspace iteration when the smallest captured singular value exceeds range_floor_tol <= 0 ->
 * 1e-11), result in page-locked memory owned by the returned object. */
int tmf_slater_sweep(tmf_ctx* ctx, const void* C, const tmf_sweep_params* par, double range_floor_tol, tmf_result** out);
typedef struct {
  int64_t x, chi, k, n_filled_left, n_filled_right, n_checked;
  const double* e;           /* k entangled eigenvalues of C_LL, descending (SchmidtModes.e)              */
  const uint64_t* masks;     /* chi x 2 words: bit i = entangled orbital i occupied on the left           */
  const double* lam_raw;     /* unnormalised Schmidt values (SchmidtVectors.schmidt_values)               */
  const int32_t* q_left;     /* particles left of the cut, ascending (idx_L)                              */
} tmf_bond_view;
typedef struct {
  int64_t site, chi_bra, chi_ket, n_blocks;
  int32_t mode, pad;         /* 0: A tensor (left of the centre), 1: B tensor                              */
  double det_always[2];
  const int32_t *bra_p, *bra_alpha;   /* 2 chi_bra entries: physical occupation / bra index of every merged row */
} tmf_site_view;
typedef struct {
  int32_t q, r0, r1, c0, c1, n;
  const void* data;          /* (r1 - r0) x (c1 - c0), row-major                                          */
} tmf_block_view;
int tmf_result_dims(const tmf_result* r, tmf_sweep_dims* dims, int64_t* L, int64_t* site_lo, int64_t* site_hi);
int tmf_result_bond(const tmf_result* r, int64_t b, tmf_bond_view* out);      /* TMF_E_ARG: not held by this range */
int tmf_result_site(const tmf_result* r, int64_t i, tmf_site_view* out);
int tmf_result_block(const tmf_result* r, int64_t i, int64_t j, tmf_block_view* out);
int tmf_result_checks(const tmf_result* r, double* checks, int32_t* n_checks);
void tmf_result_free(tmf_result* r);

#ifdef __cplusplus
}
#endif
#endif
