/*
 * eigsolve_gpu.h -- C ABI of the MI355X-native generalized symmetric/Hermitian-definite
 * eigensolver (drop-in boundary for NVIDIA/Eigensolver_gpu's dsygvdx_gpu / zhegvdx_gpu).
 *
 * Everything here is extern "C", plain pointers and ints: the Fortran modules under
 * eigensolver_gpu_amd/fortran/ (same module + procedure names and argument order as the
 * reference) are thin iso_c_binding wrappers over these symbols, and the Python host
 * mirror (eigensolver_gpu_amd/api.py) binds the same symbols with ctypes.
 *
 * Conventions (identical to the reference): column-major, fp64; complex = interleaved
 * (re,im) doubles = Fortran complex(8); "_d" pointers are DEVICE pointers, "_h" pointers
 * are HOST pointers (pinned recommended, pageable works); leading dimensions in elements.
 * Reference citations are relative to /root/reference/lib_eigsolve/.
 */
#ifndef EIGSOLVE_GPU_H
#define EIGSOLVE_GPU_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime state (replaces module eigsolve_vars, eigsolve_vars.F90:25-61) ------------ */

/* init_eigsolve_gpu (eigsolve_vars.F90:39-59): creates the per-(thread,device) context
 * (streams, events, scratch arena) for the CURRENT HIP device.  Called lazily by every
 * entry point (zhegvdx_gpu.F90:131).  Returns 0 on success. */
int eigsolve_init(void);

/* Releases the calling thread's context for the current device (the reference never frees). */
int eigsolve_finalize(void);

/* Host LAPACK used for the tridiagonal step (reference: zstedc/dstedc from the linked
 * LAPACK, zheevd_gpu.F90:101 / dsyevd_gpu.F90:99).  `path` = shared library exporting
 * dstedc_ (or scipy_dstedc_); NULL -> $EIGSOLVE_LAPACK_LIB, then the process' own symbols.
 * Returns 0 if a dstedc symbol was resolved. */
int eigsolve_set_lapack(const char *path);

/* Caps the host LAPACK thread count (OpenBLAS builds only; no-op otherwise). */
int eigsolve_set_host_threads(int nthreads);

/* Tunables (the reference hard-codes its own: trd nb=32 zheevd_gpu.F90:63, back-transform nb=64 :64, gst nb=448
 * zhegvdx_gpu.F90:156).  Every option is also read from the environment variable EIGSOLVE_<NAME> (upper case, same values;
 * TRIDIAG additionally accepts "host" / "device", POTRF "rec"; any other non-numeric value is ignored with a message on stderr) when a
 * context is created.  value < 0 restores the default everywhere; value = 0 restores it too unless 0 is itself a setting of the
 * option (tridiag, gst, potrf, overlap, batch_workers, hemv_blocks, trd_finish: 0 = the smallest cut-over, 32).
 *   "tridiag"   0 = host LAPACK dstedc exactly as the reference, 1 = device-side divide & conquer (default).
 *   "trd_nb"    panel width of the tridiagonalization, 1..64 (default 32 = the reference's, zheevd_gpu.F90:63; the caller's workspace contract bounds it).
 *   "trd_finish" order at which the blocked reduction hands the rest of the matrix to a one-workgroup kernel: -1 (default) =
 *               128 (complex) / 192 (real), the matrix then lives in the registers of one CU; 32 = the reference's cut-over
 *               (zhetrd_gpu.F90:84-87).
 *   "bt_nb"     reflectors per block of the back-transformation: 64 (the reference's larfb width), 128, 256 (default) or 512 --
 *               64-blocks whose T factors are merged pairwise, so the rank-k updates run at K = bt_nb.
 *   "gst"       reduction to standard form, 0 = symmetric recursion of zhegst_gpu.F90:51-107 down to 64x64 blocks, 1 = two full
 *               triangular solves on a Hermitian-completed copy, 2 (default) = the symmetric algorithm while the diagonal
 *               blocks are larger than "gst_thr" (default 1024), two solves below, 3 = the reference's blocked loop with
 *               nb = "trsm_base".
 *   "trsm_base" order of the inverted diagonal blocks the triangular solves outside potrf stop at: 64, 256 (default: the
 *               64-block inverses of the factorization merged), 512 or 1024.
 *   "potrf"     2 (default) right-looking Cholesky with block rows of 64 taken in pairs: a block-row kernel whose elimination is
 *               blocked by 16 and runs on MFMA, one rank-128 MFMA update per pair; 1 the round-2 form (scalar block-row kernel, one
 *               rank-64 update per block row); 0 the recursive form (no intra-grid dependency).
 *   "overlap"   bit mask of independent launch chains of one solve that run on a second stream (leased from the library's
 *               stream pool for the call): bit 0 = hegst beside the factorization, released stage by stage, bit 1 = larft T
 *               factors beside the tridiagonal eigensolver (zheevd_gpu.F90:125 overlaps the same work) and, for N*m >= 2^20,
 *               the host copy Z_h in row blocks beside the final triangular solve (zhegvdx_gpu.F90:169-180), bit 2 = look-ahead
 *               in the Cholesky factorization (the block rows of the next pair beside the rank-128 update of the rest, third
 *               stream; measured: no gain, the block-row kernel needs whole CUs and the update holds them all).  Default 3.  Same
 *               kernels, operands and order of operations per block: results are bit-identical to "overlap" 0.  Only applied
 *               to a solve that has the device to itself (best effort: no other call of this library in flight, not inside a
 *               batch call).
 *   "batch_workers" problems kept in flight inside one eigsolve_?hegvdx_batch call (default automatic: 4 when the process
 *               allows >= 5 hardware queues through GPU_MAX_HW_QUEUES, else 3; 0 = lockstep form on the caller's context).
 *   "batch_fuse" problems per launch chain of a batch call that share the per-column launches of the tridiagonalization
 *               (lockstep groups, 1..4; default automatic: min(4, problems / chains) while a matrix is <= 96 MiB, else 1).
 *   "batch_zip" inside a lockstep group the BLAS-3 phases run as ONE launch sequence for the group: every product on the MFMA engine
 *               carries all problems of the group (pointer table, blockIdx.z = problem x K-split), every other kernel is launched
 *               once per problem at its position.  Bit 0 = the phases in front of the tridiagonalization (inverse blocks, reduction
 *               to standard form), bit 1 = those behind it (back-transformation, final solve); default 3, 0 = problem after
 *               problem.  Same kernels, tiles and K order per problem: bit-identical.
 *   "real_il_reference" 1 = dsygvdx/dsyevd return eigenvectors 1..m whatever il is, as the real reference path does
 *               (dsyevd_gpu.F90:108); 0 (default) = il is honoured like in the complex path (zheevd_gpu.F90:110).
 *   "graph"     1 = the tridiagonalization's ~2N dependent launches are captured once per (type, N) on an internal working
 *               copy of A and replayed as a hipGraph; 0 (default) = eager launches (measured neutral).
 *   "tile_map"  1 (default) = XCD-aware super-tile order of the MFMA engine's workgroups, 0 = plain grids (A/B measurements).
 *   "hemv_blocks" workgroups of the panel mat-vec kernel (0 = automatic: one workgroup per CU, in every mode -- a grid that depended
 *               on the mode would change the order of the partial sums and with it the bit-identity of batch and single solves).
 *   "trace_marks" 1 = marker kernels at the phase boundaries (segments a rocprofv3 kernel trace, tools/trace_phases.py).
 *   "zs_cap_mb" largest library-side copy (MiB, default 4096) of the standard problem's eigenvectors the generalized drivers keep:
 *               they form those N x m vectors in library scratch (sizeof(T) N m bytes of device memory per context on top of the
 *               caller's buffers: 64 MiB at C3, 1 GiB at C4 full spectrum; every worker context of a batch call has its own) and the
 *               final triangular solve writes the caller's Z once.  Above the cap, or when the device cannot provide the block, the
 *               vectors are formed in the caller's Z (as the reference does) and the solve runs in column chunks through a smaller
 *               block -- same results to rounding.  Scope: eigsolve_?hegvdx and the problems a batch call solves one per launch
 *               chain; the LOCKSTEP groups of a batch call (small orders, "batch_fuse" > 1 or "batch_workers" = 0: a matrix of
 *               at most 96 MiB in the automatic setting) and the stage-level eigsolve_?trsm_lun always keep the full block.
 *   "gemm_dma"  staging path of the MFMA engine's complex 64 x 64 tiles: 0 = global -> registers -> ds_write (gemm_fast_kernel),
 *               1 = LDS-DMA (global_load_lds_dwordx4 into fragment-ordered LDS blocks, gemm_dma_kernel), 2 = LDS-DMA with persistent
 *               workgroups, 3 (default) = LDS-DMA for work items with at least 96 of K, registers below.  Same summation order and
 *               lane mapping in every form: results are bit-identical.
 *   "gemm_lean" largest K of a work item of the complex 64 x 64 tiles that runs on the lean LDS-DMA form (K-slabs of 8, the C tile fetched
 *               in the epilogue: 116 VGPRs, 32 KB of LDS, four workgroups per CU instead of two) when the launch has at least four
 *               tiles per CU -- the trailing rank-2nb updates of the tridiagonalization and the rank-128 updates of the
 *               factorization at orders >= ~3000.  Default 128, 0 = never.  Bit-identical to the other forms.
 *   "gemm_wide" complex products with fewer than one 64 x 64 tile per CU (base cases of the triangular solves, W T^H of the
 *               back-transformation, merges of inverse blocks): 0 = 32 x 32 tiles on four-wave workgroups, 1 = 32 x 32 tiles on
 *               whole-CU workgroups (16 waves, K split inside the workgroup, partial tiles summed through LDS in a fixed order:
 *               gemm_wide_kernel) for launches of at most one tile per CU, 2 = also 8-wave workgroups up to two tiles per
 *               CU.  Default 0: +0-10 % on the bare shapes, nothing in a solve, -1..2 % in batches (profiles/r06_experiments.txt
 *               section 8).  The summation order over k differs between the forms (results agree to rounding); the form is chosen from the
 *               product's shape alone; execution modes that cut a product's output differently (the potrf || hegst pipeline) then differ in the
 *               last bits, which they never do with the option off.
 *   "mv_dma"    smallest trailing order from which the panel mat-vec streams its tiles through an LDS-DMA ring instead of registers
 *               (0 = never, the default: measured slower at every order, profiles/r06_experiments.txt section 1); bit-identical results.
 * Returns 0 / -1 (unknown name). */
int eigsolve_set_option(const char *name, int value);

/* nvtxStartRange / nvtxEndRange (lib_eigsolve/toolbox.F90:71-97) -> roctx ranges when
 * librocprofiler-sdk-roctx is loadable, otherwise no-ops.  Like the reference they
 * synchronise the device before push and before pop (toolbox.F90:77,94). */
void eigsolve_range_push(const char *name, int color_id);
void eigsolve_range_pop(void);

/* Per-phase wall times (ms) of the LAST driver call on this thread/device, measured with
 * HIP events on the library's stream plus a host clock for the host LAPACK step:
 * [0] potrf [1] gst [2] trd total [3] host stedc (+ D2H d,e / H2D vectors) [4] back-transform
 * [5] trsm [6] final D2H copy [7] total.  Returns the number of entries written (<= n). */
int eigsolve_get_phase_times(double *ms, int n);

/* ---- drivers (the drop-in boundary) -------------------------------------------------- */

/* zhegvdx_gpu (zhegvdx_gpu.F90:75-182).  A x = lambda B x, eigenpairs il..iu (1-based),
 * upper triangles of A_d,B_d populated.  On exit: B_d = U (Cholesky factor), upper(A_d)
 * destroyed, strict lower(A_d) preserved, Z_d(:,1:iu-il+1) eigenvectors, w_d(1:N) ALL
 * eigenvalues ascending (zheevd_gpu.F90:111), Z_h/w_h host copies (Z_h skipped when
 * skip_host_copy != 0).  Size contract (checked, *info=-1 + message otherwise):
 * lwork >= 2*64*64+65*N, lrwork >= N, lwork_h >= N, lrwork_h >= 1+5*N+2*N*N,
 * liwork_h >= 3+5*N when the host dstedc is selected ("tridiag" = 0), >= N otherwise (the reference announces 3+5*N and
 * rejects only liwork_h < N, zhegvdx_gpu.F90:123); Z_d and Z_h need N columns.  *info = 0 ok / -1 error (bad workspace,
 * B not positive definite, tridiagonal solver failure, copy failure).  Blocking: results
 * are valid on return.  Return value == *info. */
int eigsolve_zhegvdx(int N, void *A_d, int lda, void *B_d, int ldb, void *Z_d, int ldz, int il, int iu,
                     double *w_d, void *work_d, int lwork, double *rwork_d, int lrwork, void *work_h,
                     int lwork_h, double *rwork_h, int lrwork_h, int *iwork_h, int liwork_h, void *Z_h,
                     int ldz_h, double *w_h, int *info, int skip_host_copy);

/* dsygvdx_gpu (dsygvdx_gpu.F90:71-168).  Real analogue; lwork >= 2*64*64+66*N,
 * lwork_h >= 1+6*N+2*N*N, liwork_h >= 3+5*N. */
int eigsolve_dsygvdx(int N, double *A_d, int lda, double *B_d, int ldb, double *Z_d, int ldz, int il, int iu,
                     double *w_d, double *work_d, int lwork, double *work_h, int lwork_h, int *iwork_h,
                     int liwork_h, double *Z_h, int ldz_h, double *w_h, int *info, int skip_host_copy);

/* Batch of nprob problems of ONE order (QE k-point style, BASELINE.json configs[4]) solved by one call from one host thread.
 * The library keeps "batch_workers" (automatic: 3, or 4 with GPU_MAX_HW_QUEUES >= 5) of the problems in flight on its own
 * worker threads -- the caller's thread is one of them --, each problem an ordinary single-problem solve on a context and stream of its own, so that the
 * latency-bound phases of one solve fill under the kernels of the others (C3: 16.6 problems/s against 10.3 for one call
 * per problem).  "batch_workers" = 0: everything on the caller's context, the tridiagonalizations in LOCKSTEP (every
 * per-column launch carries all problems), the other phases problem after problem.  Arguments as eigsolve_zhegvdx /
 * eigsolve_dsygvdx with one pointer per problem (host arrays of nprob device / host pointers, no null entries; Z_h may be
 * NULL only with skip_host_copy); the same workspace minima per problem (lwork, lrwork); no host workspaces: the batch
 * driver uses the device tridiagonal solver ("tridiag" = 1, the default).  info[q] = 0 / -1 per problem; return value -1 if
 * any problem failed or the arguments were rejected (then every info[q] = -1).  Per-problem results are bit-identical to
 * the single-problem driver's in both forms, all options included.  The reference has no counterpart (one problem per
 * call, zhegvdx_gpu.F90:75); this is the batch extension announced in SURVEY.md 8(b) "Threading".  Fortran:
 * modules zhegvdx_gpu_batch / dsygvdx_gpu_batch. */
int eigsolve_zhegvdx_batch(int nprob, int N, void *const *A_d, int lda, void *const *B_d, int ldb, void *const *Z_d, int ldz,
                           int il, int iu, double *const *w_d, void *const *work_d, int lwork, double *const *rwork_d,
                           int lrwork, void *const *Z_h, int ldz_h, double *const *w_h, int *info, int skip_host_copy);
int eigsolve_dsygvdx_batch(int nprob, int N, double *const *A_d, int lda, double *const *B_d, int ldb, double *const *Z_d,
                           int ldz, int il, int iu, double *const *w_d, double *const *work_d, int lwork,
                           double *const *Z_h, int ldz_h, double *const *w_h, int *info, int skip_host_copy);

/* ---- stage routines (public module procedures of the reference) ---------------------- */

/* zheevd_gpu / dsyevd_gpu (zheevd_gpu.F90:32-134, dsyevd_gpu.F90:32-132): standard problem,
 * jobz='V', uplo='U'.  Same workspace carve-up contract as the drivers. */
int eigsolve_zheevd(int il, int iu, int N, void *A_d, int lda, void *Z_d, int ldz, double *w_d, void *work_d,
                    int lwork, double *rwork_d, int lrwork, void *work_h, int lwork_h, double *rwork_h,
                    int lrwork_h, int *iwork_h, int liwork_h, void *Z_h, int ldz_h, double *w_h, int *info);
int eigsolve_dsyevd(int il, int iu, int N, double *A_d, int lda, double *Z_d, int ldz, double *w_d,
                    double *work_d, int lwork, double *work_h, int lwork_h, int *iwork_h, int liwork_h,
                    double *Z_h, int ldz_h, double *w_h, int *info);

/* zhegst_gpu / dsygst_gpu (zhegst_gpu.F90:31-109): itype=1, uplo='U': A <- U^-H A U^-1,
 * B_d holds U.  nb is accepted for signature compatibility (the recursion picks its own
 * blocking).  Only the upper triangle of A is read or written. */
int eigsolve_zhegst(int N, void *A_d, int lda, const void *B_d, int ldb, int nb);
int eigsolve_dsygst(int N, double *A_d, int lda, const double *B_d, int ldb, int nb);

/* zhetrd_gpu / dsytrd_gpu (zhetrd_gpu.F90:30-96): uplo='U'.  d[N], e[N-1], tau[N-1] device
 * outputs; reflectors in upper(A) exactly as the reference leaves them (explicit 1 at
 * A(i-1,i) for the blocked part, e written back only inside the final 32x32 block).
 * work_d/lwork may be NULL/0 (internal scratch is used); nb<=0 -> default. */
int eigsolve_zhetrd(int N, void *A_d, int lda, double *d_d, double *e_d, void *tau_d, void *work_d, int lwork,
                    int nb);
int eigsolve_dsytrd(int N, double *A_d, int lda, double *d_d, double *e_d, double *tau_d, double *work_d,
                    int lwork, int nb);

/* zlarft_gpu + finish_T_block_kernel (zheevd_gpu.F90:136-176,215-279; dsyevd_gpu.F90:134-174,212-276) for ALL
 * reflector blocks of a tridiagonalized A_d (reflector j in column j+1 of upper(A_d), as ?hetrd leaves them, tau_d[N-1]):
 * block b covers reflectors b*nb .. min((b+1)*nb, N-1)-1, nb = 64 (the reference's larfb width) or 128 (two 64-blocks
 * whose T factors are merged, T10 = -T1 (V1^H V0) T0).  T_d receives ceil((N-1)/nb) lower-triangular factors, block b
 * at T_d + b*ldt*ldt with leading dimension ldt >= min(nb, N) rounded up to 64/128.  Stage-level parity hook. */
int eigsolve_zlarft(int N, const void *A_d, int lda, const void *tau_d, int nb, void *T_d, int ldt);
int eigsolve_dlarft(int N, const double *A_d, int lda, const double *tau_d, int nb, double *T_d, int ldt);

/* The back-transformation loop of zheevd_gpu.F90:113-131 (zlarft_gpu + zlarfb_gpu per block, i.e. LAPACK
 * ZUNMTR('L','U','N')): Z_d(0:N, 0:m) <- Q Z_d with Q = H(N-2)...H(0) from the reflectors in upper(A_d). nb as above. */
int eigsolve_zunmtr(int N, int m, const void *A_d, int lda, const void *tau_d, void *Z_d, int ldz, int nb);
int eigsolve_dormtr(int N, int m, const double *A_d, int lda, const double *tau_d, double *Z_d, int ldz, int nb);

/* Upper Cholesky B = U^H U (replaces cusolverDn?potrf, zhegvdx_gpu.F90:135).  *info_h = 0
 * or the 1-based index of the first non-positive pivot (LAPACK convention). */
int eigsolve_zpotrf(int N, void *B_d, int ldb, int *info_h);
int eigsolve_dpotrf(int N, double *B_d, int ldb, int *info_h);

/* ---- kernel-level entry points (parity tests, micro-benchmarks) ---------------------- */

/* zhemv_gpu / dsymv_gpu (zhemv_gpu.F90:33-193): y = A x, A n x n Hermitian, upper stored.
 * Unlike the reference kernel y need not be pre-zeroed.  Blocking. */
int eigsolve_zhemv(int n, const void *A_d, int lda, const void *x_d, void *y_d);
int eigsolve_dsymv(int n, const double *A_d, int lda, const double *x_d, double *y_d);

/* Times `reps` back-to-back launches of the hemv/symv kernel with HIP events on the
 * library stream; returns the average ms per launch in *ms_avg (roofline leg of bench.py). */
int eigsolve_zhemv_bench(int n, const void *A_d, int lda, const void *x_d, void *y_d, int reps, double *ms_avg);
int eigsolve_dsymv_bench(int n, const double *A_d, int lda, const double *x_d, double *y_d, int reps,
                         double *ms_avg);

/* Roofline leg of bench.py: launches, back to back on the library stream, exactly the sequence
 * of panel mat-vec kernels (hemv + stacked gemv, the HBM-bound kernel) that one ?hetrd of order N
 * issues -- same grids and arguments, the row kernels and her2k updates skipped, so A_d is left
 * numerically meaningless.  *ms_total = HIP-event time of one sweep (average over reps),
 * *nlaunch = kernel launches per sweep, *algo_bytes = sum over launches of s*n(n+1)/2
 * (SURVEY.md 8(d)).  Host clock / events only; results valid on return. */
int eigsolve_zhetrd_mv_sweep(int N, void *A_d, int lda, int nb, int reps, double *ms_total, long *nlaunch,
                             double *algo_bytes);
int eigsolve_dsytrd_mv_sweep(int N, double *A_d, int lda, int nb, int reps, double *ms_total, long *nlaunch,
                             double *algo_bytes);

/* Roofline leg of bench.py: the sequence of trailing rank-2nb updates (zher2k / dsyr2k, zhetrd_gpu.F90:67) that one ?hetrd of
 * order N issues, back to back on the library stream -- same orders, panel widths and operand placement (V = the panel's columns
 * of A_d, W_d = an N x nb panel workspace, ld N).  *ms_total = HIP-event time of one sweep (average over reps), *nlaunch = updates
 * per sweep, *flops = sum of c * 2 * n^2 * k.  A_d is overwritten (each update adds O(|V||W|) to it). */
int eigsolve_zhetrd_her2k_sweep(int N, void *A_d, int lda, void *W_d, int nb, int reps, double *ms_total, long *nlaunch,
                                double *flops);
int eigsolve_dsytrd_her2k_sweep(int N, double *A_d, int lda, double *W_d, int nb, int reps, double *ms_total, long *nlaunch,
                                double *flops);

/* C = alpha op(A) op(B) + beta C on the fp64 MFMA tile engine (replaces cublas?gemm_v2 call
 * sites).  ta/tb in {'N','T','C'}.  alpha/beta: pointer to 1 (d) or 2 (z)
 * doubles on the host. */
int eigsolve_zgemm(char ta, char tb, int M, int N, int K, const double *alpha, const void *A_d, int lda,
                   const void *B_d, int ldb, const double *beta, void *C_d, int ldc);
int eigsolve_dgemm(char ta, char tb, int M, int N, int K, const double *alpha, const double *A_d, int lda,
                   const double *B_d, int ldb, const double *beta, double *C_d, int ldc);
/* Same, timed: average ms over reps launches. */
int eigsolve_zgemm_bench(char ta, char tb, int M, int N, int K, const void *A_d, int lda, const void *B_d, int ldb,
                         void *C_d, int ldc, int reps, double *ms_avg);
int eigsolve_dgemm_bench(char ta, char tb, int M, int N, int K, const double *A_d, int lda, const double *B_d,
                         int ldb, double *C_d, int ldc, int reps, double *ms_avg);

/* her2k/syr2k, uplo='U', trans='N': C <- C - V W^H - W V^H (the trd trailing update,
 * zhetrd_gpu.F90:67,82).  C n x n, V,W n x k. */
int eigsolve_zher2k(int n, int k, const void *V_d, int ldv, const void *W_d, int ldw, void *C_d, int ldc);
int eigsolve_dsyr2k(int n, int k, const double *V_d, int ldv, const double *W_d, int ldw, double *C_d, int ldc);
int eigsolve_zher2k_bench(int n, int k, const void *V_d, int ldv, const void *W_d, int ldw, void *C_d, int ldc,
                          int reps, double *ms_avg);
int eigsolve_dsyr2k_bench(int n, int k, const double *V_d, int ldv, const double *W_d, int ldw, double *C_d,
                          int ldc, int reps, double *ms_avg);

/* Z(:,0:m) <- U^-1 Z (cublasZtrsm L,U,N,N, zhegvdx_gpu.F90:169).  U = upper Cholesky factor
 * as left in B_d by ?potrf above (its inverted diagonal blocks are rebuilt here). */
int eigsolve_ztrsm_lun(int N, int m, const void *U_d, int ldu, void *Z_d, int ldz);
int eigsolve_dtrsm_lun(int N, int m, const double *U_d, int ldu, double *Z_d, int ldz);

/* Device-side divide & conquer for the symmetric tridiagonal (d_d[N], e_d[N-1]) -- the "next" row
 * replacing the host zstedc/dstedc('I') of zheevd_gpu.F90:101.  w_d[N] ascending, Q_d (N x N, ldq)
 * eigenvectors (may be NULL), *ms host wall time.  d_d/e_d are not modified (w_d may alias d_d). */
int eigsolve_dstedc_device(int N, const double *d_d, const double *e_d, double *w_d, double *Q_d, int ldq, double *ms);

/* Library version / build info string. */
const char *eigsolve_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EIGSOLVE_GPU_H */
