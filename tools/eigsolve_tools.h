/* eigsolve_tools.h -- experiment hooks of the TOOLS-side build of the library.
 *
 * These entry points are NOT part of libeigsolve_gpu.so and not of the reference's interface.  They exist only in
 *     tools/_lib/libeigsolve_gpu.so      (make -C eigensolver_gpu_amd/csrc tools  ==  the same sources with -DEIG_TOOLS)
 * which tools/gemm_shapes.py, tools/dgemm_shapes.py and tools/two_stage_model.py load through tools/_toolslib.py. */
#ifndef EIGSOLVE_TOOLS_H
#define EIGSOLVE_TOOLS_H
#include "../include/eigsolve_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ?gemm_probe = ?gemm_bench with beta = 1 when beta_one != 0 and operand masks (0 none, 1 upper, 2 strictly upper, 3 lower,
 * 4 unit trapezoid with offset moff, in stored coordinates) -- the forms the solve's triangular / trapezoidal products take;
 * debug_two_stage_model times the launch skeleton of stage 1 of a two-stage reduction (full -> band 64) of order N on
 * pseudo-random data (what: 0 whole stage, 1 panels only, 2 trailing updates only): the go / no-go measurement of round 5. */
int eigsolve_zgemm_probe(char ta, char tb, int M, int N, int K, const void *A_d, int lda, const void *B_d, int ldb,
                         void *C_d, int ldc, int reps, int beta_one, int maskA, int moffA, int maskB, int moffB,
                         double *ms_avg);
int eigsolve_dgemm_probe(char ta, char tb, int M, int N, int K, const double *A_d, int lda, const double *B_d,
                         int ldb, double *C_d, int ldc, int reps, int beta_one, int maskA, int moffA, int maskB,
                         int moffB, double *ms_avg);
int eigsolve_debug_two_stage_model(int N, int cplx, int what, int reps, double *ms_avg);

#ifdef __cplusplus
}
#endif
#endif
