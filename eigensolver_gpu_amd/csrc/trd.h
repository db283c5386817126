// trd.h -- tridiagonalization layer (internal).
#pragma once
#include "blas3.h"

namespace eig {

// Blocked Householder tridiagonalization, uplo='U' (zhetrd_gpu.F90:30-96).  W = N x nb
// workspace (ld N).  d[N], e[N-1], tau[N-1] on device.
template <class T>
void hetrd_upper(Ctx& c, hipStream_t st, int N, T* A, int lda, double* d, double* e, T* tau, T* W, int nb);

// nprob problems of the same order in lockstep: every per-column launch of the panels carries all of them (arrays of
// per-problem pointers; workspaces as for hetrd_upper).  Results per problem are bit-identical to hetrd_upper's.
template <class T>
void hetrd_upper_batch(Ctx& c, hipStream_t st, int N, int nprob, T* const* A, int lda, double* const* d, double* const* e,
                       T* const* tau, T* const* W, int nb);

// y = A x, A Hermitian upper (zhemv_gpu.F90:33-193).  gather=false leaves only the tile
// partials in scratch (used to time the HBM-bound kernel alone).
template <class T> void hemv_upper(Ctx& c, hipStream_t st, int n, const T* A, int lda, const T* x, T* y, bool gather);

// Launches only the panel mat-vec kernels of a full tridiagonalization (bench.py roofline leg).
template <class T>
void hetrd_mv_sweep(Ctx& c, hipStream_t st, int N, T* A, int lda, T* W, int nb, double* e, T* tau, long* nlaunch, double* algo_bytes);
template <class T>
void hetrd_her2k_sweep(Ctx& c, hipStream_t st, int N, T* A, int lda, T* W, int nb, long* nlaunch, double* flops);

// Allocates (if needed) every scratch slot hetrd_upper uses for order N and returns one of the pointers
// (graph capture must not allocate; the pointer doubles as a cache-validity token).
template <class T> const void* hemv_scratch_touch(Ctx& c, int N, const void** all6 = nullptr);

}  // namespace eig
