// stedc.h -- device-side divide & conquer tridiagonal eigensolver (internal).
#pragma once
#include <functional>

#include "blas3.h"

namespace eig {

// Eigen-decomposition of the symmetric tridiagonal (d_d[N], e_d[N-1]) on the device.
// w_d[N] <- eigenvalues ascending; *Q_out <- device pointer (scratch owned by the context) to the
// N x N eigenvector matrix (column j <-> w[j]), leading dimension *ldq_out.  Returns 0 / -1.
// il, iu: eigenvectors wanted by the caller (1-based, inclusive; iu < 0 = all): columns outside il..iu of Q_out may be left
// uncomputed by the root merge.  All N eigenvalues are always returned.
int stedc_device(Ctx& c, hipStream_t st, int N, const double* d_d, const double* e_d, double* w_d, double** Q_out, int* ldq_out,
                 int il = 1, int iu = -1);

}  // namespace eig
