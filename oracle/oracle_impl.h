/*
 * oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by oracle.c: once with ORACLE_COMPLEX undefined (real(8) path,
 * prefix d_) and once with it defined (complex(8) path, prefix z_).
 *
 * This is a plain-C restatement of the *algorithm* of NVIDIA/Eigensolver_gpu's
 * dsygvdx_gpu / zhegvdx_gpu path, written from the reference's behaviour (file:line
 * citations on every function, all relative to /root/reference/lib_eigsolve/).
 * It is deliberately naive (triple loops, no blocking for speed): its only job is to
 * be an executable specification the HIP kernels are diffed against.
 *
 * Nothing in the product (eigensolver_gpu_amd/) may include, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Conventions: column-major, 0-based C indices; comments quote the reference's 1-based
 * Fortran indices where that helps cross-reading.
 */

#ifdef ORACLE_COMPLEX
#define T double _Complex
#define PFX(n) z_##n
#define CONJ(x) conj(x)
#define REAL(x) creal(x)
#define IMAG(x) cimag(x)
#define MK(re, im) ((re) + (im)*I)
#else
#define T double
#define PFX(n) d_##n
#define CONJ(x) (x)
#define REAL(x) (x)
#define IMAG(x) (0.0)
#define MK(re, im) (re)
#endif

#define A_(i, j) A[(size_t)(i) + (size_t)(j) * lda]
#define B_(i, j) B[(size_t)(i) + (size_t)(j) * ldb]
#define W_(i, j) W[(size_t)(i) + (size_t)(j) * ldw]
#define Z_(i, j) Z[(size_t)(i) + (size_t)(j) * ldz]

/* ------------------------------------------------------------------------------------
 * Input generator.  Reference recipe: test_driver/test_zhegvdx.F90:28-66 and
 * test_driver/test_dsygvdx.F90:28-64: temp = Hermitian with strict-lower entries
 * U[0,1) (+ i U[0,1)), real U[0,1) diagonal; matrix = temp * temp^H.  The reference uses
 * an unseeded Fortran random_number; we use a counter-based splitmix64 stream
 * (oracle_u01) so host and device agree bit-for-bit on the uniform draws.
 * `shift` adds shift*I (the well-conditioned family of SURVEY.md 8(c)).
 * ------------------------------------------------------------------------------------ */
void PFX(gen_spd)(int n, uint64_t seed, double shift, T *A, int lda) {
    T *tmp = (T *)malloc(sizeof(T) * (size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) {
            if (i > j) {
                double re = oracle_u01(seed, (uint64_t)i, (uint64_t)j, 0);
                double im = oracle_u01(seed, (uint64_t)i, (uint64_t)j, 1);
                (void)im;
                T v = MK(re, im);
                tmp[i + (size_t)j * n] = v;
                tmp[j + (size_t)i * n] = CONJ(v);
            } else {
                tmp[i + (size_t)j * n] = MK(oracle_u01(seed, (uint64_t)i, (uint64_t)j, 0), 0.0);
            }
        }
    /* A = tmp * tmp^H   (cublaszgemm 'N','C', test_zhegvdx.F90:59) */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            T s = 0;
            for (int k = 0; k < n; ++k) s += tmp[i + (size_t)k * n] * CONJ(tmp[j + (size_t)k * n]);
            if (i == j) s = MK(REAL(s) + shift, 0.0);
            A_(i, j) = s;
        }
    free(tmp);
}

/* ------------------------------------------------------------------------------------
 * potrf, upper:  B = U^H U, U overwrites the upper triangle of B.
 * Reference call site: cusolverDnZpotrf / cusolverDnDpotrf, zhegvdx_gpu.F90:135,
 * dsygvdx_gpu.F90:121 (closed-source vendor routine; restated as the textbook LAPACK
 * ?potf2('U') column algorithm).  Returns 0, or k+1 if the leading minor k+1 is not PD.
 * The strict lower triangle is not referenced.
 * ------------------------------------------------------------------------------------ */
int PFX(potrf_upper)(int n, T *B, int ldb) {
    for (int j = 0; j < n; ++j) {
        double ajj = REAL(B_(j, j));
        for (int k = 0; k < j; ++k) ajj -= REAL(CONJ(B_(k, j)) * B_(k, j));
        if (!(ajj > 0.0)) return j + 1;
        ajj = sqrt(ajj);
        B_(j, j) = MK(ajj, 0.0);
        for (int c = j + 1; c < n; ++c) {
            T s = B_(j, c);
            for (int k = 0; k < j; ++k) s -= CONJ(B_(k, j)) * B_(k, c);
            B_(j, c) = s / ajj;
        }
    }
    return 0;
}

/* --- small dense helpers used by hegst (restating the cuBLAS calls it makes) -------- */

/* X <- U^{-H} X, U = kb x kb upper, X = kb x nc.   (ztrsm L,U,C,N: zhegst_gpu.F90:68,87) */
static void PFX(trsm_LUC)(int kb, int nc, const T *U, int ldu, T *X, int ldx) {
    for (int c = 0; c < nc; ++c)
        for (int i = 0; i < kb; ++i) {
            T s = X[i + (size_t)c * ldx];
            for (int k = 0; k < i; ++k) s -= CONJ(U[k + (size_t)i * ldu]) * X[k + (size_t)c * ldx];
            X[i + (size_t)c * ldx] = s / CONJ(U[i + (size_t)i * ldu]);
        }
}
/* X <- X U^{-1}, U = nc x nc upper, X = kb x nc.   (ztrsm R,U,N,N: zhegst_gpu.F90:70,103) */
static void PFX(trsm_RUN)(int kb, int nc, const T *U, int ldu, T *X, int ldx) {
    for (int c = 0; c < nc; ++c) {
        for (int k = 0; k < c; ++k) {
            T u = U[k + (size_t)c * ldu];
            for (int i = 0; i < kb; ++i) X[i + (size_t)c * ldx] -= X[i + (size_t)k * ldx] * u;
        }
        T d = U[c + (size_t)c * ldu];
        for (int i = 0; i < kb; ++i) X[i + (size_t)c * ldx] /= d;
    }
}
/* X <- U^{-1} X, U = n x n upper, X = n x nc.      (cublasZtrsm L,U,N,N: zhegvdx_gpu.F90:169) */
void PFX(trsm_LUN)(int n, int nc, const T *U, int ldu, T *X, int ldx) {
    for (int c = 0; c < nc; ++c)
        for (int i = n - 1; i >= 0; --i) {
            T s = X[i + (size_t)c * ldx];
            for (int k = i + 1; k < n; ++k) s -= U[i + (size_t)k * ldu] * X[k + (size_t)c * ldx];
            X[i + (size_t)c * ldx] = s / U[i + (size_t)i * ldu];
        }
}

/* ------------------------------------------------------------------------------------
 * hegst / sygst, itype=1, uplo='U':  A <- U^{-H} A U^{-1}.
 * Follows zhegst_gpu.F90:51-107 / dsygst_gpu.F90:48-96 step by step (block size nb,
 * reference uses 448).  Like the reference, the strict lower triangle of each *diagonal
 * block* of A is overwritten (Hermitian completion, :57-65); everything else below the
 * diagonal is untouched.  B holds U (upper Cholesky factor).
 * ------------------------------------------------------------------------------------ */
void PFX(hegst)(int n, T *A, int lda, const T *B, int ldb, int nb) {
    for (int k = 0; k < n; k += nb) {
        int kb = (n - k < nb) ? n - k : nb;
        int r = n - k - kb;
        T *Akk = &A_(k, k);
        /* :57-65 complete the diagonal block */
        for (int j = 0; j < kb; ++j)
            for (int i = j + 1; i < kb; ++i) Akk[i + (size_t)j * lda] = CONJ(Akk[j + (size_t)i * lda]);
        /* :68-71 two full-block trsm */
        PFX(trsm_LUC)(kb, kb, &B_(k, k), ldb, Akk, lda);
        PFX(trsm_RUN)(kb, kb, &B_(k, k), ldb, Akk, lda);
        /* :73-81 force real diagonal */
        for (int j = 0; j < kb; ++j) Akk[j + (size_t)j * lda] = MK(REAL(Akk[j + (size_t)j * lda]), 0.0);
        if (r > 0) {
            T *Akr = &A_(k, k + kb);
            const T *Ukr = &B_(k, k + kb);
            /* :87-88 */
            PFX(trsm_LUC)(kb, r, &B_(k, k), ldb, Akr, lda);
            /* :93-94  A_kr -= 1/2 A_kk U_kr (gemm on the fully populated block) */
            for (int pass = 0; pass < 2; ++pass) {
                T *tmp = (T *)malloc(sizeof(T) * (size_t)kb * r);
                for (int c = 0; c < r; ++c)
                    for (int i = 0; i < kb; ++i) {
                        T s = 0;
                        for (int p = 0; p < kb; ++p) s += Akk[i + (size_t)p * lda] * Ukr[p + (size_t)c * ldb];
                        tmp[i + (size_t)c * kb] = s;
                    }
                for (int c = 0; c < r; ++c)
                    for (int i = 0; i < kb; ++i) Akr[i + (size_t)c * lda] -= 0.5 * tmp[i + (size_t)c * kb];
                free(tmp);
                if (pass == 0) {
                    /* :95-96 her2k 'C', upper: A_rr -= A_kr^H U_kr + U_kr^H A_kr */
                    T *Arr = &A_(k + kb, k + kb);
                    for (int c = 0; c < r; ++c)
                        for (int i = 0; i <= c; ++i) {
                            T s = 0;
                            for (int p = 0; p < kb; ++p)
                                s += CONJ(Akr[p + (size_t)i * lda]) * Ukr[p + (size_t)c * ldb] +
                                     CONJ(Ukr[p + (size_t)i * ldb]) * Akr[p + (size_t)c * lda];
                            if (i == c) s = MK(REAL(s), 0.0);
                            Arr[i + (size_t)c * lda] -= s;
                            if (i == c) Arr[i + (size_t)c * lda] = MK(REAL(Arr[i + (size_t)c * lda]), 0.0);
                        }
                }
                /* pass 1 == :100-101, the second half-update */
            }
            /* :103-104 */
            PFX(trsm_RUN)(kb, r, &B_(k + kb, k + kb), ldb, Akr, lda);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * larfg as the reference's kernels do it (zhetrd_gpu.F90:211-333 / :400-509,
 * dsytrd_gpu.F90:200-301): x has n entries, alpha = x[n-1]; no safe-minimum rescaling
 * loop, scaling by max(|ar|,|ai|,xnorm) only (:285-295).  On exit x[0..n-2] is the
 * scaled vector, x[n-1] = 1 (stored explicitly), *e_out = beta, *tau_out = tau.
 * Degenerate case (xnorm==0 and Im(alpha)==0): the reference only sets tau=0 and leaves
 * e and x untouched (:275-279) -- an omission; LAPACK ?latrd sets e=Re(alpha), x[n-1]=1.
 * We follow LAPACK there and document it (DESIGN.md "deviations").
 * ------------------------------------------------------------------------------------ */
void PFX(larfg_ref)(int n, T *x, double *e_out, T *tau_out) {
    T alpha = x[n - 1];
    double alphar = REAL(alpha), alphai = IMAG(alpha);
    double ss = 0.0;
    for (int i = 0; i < n - 1; ++i) ss += REAL(x[i]) * REAL(x[i]) + IMAG(x[i]) * IMAG(x[i]);
    if (ss == 0.0 && alphai == 0.0) {
        *tau_out = 0;
        *e_out = alphar;
        x[n - 1] = MK(1.0, 0.0);
        return;
    }
    double xnorm = sqrt(ss);
    double rv1 = fabs(alphar), rv2 = fabs(alphai);
    double scal = fmax(fmax(rv1, rv2), xnorm);
    double inv = 1.0 / scal;
    rv1 *= inv; rv2 *= inv; xnorm *= inv;
    double beta = -copysign(scal * sqrt(rv1 * rv1 + rv2 * rv2 + xnorm * xnorm), alphar);
    *tau_out = MK((beta - alphar) / beta, -alphai / beta);
    T s;
#ifdef ORACLE_COMPLEX
    /* inline zladiv of 1/(alpha-beta)  (:299-311) */
    double xr = alphar - beta, xi = alphai;
    if (fabs(xi) < fabs(xr)) {
        double q = xi / xr, d = 1.0 / (xr + xi * q);
        s = MK(d, -q * d);
    } else {
        double q = xr / xi, d = 1.0 / (xi + xr * q);
        s = MK(q * d, -d);
    }
#else
    s = 1.0 / (alphar - beta);
#endif
    for (int i = 0; i < n - 1; ++i) x[i] = s * x[i];
    x[n - 1] = MK(1.0, 0.0);
    *e_out = beta;
}

/* ------------------------------------------------------------------------------------
 * latrd panel, uplo='U'.  zhetrd_gpu.F90:99-165 (+ kernel math :365-378 column update,
 * :540-557 & :778-790 gemv pair, :827-877 finish W; hemv semantics zhemv_gpu.F90:102-135).
 * np = order of the active leading submatrix (the reference's N), nb columns
 * np-nb .. np-1 are reduced right to left.  W is np x nb (ldw).  e/tau are the global
 * arrays (entry i-1 written for column i).
 * ------------------------------------------------------------------------------------ */
void PFX(latrd)(int np, int nb, T *A, int lda, double *e, T *tau, T *W, int ldw) {
    if (np <= 0) return;
    for (int i = np - 1; i >= np - nb; --i) { /* 0-based column */
        int iw = i - np + nb;
        if (i < np - 1) {
            /* step 1: A(0:i,i) -= V W(i,:)^H + W V(i,:)^H   (:365-378) */
            for (int r = 0; r <= i; ++r) {
                T s = A_(r, i);
                for (int k = i + 1; k < np; ++k) {
                    int kw = k - np + nb;
                    s -= A_(r, k) * CONJ(W_(i, kw)) + W_(r, kw) * CONJ(A_(i, k));
                }
                if (r == i) s = MK(REAL(s), 0.0);
                A_(r, i) = s;
            }
        }
        if (i > 0) {
            /* step 2: larfg on A(0:i-1, i), alpha = A(i-1,i) */
            PFX(larfg_ref)(i, &A_(0, i), &e[i - 1], &tau[i - 1]);
            /* step 3: w = A(0:i-1,0:i-1) v, upper Hermitian (zhemv_gpu.F90:102-135) */
            for (int r = 0; r < i; ++r) {
                T s = 0;
                for (int c = 0; c < i; ++c) {
                    T a = (r < c) ? A_(r, c) : (r == c ? MK(REAL(A_(r, r)), 0.0) : CONJ(A_(c, r)));
                    s += a * A_(c, i);
                }
                W_(r, iw) = s;
            }
            /* step 4 (:540-557, :778-790): z1 = V^H v, z2 = W^H v; w -= W z1 + V z2 */
            for (int k = i + 1; k < np; ++k) {
                int kw = k - np + nb;
                T z1 = 0, z2 = 0;
                for (int r = 0; r < i; ++r) {
                    z1 += CONJ(A_(r, k)) * A_(r, i);
                    z2 += CONJ(W_(r, kw)) * A_(r, i);
                }
                for (int r = 0; r < i; ++r) W_(r, iw) -= W_(r, kw) * z1 + A_(r, k) * z2;
            }
            /* step 5 (:827-877): w = tau w; alpha = -1/2 tau (w^H v); w += alpha v */
            T t = tau[i - 1];
            T dot = 0;
            for (int r = 0; r < i; ++r) {
                W_(r, iw) = t * W_(r, iw);
                dot += CONJ(W_(r, iw)) * A_(r, i);
            }
            T alpha = -0.5 * t * dot;
            for (int r = 0; r < i; ++r) W_(r, iw) += alpha * A_(r, i);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * hetd2 / sytd2, uplo='U', on the leading n x n block (n <= 32 in the reference).
 * zhetd2_gpu.F90:41-188.  Unlike the blocked part the superdiagonal e(i) IS written back
 * into A and the explicit 1 is not kept (:180-184).  Only the upper triangle of A is
 * read; we write back only the upper triangle (the reference's kernel also rewrites the
 * lower part with the mirrored values, which its driver later overwrites from Z).
 * ------------------------------------------------------------------------------------ */
void PFX(hetd2)(int n, T *A, int lda, double *d, double *e, T *tau) {
    if (n <= 0) return;
    T *S = (T *)malloc(sizeof(T) * (size_t)n * n);
    T *p = (T *)malloc(sizeof(T) * (size_t)n);
#define S_(i, j) S[(i) + (size_t)(j) * n]
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) {
            S_(i, j) = A_(i, j);
            if (i < j) S_(j, i) = CONJ(A_(i, j));
        }
    S_(n - 1, n - 1) = MK(REAL(S_(n - 1, n - 1)), 0.0);
    for (int i = n - 2; i >= 0; --i) { /* reference i = n-1..1 (1-based); column i+1 here */
        double ei;
        T taui;
        /* larfg on S(0:i, i+1) with alpha = S(i, i+1) (:45-97), same formulae as above */
        T *x = &S_(0, i + 1);
        T alpha0 = x[i];
        PFX(larfg_ref)(i + 1, x, &ei, &taui);
        e[i] = ei;
        if (taui != 0) {
            /* p = taui * A(0:i,0:i) v   (:119-124) */
            for (int r = 0; r <= i; ++r) {
                T s = 0;
                for (int c = 0; c <= i; ++c) s += S_(r, c) * x[c];
                p[r] = taui * s;
            }
            /* alpha = -1/2 taui (p^H v)  (:128-167) */
            T dot = 0;
            for (int r = 0; r <= i; ++r) dot += CONJ(p[r]) * x[r];
            T al = -0.5 * taui * dot;
            for (int r = 0; r <= i; ++r) p[r] += al * x[r];
            /* rank-2 update (:171-173) */
            for (int c = 0; c <= i; ++c)
                for (int r = 0; r <= i; ++r) S_(r, c) -= x[r] * CONJ(p[c]) + p[r] * CONJ(x[c]);
        } else {
            S_(i, i) = MK(REAL(S_(i, i)), 0.0);
            (void)alpha0;
        }
        S_(i, i + 1) = MK(e[i], 0.0);
        d[i + 1] = REAL(S_(i + 1, i + 1));
        tau[i] = taui;
    }
    d[0] = REAL(S_(0, 0));
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) A_(i, j) = S_(i, j);
#undef S_
    free(S);
    free(p);
}

/* ------------------------------------------------------------------------------------
 * hetrd / sytrd blocked driver, uplo='U'.  zhetrd_gpu.F90:56-94 / dsytrd_gpu.F90:55-93:
 * panels of nb from the right while at least 32 columns remain (:60-71), a remainder
 * panel (:73-83), final min(32,N) block by hetd2 (:86-87), d(j)=A(j,j) for j>32 (:89-94;
 * e is NOT copied back into A for the blocked part, the explicit 1 stays).
 * W is N x nb workspace with ldw = N.
 * ------------------------------------------------------------------------------------ */
void PFX(hetrd)(int n, T *A, int lda, double *d, double *e, T *tau, T *W, int nb) {
    int ldw = n > 1 ? n : 1;
    int nx = 32;
    int np = n; /* order of active matrix */
    while (np - nb >= nx) {
        PFX(latrd)(np, nb, A, lda, e, tau, W, ldw);
        int m = np - nb;
        /* her2k 'U','N': A(0:m,0:m) -= V W^H + W V^H   (:67) */
        for (int c = 0; c < m; ++c)
            for (int r = 0; r <= c; ++r) {
                T s = 0;
                for (int k = 0; k < nb; ++k)
                    s += A_(r, m + k) * CONJ(W_(c, k)) + W_(r, k) * CONJ(A_(c, m + k));
                if (r == c) s = MK(REAL(s), 0.0);
                A_(r, c) -= s;
                if (r == c) A_(r, c) = MK(REAL(A_(r, c)), 0.0);
            }
        np -= nb;
    }
    int nbr = np - nx;
    if (nbr > 0) {
        PFX(latrd)(np, nbr, A, lda, e, tau, W, ldw);
        int m = np - nbr;
        for (int c = 0; c < m; ++c)
            for (int r = 0; r <= c; ++r) {
                T s = 0;
                for (int k = 0; k < nbr; ++k)
                    s += A_(r, m + k) * CONJ(W_(c, k)) + W_(r, k) * CONJ(A_(c, m + k));
                if (r == c) s = MK(REAL(s), 0.0);
                A_(r, c) -= s;
                if (r == c) A_(r, c) = MK(REAL(A_(r, c)), 0.0);
            }
        np = nx;
    }
    int n0 = n < nx ? n : nx;
    PFX(hetd2)(n0, A, lda, d, e, tau);
    for (int j = n0; j < n; ++j) d[j] = REAL(A_(j, j));
}

/* ------------------------------------------------------------------------------------
 * larft (Backward, Columnwise -> LOWER triangular T) for one block of K reflectors whose
 * vectors are the columns of V (mi x K, ldv) with the bottom K x K square treated as unit
 * UPPER triangular.  zheevd_gpu.F90:136-176 (stash/zero/unit: :154-164; herk: :170) and
 * finish_T_block_kernel :215-279 (scale by -tau: :233-246; recurrence: :248-265).
 * The masking is done on the fly here; V is not modified.
 * ------------------------------------------------------------------------------------ */
static T PFX(vmask)(const T *V, int ldv, int mi, int K, int r, int j) {
    int rr = r - (mi - K);
    if (rr == j) return MK(1.0, 0.0);
    if (rr > j) return 0;
    return V[r + (size_t)j * ldv];
}
void PFX(larft)(int mi, int K, const T *V, int ldv, const T *tau, T *Tm, int ldt) {
    /* S = V^H V, lower */
    for (int j = 0; j < K; ++j)
        for (int r = j; r < K; ++r) {
            T s = 0;
            for (int p = 0; p < mi; ++p) s += CONJ(PFX(vmask)(V, ldv, mi, K, p, r)) * PFX(vmask)(V, ldv, mi, K, p, j);
            Tm[r + (size_t)j * ldt] = s;
        }
    for (int j = 0; j < K; ++j) {
        for (int r = j + 1; r < K; ++r) Tm[r + (size_t)j * ldt] = -tau[j] * Tm[r + (size_t)j * ldt];
        Tm[j + (size_t)j * ldt] = tau[j];
    }
    T *cv = (T *)malloc(sizeof(T) * (size_t)K);
    for (int c = K - 2; c >= 0; --c) {
        for (int r = c + 1; r < K; ++r) {
            T s = 0;
            for (int j = c + 1; j <= r; ++j) s += Tm[j + (size_t)c * ldt] * Tm[r + (size_t)j * ldt];
            cv[r] = s;
        }
        for (int r = c + 1; r < K; ++r) Tm[r + (size_t)c * ldt] = cv[r];
    }
    free(cv);
}

/* larfb: C <- (I - V T V^H) C, C = mi x m.  zheevd_gpu.F90:178-213 (:193 gemm C^H V,
 * :197 trmm by T^H, :201 gemm). */
void PFX(larfb)(int mi, int m, int K, const T *V, int ldv, const T *Tm, int ldt, T *C, int ldc) {
    T *Wk = (T *)calloc((size_t)m * K, sizeof(T));
    T *W2 = (T *)calloc((size_t)m * K, sizeof(T));
    for (int j = 0; j < K; ++j)
        for (int c = 0; c < m; ++c) {
            T s = 0;
            for (int p = 0; p < mi; ++p) s += CONJ(C[p + (size_t)c * ldc]) * PFX(vmask)(V, ldv, mi, K, p, j);
            Wk[c + (size_t)j * m] = s;
        }
    /* Wk <- Wk * T^H, T lower: (Wk T^H)(c,j) = sum_{l<=j} Wk(c,l) conj(T(j,l)) */
    for (int j = 0; j < K; ++j)
        for (int c = 0; c < m; ++c) {
            T s = 0;
            for (int l = 0; l <= j; ++l) s += Wk[c + (size_t)l * m] * CONJ(Tm[j + (size_t)l * ldt]);
            W2[c + (size_t)j * m] = s;
        }
    for (int c = 0; c < m; ++c)
        for (int p = 0; p < mi; ++p) {
            T s = 0;
            for (int j = 0; j < K; ++j) s += PFX(vmask)(V, ldv, mi, K, p, j) * CONJ(W2[c + (size_t)j * m]);
            C[p + (size_t)c * ldc] -= s;
        }
    free(Wk);
    free(W2);
}

/* ------------------------------------------------------------------------------------
 * heevd / syevd:  zheevd_gpu.F90:63-131 / dsyevd_gpu.F90:63-129.
 *   trd (nb1=32) -> tridiagonal eigensolver (reference: host LAPACK zstedc/dstedc 'I',
 *   :101, third-party; restated here by the implicit-QL routine oracle_steql, any
 *   accurate tridiagonal solver gives the same eigenpairs up to sign/rounding)
 *   -> Z(:,0:m) = Q(:, il-1 : iu)   (:110; the real reference copies from column 1
 *   regardless of il, dsyevd_gpu.F90:108 -- we honour il in both)
 *   -> back-transform in blocks of nb2 = min(64,N) ascending (:121-130).
 * w receives all N eigenvalues ascending (:111).  Returns 0 or -1 (tridiagonal failure).
 * A's upper triangle holds the reflectors on exit.
 * ------------------------------------------------------------------------------------ */
int PFX(heevd)(int n, int il, int iu, T *A, int lda, T *Z, int ldz, double *w, int nb1, int nb2) {
    int m = iu - il + 1;
    double *d = (double *)calloc((size_t)n, sizeof(double));
    double *e = (double *)calloc((size_t)n, sizeof(double));
    T *tau = (T *)calloc((size_t)n, sizeof(T));
    T *Wp = (T *)calloc((size_t)n * (nb1 > 0 ? nb1 : 1), sizeof(T));
    PFX(hetrd)(n, A, lda, d, e, tau, Wp, nb1);
    double *Q = (double *)calloc((size_t)n * n, sizeof(double));
    int info = oracle_steql(n, d, e, Q, n);
    if (info == 0) {
        for (int j = 0; j < n; ++j) w[j] = d[j];
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < n; ++i) Z_(i, j) = MK(Q[i + (size_t)(il - 1 + j) * n], 0.0);
        int k = n - 1;
        if (nb2 > n) nb2 = n;
        T *Tm = (T *)calloc((size_t)nb2 * nb2, sizeof(T));
        for (int i = 0; i < k; i += nb2) { /* 0-based first reflector of the block */
            int ib = (k - i < nb2) ? k - i : nb2;
            int mi = i + ib;
            const T *V = &A_(0, i + 1);
            PFX(larft)(mi, ib, V, lda, &tau[i], Tm, nb2);
            PFX(larfb)(mi, m, ib, V, lda, Tm, nb2, Z, ldz);
        }
        free(Tm);
    }
    free(d); free(e); free(tau); free(Wp); free(Q);
    return info ? -1 : 0;
}

/* ------------------------------------------------------------------------------------
 * hegvdx / sygvdx driver.  zhegvdx_gpu.F90:129-180 / dsygvdx_gpu.F90:115-166:
 * potrf(B) -> (save lower(A) in Z; not needed here beyond preserving it) -> hegst(nb=448)
 * -> heevd -> Z <- U^{-1} Z.  On exit: B = U, A upper destroyed, strict lower(A)
 * preserved, w = all N eigenvalues, Z(:,0:m) eigenvectors il..iu.  info 0 / -1.
 * ------------------------------------------------------------------------------------ */
int PFX(hegvdx)(int n, T *A, int lda, T *B, int ldb, T *Z, int ldz, int il, int iu, double *w) {
    if (PFX(potrf_upper)(n, B, ldb) != 0) return -1;
    /* :144-152 save strict lower(A) */
    T *L = (T *)malloc(sizeof(T) * (size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) L[i + (size_t)j * n] = A_(i, j);
    PFX(hegst)(n, A, lda, B, ldb, 448);
    int info = PFX(heevd)(n, il, iu, A, lda, Z, ldz, w, 32, 64);
    /* zheevd_gpu.F90:88-96 restore strict lower(A) */
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) A_(i, j) = L[i + (size_t)j * n];
    free(L);
    if (info) return -1;
    PFX(trsm_LUN)(n, iu - il + 1, B, ldb, Z, ldz);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * compare (test_driver/toolbox.F90:36-176): relative l2 error and max % error; for the
 * 2-D variants on ABSOLUTE values of the entries (sign/phase insensitive, :101-103,
 * :150-152); entries with |ref| < 1e-10 are skipped (:53).
 * out[0] = l2 relative error, out[1] = max percent error.
 * ------------------------------------------------------------------------------------ */
void PFX(compare_abs2d)(int n, int m, const T *R, int ldr, const T *G, int ldg, double *out) {
    double l2 = 0, nrm = 0, mx = 0;
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
#ifdef ORACLE_COMPLEX
            double a = cabs(R[i + (size_t)j * ldr]), b = cabs(G[i + (size_t)j * ldg]);
#else
            double a = fabs(R[i + (size_t)j * ldr]), b = fabs(G[i + (size_t)j * ldg]);
#endif
            if (a >= 1e-10) {
                double perr = fabs(a - b) / a * 100.0;
                nrm += a * a;
                l2 += (a - b) * (a - b);
                if (perr > mx && a != 0.0 && b != 0.0) mx = perr;
            }
        }
    nrm = sqrt(nrm);
    l2 = sqrt(l2);
    out[0] = (l2 != 0.0) ? l2 / nrm : 0.0;
    out[1] = mx;
}

#undef T
#undef PFX
#undef CONJ
#undef REAL
#undef IMAG
#undef MK
#undef A_
#undef B_
#undef W_
#undef Z_
