// trd.hip -- blocked Householder tridiagonalization (uplo='U') for gfx950.
//
// Replaces zhetrd_gpu / dsytrd_gpu + zlatrd_gpu / dlatrd_gpu and their eight CUDA-Fortran
// kernels (zhetrd_gpu.F90:30-879, dsytrd_gpu.F90:30-725, zhemv_gpu.F90, dsymv_gpu.F90,
// zhetd2_gpu.F90, dsytd2_gpu.F90).  Same mathematics (LAPACK ?latrd 'U' / ?hetd2 'U', same
// reflector/tau/e conventions, explicit 1 left at A(i-1,i), e not copied back for the
// blocked part), different decomposition, designed for wave64 / 256 CUs / deterministic sums:
//
//   per column i of a panel the reference launches 4 kernels that communicate through fp64
//   atomics and a "last block done" counter; here a column is exactly TWO kernels and no
//   atomics at all (bit-reproducible results):
//
//   panel_row_kernel (one row per lane, 64 rows per workgroup, the 4 waves split the
//     pending panel columns):   finishes W(:,c) of the previous column c=i+1 from the hemv
//     partials, applies the pending rank-2(np-1-i) update to column i, and emits per-workgroup
//     partial sums of ||A(0:i-2,i)||^2 -- i.e. zher2_mv + the tail of stacked_zgemv_N_finish_W.
//   panel_mv_kernel: every workgroup re-derives (beta, tau, 1/(alpha-beta)) from those partials
//     (larfg without a grid sync), scales v on the fly, and does the Hermitian mat-vec
//     w = A v reading each upper-triangle element ONCE (HBM-bound, the dominant bytes of the
//     whole solver) plus the stacked V^H v / W^H v products.
//     The scalar alpha = -1/2 tau (w^H v) needs no extra global reduction:
//         w^H v = conj(tau) * conj( v^H A v - 2 Re(z1^H z2) ),
//     and v^H A v is accumulated tile by tile inside the mat-vec.
#include "trd.h"

namespace eig {

#ifndef EIG_MV_PREFETCH
#define EIG_MV_PREFETCH 1
#endif
constexpr int HT = 64;     // hemv tile: 64 rows x 64 cols per workgroup step (one row per lane)
constexpr int CH = 512;    // rows per gemv partial chunk (8 rows per lane)
constexpr int NBMAX = 64;  // maximum panel width

template <class T> struct PanelArgs {
    T* A; int lda;
    T* W; int ldw;     // panel W (np x nb), column (k - wbase) belongs to matrix column k
    int np, nb, i;
    double* e; T* tau;
    T* xbuf;           // updated, unscaled column i (length >= np)
    T* P; int ldp;     // hemv partials P[q*ldp + row]
    T* S;              // per-hemv-workgroup partial of v^H A v
    T* Zp;             // gemv partials Zp[(chunk*2 + which)*NBMAX + kk]
    double* NP;        // per-row-workgroup partial of the squared norm
    T* alphaSlot;      // A(i-1,i) after the update
    int nblkA;         // row-kernel workgroups that produced NP for column i
    int gh;            // hemv workgroups used for the column being finished / generated
    int nchunk;        // gemv row chunks for that column
    int ablate;        // debug/timing only (EIGSOLVE_ABLATE): skips parts of the row kernel, results invalid
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ cplx wave_sum(cplx v) { return cplx{wave_sum(v.x), wave_sum(v.y)}; }
__device__ __forceinline__ double shx(double v, int o) { return __shfl_xor(v, o); }
__device__ __forceinline__ cplx shx(cplx v, int o) { return cplx{__shfl_xor(v.x, o), __shfl_xor(v.y, o)}; }

// larfg scalars exactly as the reference computes them (zhetrd_gpu.F90:275-311): scaling by
// max(|ar|,|ai|,xnorm), no safe-minimum loop.  Degenerate case follows LAPACK (tau=0, beta=ar).
template <class T> __device__ void larfg_scalars(double ss, T alpha, double& beta, T& tau, T& scale) {
    double ar = real_(alpha), ai = imag_(alpha);
    if (ss == 0.0 && ai == 0.0) {
        beta = ar;
        tau = Tr<T>::zero();
        scale = Tr<T>::zero();
        return;
    }
    double xnorm = sqrt(ss);
    double rv1 = fabs(ar), rv2 = fabs(ai);
    double scal = fmax(fmax(rv1, rv2), xnorm);
    double inv = 1.0 / scal;
    rv1 *= inv; rv2 *= inv; xnorm *= inv;
    beta = -copysign(scal * sqrt(rv1 * rv1 + rv2 * rv2 + xnorm * xnorm), ar);
    tau = Tr<T>::make((beta - ar) / beta, -ai / beta);
    if constexpr (Tr<T>::cx) {
        double xr = ar - beta, xi = ai;
        if (fabs(xi) < fabs(xr)) {
            double q = xi / xr, d = 1.0 / (xr + xi * q);
            scale = cplx{d, -q * d};
        } else {
            double q = xr / xi, d = 1.0 / (xi + xr * q);
            scale = cplx{q * d, -d};
        }
    } else {
        scale = 1.0 / (ar - beta);
    }
}

// ------------------------------------------------------------------------------------------
// panel_row_kernel : 16 rows x 16 column-groups per workgroup.  Every global load the kernel
// needs (gemv partials, v^H A v partials, the row-i data for w_i, this thread's slice of the V/W
// panel and of the hemv partials) is issued before the first barrier, so the kernel costs one
// memory round trip plus a handful of LDS reductions instead of a chain of dependent loads.
// ------------------------------------------------------------------------------------------
constexpr int RR = 16;   // rows per workgroup
constexpr int RG = 16;   // column groups per workgroup
constexpr int RU = 4;    // panel columns per group (NBMAX / RG)
constexpr int RP = 8;    // hemv stripes per group (supports N <= RP*RG*HT = 8192; larger loops)

template <class T>
__global__ void __launch_bounds__(256) panel_row_kernel(PanelArgs<T> a, int do_finish, int do_update) {
    const int i = a.i, c = i + 1;
    const int npo = do_finish ? a.np - 1 - c : 0;  // columns older than c inside the panel
    const int wbase = a.np - a.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rl = tid & (RR - 1), g = tid >> 4;
    const int r = blockIdx.x * RR + rl;
    const int rows = i + 1;
    const bool active = r < rows;
    const int ntc = (c + HT - 1) / HT;  // hemv stripes of the column being finished (n = c)

    __shared__ T z1s[NBMAX], z2s[NBMAX], rowW[NBMAX + 1], rowV[NBMAX + 1];
    __shared__ T red2[RG][RR], red3[RG][RR];
    __shared__ T s4[4];
    __shared__ T sc_tau, sc_alpha;

    // ---------------- phase 0: issue every load ----------------
    T zs = Tr<T>::zero(), Ssum = Tr<T>::zero();
    T wi_w = Tr<T>::zero(), wi_v = Tr<T>::zero(), wi_p = Tr<T>::zero();
    T vv[RU], wv[RU], pp[RP];
    T vr = Tr<T>::zero(), acur = Tr<T>::zero(), vic = Tr<T>::zero();
    if (do_finish && do_update && wave == 0) vic = a.A[(size_t)i + (size_t)c * a.lda];   // needed in phase 2: issue now
    T tau_early = Tr<T>::zero();
    if (do_finish && tid == 0) tau_early = a.tau[c - 1];
#pragma unroll
    for (int u = 0; u < RU; ++u) { vv[u] = Tr<T>::zero(); wv[u] = Tr<T>::zero(); }
#pragma unroll
    for (int u = 0; u < RP; ++u) pp[u] = Tr<T>::zero();
    if (do_finish) {
        // independent, branch-free loads (a `for (...) s += x[q]` loop is a chain of dependent round trips)
        if (tid < 2 * NBMAX) {
            int which = tid >> 6, kk = tid & 63;
            if (kk < npo) {
                T zt[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int ch = min(u, a.nchunk - 1);
                    T t = a.Zp[(size_t)(ch * 2 + which) * NBMAX + kk];
                    zt[u] = (u < a.nchunk) ? t : Tr<T>::zero();
                }
                for (int ch = 8; ch < a.nchunk; ++ch) zs = zs + a.Zp[(size_t)(ch * 2 + which) * NBMAX + kk];
#pragma unroll
                for (int u = 0; u < 8; ++u) zs = zs + zt[u];
            }
        }
        {
            T st4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int q = tid + 256 * u;
                T t = a.S[min(q, a.gh - 1)];
                st4[u] = (q < a.gh) ? t : Tr<T>::zero();
            }
            for (int q = tid + 1024; q < a.gh; q += 256) Ssum = Ssum + a.S[q];
            Ssum = Ssum + ((st4[0] + st4[1]) + (st4[2] + st4[3]));
        }
        if (do_update && wave == 0) {
            if (lane < npo) {
                int k = c + 1 + lane;
                wi_w = a.W[(size_t)i + (size_t)(k - wbase) * a.ldw];
                wi_v = a.A[(size_t)i + (size_t)k * a.lda];
            }
            T p0 = a.P[(size_t)min(lane, ntc - 1) * a.ldp + i];
            T p1 = a.P[(size_t)min(lane + 64, ntc - 1) * a.ldp + i];
            wi_p = ((lane < ntc) ? p0 : Tr<T>::zero()) + ((lane + 64 < ntc) ? p1 : Tr<T>::zero());
            for (int q = lane + 128; q < ntc; q += 64) wi_p = wi_p + a.P[(size_t)q * a.ldp + i];
        }
        if (active) {
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                int kk = g + RG * u;
                if (kk < npo && !(a.ablate & 1)) {
                    int k = c + 1 + kk;
                    vv[u] = a.A[(size_t)r + (size_t)k * a.lda];
                    wv[u] = a.W[(size_t)r + (size_t)(k - wbase) * a.ldw];
                }
            }
#pragma unroll
            for (int u = 0; u < RP; ++u) {
                int q = g + RG * u;
                if (q < ntc && !(a.ablate & 2)) pp[u] = a.P[(size_t)q * a.ldp + r];
            }
            for (int q = g + RG * RP; q < ntc; q += RG) pp[0] = pp[0] + a.P[(size_t)q * a.ldp + r];
            if (g == 0) vr = a.A[(size_t)r + (size_t)c * a.lda];
        }
    }
    if (do_update && active && g == 0) acur = a.A[(size_t)r + (size_t)i * a.lda];

    if (do_finish) {
        // ---------------- phase 1: gemv sums, v^H A v ----------------
        if (tid < 2 * NBMAX) {
            int which = tid >> 6, kk = tid & 63;
            if (kk < npo) { if (which == 0) z1s[kk] = zs; else z2s[kk] = zs; }
        }
        Ssum = wave_sum(Ssum);
        if (lane == 0) s4[wave] = Ssum;
        if (tid == 0) sc_tau = tau_early;
        __syncthreads();
        // ---------------- phase 2: alpha, w_i ----------------
        if (wave == 0) {
            double zz = 0.0;
            T u = wi_p;
            if (lane < npo) {
                T t = Tr<T>::zero();
                fmac_(t, z1s[lane], z2s[lane]);
                zz = real_(t);
                u = u - (wi_w * z1s[lane] + wi_v * z2s[lane]);
                rowW[lane] = conj_(wi_w);
                rowV[lane] = conj_(wi_v);
            }
            zz = wave_sum(zz);
            if (do_update) u = wave_sum(u);
            if (lane == 0) {
                T tau = sc_tau;
                T S = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                // alpha = -1/2 tau (w'^H v),  w'^H v = conj(tau) conj(S - 2 Re(z1^H z2))
                T alpha = (-0.5 * abs2_(tau)) * conj_(S - Tr<T>::make(2.0 * zz, 0.0));
                sc_alpha = alpha;
                if (do_update) {
                    T wi = tau * u + alpha * vic;
                    rowW[npo] = conj_(wi);
                    rowV[npo] = conj_(vic);
                }
            }
        }
        __syncthreads();
    }

    // ---------------- phase 3: this thread's slice ----------------
    T acc2 = Tr<T>::zero(), acc3 = Tr<T>::zero();
    if (do_finish && active) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            int kk = g + RG * u;
            if (kk < npo) {
                acc2 = acc2 - (wv[u] * z1s[kk] + vv[u] * z2s[kk]);
                if (do_update) acc3 = acc3 + (vv[u] * rowW[kk] + wv[u] * rowV[kk]);
            }
        }
#pragma unroll
        for (int u = 0; u < RP; ++u) acc2 = acc2 + pp[u];
    }
    red2[g][rl] = acc2;
    red3[g][rl] = acc3;
    __syncthreads();
    // ---------------- phase 4: one thread per row finishes ----------------
    if (tid < RR) {
        double contrib = 0.0;
        if (active) {
            T anew = acur;
            if (do_finish) {
                T u = Tr<T>::zero(), upd = Tr<T>::zero();
#pragma unroll
                for (int q = 0; q < RG; ++q) { u = u + red2[q][rl]; upd = upd + red3[q][rl]; }
                T wr = sc_tau * u + sc_alpha * vr;
                a.W[(size_t)r + (size_t)(c - wbase) * a.ldw] = wr;
                if (do_update) {
                    upd = upd + (vr * rowW[npo] + wr * rowV[npo]);
                    anew = acur - upd;
                    if (r == i) anew = Tr<T>::realpart(anew);
                    a.A[(size_t)r + (size_t)i * a.lda] = anew;
                }
            }
            if (do_update) {
                a.xbuf[r] = anew;
                if (r <= i - 2) contrib = abs2_(anew);
                if (r == i - 1) *a.alphaSlot = anew;
            }
        }
        if (do_update) {
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) contrib += __shfl_xor(contrib, o);
            if (tid == 0) a.NP[blockIdx.x] = contrib;
        }
    }
}

// ------------------------------------------------------------------------------------------
// panel_mv_kernel : larfg scalars + Hermitian mat-vec (upper, each element read once) + stacked
// conjugate-transposed panel products.  Also used stand-alone (bench / zhemv entry point) with
// `plain` != 0: v = xbuf as is, no scalars, no gemv part.
// Grid = [gemv workgroups | hemv workgroups]: the (short, latency-bound) gemv items start first
// and hide under the bandwidth-bound tiles.  Loads are issued before the scalar prologue.
// ------------------------------------------------------------------------------------------
template <class T, int NCOL>
__device__ __forceinline__ void transpose_reduce16(T (&t)[NCOL], int lane, T& out) {
    // 16 per-lane column partials -> every 4-lane quad ends with the full 64-lane sum of column (lane>>2)&15
    static_assert(NCOL == 16, "");
    T u8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bool hi = lane & 32;
        T send = hi ? t[j] : t[j + 8];
        T keep = hi ? t[j + 8] : t[j];
        u8[j] = keep + shx(send, 32);
    }
    T u4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bool hi = lane & 16;
        T send = hi ? u8[j] : u8[j + 4];
        T keep = hi ? u8[j + 4] : u8[j];
        u4[j] = keep + shx(send, 16);
    }
    T u2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bool hi = lane & 8;
        T send = hi ? u4[j] : u4[j + 2];
        T keep = hi ? u4[j + 2] : u4[j];
        u2[j] = keep + shx(send, 8);
    }
    {
        bool hi = lane & 4;
        T send = hi ? u2[0] : u2[1];
        T keep = hi ? u2[1] : u2[0];
        out = keep + shx(send, 4);
    }
    out = out + shx(out, 2);
    out = out + shx(out, 1);
}

__device__ __forceinline__ void tile_decode(int t, int& I, int& J) {
    // t = J(J+1)/2 + I, I <= J
    J = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((J + 1) * (J + 2) / 2 <= t) ++J;
    while (J * (J + 1) / 2 > t) --J;
    I = t - J * (J + 1) / 2;
}

template <class T>
__global__ void __launch_bounds__(256) panel_mv_kernel(PanelArgs<T> a, int plain, int gg) {
    const int i = a.i, n = i;  // v has n entries (rows 0..i-1), v(n-1) = 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ T redy[4][64];
    __shared__ T redt[64];
    __shared__ T sc_scale;
    const bool is_gemv = (int)blockIdx.x < gg;
    const int hb = (int)blockIdx.x - gg;  // hemv workgroup index

    // raw (unscaled) entry of the column; the scale is applied after the prologue
    auto xraw = [&](int r) -> T { return (r < n) ? a.xbuf[r] : Tr<T>::zero(); };
    auto vfix = [&](T x, int r, T scale) -> T {
        if (plain) return x;
        if (r < n - 1) return scale * x;
        return (r == n - 1) ? Tr<T>::one() : Tr<T>::zero();
    };

    // ---------------- issue the first batch of loads ----------------
    T alpha_early = Tr<T>::zero();
    if (!plain && wave == 0) alpha_early = *a.alphaSlot;   // consumed by the scalar prologue: issue with everything else
    const int nt = (n + HT - 1) / HT;
    const int ntiles = nt * (nt + 1) / 2;
    __shared__ T xcs[2][HT];   // raw column entries of v for the current / next tile
    T av[16];
    T xr = Tr<T>::zero();
    int I = 0, J = 0, t = hb, buf = 0;
    // gemv item
    int g_which = 0, g_kk = 0, g_rbeg = 0, g_rend = 0, g_ch = 0;
    bool g_ok = false;
    const int wbase = a.np - a.nb;
    auto load_tile = [&](int b) {
        tile_decode(t, I, J);
        const int r0 = I * HT, c0 = J * HT, r = r0 + lane;
        const bool diag = (I == J);
        xr = xraw(r);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int cc = c0 + wave * 16 + j;
            bool ok = (r < n) && (cc < n) && (!diag || r <= cc);
            T v = Tr<T>::zero();
            if (ok) v = a.A[(size_t)r + (size_t)cc * a.lda];
            if (diag && r == cc) v = Tr<T>::realpart(v);
            av[j] = v;
        }
        if (tid < HT) xcs[b][tid] = xraw(c0 + tid);
    };
    if (is_gemv) {
        const int npo = a.np - 1 - i;
        const int item = (int)blockIdx.x * 4 + wave;
        g_ok = item < 2 * npo * a.nchunk;
        if (g_ok) {
            g_ch = item / (2 * npo);
            int rem = item % (2 * npo);
            g_which = rem / npo; g_kk = rem % npo;
            int k = i + 1 + g_kk;
            const T* src = g_which == 0 ? a.A + (size_t)k * a.lda : a.W + (size_t)(k - wbase) * a.ldw;
            g_rbeg = g_ch * CH; g_rend = min(n, g_rbeg + CH);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int r = g_rbeg + lane + 64 * j;
                av[j] = (r < g_rend) ? src[r] : Tr<T>::zero();
                av[8 + j] = (r < g_rend) ? a.xbuf[r] : Tr<T>::zero();
            }
        }
    } else if (t < ntiles) {
        load_tile(0);
    }

    // ---------------- scalar prologue (every workgroup, deterministic) ----------------
    if (!plain) {
        if (wave == 0) {
            double ss = 0.0, np4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int q = lane + 64 * u;
                double t = a.NP[min(q, a.nblkA - 1)];
                np4[u] = (q < a.nblkA) ? t : 0.0;
            }
            for (int q = lane + 256; q < a.nblkA; q += 64) ss += a.NP[q];
            ss += (np4[0] + np4[1]) + (np4[2] + np4[3]);
            ss = wave_sum(ss);
            if (lane == 0) {
                double beta;
                T tau, scale;
                larfg_scalars<T>(ss, alpha_early, beta, tau, scale);
                sc_scale = scale;
                if (blockIdx.x == 0) {
                    a.e[i - 1] = beta;
                    a.tau[i - 1] = tau;
                }
            }
        }
    }
    __syncthreads();
    const T scale = plain ? Tr<T>::one() : sc_scale;

    if (is_gemv) {
        // stacked conjugate-transposed products z1 = V^H v, z2 = W^H v (partials per row chunk)
        if (g_ok) {
            T s = Tr<T>::zero();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int r = g_rbeg + lane + 64 * j;
                fmac_(s, av[j], vfix(av[8 + j], r, scale));
            }
            s = wave_sum(s);
            if (lane == 0) a.Zp[(size_t)(g_ch * 2 + g_which) * NBMAX + g_kk] = s;
        }
        return;
    }

    T Sacc = Tr<T>::zero();
    while (t < ntiles) {
        const int r0 = I * HT, c0 = J * HT;
        const bool diag = (I == J);
        const int r = r0 + lane;
        const T vr = vfix(xr, r, scale);
        T yI = Tr<T>::zero();
        T tj[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int cc = c0 + wave * 16 + j;
            T vc = vfix(xcs[buf][wave * 16 + j], cc, scale);
            fma_(yI, av[j], vc);
            T p = Tr<T>::zero();
            if (!(diag && r == cc)) fmac_(p, av[j], vr);
            tj[j] = p;
        }
        T tval;
        transpose_reduce16<T, 16>(tj, lane, tval);
        redy[wave][lane] = yI;
        if ((lane & 3) == 0) redt[wave * 16 + (lane >> 2)] = tval;
        // own row/column entries of v for the S partial (tid < 64)
        T vI = Tr<T>::zero(), vJ = Tr<T>::zero();
        if (tid < 64) { vI = vfix(xraw(r0 + tid), r0 + tid, scale); vJ = vfix(xcs[buf][tid], c0 + tid, scale); }
        const int Ic = I, Jc = J;
        t += a.gh;
#if EIG_MV_PREFETCH
        if (t < ntiles) load_tile(buf ^ 1);   // next tile's loads fly while the barriers drain
#endif
        __syncthreads();
        if (tid < 64) {
            T yv = (redy[0][tid] + redy[1][tid]) + (redy[2][tid] + redy[3][tid]);
            T tv = redt[tid];
            if (diag) {
                T s = yv + tv;
                a.P[(size_t)Jc * a.ldp + r0 + tid] = s;
                fmac_(Sacc, vI, s);
                if (!plain && c0 + tid < n) a.A[(size_t)(c0 + tid) + (size_t)i * a.lda] = vJ;
            } else {
                a.P[(size_t)Jc * a.ldp + r0 + tid] = yv;
                a.P[(size_t)Ic * a.ldp + c0 + tid] = tv;
                fmac_(Sacc, vI, yv);
                fmac_(Sacc, vJ, tv);
            }
        }
        buf ^= 1;
#if !EIG_MV_PREFETCH
        if (t < ntiles) load_tile(buf);
#endif
        __syncthreads();
    }
    if (wave == 0) {
        Sacc = wave_sum(Sacc);
        if (lane == 0) a.S[hb] = Sacc;
    }
}

// y = sum of the hemv partials (stand-alone hemv entry point only)
template <class T> __global__ void __launch_bounds__(256) hemv_gather_kernel(int n, int nt, const T* P, int ldp, T* y) {
    int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    T s = Tr<T>::zero();
    for (int q = 0; q < nt; ++q) s = s + P[(size_t)q * ldp + r];
    y[r] = s;
}

// ------------------------------------------------------------------------------------------
// final <=32 x 32 block: unblocked ?hetd2 'U' in LDS (zhetd2_gpu.F90:41-188).  Writes d, e, tau and
// the upper triangle of the block (superdiagonal = e, as the reference does for this block).
// ------------------------------------------------------------------------------------------
constexpr int TD = 32;
template <class T> __global__ void __launch_bounds__(256) hetd2_kernel(int n, T* A, int lda, double* d, double* e, T* tau) {
    __shared__ T s[TD][TD + 1];  // s[c][r]
    __shared__ T p[TD];
    __shared__ T sc_tau, sc_scale, sc_al;
    __shared__ double sc_beta;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int el = tid; el < TD * TD; el += 256) {
        int r = el % TD, cc = el / TD;
        T v = Tr<T>::zero();
        if (r < n && cc < n) {
            if (r < cc) v = A[(size_t)r + (size_t)cc * lda];
            else if (r == cc) v = A[(size_t)r + (size_t)cc * lda];
            else v = conj_(A[(size_t)cc + (size_t)r * lda]);
        }
        s[cc][r] = v;
    }
    __syncthreads();
    if (tid == 0 && n > 0) s[n - 1][n - 1] = Tr<T>::realpart(s[n - 1][n - 1]);
    __syncthreads();
    for (int i = n - 2; i >= 0; --i) {
        // reflector from x = s(0:i, i+1), alpha = x(i)
        if (wave == 0) {
            double w = (lane < i) ? abs2_(s[i + 1][lane]) : 0.0;
            w = wave_sum(w);
            if (lane == 0) {
                double beta;
                T t, sc;
                larfg_scalars<T>(w, s[i + 1][i], beta, t, sc);
                sc_beta = beta; sc_tau = t; sc_scale = sc;
            }
        }
        __syncthreads();
        const T taui = sc_tau;
        if (tid < i) s[i + 1][tid] = sc_scale * s[i + 1][tid];
        if (tid == i) s[i + 1][i] = Tr<T>::one();
        if (tid == 0) e[i] = sc_beta;
        __syncthreads();
        const bool nz = !(real_(taui) == 0.0 && imag_(taui) == 0.0);
        if (nz) {
            if (tid <= i) {
                T acc = Tr<T>::zero();
                for (int cc = 0; cc <= i; ++cc) fma_(acc, s[cc][tid], s[i + 1][cc]);
                p[tid] = taui * acc;
            }
            __syncthreads();
            if (wave == 0) {
                T dd = Tr<T>::zero();
                if (lane <= i) fmac_(dd, p[lane], s[i + 1][lane]);
                dd = wave_sum(dd);
                if (lane == 0) sc_al = (-0.5 * taui) * dd;
            }
            __syncthreads();
            if (tid <= i) p[tid] = p[tid] + sc_al * s[i + 1][tid];
            __syncthreads();
            for (int el = tid; el < (i + 1) * (i + 1); el += 256) {
                int r = el % (i + 1), cc = el / (i + 1);
                T xr = s[i + 1][r], xc = s[i + 1][cc];
                s[cc][r] = s[cc][r] - (xr * conj_(p[cc]) + p[r] * conj_(xc));
            }
            __syncthreads();
        } else {
            if (tid == 0) s[i][i] = Tr<T>::realpart(s[i][i]);
        }
        if (tid == 0) {
            s[i + 1][i] = Tr<T>::make(sc_beta, 0.0);
            d[i + 1] = real_(s[i + 1][i + 1]);
            tau[i] = taui;
        }
        __syncthreads();
    }
    if (tid == 0 && n > 0) d[0] = real_(s[0][0]);
    __syncthreads();
    for (int el = tid; el < TD * TD; el += 256) {
        int r = el % TD, cc = el / TD;
        if (r < n && cc < n && r <= cc) A[(size_t)r + (size_t)cc * lda] = s[cc][r];
    }
}

template <class T> __global__ void __launch_bounds__(256) diag_extract_kernel(int n0, int n, const T* A, int lda, double* d) {
    int j = n0 + blockIdx.x * 256 + threadIdx.x;
    if (j < n) d[j] = real_(A[(size_t)j + (size_t)j * lda]);  // zhetrd_gpu.F90:89-94
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int hemv_grid(const Ctx& c, int n) {
    int nt = (n + HT - 1) / HT;
    long ntiles = (long)nt * (nt + 1) / 2;
    long cap = c.hemv_blocks > 0 ? c.hemv_blocks : 2L * c.n_cu;  // = resident workgroups (253 VGPRs -> 2 per CU): one wave of blocks, no tail
    return (int)(ntiles < cap ? ntiles : cap);
}

template <class T> struct TrdScratch {
    T *xbuf, *P, *S, *Zp, *alphaSlot;
    double* NP;
    int ldp;
};

template <class T> static TrdScratch<T> trd_scratch(Ctx& c, int N) {
    TrdScratch<T> s;
    int nt = (N + HT - 1) / HT;
    s.ldp = nt * HT;
    s.xbuf = c.scratch<T>("trd_xbuf", (size_t)nt * HT + 64);
    s.P = c.scratch<T>("trd_P", (size_t)nt * s.ldp);
    s.S = c.scratch<T>("trd_S", 8192);
    int nchunk = (N + CH - 1) / CH;
    s.Zp = c.scratch<T>("trd_Zp", (size_t)(nchunk + 1) * 2 * NBMAX);
    s.NP = c.scratch<double>("trd_NP", (size_t)(N / RR) + 64);
    s.alphaSlot = c.scratch<T>("trd_alpha", 8);
    return s;
}

template <class T>
static void latrd_panel(Ctx& c, hipStream_t st, const TrdScratch<T>& sc, int np, int nb, T* A, int lda, double* e, T* tau,
                        T* W, int ldw, bool mv_only = false, long* nlaunch = nullptr, double* algo_bytes = nullptr) {
    PanelArgs<T> a;
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.np = np; a.nb = nb; a.e = e; a.tau = tau;
    a.xbuf = sc.xbuf; a.P = sc.P; a.ldp = sc.ldp; a.S = sc.S; a.Zp = sc.Zp; a.NP = sc.NP; a.alphaSlot = sc.alphaSlot;
    static const int ablate_env = getenv("EIGSOLVE_ABLATE") ? atoi(getenv("EIGSOLVE_ABLATE")) : 0;
    a.ablate = ablate_env;
    int gh_prev = 0, nchunk_prev = 0;
    for (int i = np - 1; i >= np - nb - 1; --i) {
        const bool last = (i == np - nb - 1);  // finish-only pass for the panel's leftmost column
        const int do_finish = (i < np - 1), do_update = !last;
        a.i = i;
        a.gh = gh_prev; a.nchunk = nchunk_prev;
        int gA = (i + 1 + RR - 1) / RR;
        if (!mv_only) hipLaunchKernelGGL((panel_row_kernel<T>), dim3(gA), dim3(256), 0, st, a, do_finish, do_update);
        if (last) break;
        // mat-vec for column i (v has i entries)
        int n = i;
        int gh = hemv_grid(c, n);
        int nchunk = (n + CH - 1) / CH;
        int npo = np - 1 - i;
        int gg = (2 * npo * nchunk + 3) / 4;
        a.nblkA = gA; a.gh = gh; a.nchunk = nchunk;
        hipLaunchKernelGGL((panel_mv_kernel<T>), dim3(gh + gg), dim3(256), 0, st, a, 0, gg);
        if (nlaunch) ++*nlaunch;
        if (algo_bytes) *algo_bytes += (double)sizeof(T) * (double)n * (double)(n + 1) * 0.5;
        gh_prev = gh; nchunk_prev = nchunk;
    }
    EIG_HIP(hipGetLastError());
}

template <class T>
void hetrd_upper(Ctx& c, hipStream_t st, int N, T* A, int lda, double* d, double* e, T* tau, T* W, int nb) {
    if (N <= 0) return;
    if (nb <= 0 || nb > NBMAX) nb = NBMAX;
    const int nx = TD;
    TrdScratch<T> sc = trd_scratch<T>(c, N);
    const int ldw = N;
    int np = N;
    while (np - nb >= nx) {  // zhetrd_gpu.F90:60-71
        latrd_panel(c, st, sc, np, nb, A, lda, e, tau, W, ldw);
        her2k_un<T>(c, st, np - nb, nb, A + (size_t)(np - nb) * lda, lda, W, ldw, A, lda);
        np -= nb;
    }
    int nbr = np - nx;  // remainder panel, :73-83
    if (nbr > 0) {
        latrd_panel(c, st, sc, np, nbr, A, lda, e, tau, W, ldw);
        her2k_un<T>(c, st, np - nbr, nbr, A + (size_t)(np - nbr) * lda, lda, W, ldw, A, lda);
        np = nx;
    }
    int n0 = N < nx ? N : nx;
    hipLaunchKernelGGL((hetd2_kernel<T>), dim3(1), dim3(256), 0, st, n0, A, lda, d, e, tau);
    if (N > n0)
        hipLaunchKernelGGL((diag_extract_kernel<T>), dim3((N - n0 + 255) / 256), dim3(256), 0, st, n0, N, (const T*)A, lda, d);
    EIG_HIP(hipGetLastError());
}

// Roofline leg: the exact sequence of panel_mv_kernel launches of a full tridiagonalization
// (same grids, same panel state layout, row kernels and her2k skipped; A is left numerically
// meaningless).  Returns launches and algorithmic bytes sum_n s*n(n+1)/2.
template <class T>
void hetrd_mv_sweep(Ctx& c, hipStream_t st, int N, T* A, int lda, T* W, int nb, double* e, T* tau, long* nlaunch, double* algo_bytes) {
    if (nb <= 0 || nb > NBMAX) nb = NBMAX;
    TrdScratch<T> sc = trd_scratch<T>(c, N);
    std::vector<double> ones((size_t)(N / RR) + 64, 1.0);
    EIG_HIP(hipMemcpyAsync(sc.NP, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice, st));
    T one = Tr<T>::one();
    EIG_HIP(hipMemcpyAsync(sc.alphaSlot, &one, sizeof(T), hipMemcpyHostToDevice, st));
    EIG_HIP(hipMemcpyAsync(sc.xbuf, A, sizeof(T) * N, hipMemcpyDeviceToDevice, st));
    EIG_HIP(hipStreamSynchronize(st));
    *nlaunch = 0; *algo_bytes = 0.0;
    const int nx = TD;
    int np = N;
    while (np - nb >= nx) {
        latrd_panel(c, st, sc, np, nb, A, lda, e, tau, W, N, true, nlaunch, algo_bytes);
        np -= nb;
    }
    int nbr = np - nx;
    if (nbr > 0) latrd_panel(c, st, sc, np, nbr, A, lda, e, tau, W, N, true, nlaunch, algo_bytes);
}

template <class T> void hemv_upper(Ctx& c, hipStream_t st, int n, const T* A, int lda, const T* x, T* y, bool gather) {
    if (n <= 0) return;
    TrdScratch<T> sc = trd_scratch<T>(c, n);
    PanelArgs<T> a;
    a.A = const_cast<T*>(A); a.lda = lda; a.W = nullptr; a.ldw = 0; a.np = n + 1; a.nb = 1; a.i = n;
    a.e = nullptr; a.tau = nullptr; a.xbuf = const_cast<T*>(x); a.P = sc.P; a.ldp = sc.ldp; a.S = sc.S; a.Zp = sc.Zp;
    a.NP = sc.NP; a.alphaSlot = sc.alphaSlot; a.nblkA = 0; a.nchunk = 0; a.ablate = 0;
    a.gh = hemv_grid(c, n);
    hipLaunchKernelGGL((panel_mv_kernel<T>), dim3(a.gh), dim3(256), 0, st, a, 1, 0);
    if (gather) {
        int nt = (n + HT - 1) / HT;
        hipLaunchKernelGGL((hemv_gather_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, n, nt, (const T*)sc.P, sc.ldp, y);
    }
    EIG_HIP(hipGetLastError());
}

template void hetrd_upper<double>(Ctx&, hipStream_t, int, double*, int, double*, double*, double*, double*, int);
template void hetrd_upper<cplx>(Ctx&, hipStream_t, int, cplx*, int, double*, double*, cplx*, cplx*, int);
template <class T> const void* hemv_scratch_touch(Ctx& c, int N, const void** all6) {
    TrdScratch<T> sc = trd_scratch<T>(c, N);
    if (all6) {
        all6[0] = sc.xbuf; all6[1] = sc.P; all6[2] = sc.S; all6[3] = sc.Zp; all6[4] = sc.NP; all6[5] = sc.alphaSlot;
    }
    return sc.P;
}
template const void* hemv_scratch_touch<double>(Ctx&, int, const void**);
template const void* hemv_scratch_touch<cplx>(Ctx&, int, const void**);
template void hetrd_mv_sweep<double>(Ctx&, hipStream_t, int, double*, int, double*, int, double*, double*, long*, double*);
template void hetrd_mv_sweep<cplx>(Ctx&, hipStream_t, int, cplx*, int, cplx*, int, double*, cplx*, long*, double*);
template void hemv_upper<double>(Ctx&, hipStream_t, int, const double*, int, const double*, double*, bool);
template void hemv_upper<cplx>(Ctx&, hipStream_t, int, const cplx*, int, const cplx*, cplx*, bool);

}  // namespace eig
