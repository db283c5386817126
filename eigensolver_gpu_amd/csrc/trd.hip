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
#include "lanes.h"

namespace eig {

#ifndef EIG_MV_PREFETCH
#define EIG_MV_PREFETCH 1
#endif
#ifndef EIG_TRD_TIMING
#define EIG_TRD_TIMING 0
#endif
#if EIG_TRD_TIMING
__device__ unsigned long long g_trd_stamp[4][16];   // [kernel][phase] accumulated shader cycles, block 0 lane 0
__device__ unsigned long long g_trd_count[4];       // kernel 0: panel_mv_kernel, 1: panel_row_kernel
#define TSTAMP(KID, PH, T0) do { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_trd_stamp[KID][PH], (unsigned long long)(__builtin_readcyclecounter() - (T0))); } while (0)
#define TSTAMPW(KID, PH, T0, W) do { if (blockIdx.x == 0 && threadIdx.x == 64 * (W)) atomicAdd(&g_trd_stamp[KID][PH], (unsigned long long)(__builtin_readcyclecounter() - (T0))); } while (0)
#else
#define TSTAMP(KID, PH, T0) do { } while (0)
#define TSTAMPW(KID, PH, T0, W) do { } while (0)
#endif
#ifndef EIG_MV_SNAKE
#define EIG_MV_SNAKE 1    // the mat-vec's tiles walked in alternating direction from column to column (0: always ascending, rounds 1-5)
#endif
#ifndef EIG_MV_PADLDS
#define EIG_MV_PADLDS 0   // (measurement variant: the register-staged kernel with the LDS footprint of the DMA form)
#endif
// The partial sums P the mat-vec hands to the next row kernel, stored / loaded with the non-temporal hint (EIG_MV_NT_P = 1): they are
// written once and read once, and with the alternating tile order every L2 line they do not take is a line of the matrix stream
// that the next column finds.
#ifndef EIG_MV_NT_P
#define EIG_MV_NT_P 0
#endif
typedef double d2v_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_p(double* p, double v) { if (EIG_MV_NT_P) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void st_p(cplx* p, cplx v) {
    if (EIG_MV_NT_P) __builtin_nontemporal_store(d2v_{v.x, v.y}, reinterpret_cast<d2v_*>(p)); else *p = v;
}
__device__ __forceinline__ double ld_p(const double* p) { return EIG_MV_NT_P ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ cplx ld_p(const cplx* p) {
    if (EIG_MV_NT_P) { const d2v_ v = __builtin_nontemporal_load(reinterpret_cast<const d2v_*>(p)); return cplx{v.x, v.y}; }
    return *p;
}
constexpr int HT = 64;     // hemv tile: 64 rows x 64 cols per workgroup step (one row per lane)
constexpr int CH = 512;    // rows per gemv partial chunk (8 rows per lane)
constexpr int NBMAX = 64;  // maximum panel width
constexpr int MVT = 320;   // panel_mv_kernel: four streaming waves + one finishing wave

template <class T> struct PanelArgs {
    T* A; int lda;
    T* W; int ldw;     // panel W (np x nb), column (k - wbase) belongs to matrix column k
    int np, nb, i;
    double* e; T* tau;
    T* xbuf;           // updated, unscaled column i (length >= np)
    T* P; int ldp;     // hemv partials P[q*ldp + row]
    T* S;              // per-hemv-workgroup partial of v^H A v
    T* Zp;             // gemv partials Zp[(chunk*2 + which)*NBMAX + kk]
    double* NP;        // per-wave (4 rows) partials of the squared norm
    T* alphaSlot;      // A(i-1,i) after the update
    int nblkA;         // norm partials (one per row-kernel wave) produced for column i
    int gh;            // hemv workgroups used for the column being finished / generated
    int nchunk;        // gemv row chunks for that column
};

// A launch serves NB independent problems of the same order in lockstep (blockIdx.y = problem): the per-column kernels of
// a tridiagonalization are latency-bound for most of the reduction (4.3-4.8 us per mat-vec launch up to
// n = 1280 whatever the work), so two problems per launch cost little more than one.  NB = 1 is the single-problem path.
constexpr int MAXB = 4;
template <class T, int NB> struct PanelBatch {
    PanelArgs<T> p[NB];
};

// larfg scalars with the reference's scaling (zhetrd_gpu.F90:275-311: scale by max(|ar|,|ai|,xnorm), no
// safe-minimum loop).  Degenerate case follows LAPACK (tau=0, beta=ar).  This sits on the critical path of
// every column (all lanes of a wave evaluate it redundantly), so the three quotients are multiplications by
// Newton-refined reciprocals instead of IEEE division sequences.
template <class T> __device__ __forceinline__ void larfg_scalars(double ss, T alpha, double& beta, T& tau, T& scale) {
    double ar = real_(alpha), ai = imag_(alpha);
    if (ss == 0.0 && ai == 0.0) {
        beta = ar;
        tau = Tr<T>::zero();
        scale = Tr<T>::zero();
        return;
    }
    double xnorm = fast_sqrt(ss);
    double rv1 = fabs(ar), rv2 = fabs(ai);
    double scal = fmax(fmax(rv1, rv2), xnorm);
    double inv = fast_rcp(scal);
    rv1 *= inv; rv2 *= inv; xnorm *= inv;
    beta = -copysign(scal * fast_sqrt(rv1 * rv1 + rv2 * rv2 + xnorm * xnorm), ar);
    double rb = fast_rcp(beta);
    tau = Tr<T>::make((beta - ar) * rb, -ai * rb);
    if constexpr (Tr<T>::cx) {
        // 1/(alpha - beta) = conj(x) / |x|^2 on the scaled x (|x| in [1, 3])
        double xr = (ar - beta) * inv, xi = ai * inv;
        double r = fast_rcp(xr * xr + xi * xi) * inv;
        scale = cplx{xr * r, -xi * r};
    } else {
        scale = fast_rcp(ar - beta);
    }
}

// ------------------------------------------------------------------------------------------
// panel_row_kernel : 16 rows per workgroup, one matrix row per 16-lane DPP row (4 rows per wave), the 16
// lanes of a row split the pending panel columns / hemv stripes.  Every global load the kernel needs is
// issued up front, branch-free (clamped address + select), chain-critical ones first.  One workgroup
// barrier (the gemv / v^H A v partial sums are gathered by the whole workgroup); after it each wave works
// alone: alpha and w_i are evaluated redundantly per wave, the per-row sums are DPP row reductions.
//   do_finish: finish W(:,c), c = i+1, from the hemv partials of the previous mat-vec
//   do_update: apply the pending rank-2 updates to column i, emit xbuf / norm partials / alpha slot
// ------------------------------------------------------------------------------------------
constexpr int RR = 16;   // rows per workgroup
constexpr int RG = 16;   // lanes per row
constexpr int RUMAX = 4; // panel columns per lane (NBMAX / RG)
constexpr int RPMAX = 8; // hemv stripes per lane without the tail loop (N <= RPMAX*RG*HT = 8192)
constexpr int NPW = 4;   // norm partials per workgroup (one per wave)
constexpr int ZC = 4;    // gemv chunk partials per lane without the tail loop (256 lanes x 4 = 128 sums x 8 chunks: N <= 4096)
constexpr int SC = 2;    // v^H A v partials per lane without the tail loop (512 hemv workgroups)

// FIN / UPD are compile-time, and RU / RP (panel columns / hemv stripes per lane, rounded up by the host) size
// the unconditional load groups: every load is issued, in a fixed order, so the compiler can wait for exactly
// the values it needs (s_waitcnt vmcnt(k) with the later loads still in flight) instead of for everything.
template <class T, bool FIN, bool UPD, int RU, int RP, int NB>
__global__ void __launch_bounds__(256) panel_row_kernel(PanelBatch<T, NB> ab) {
    const PanelArgs<T>& a = ab.p[NB == 1 ? 0 : blockIdx.y];
    constexpr bool do_finish = FIN, do_update = UPD;
    const int i = a.i, c = i + 1;
    const int npo = do_finish ? a.np - 1 - c : 0;  // columns older than c inside the panel
    const int wbase = a.np - a.nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane & (RG - 1), rr = lane >> 4;
#if EIG_TRD_TIMING
    const int gg = 0;
    const long long T0 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_trd_count[1], 1ULL);
#endif
    const int r = blockIdx.x * RR + wave * 4 + rr;
    const int rows = i + 1;
    const bool active = r < rows;
    const size_t rc = (size_t)min(r, rows - 1);   // clamped row: loads stay in bounds, results are masked
    const int ntc = (c + HT - 1) / HT;            // hemv stripes of the column being finished (n = c)

    __shared__ T z1s[2][NBMAX], z2s[2][NBMAX];             // two half sums each (chunks split over the wave pairs)
    __shared__ T rowW[4][NBMAX + 1], rowV[4][NBMAX + 1];   // private to each wave
    __shared__ T s4[4];

    // ---------------- phase 0: issue every load (raw, unconditional, clamped), chain-critical ones first -------
    const T zero = Tr<T>::zero();
    T tau = zero, vic = zero, l_ww = zero, l_wv = zero, l_p0 = zero, vr = zero, acur = zero;
    T zt[ZC], st[SC], tv[RU], tw[RU], pt[RP];
    const int kz = min(lane, max(npo - 1, 0));
    const int zwhich = wave & 1, zhalf = wave >> 1;   // waves 0,2: z1 (A columns); 1,3: z2 (W columns)
    if constexpr (do_finish) {
        // (a) what alpha and w_i need, per wave
        tau = a.tau[c - 1];
        if constexpr (do_update) {
            vic = a.A[(size_t)i + (size_t)c * a.lda];
            const int kl = min(c + 1 + kz, a.np - 1);
            l_ww = a.W[(size_t)i + (size_t)(kl - wbase) * a.ldw];
            l_wv = a.A[(size_t)i + (size_t)kl * a.lda];
            l_p0 = ld_p(&a.P[(size_t)min(lane, ntc - 1) * a.ldp + i]);
        }
        // (b) partial sums gathered by the workgroup: stacked gemv and v^H A v
#pragma unroll
        for (int u = 0; u < ZC; ++u) zt[u] = a.Zp[(size_t)(min(zhalf * ZC + u, a.nchunk - 1) * 2 + zwhich) * NBMAX + kz];
#pragma unroll
        for (int u = 0; u < SC; ++u) st[u] = a.S[min(tid + 256 * u, a.gh - 1)];
        // (c) this lane's slice of its row
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int k = min(c + 1 + min(g + RG * u, max(npo - 1, 0)), a.np - 1);
            tv[u] = a.A[rc + (size_t)k * a.lda];
            tw[u] = a.W[rc + (size_t)(k - wbase) * a.ldw];
        }
#pragma unroll
        for (int u = 0; u < RP; ++u) pt[u] = ld_p(&a.P[(size_t)min(g + RG * u, ntc - 1) * a.ldp + rc]);
        vr = a.A[rc + (size_t)c * a.lda];
    }
    if constexpr (do_update) acur = a.A[rc + (size_t)i * a.lda];
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(1, 0, T0);   // loads issued

    T alpha = zero;
    if constexpr (do_finish) {
        // ---------------- phase 1: gather the partial sums (the only workgroup barrier) ----------------
        T zs = zero, Ssum = zero;
#pragma unroll
        for (int u = 0; u < ZC; ++u) zs = zs + sel(zhalf * ZC + u < a.nchunk, zt[u], zero);
#pragma unroll
        for (int u = 0; u < SC; ++u) Ssum = Ssum + sel(tid + 256 * u < a.gh, st[u], zero);
        // tails beyond the unrolled counts (N > 4096 or more than 512 hemv workgroups)
        if (zhalf == 1)
            for (int ch = 2 * ZC; ch < a.nchunk; ++ch) zs = zs + a.Zp[(size_t)(ch * 2 + zwhich) * NBMAX + kz];
        for (int q = tid + 256 * SC; q < a.gh; q += 256) Ssum = Ssum + a.S[q];
        if (lane < npo) { if (zwhich == 0) z1s[zhalf][lane] = zs; else z2s[zhalf][lane] = zs; }
        Ssum = wave_sum(Ssum);
        if (lane == 0) s4[wave] = Ssum;
        TSTAMP(1, 1, T0);   // critical loads arrived
        __syncthreads();
        TSTAMP(1, 2, T0);
        // ---------------- phase 2: alpha and w_i, redundantly per wave ----------------
        const T z1l = sel(lane < npo, z1s[0][kz] + z1s[1][kz], zero), z2l = sel(lane < npo, z2s[0][kz] + z2s[1][kz], zero);
        T t = zero;
        fmac_(t, z1l, z2l);
        const double zz = wave_sum(real_(t));
        const T S = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        // alpha = -1/2 tau (w'^H v),  w'^H v = conj(tau) conj(S - 2 Re(z1^H z2))
        alpha = (-0.5 * abs2_(tau)) * conj_(S - Tr<T>::make(2.0 * zz, 0.0));
        if constexpr (do_update) {
            const T wi_w = sel(lane < npo, l_ww, zero), wi_v = sel(lane < npo, l_wv, zero);
            T wi_p = sel(lane < ntc, l_p0, zero);
            for (int q = lane + 64; q < ntc; q += 64) wi_p = wi_p + ld_p(&a.P[(size_t)q * a.ldp + i]);   // N > 4096 only
            const T u = wave_sum(wi_p - (wi_w * z1l + wi_v * z2l));
            const T wi = tau * u + alpha * vic;
            if (lane < npo) { rowW[wave][lane] = conj_(wi_w); rowV[wave][lane] = conj_(wi_v); }
            if (lane == 0) { rowW[wave][npo] = conj_(wi); rowV[wave][npo] = conj_(vic); }
            __builtin_amdgcn_wave_barrier();   // wave-private LDS: program order is enough, keep the compiler from reordering
        }
    }

    // ---------------- phase 3: this lane's slice, then the DPP row sums ----------------
    // (the row data -- the bulk of the loads, issued last -- is first touched here, after the scalar chain)
    __builtin_amdgcn_sched_barrier(0);
    T acc2 = zero, acc3 = zero;
    if constexpr (do_finish) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int kk = g + RG * u;
            if (kk < npo) {
                const T vvu = sel(active, tv[u], zero), wvu = sel(active, tw[u], zero);
                acc2 = acc2 - (wvu * (z1s[0][kk] + z1s[1][kk]) + vvu * (z2s[0][kk] + z2s[1][kk]));
                if constexpr (do_update) acc3 = acc3 + (vvu * rowW[wave][kk] + wvu * rowV[wave][kk]);
            }
        }
#pragma unroll
        for (int u = 0; u < RP; ++u) acc2 = acc2 + sel((g + RG * u < ntc) && active, pt[u], zero);
        for (int q = g + RG * RP; q < ntc; q += RG) acc2 = acc2 + sel(active, ld_p(&a.P[(size_t)q * a.ldp + rc]), zero);
    }
    TSTAMP(1, 3, T0);       // row data arrived, slices done
    const T urow = row_sum16(acc2);
    T upd = row_sum16(acc3);
    // ---------------- phase 4: finish the row (all 16 lanes hold the same values, lane g == 0 stores) ----------------
    const bool writer = active && g == 0;
    T anew = acur;
    if constexpr (do_finish) {
        const T wr = tau * urow + alpha * vr;
        if (writer) a.W[(size_t)r + (size_t)(c - wbase) * a.ldw] = wr;
        if constexpr (do_update) {
            upd = upd + (vr * rowW[wave][npo] + wr * rowV[wave][npo]);
            anew = acur - upd;
            if (r == i) anew = Tr<T>::realpart(anew);
            if (writer) a.A[(size_t)r + (size_t)i * a.lda] = anew;
        }
    }
    TSTAMP(1, 4, T0);
    if constexpr (do_update) {
        if (writer) {
            a.xbuf[r] = anew;
            if (r == i - 1) *a.alphaSlot = anew;
        }
        double contrib = (writer && r <= i - 2) ? abs2_(anew) : 0.0;
        contrib = (read_lane(contrib, 0) + read_lane(contrib, 16)) + (read_lane(contrib, 32) + read_lane(contrib, 48));
        if (lane == 0) a.NP[blockIdx.x * NPW + wave] = contrib;
    }
    TSTAMP(1, 5, T0);       // end
}

// host-side choice of the load-group sizes
template <class T, bool FIN, bool UPD, int NB>
static void launch_row(hipStream_t st, int grid, int nprob, const PanelBatch<T, NB>& a, int npo, int ntc) {
    const int ru = npo <= 16 ? 1 : (npo <= 32 ? 2 : (npo <= 48 ? 3 : 4));
    const int rp = ntc <= 16 ? 1 : (ntc <= 32 ? 2 : (ntc <= 64 ? 4 : 8));
#define EIG_ROW(RU_, RP_) hipLaunchKernelGGL((panel_row_kernel<T, FIN, UPD, RU_, RP_, NB>), dim3(grid, nprob), dim3(256), 0, st, a)
#define EIG_ROW_RP(RU_) do { if (rp == 1) EIG_ROW(RU_, 1); else if (rp == 2) EIG_ROW(RU_, 2); else if (rp == 4) EIG_ROW(RU_, 4); else EIG_ROW(RU_, 8); } while (0)
    if constexpr (!FIN) { EIG_ROW(1, 1); }
    else if constexpr (!UPD) { EIG_ROW_RP(4); }   // once per panel: only the stripe count is specialised
    else {
        if (ru == 1) EIG_ROW_RP(1); else if (ru == 2) EIG_ROW_RP(2); else if (ru == 3) EIG_ROW_RP(3); else EIG_ROW_RP(4);
    }
#undef EIG_ROW_RP
#undef EIG_ROW
}

#if EIG_TRD_TIMING
extern "C" int eigsolve_debug_trd_timing(unsigned long long* out36) {
    unsigned long long st[4][16], cn[4];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_trd_stamp), sizeof st) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(cn, HIP_SYMBOL(g_trd_count), sizeof cn) != hipSuccess) return -1;
    for (int k = 0; k < 4; ++k) { out36[k * 9] = cn[k]; for (int p = 0; p < 8; ++p) out36[k * 9 + 1 + p] = st[k][p]; }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------
// panel_mv_kernel : larfg scalars + Hermitian mat-vec (upper, each element read once) + stacked
// conjugate-transposed panel products.  Also used stand-alone (bench / zhemv entry point) with
// `plain` != 0: v = xbuf as is, no scalars, no gemv part.
// Grid = [gemv workgroups | hemv workgroups]: the (short, latency-bound) gemv items start first
// and hide under the bandwidth-bound tiles.  Loads are issued before the scalar prologue.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_decode(int t, int& I, int& J) {
    // t = J(J+1)/2 + I, I <= J.  Single-precision estimate (one v_sqrt_f32) + exact integer correction.
    J = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((J + 1) * (J + 2) / 2 <= t) ++J;
    while (J * (J + 1) / 2 > t) --J;
    I = t - J * (J + 1) / 2;
}

// (LDS-DMA primitives: lanes.h)
// NI instructions of one slot in ONE statement (M0 saved once): instruction q copies [base + voff + q * stride] -> [dst + 1024 q].
// base, stride, dst wave-uniform; voff = the lane's byte offset.  Three instructions per piece: the M0 update, the offset update
// (which also is the wait state an M0 write needs in front of an LDS-DMA) and the copy.
#define EIG_DMA_STEP "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, %3, %1\n\t"
#define EIG_DMA_STEP4 EIG_DMA_STEP EIG_DMA_STEP EIG_DMA_STEP EIG_DMA_STEP
template <int NI> __device__ __forceinline__ void lds_dma16_run(const void* base, unsigned voff, unsigned stride, unsigned dst) {
    static_assert(NI == 8 || NI == 16, "pieces per slot");
    unsigned keep;
    if constexpr (NI == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t" EIG_DMA_STEP4 EIG_DMA_STEP4 EIG_DMA_STEP4 EIG_DMA_STEP4 "s_mov_b32 m0, %0"
                     : "=&s"(keep), "+v"(voff)
                     : "s"(base), "s"(stride), "s"(dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t" EIG_DMA_STEP4 EIG_DMA_STEP4 "s_mov_b32 m0, %0"
                     : "=&s"(keep), "+v"(voff)
                     : "s"(base), "s"(stride), "s"(dst)
                     : "memory");
}

// Geometry of the LDS ring of panel_mv_kernel<T, NB, true>: every streaming wave owns DEPTH slots; a slot holds the wave's 16
// columns of one 64 x 64 tile ([column][64 rows], exactly the image the DMA writes: a column of 64 complex numbers is one
// instruction, two columns of 64 doubles are one) followed by the tile's 64 column entries and 64 row entries of v.
constexpr int MVD = 384;   // four streaming waves + the finishing wave + the wave of the stacked products
template <class T> struct MvRing {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int CPI = 1024 / (HT * ES);        // tile columns per DMA instruction: 1 (complex) / 2 (real)
    static constexpr int NDA = 16 / CPI;                // instructions per slot for the matrix
    static constexpr int NDV = Tr<T>::cx ? 2 : 1;       // ... for the entries of v
    static constexpr int NDT = NDA + NDV;
    static constexpr int DEPTH = Tr<T>::cx ? 2 : 4;     // slots (wave-tiles in flight) per streaming wave
    static constexpr int SLOT = 18 * HT * ES;           // bytes
    static constexpr int RING = 4 * DEPTH * SLOT;       // 144 KB
    static constexpr int REDY = RING, REDT = REDY + 2 * 4 * HT * ES, XCS = REDT + 2 * HT * ES, TOTAL = XCS + 3 * HT * ES;
    static_assert(TOTAL <= 160 * 1024, "one workgroup per CU");
};

// The products of one 64 x 64 tile for the wave that holds 16 of its columns (av[j] = A(r, c0 + 16 wave + j), one row per lane):
// yI = sum_j A(r, c) xc(c) (this wave's part of the row-direction sum) and tval = the 64-lane sum of conj(A(r, c)) xr for the column
// transpose_col_of_lane(lane) -- 8 columns at a time: products, then the first two levels of the column reduction.  Shared by the
// register-staged and the LDS-DMA data path (panel_mv_kernel / panel_mv_dma_kernel): same operations in the same order, i.e. the
// two kernels give bit-identical results.
template <class T, class XC>
__device__ __forceinline__ void mv_tile_products(const T (&av)[16], T xr, XC xc, int wave, int lane, int r, int c0, int n, bool diag,
                                                 bool interior, T& yI, T& tval) {   // xc(j) = entry of v for the wave's column j
    const T zero = Tr<T>::zero();
    yI = zero;
    auto half = [&](int jb, T& w0, T& w1) {
        T tj[8];
        if (interior) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                fma_(yI, av[jb + j], xc(jb + j));
                T p = Tr<T>::zero();
                fmac_(p, av[jb + j], xr);
                tj[j] = p;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cc = c0 + wave * 16 + jb + j;
                const bool ok = (r < n) && (cc < n) && (!diag || r <= cc);
                const bool dg = diag && r == cc;
                T v = sel(ok, av[jb + j], zero);
                v = sel(dg, Tr<T>::realpart(v), v);
                fma_(yI, v, xc(jb + j));
                T p = Tr<T>::zero();
                fmac_(p, sel(dg, zero, v), xr);
                tj[j] = p;
            }
        }
        transpose_reduce8_phase1<T>(tj, w0, w1);
    };
    T wa0, wa1, wb0, wb1;
    half(0, wa0, wa1);
    half(8, wb0, wb1);
    tval = transpose_reduce_phase2<T>(wa0, wa1, wb0, wb1, lane);
}

template <class T, int NB, bool DMA>
__global__ void __launch_bounds__(DMA ? MVD : MVT, DMA ? 2 : 3) panel_mv_kernel(PanelBatch<T, NB> ab, int plain, int gg) {
    const PanelArgs<T>& a = ab.p[NB == 1 ? 0 : blockIdx.y];
    const int i = a.i, n = i;  // v has n entries (rows 0..i-1), v(n-1) = 1
    // (DMA form: the wave index is made wave-uniform FOR THE COMPILER, so that the wave roles below are scalar branches.  With a
    //  per-lane condition hipcc lays the roles out one after the other under exec masks and its s_waitcnt bookkeeping flows from
    //  one role into the next: the streaming waves then waited vmcnt(12) in front of their LDS reads -- for registers the OTHER
    //  roles load into -- which drains the DMA ring every tile.)
    const int tid = threadIdx.x, lane = tid & 63, wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
#if EIG_TRD_TIMING
    const long long T0 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&g_trd_count[0], 1ULL); atomicAdd(&g_trd_count[2], 1ULL); }
#endif
    // Five waves: 0-3 stream and multiply the tiles, wave 4 evaluates the larfg scalars (a serial chain of ~1500 cycles)
    // while the first tile is being multiplied and finishes every tile (partial sums, S, the stored v) while the
    // others are already on the next one.  LDS hand-over buffers are double (partials) / triple (column entries of
    // v) buffered so that one barrier per tile is enough.  Two workgroups per CU need <= 168 VGPRs per wave (10 waves
    // on 4 SIMDs): the column reduction runs in two halves of 8 columns to stay below that.
    // (ONE LDS object: a second one makes hipcc wait vmcnt(0) in front of LDS reads, cdna_hip_programming.md 5 "three traps")
    using RG = MvRing<T>;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(DMA || EIG_MV_PADLDS) ? RG::TOTAL : RG::TOTAL - RG::RING];
    constexpr int SM0 = DMA ? RG::RING : 0;
    T (&redy)[2][4][64] = *reinterpret_cast<T (*)[2][4][64]>(smem + SM0);
    T (&redt)[2][64] = *reinterpret_cast<T (*)[2][64]>(smem + SM0 + (RG::REDT - RG::REDY));
    T (&xcs)[3][HT] = *reinterpret_cast<T (*)[3][HT]>(smem + SM0 + (RG::XCS - RG::REDY));   // xh entries of the tile columns: tile k of this workgroup uses slot k % 3
    constexpr int NPL = 16;    // norm partials per lane loaded up front (N <= 4096 without the tail loop)
    // The mat-vec workgroups come FIRST in the grid: tile t is then always taken by workgroup t mod gh, i.e. (gh = one per CU, a
    // multiple of 8) by the same XCD in every column of the sweep, and what that XCD's L2 still holds of the tile from the previous
    // column is a hit -- the whole stored triangle once it is below 8 x 4 MB (round 4: zhetrd N=2048 21.2 -> 20.2 ms, dsytrd
    // N=2048 17.0 -> 16.8, zhetrd N=4096 66.6 -> 66.2; with the gemv workgroups in front the XCD of a tile moved with their count).
    // DMA form: no separate workgroups for the stacked products (a workgroup holds the whole LDS of its CU, the short ones could
    // not start beside the streaming ones): wave 5 of every mat-vec workgroup takes the items hb, hb + gh, ... and exits.
    const bool is_gemv = DMA ? wave == 5 : (int)blockIdx.x >= a.gh;
    const int hb = (int)blockIdx.x;          // hemv workgroup index
    const int gb = (int)blockIdx.x - a.gh;   // gemv workgroup index
    if (!DMA && is_gemv && wave == 4) return;

    // v = scale * xh + e_(n-1), xh = raw column with the entries >= nz zeroed.  Everything below is linear in
    // v, so the products are formed with xh (known at launch) and the larfg scalars -- the end of a chain
    // load -> reduce -> sqrt/reciprocals -- are only applied to the reduced partial sums.
    const int nz = plain ? n : n - 1;
    const int one_at = plain ? -1 : n - 1;
    const T zero = Tr<T>::zero();
    auto unit = [&](int r) -> T { return sel(r == one_at, Tr<T>::one(), zero); };

    // ---------------- Hermitian mat-vec tiles: the four streaming waves ----------------
    const int nt = (n + HT - 1) / HT;
    const int ntiles = nt * (nt + 1) / 2;
    // Workgroup hb owns the tiles hb, hb + gh, hb + 2 gh, ... -- the same tiles in every column of the sweep (and, the grid being a
    // multiple of 8, on the same XCD).  It walks them in ALTERNATING direction from column to column (round 6): what a column reads
    // last is what the next column reads first, i.e. the part of the stream that is still in that XCD's 4 MB L2 (L2 contents survive
    // kernel boundaries, profiles/r04_experiments.txt 15; walked in one direction an LRU cache smaller than the stream never hits).
    // Every tile writes its own slots of the partial array, so only the order of the per-workgroup sum S changes with the direction.
    const int KT = (hb < ntiles && !is_gemv) ? (ntiles - hb + a.gh - 1) / a.gh : 0;
    // (only where ONE problem's stream exceeds what the L2s keep anyway: below ~24 MB every order hits, and the ascending one measured
    //  2 % faster there -- dsytrd N=2048 sweep 9.26 vs 9.46 ms.  The rule must not look at NB: the direction decides the order of
    //  the sum S, and a problem solved inside a lockstep group has to give the bits it gives alone -- with the group's stream in
    //  the rule the problems of a C5 batch differed from their single solves in the last bits: test_c5_full_size_batch_all_64_problems)
    const bool rev = EIG_MV_SNAKE && !plain && (i & 1) && (size_t)ntiles * (HT * HT * sizeof(T)) > ((size_t)24 << 20);
    auto tile_at = [&](int k) -> int { return hb + (rev ? KT - 1 - k : k) * a.gh; };
    int I = 0, J = 0, t = KT > 0 ? tile_at(0) : ntiles;
    if constexpr (DMA) {
        if (wave < 4) {
            // Streaming waves, LDS-DMA form: DEPTH wave-tiles in flight per wave whatever the wave computes meanwhile.  Per tile:
            // wait for its slot (the younger tiles' instructions stay in flight), copy the slot into registers, hand the slot straight
            // back to the DMA for the tile DEPTH steps ahead, and only then multiply -- the memory pipe never waits for the products.
            constexpr int DEPTH = RG::DEPTH, ES = RG::ES;
            const int wu = wave;
            const unsigned ring_w = lds_offset_of(smem) + (unsigned)(wu * DEPTH * RG::SLOT);
            const int vcl = max(nz - 1, 0);
            auto issue = [&](int slot, int tt) {
                int Iq, Jq;
                tile_decode(tt, Iq, Jq);
                const int r0 = Iq * HT, c0 = Jq * HT;
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_w + (unsigned)(slot * RG::SLOT)));
                // interior tile (every row and column inside the matrix): the columns are lda apart -> one statement, 3 instructions
                // per piece instead of a 64-bit address computation each
                const bool inside = r0 + HT <= n && c0 + HT <= n && (long)a.lda * ES * 16 < (1L << 31);
                if (inside) {
                    if constexpr (Tr<T>::cx) {
                        lds_dma16(a.xbuf + min(c0 + lane, vcl), dst + 16 * HT * ES);
                        lds_dma16(a.xbuf + min(r0 + lane, vcl), dst + 17 * HT * ES);
                        lds_dma16_run<16>(a.A + (size_t)r0 + (size_t)(c0 + wu * 16) * a.lda, (unsigned)(lane * ES), (unsigned)(a.lda * ES), dst);
                    } else {
                        const int l2 = 2 * (lane & 31), hi = lane >> 5;
                        lds_dma16(a.xbuf + (hi ? r0 + l2 : c0 + l2), dst + 16 * HT * ES);
                        lds_dma16_run<8>(a.A + (size_t)r0 + (size_t)(c0 + wu * 16) * a.lda, (unsigned)(l2 * ES + hi * a.lda * ES),
                                         (unsigned)(2 * a.lda * ES), dst);
                    }
                    return;
                }
                if constexpr (Tr<T>::cx) {
                    lds_dma16(a.xbuf + min(c0 + lane, vcl), dst + 16 * HT * ES);
                    lds_dma16(a.xbuf + min(r0 + lane, vcl), dst + 17 * HT * ES);
                    const size_t roff = (size_t)min(r0 + lane, n - 1);
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        lds_dma16(a.A + roff + (size_t)min(c0 + wu * 16 + q, n - 1) * a.lda, dst + q * 1024);
                } else {
                    // 16 B = two consecutive rows; lanes 0-31 / 32-63 = two consecutive columns (v: column entries / row entries).
                    // Raw, clamped pairs: everything beyond n or nz is masked when the tile is consumed.  (lda even, A and xbuf
                    // 16-byte aligned: checked by the host; xbuf is the library's own buffer of nt * 64 + 64 entries.)
                    const int l2 = 2 * (lane & 31), hi = lane >> 5;
                    lds_dma16(a.xbuf + (hi ? r0 + l2 : c0 + l2), dst + 16 * HT * ES);
                    const size_t roff = (size_t)min(r0 + l2, (n - 1) & ~1);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        lds_dma16(a.A + roff + (size_t)min(c0 + wu * 16 + 2 * q + hi, n - 1) * a.lda, dst + q * 1024);
                }
            };
            // (Measured and dropped: a barrier here, so that the loads of the finishing wave and of the stacked products are queued
            //  in front of the tile pieces -- a CU serves its vector-memory requests in order and behind 72 KB of pieces the stacked
            //  products' loads come back after 9000 cycles.  The barrier costs the tiles 1400 cycles and the launch 0.3 us:
            //  profiles/r06_experiments.txt section 1.)
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                if (d < KT) issue(d, tile_at(d));
            TSTAMP(0, 0, T0);   // first DEPTH tiles requested
            int k = 0;
            while (k < KT) {
                tile_decode(t, I, J);
                const int r0 = I * HT, c0 = J * HT;
                const bool diag = (I == J);
                const int r = r0 + lane;
                const int rb = k & 1, xs = k % 3, slot = k & (DEPTH - 1);
                const int ahead = min(DEPTH - 1, KT - 1 - k);   // younger tiles of this wave in flight
                if (ahead == 0) wait_vmcnt<0>();
                else if (ahead == 1) wait_vmcnt<RG::NDT>();
                else if (ahead == 2) wait_vmcnt<2 * RG::NDT>();
                else wait_vmcnt<3 * RG::NDT>();
                if (k == 0) TSTAMP(0, 1, T0);   // first tile landed
                const T* sp = reinterpret_cast<const T*>(smem + (wu * DEPTH + slot) * RG::SLOT);
                T av[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) av[j] = sp[j * HT + lane];
                const T xc_raw = sp[16 * HT + lane], xr_raw = sp[17 * HT + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is in registers: it may be overwritten
                if (k == 0) TSTAMP(0, 6, T0);   // ... and copied into registers
                if (k + DEPTH < KT) issue(slot, tile_at(k + DEPTH));
                if (k == 0) TSTAMP(0, 7, T0);   // ... slot handed back to the DMA
                xcs[xs][lane] = sel(c0 + lane < nz, xc_raw, zero);
                const T xr = sel(r < nz, xr_raw, zero);
                const bool interior = !diag && r0 + HT <= n && c0 + HT <= n;
                T yI, tval;
                T xcv[16];      // the wave's 16 column entries of v: all LDS reads in flight at once, not one round trip per column
#pragma unroll
                for (int j = 0; j < 16; ++j) xcv[j] = xcs[xs][wave * 16 + j];
                mv_tile_products<T>(av, xr, [&](int j) -> T { return xcv[j]; }, wave, lane, r, c0, n, diag, interior, yI, tval);
                if (k == 0) { TSTAMP(0, 3, T0); TSTAMPW(2, 0, T0, 3); }   // first tile multiplied: wave 0, wave 3
                redy[rb][wave][lane] = yI;
                if ((lane & 3) == 0) redt[rb][wave * 16 + transpose_col_of_lane(lane)] = tval;
                ++k;
                t = k < KT ? tile_at(k) : ntiles;
                __syncthreads();
                if (k == 1) TSTAMP(2, 1, T0);   // first barrier passed (streaming wave 0)
            }
            TSTAMP(2, 2, T0);   // streaming wave 0 done
            return;
        }
    }
    if (!DMA && !is_gemv && wave < 4) {
        T av[16], xr_raw = zero;   // raw loads of the tile in flight
        // Issue the loads of tile t: unconditional, clamped addresses, nothing else in between (a select on a
        // loaded value is where the compiler waits; masks are applied when the tile is consumed).  The column
        // entries of v go first: their hand-over to LDS then waits for that one load and leaves the rest in flight.
        // All four waves store the same 64 values (a store only one wave executes lets the compiler sink the load).
        auto issue_tile = [&](int slot) {
            tile_decode(t, I, J);
            const int r0 = I * HT, c0 = J * HT, r = r0 + lane;
            const T xc_raw = a.xbuf[min(c0 + lane, max(nz - 1, 0))];
            xr_raw = a.xbuf[min(r, max(nz - 1, 0))];
            const size_t roff = (size_t)min(r, n - 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) av[j] = a.A[roff + (size_t)min(c0 + wave * 16 + j, n - 1) * a.lda];
            __builtin_amdgcn_sched_barrier(0);
            xcs[slot][lane] = sel(c0 + lane < nz, xc_raw, zero);
        };
        if (t < ntiles) issue_tile(0);
        TSTAMP(0, 0, T0);   // loads issued, xcs written
        __syncthreads();
        TSTAMP(0, 1, T0);
        int k = 0;
        while (k < KT) {
            const int r0 = I * HT, c0 = J * HT;
            const bool diag = (I == J);
            const int r = r0 + lane;
            const int rb = k & 1, xs = k % 3;
            const T xr = sel(r < nz, xr_raw, zero);
            const bool interior = !diag && r0 + HT <= n && c0 + HT <= n;
            T yI, tval;
            mv_tile_products<T>(av, xr, [&](int j) -> T { return xcs[xs][wave * 16 + j]; }, wave, lane, r, c0, n, diag, interior, yI, tval);
            TSTAMP(0, 3, T0);   // tile loads arrived, FMAs + transpose-reduce done
            redy[rb][wave][lane] = yI;
            if ((lane & 3) == 0) redt[rb][wave * 16 + transpose_col_of_lane(lane)] = tval;
            ++k;
            t = k < KT ? tile_at(k) : ntiles;
            if (t < ntiles) issue_tile(k % 3);   // next tile's loads fly while wave 4 finishes this one
            TSTAMP(0, 4, T0);
            __syncthreads();
            if (k == 1) TSTAMP(2, 1, T0);
        }
        TSTAMP(2, 2, T0);
        return;
    }

    // ---------------- scalar loads (gemv waves and the finishing wave) ----------------
    // raw, unconditional (clamped) loads; selects and sums happen in scalars(), after everything is issued
    const int nnp = plain ? 1 : a.nblkA;
    T alpha_early = zero;
    double npl[NPL];
    {
        const T* ap = plain ? a.xbuf : a.alphaSlot;          // any valid address in plain mode, value unused
        const double* np = plain ? reinterpret_cast<const double*>(a.xbuf) : a.NP;
        alpha_early = *ap;
#pragma unroll
        for (int u = 0; u < NPL; ++u) npl[u] = np[min(lane + 64 * u, nnp - 1)];
    }
    // larfg scalars, evaluated redundantly by every lane of the calling wave (wave-uniform result)
    auto scalars = [&]() -> T {
        double ss = 0.0;
        for (int q = lane + 64 * NPL; q < a.nblkA; q += 64) ss += a.NP[q];
#pragma unroll
        for (int u = 0; u < NPL; ++u) npl[u] = (lane + 64 * u < a.nblkA) ? npl[u] : 0.0;
#pragma unroll
        for (int u = 0; u < NPL; u += 4) ss += (npl[u] + npl[u + 1]) + (npl[u + 2] + npl[u + 3]);
        ss = wave_sum(ss);
        double beta;
        T tau, scale;
        larfg_scalars<T>(ss, alpha_early, beta, tau, scale);
        if (blockIdx.x == 0 && lane == 0 && (is_gemv ? wave == 0 : wave == 4)) {
            a.e[i - 1] = beta;
            a.tau[i - 1] = tau;
        }
        return scale;
    };

    if (is_gemv) {
        // stacked conjugate-transposed products z1 = V^H v, z2 = W^H v (partials per row chunk), one item per wave
        // (DMA form: wave 5 of workgroup hb takes the items hb, hb + gh, ...; the larfg scalars are derived once)
        const int npo = a.np - 1 - i;
        const int wbase = a.np - a.nb;
        const int nitems = 2 * npo * a.nchunk;
        T scale = Tr<T>::one();
        bool have_scale = false;
        for (int item = DMA ? hb : gb * 4 + wave; item < nitems; item += DMA ? a.gh : nitems) {
            const int ch = item / (2 * npo);
            const int rem = item % (2 * npo);
            const int which = rem / npo, kk = rem % npo;
            const int kcol = i + 1 + kk;
            const T* src = which == 0 ? a.A + (size_t)kcol * a.lda : a.W + (size_t)(kcol - wbase) * a.ldw;
            const int rbeg = ch * CH;
            T s = Tr<T>::zero(), eone = Tr<T>::zero();
            T sv[8], xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {   // raw loads, all in flight before the first use
                int r = rbeg + lane + 64 * j;
                sv[j] = src[min(r, n - 1)];
                xv[j] = a.xbuf[min(r, max(nz - 1, 0))];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int r = rbeg + lane + 64 * j;
                const T sj = sel(r < n, sv[j], zero);
                fmac_(s, sj, sel(r < nz, xv[j], zero));
                eone = sel(r == one_at, conj_(sj), eone);
            }
            if (!have_scale) { scale = scalars(); have_scale = true; }
            s = wave_sum(scale * s + eone);
            if (lane == 0) a.Zp[(size_t)(ch * 2 + which) * NBMAX + kk] = s;
        }
        TSTAMPW(2, 4, T0, 5);   // wave of the stacked products done
        if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), see the end of the kernel
        return;
    }

    // ---------------- the finishing wave (wave 4 of a mat-vec workgroup) ----------------
    T xr_raw = zero, lc_raw = zero;
    auto issue_mine = [&]() {
        tile_decode(t, I, J);
        const int r = I * HT + lane;
        lc_raw = a.A[(size_t)min(r, n - 1) + (size_t)(n - 1) * a.lda];   // column n-1: the e_(n-1) part of v
        xr_raw = a.xbuf[min(r, max(nz - 1, 0))];
    };
    if (t < ntiles) issue_mine();
    if (!DMA) __syncthreads();       // (pairs with the streaming waves' barrier behind their first issue_tile)
    T scale = Tr<T>::one();
    if (!plain) scale = scalars();   // while waves 0-3 multiply the first tile
    TSTAMPW(0, 2, T0, 4);
    T Sacc = Tr<T>::zero();
    int k = 0;
    while (k < KT) {
        const int r0 = I * HT, c0 = J * HT, Ic = I, Jc = J;
        const bool diag = (I == J);
        const int r = r0 + lane;
        const T xr = sel(r < nz, xr_raw, zero);
        T lastc = sel(!plain && J == nt - 1 && r <= n - 1, lc_raw, zero);
        lastc = sel(r == n - 1, Tr<T>::realpart(lastc), lastc);
        t = k + 1 < KT ? tile_at(k + 1) : ntiles;
        if (t < ntiles) issue_mine();
        __syncthreads();   // partial sums of tile k are in redy / redt [k & 1]
        if (k == 0) TSTAMPW(2, 3, T0, 4);   // finishing wave: first barrier passed
        const int rb = k & 1, xs = k % 3;
        T yv = scale * ((redy[rb][0][lane] + redy[rb][1][lane]) + (redy[rb][2][lane] + redy[rb][3][lane])) + lastc;
        T tv = scale * redt[rb][lane];
        T vI = scale * xr + unit(r0 + lane);
        T vJ = scale * xcs[xs][lane] + unit(c0 + lane);
        if (diag) {
            T s = yv + tv;
            st_p(&a.P[(size_t)Jc * a.ldp + r0 + lane], s);
            fmac_(Sacc, vI, s);
            if (!plain && c0 + lane < n) a.A[(size_t)(c0 + lane) + (size_t)i * a.lda] = vJ;
        } else {
            st_p(&a.P[(size_t)Jc * a.ldp + r0 + lane], yv);
            st_p(&a.P[(size_t)Ic * a.ldp + c0 + lane], tv);
            fmac_(Sacc, vI, yv);
            fmac_(Sacc, vJ, tv);
        }
        ++k;
    }
    Sacc = wave_sum(Sacc);
    if (lane == 0) a.S[hb] = Sacc;
    TSTAMPW(0, 5, T0, 4);
    // DMA form: this role and the one above may leave loads unconsumed (the up-front norm partials in plain mode / without items).
    // hipcc merges the pending-load state of every role at the common exit block, which sits in FRONT of the streaming role in
    // its control-flow graph: the streaming waves then waited for those registers (vmcnt(15), 14, ...) between their DMA
    // instructions.  An explicit wait the compiler can see (the builtin, not inline asm) empties that state.
    if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);
}

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

// ------------------------------------------------------------------------------------------
// hetd2_wide_kernel: the END of the reduction in ONE workgroup, matrix resident in REGISTERS.
//
// The reference hands only the last 32 columns to a one-block kernel (zhetrd_gpu.F90:84-87, zhetd2_gpu.F90:3-203); every
// column above that costs two chip-wide dependent launches here (~9 us) however small the trailing matrix is.  A CU has
// 512 KB of vector registers (and 160 KB of LDS): the full Hermitian matrix of order 128 (complex) / 192 (real) fits in the
// registers of 512 threads.  Thread (tr, tc) of a 16 x 32 grid owns the entries H(tr + 16 p, tc + 32 q) -- a cyclic layout, so
// the work per column shrinks with the trailing order for every thread alike; BOTH triangles are kept and updated (the
// rank-2 update is symmetric), so y = H v needs no transposed products: a thread multiplies its rows with its columns
// of v and the 32 threads of a half wave (fixed tr, all tc) sum with DPP / permlane16 moves.  Column j, right to left, is
// LAPACK's zhetd2 'U' step: x = H(0:j-1, j) -> larfg -> y = tau H v -> w = y - 1/2 tau (y^H v) v -> H -= v w^H + w v^H.
// Two workgroup barriers per column: x published (every wave then derives the larfg scalars redundantly, as the panel
// kernels do), y published (every wave derives alpha and its w entries).  Vectors go through double-buffered LDS lines.
// Outputs as zhetd2_gpu: d, e, tau, reflectors in the upper triangle; the superdiagonal holds e inside the reference's final
// 32x32 block and the explicit 1 of the blocked part above it (zhetrd_gpu.F90:92), i.e. exactly the reference's layout.
// ------------------------------------------------------------------------------------------
constexpr int WTR = 16, WTC = 32;   // thread grid of hetd2_wide_kernel
// sum over the 32 lanes of each half wave, result in every lane of the half
__device__ __forceinline__ double half_sum32(double v) {
    v = row_sum16(v);
    double t = v;
    swap_rows(v, t);      // v = {r0, r0, r2, r2}, t = {r1, r1, r3, r3}
    return v + t;
}
__device__ __forceinline__ cplx half_sum32(cplx v) { return cplx{half_sum32(v.x), half_sum32(v.y)}; }

template <class T, int PC>
__global__ void __launch_bounds__(WTR * WTC) hetd2_wide_kernel(int n, T* A, int lda, double* d, double* e, T* tau) {
    constexpr int PR = 2 * PC, NMAX = WTC * PC;
    static_assert(WTR * PR == NMAX, "square matrix");
    __shared__ T xs[2][NMAX], ys[2][NMAX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int tr = tid >> 5, tc = tid & 31;
    const T zero = Tr<T>::zero();

    T a[PR][PC];
#pragma unroll
    for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            const int r = tr + WTR * p, cc = tc + WTC * q;
            const bool in = r < n && cc < n;
            const int rl = min(r, n - 1), cl = min(cc, n - 1);
            const T v = A[(size_t)min(rl, cl) + (size_t)max(rl, cl) * lda];      // stored upper triangle
            T h = sel(r <= cc, v, conj_(v));
            h = sel(r == cc, Tr<T>::realpart(h), h);
            a[p][q] = sel(in, h, zero);
        }

    for (int j = n - 1; j >= 1; --j) {
        const int buf = j & 1;
        const int qj = j >> 5, tcj = j & 31;
        const int kp = (j + WTR - 1) / WTR, kq = (j + WTC - 1) / WTC;   // register rows / columns that still hold rows, columns < j
        // ---- x = H(0:j-1, j): its owners publish it ----
        // (qj is uniform: a scalar branch per register column, static register indices inside)
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            if (q == qj && tc == tcj) {
#pragma unroll
                for (int p = 0; p < PR; ++p)
                    if (p < kp && tr + WTR * p < j) xs[buf][tr + WTR * p] = a[p][q];
            }
        }
        __syncthreads();
        // ---- larfg scalars, redundantly per wave (zhetd2_gpu.F90:60-100) ----
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < NMAX / 64; ++u) {
            const int r = lane + 64 * u;
            const T xv = xs[buf][min(r, j - 1)];
            ss += (r < j - 1) ? abs2_(xv) : 0.0;
        }
        ss = wave_sum(ss);
        double beta;
        T tauj, scale;
        larfg_scalars<T>(ss, xs[buf][j - 1], beta, tauj, scale);
        auto vat = [&](int r) -> T {      // v(r): scale * x(r), 1 at r = j-1, 0 from j on
            const T xv = xs[buf][min(r, j - 1)];
            return sel(r < j - 1, scale * xv, sel(r == j - 1, Tr<T>::one(), zero));
        };
        T vr[PR], vc[PC], wr[PR];
#pragma unroll
        for (int p = 0; p < PR; ++p) { vr[p] = zero; if (p < kp) vr[p] = vat(tr + WTR * p); }
#pragma unroll
        for (int q = 0; q < PC; ++q) { vc[q] = zero; if (q < kq) vc[q] = vat(tc + WTC * q); }
        const bool nz = !(real_(tauj) == 0.0 && imag_(tauj) == 0.0);
        if (nz) {
            // ---- y = H v on rows < j: own rows x own columns, then the half-wave sum over tc ----
#pragma unroll
            for (int p = 0; p < PR; ++p) {
                wr[p] = zero;
                if (p < kp) {
                    T acc = zero;
#pragma unroll
                    for (int q = 0; q < PC; ++q)
                        if (q < kq) fma_(acc, a[p][q], vc[q]);
                    wr[p] = half_sum32(acc);
                }
            }
            if (tc == 0) {
#pragma unroll
                for (int p = 0; p < PR; ++p)
                    if (tr + WTR * p < j) ys[buf][tr + WTR * p] = wr[p];
            }
            __syncthreads();
            // ---- alpha = -1/2 tau (p^H v), p = tau y; w = p + alpha v  (zhetd2_gpu.F90:120-160) ----
            T dd = zero;
#pragma unroll
            for (int u = 0; u < NMAX / 64; ++u) {
                const int r = lane + 64 * u;
                const T pv = sel(r < j, tauj * ys[buf][min(r, j - 1)], zero);
                fmac_(dd, pv, vat(r));
            }
            dd = wave_sum(dd);
            const T al = (-0.5 * tauj) * dd;
#pragma unroll
            for (int p = 0; p < PR; ++p)
                if (p < kp) wr[p] = sel(tr + WTR * p < j, tauj * wr[p] + al * vr[p], zero);
            // ---- H -= v w^H + w v^H on rows, columns < j ----
#pragma unroll
            for (int q = 0; q < PC; ++q) {
                if (q < kq) {
                    const int cc = tc + WTC * q;
                    const T wcq = sel(cc < j, tauj * ys[buf][min(cc, j - 1)] + al * vc[q], zero);
                    const T vcc = conj_(vc[q]), wcc = conj_(wcq);
#pragma unroll
                    for (int p = 0; p < PR; ++p) {
                        if (p < kp) {
                            T t = a[p][q];
                            t = t - (vr[p] * wcc + wr[p] * vcc);
                            a[p][q] = sel(tr + WTR * p == cc, Tr<T>::realpart(t), t);
                        }
                    }
                }
            }
        }
        // ---- column j now stores the reflector; superdiagonal as the reference leaves it ----
        const T sup = j < 32 ? Tr<T>::make(beta, 0.0) : Tr<T>::one();
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            if (q == qj) {
#pragma unroll
                for (int p = 0; p < PR; ++p) {
                    if (p < kp) {
                        const int r = tr + WTR * p;
                        a[p][q] = sel(tc == tcj && r < j, sel(r == j - 1, sup, vr[p]), a[p][q]);
                    }
                }
            }
        }
        if (tid == 0) { e[j - 1] = beta; tau[j - 1] = tauj; }
    }
    // ---- d, and the upper triangle back to A ----
#pragma unroll
    for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            const int r = tr + WTR * p, cc = tc + WTC * q;
            if (r < n && cc < n && r <= cc) {
                A[(size_t)r + (size_t)cc * lda] = a[p][q];
                if (r == cc) d[r] = real_(a[p][q]);
            }
        }
}
template <class T> constexpr int wide_pc() { return Tr<T>::cx ? 4 : 6; }          // order 128 (complex) / 192 (real)
template <class T> constexpr int wide_nmax() { return WTC * wide_pc<T>(); }

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
    // Workgroup b takes tiles b, b+G, b+2G, ... with G = ONE workgroup per CU: every workgroup streams through its tiles with the
    // next tile's loads in flight while the finishing wave closes the current one (round 4: G = 256 against the two-per-CU grid of
    // rounds 1-3: trd 68.0 -> 67.4 ms at C3, sweep 0.502 -> 0.509 of the HBM peak, batch rate 17.32 -> 17.65 problems/s; G = 272 /
    // 288 / 320 leave a ragged second round: 73-76 ms).  Inside a batch call an even smaller grid would gain another 0.8 % (G = 192:
    // 17.8 problems/s -- a launch that asks for fewer CUs leaves the other chains' kernels more room), but the per-workgroup
    // partial sums of v^H A v are added up in grid order, so a grid that depends on the calling mode would cost the bit-identity of
    // batch and single-problem results; one grid everywhere (profiles/r04_experiments.txt section 6).
    // (Spreading the tiles evenly over the minimum number of rounds was measured SLOWER in round 2.)
    long cap = c.hemv_blocks > 0 ? c.hemv_blocks : (long)c.n_cu;
    if (ntiles <= cap) return (int)ntiles;
    return (int)cap;
}

// Which data path the panel mat-vec of trailing order n takes: option "mv_dma" = the smallest order streamed through the LDS-DMA
// ring.  The ring's 16-byte pieces are one complex number or a PAIR of rows of a real column, so the real form wants an even
// leading dimension and 16-byte aligned bases.  Same operations in the same order either way: bit-identical results.
template <class T> static bool mv_dma_ok(const Ctx& c, int n, int lda, bool aligned) {
    if (c.mv_dma <= 0 || n < c.mv_dma) return false;
    return Tr<T>::cx ? true : (aligned && lda % 2 == 0);
}

template <class T> struct TrdScratch {
    T *xbuf, *P, *S, *Zp, *alphaSlot;
    double* NP;
    int ldp;
};

template <class T> static TrdScratch<T> trd_scratch(Ctx& c, int N, int prob = 0) {
    TrdScratch<T> s;
    int nt = (N + HT - 1) / HT;
    s.ldp = nt * HT;
    char nm[6][24];
    const char* base[6] = {"trd_xbuf", "trd_P", "trd_S", "trd_Zp", "trd_NP", "trd_alpha"};
    for (int q = 0; q < 6; ++q) {
        if (prob == 0) snprintf(nm[q], sizeof nm[q], "%s", base[q]);     // (the single-problem slot names of round 1)
        else snprintf(nm[q], sizeof nm[q], "%s#%d", base[q], prob);
    }
    s.xbuf = c.scratch<T>(nm[0], (size_t)nt * HT + 64);
    s.P = c.scratch<T>(nm[1], (size_t)nt * s.ldp);
    s.S = c.scratch<T>(nm[2], 8192);
    int nchunk = (N + CH - 1) / CH;
    s.Zp = c.scratch<T>(nm[3], (size_t)(nchunk + 1) * 2 * NBMAX);
    s.NP = c.scratch<double>(nm[4], (size_t)(N / RR + 1) * NPW + 64);
    s.alphaSlot = c.scratch<T>(nm[5], 8);
    return s;
}

// one problem of a lockstep batch: its matrix, outputs, panel workspace and private scratch
template <class T> struct TrdProb {
    T* A; double* d; double* e; T* tau; T* W;
    TrdScratch<T> sc;
};

template <class T, int NB>
static void latrd_panel(Ctx& c, hipStream_t st, int nprob, const TrdProb<T>* pr, int np, int nb, int lda, int ldw,
                        bool mv_only = false, long* nlaunch = nullptr, double* algo_bytes = nullptr) {
    PanelBatch<T, NB> ab;
    for (int q = 0; q < NB; ++q) {
        const TrdProb<T>& P = pr[q < nprob ? q : 0];
        PanelArgs<T>& a = ab.p[q];
        a.A = P.A; a.lda = lda; a.W = P.W; a.ldw = ldw; a.np = np; a.nb = nb; a.e = P.e; a.tau = P.tau;
        a.xbuf = P.sc.xbuf; a.P = P.sc.P; a.ldp = P.sc.ldp; a.S = P.sc.S; a.Zp = P.sc.Zp; a.NP = P.sc.NP; a.alphaSlot = P.sc.alphaSlot;
    }
    auto set_all = [&](auto f) { for (int q = 0; q < NB; ++q) f(ab.p[q]); };
    bool dma_aligned = true;    // (real path of the LDS-DMA form: 16-byte pieces = pairs of rows)
    for (int q = 0; q < nprob && q < NB; ++q) dma_aligned = dma_aligned && ((uintptr_t)pr[q].A % 16 == 0) && ((uintptr_t)pr[q].sc.xbuf % 16 == 0);
    int gh_prev = 0, nchunk_prev = 0;
    for (int i = np - 1; i >= np - nb - 1; --i) {
        const bool last = (i == np - nb - 1);  // finish-only pass for the panel's leftmost column
        const int do_finish = (i < np - 1), do_update = !last;
        set_all([&](PanelArgs<T>& a) { a.i = i; a.gh = gh_prev; a.nchunk = nchunk_prev; });
        int gA = (i + 1 + RR - 1) / RR;
        if (!mv_only) {
            const int cc = i + 1, npo = do_finish ? np - 1 - cc : 0, ntc = (cc + HT - 1) / HT;
            if (!do_finish) launch_row<T, false, true, NB>(st, gA, nprob, ab, npo, ntc);
            else if (do_update) launch_row<T, true, true, NB>(st, gA, nprob, ab, npo, ntc);
            else launch_row<T, true, false, NB>(st, gA, nprob, ab, npo, ntc);
        }
        if (last) break;
        // mat-vec for column i (v has i entries)
        int n = i;
        int gh = hemv_grid(c, n);
        int nchunk = (n + CH - 1) / CH;
        int npo = np - 1 - i;
        int gg = (2 * npo * nchunk + 3) / 4;
        set_all([&](PanelArgs<T>& a) { a.nblkA = gA * NPW; a.gh = gh; a.nchunk = nchunk; });
        if (mv_dma_ok<T>(c, n, lda, dma_aligned))
            hipLaunchKernelGGL((panel_mv_kernel<T, NB, true>), dim3(gh, nprob), dim3(MVD), 0, st, ab, 0, 0);
        else
            hipLaunchKernelGGL((panel_mv_kernel<T, NB, false>), dim3(gh + gg, nprob), dim3(MVT), 0, st, ab, 0, gg);
        if (nlaunch) ++*nlaunch;
        if (algo_bytes) *algo_bytes += (double)sizeof(T) * (double)n * (double)(n + 1) * 0.5;
        gh_prev = gh; nchunk_prev = nchunk;
    }
    EIG_HIP(hipGetLastError());
}

// Order at which the blocked reduction stops and the one-workgroup kernel takes over: option "trd_finish" (-1 = the largest
// order hetd2_wide_kernel holds, 32 = the reference's cut-over with the LDS kernel hetd2_kernel; values in between are allowed).
// (A multi-workgroup version of the register-resident finish -- up to 48 co-resident workgroups, one in-launch exchange per
// column, order <= 768 / 1024 -- was built in round 4, is parity-green, and costs 6.8-8.2 us per column against the 8.5-9.3 of
// the two-launch path: not kept.  profiles/r04_experiments.txt section 3, stamps in profiles/r04_multi_finish_stamps.txt.)
template <class T> static int finish_order(const Ctx& c) {
    const int nmax = wide_nmax<T>();
    if (c.trd_finish < 0) return nmax;
    return c.trd_finish <= TD ? TD : (c.trd_finish > nmax ? nmax : c.trd_finish);
}
template <class T> static void launch_finish(hipStream_t st, int nx, int n0, T* A, int lda, double* d, double* e, T* tau) {
    if (nx <= TD) hipLaunchKernelGGL((hetd2_kernel<T>), dim3(1), dim3(256), 0, st, n0, A, lda, d, e, tau);
    else hipLaunchKernelGGL((hetd2_wide_kernel<T, wide_pc<T>()>), dim3(1), dim3(WTR * WTC), 0, st, n0, A, lda, d, e, tau);
}

// nprob problems of order N reduced in lockstep: every per-column launch carries all of them (blockIdx.y); the trailing
// rank-2nb updates and the final blocks are launched per problem.  nprob = 1 is the plain zhetrd_gpu path.
template <class T, int NB>
static void hetrd_lockstep(Ctx& c, hipStream_t st, int N, int nprob, const TrdProb<T>* pr, int lda, int nb) {
    if (N <= 0) return;
    if (nb <= 0 || nb > NBMAX) nb = NBMAX;
    const int nx = finish_order<T>(c);
    const int ldw = N;
    int np = N;
    auto trailing = [&](int npn, int nbn) {
        for (int q = 0; q < nprob; ++q)
            her2k_un<T>(c, st, npn - nbn, nbn, pr[q].A + (size_t)(npn - nbn) * lda, lda, pr[q].W, ldw, pr[q].A, lda);
    };
    while (np - nb >= nx) {  // zhetrd_gpu.F90:60-71
        latrd_panel<T, NB>(c, st, nprob, pr, np, nb, lda, ldw);
        trailing(np, nb);
        np -= nb;
    }
    int nbr = np - nx;  // remainder panel, :73-83
    if (nbr > 0) {
        latrd_panel<T, NB>(c, st, nprob, pr, np, nbr, lda, ldw);
        trailing(np, nbr);
        np = nx;
    }
    int n0 = N < nx ? N : nx;
    for (int q = 0; q < nprob; ++q) {
        launch_finish<T>(st, nx, n0, pr[q].A, lda, pr[q].d, pr[q].e, pr[q].tau);   // :86-87
        if (N > n0)
            hipLaunchKernelGGL((diag_extract_kernel<T>), dim3((N - n0 + 255) / 256), dim3(256), 0, st, n0, N, (const T*)pr[q].A, lda,
                               pr[q].d);
    }
    EIG_HIP(hipGetLastError());
}

template <class T>
void hetrd_upper(Ctx& c, hipStream_t st, int N, T* A, int lda, double* d, double* e, T* tau, T* W, int nb) {
    if (N <= 0) return;
    TrdProb<T> pr{A, d, e, tau, W, trd_scratch<T>(c, N)};
    hetrd_lockstep<T, 1>(c, st, N, 1, &pr, lda, nb);
}

template <class T>
void hetrd_upper_batch(Ctx& c, hipStream_t st, int N, int nprob, T* const* A, int lda, double* const* d, double* const* e,
                       T* const* tau, T* const* W, int nb) {
    if (N <= 0 || nprob <= 0) return;
    if (nprob == 1) { hetrd_upper<T>(c, st, N, A[0], lda, d[0], e[0], tau[0], W[0], nb); return; }
    for (int q0 = 0; q0 < nprob; q0 += MAXB) {
        const int nq = nprob - q0 < MAXB ? nprob - q0 : MAXB;
        TrdProb<T> pr[MAXB];
        for (int q = 0; q < nq; ++q)
            pr[q] = TrdProb<T>{A[q0 + q], d[q0 + q], e[q0 + q], tau[q0 + q], W[q0 + q], trd_scratch<T>(c, N, q)};
        if (nq == 1) hetrd_lockstep<T, 1>(c, st, N, 1, pr, lda, nb);
        else hetrd_lockstep<T, MAXB>(c, st, N, nq, pr, lda, nb);
    }
}

// Roofline leg: the exact sequence of panel_mv_kernel launches of a full tridiagonalization
// (same grids, same panel state layout, row kernels and her2k skipped; A is left numerically
// meaningless).  Returns launches and algorithmic bytes sum_n s*n(n+1)/2.
template <class T>
void hetrd_mv_sweep(Ctx& c, hipStream_t st, int N, T* A, int lda, T* W, int nb, double* e, T* tau, long* nlaunch, double* algo_bytes) {
    if (nb <= 0 || nb > NBMAX) nb = NBMAX;
    TrdScratch<T> sc = trd_scratch<T>(c, N);
    std::vector<double> ones((size_t)(N / RR + 1) * NPW + 64, 1.0);
    EIG_HIP(hipMemcpyAsync(sc.NP, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice, st));
    T one = Tr<T>::one();
    EIG_HIP(hipMemcpyAsync(sc.alphaSlot, &one, sizeof(T), hipMemcpyHostToDevice, st));
    EIG_HIP(hipMemcpyAsync(sc.xbuf, A, sizeof(T) * N, hipMemcpyDeviceToDevice, st));
    c.sync(st);
    *nlaunch = 0; *algo_bytes = 0.0;
    const int nx = finish_order<T>(c);
    int np = N;
    double* dsink = c.scratch<double>("sweep_d", (size_t)N + 8);
    TrdProb<T> pr{A, dsink, e, tau, W, sc};
    auto panel = [&](int npn, int nbn) { latrd_panel<T, 1>(c, st, 1, &pr, npn, nbn, lda, N, true, nlaunch, algo_bytes); };
    while (np - nb >= nx) {
        panel(np, nb);
        np -= nb;
    }
    int nbr = np - nx;
    if (nbr > 0) panel(np, nbr);
}

// Roofline leg: the exact sequence of trailing rank-2nb updates of a full tridiagonalization (zhetrd_gpu.F90:67), back to back:
// same orders, panel widths, operand placement (V = the panel's columns of A, W = the panel workspace) as hetrd_lockstep issues.
// A and W hold whatever the caller put there.  Returns launches and the model flops c * 2 * n^2 * k per update.
template <class T>
void hetrd_her2k_sweep(Ctx& c, hipStream_t st, int N, T* A, int lda, T* W, int nb, long* nlaunch, double* flops) {
    if (nb <= 0 || nb > NBMAX) nb = NBMAX;
    const int nx = finish_order<T>(c);
    const double cm = Tr<T>::cx ? 4.0 : 1.0;
    *nlaunch = 0; *flops = 0.0;
    auto trailing = [&](int npn, int nbn) {
        const int n = npn - nbn;
        if (n <= 0) return;
        her2k_un<T>(c, st, n, nbn, A + (size_t)n * lda, lda, W, N, A, lda);
        *nlaunch += 1;
        *flops += cm * 2.0 * (double)n * n * nbn;
    };
    int np = N;
    while (np - nb >= nx) { trailing(np, nb); np -= nb; }
    if (np - nx > 0) trailing(np, np - nx);
}
template void hetrd_her2k_sweep<double>(Ctx&, hipStream_t, int, double*, int, double*, int, long*, double*);
template void hetrd_her2k_sweep<cplx>(Ctx&, hipStream_t, int, cplx*, int, cplx*, int, long*, double*);

template <class T> void hemv_upper(Ctx& c, hipStream_t st, int n, const T* A, int lda, const T* x, T* y, bool gather) {
    if (n <= 0) return;
    TrdScratch<T> sc = trd_scratch<T>(c, n);
    PanelBatch<T, 1> ab;
    PanelArgs<T>& a = ab.p[0];
    a.A = const_cast<T*>(A); a.lda = lda; a.W = nullptr; a.ldw = 0; a.np = n + 1; a.nb = 1; a.i = n;
    a.e = nullptr; a.tau = nullptr; a.xbuf = const_cast<T*>(x); a.P = sc.P; a.ldp = sc.ldp; a.S = sc.S; a.Zp = sc.Zp;
    a.NP = sc.NP; a.alphaSlot = sc.alphaSlot; a.nblkA = 0; a.nchunk = 0;
    a.gh = hemv_grid(c, n);
    if (mv_dma_ok<T>(c, n, lda, (uintptr_t)A % 16 == 0)) {
        // (the DMA reads v in 16-byte pieces up to the end of the last tile: from the library's own padded buffer, not the caller's x)
        EIG_HIP(hipMemcpyAsync(sc.xbuf, x, sizeof(T) * (size_t)n, hipMemcpyDeviceToDevice, st));
        a.xbuf = sc.xbuf;
        hipLaunchKernelGGL((panel_mv_kernel<T, 1, true>), dim3(a.gh), dim3(MVD), 0, st, ab, 1, 0);
    } else {
        hipLaunchKernelGGL((panel_mv_kernel<T, 1, false>), dim3(a.gh), dim3(MVT), 0, st, ab, 1, 0);
    }
    if (gather) {
        int nt = (n + HT - 1) / HT;
        hipLaunchKernelGGL((hemv_gather_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, n, nt, (const T*)sc.P, sc.ldp, y);
    }
    EIG_HIP(hipGetLastError());
}

template void hetrd_upper<double>(Ctx&, hipStream_t, int, double*, int, double*, double*, double*, double*, int);
template void hetrd_upper<cplx>(Ctx&, hipStream_t, int, cplx*, int, double*, double*, cplx*, cplx*, int);
template void hetrd_upper_batch<double>(Ctx&, hipStream_t, int, int, double* const*, int, double* const*, double* const*, double* const*,
                                        double* const*, int);
template void hetrd_upper_batch<cplx>(Ctx&, hipStream_t, int, int, cplx* const*, int, double* const*, double* const*, cplx* const*,
                                      cplx* const*, int);
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
