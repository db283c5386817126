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
__device__ unsigned long long g_trd_count[4];       // kernel 2 / 3: panel_col_kernel, an owner tile (block 0) / an off-diagonal tile (block 1)
#define CSTAMP(PH) do { if (blockIdx.x < 2 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&g_trd_stamp[2 + blockIdx.x][PH], (unsigned long long)(__builtin_readcyclecounter() - CT0)); } while (0)
#define TSTAMP(KID, PH, T0) do { if (blockIdx.x == (KID == 0 ? gg : 0) && threadIdx.x == 0) atomicAdd(&g_trd_stamp[KID][PH], (unsigned long long)(__builtin_readcyclecounter() - (T0))); } while (0)
#else
#define TSTAMP(KID, PH, T0) do { } while (0)
#define CSTAMP(PH) do { } while (0)
#endif
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
    int wt;            // 1: the hemv partials are stored write-through (sc1), see store_partial
};

// The hemv partials P (up to nt*n*s bytes = 4 MB at n = 4096) are the only sizeable data the mat-vec kernel writes, and
// the next kernel (on other XCDs) reads all of it.  A plain store leaves the lines dirty in this XCD's L2 and the kernel
// boundary pays for their write-back (MI355X_MICROARCH.md, row "boundary": + B / 6 TB/s); an sc1 store writes through
// while the kernel is still streaming.
__device__ __forceinline__ void store_partial(cplx* p, cplx v, int wt) {
    if (wt) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 t = {v.x, v.y};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
    } else {
        *p = v;
    }
}
__device__ __forceinline__ void store_partial(double* p, double v, int wt) {
    if (wt) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}

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
            l_p0 = a.P[(size_t)min(lane, ntc - 1) * a.ldp + i];
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
        for (int u = 0; u < RP; ++u) pt[u] = a.P[(size_t)min(g + RG * u, ntc - 1) * a.ldp + rc];
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
            for (int q = lane + 64; q < ntc; q += 64) wi_p = wi_p + a.P[(size_t)q * a.ldp + i];   // N > 4096 only
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
        for (int q = g + RG * RP; q < ntc; q += RG) acc2 = acc2 + sel(active, a.P[(size_t)q * a.ldp + rc], zero);
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

template <class T, int NB>
__global__ void __launch_bounds__(MVT, 3) panel_mv_kernel(PanelBatch<T, NB> ab, int plain, int gg) {
    const PanelArgs<T>& a = ab.p[NB == 1 ? 0 : blockIdx.y];
    const int i = a.i, n = i;  // v has n entries (rows 0..i-1), v(n-1) = 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if EIG_TRD_TIMING
    const long long T0 = __builtin_readcyclecounter();
    if (blockIdx.x == gg && threadIdx.x == 0) atomicAdd(&g_trd_count[0], 1ULL);
#endif
    // Five waves: 0-3 stream and multiply the tiles, wave 4 evaluates the larfg scalars (a serial chain of ~1500 cycles)
    // while the first tile is being multiplied and finishes every tile (partial sums, S, the stored v) while the
    // others are already on the next one.  LDS hand-over buffers are double (partials) / triple (column entries of
    // v) buffered so that one barrier per tile is enough.  Two workgroups per CU need <= 168 VGPRs per wave (10 waves
    // on 4 SIMDs): the column reduction runs in two halves of 8 columns to stay below that.
    __shared__ T redy[2][4][64];
    __shared__ T redt[2][64];
    __shared__ T xcs[3][HT];   // xh entries of the tile columns: tile k of this workgroup uses slot k % 3
    constexpr int NPL = 16;    // norm partials per lane loaded up front (N <= 4096 without the tail loop)
    const bool is_gemv = (int)blockIdx.x < gg;
    const int hb = (int)blockIdx.x - gg;  // hemv workgroup index
    if (is_gemv && wave == 4) return;

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
    int I = 0, J = 0, t = hb;
    if (!is_gemv && wave < 4) {
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
        while (t < ntiles) {
            const int r0 = I * HT, c0 = J * HT;
            const bool diag = (I == J);
            const int r = r0 + lane;
            const int rb = k & 1, xs = k % 3;
            const T xr = sel(r < nz, xr_raw, zero);
            const bool interior = !diag && r0 + HT <= n && c0 + HT <= n;
            T yI = Tr<T>::zero();
            // 8 columns at a time: products, then the first two levels of the column reduction
            auto half = [&](int jb, T& w0, T& w1) {
                T tj[8];
                if (interior) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fma_(yI, av[jb + j], xcs[xs][wave * 16 + jb + j]);
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
                        fma_(yI, v, xcs[xs][wave * 16 + jb + j]);
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
            const T tval = transpose_reduce_phase2<T>(wa0, wa1, wb0, wb1, lane);
            TSTAMP(0, 3, T0);   // tile loads arrived, FMAs + transpose-reduce done
            redy[rb][wave][lane] = yI;
            if ((lane & 3) == 0) redt[rb][wave * 16 + transpose_col_of_lane(lane)] = tval;
            t += a.gh;
            ++k;
            if (t < ntiles) issue_tile(k % 3);   // next tile's loads fly while wave 4 finishes this one
            TSTAMP(0, 4, T0);
            __syncthreads();
        }
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
        const int npo = a.np - 1 - i;
        const int wbase = a.np - a.nb;
        const int item = (int)blockIdx.x * 4 + wave;
        if (item >= 2 * npo * a.nchunk) return;
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
        const T scale = scalars();
        s = wave_sum(scale * s + eone);
        if (lane == 0) a.Zp[(size_t)(ch * 2 + which) * NBMAX + kk] = s;
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
    __syncthreads();
    T scale = Tr<T>::one();
    if (!plain) scale = scalars();   // while waves 0-3 multiply the first tile
    TSTAMP(0, 2, T0);
    T Sacc = Tr<T>::zero();
    int k = 0;
    while (t < ntiles) {
        const int r0 = I * HT, c0 = J * HT, Ic = I, Jc = J;
        const bool diag = (I == J);
        const int r = r0 + lane;
        const T xr = sel(r < nz, xr_raw, zero);
        T lastc = sel(!plain && J == nt - 1 && r <= n - 1, lc_raw, zero);
        lastc = sel(r == n - 1, Tr<T>::realpart(lastc), lastc);
        t += a.gh;
        if (t < ntiles) issue_mine();
        __syncthreads();   // partial sums of tile k are in redy / redt [k & 1]
        const int rb = k & 1, xs = k % 3;
        T yv = scale * ((redy[rb][0][lane] + redy[rb][1][lane]) + (redy[rb][2][lane] + redy[rb][3][lane])) + lastc;
        T tv = scale * redt[rb][lane];
        T vI = scale * xr + unit(r0 + lane);
        T vJ = scale * xcs[xs][lane] + unit(c0 + lane);
        if (diag) {
            T s = yv + tv;
            store_partial(&a.P[(size_t)Jc * a.ldp + r0 + lane], s, a.wt);
            fmac_(Sacc, vI, s);
            if (!plain && c0 + lane < n) a.A[(size_t)(c0 + lane) + (size_t)i * a.lda] = vJ;
        } else {
            store_partial(&a.P[(size_t)Jc * a.ldp + r0 + lane], yv, a.wt);
            store_partial(&a.P[(size_t)Ic * a.ldp + c0 + lane], tv, a.wt);
            fmac_(Sacc, vI, yv);
            fmac_(Sacc, vJ, tv);
        }
        ++k;
    }
    Sacc = wave_sum(Sacc);
    if (lane == 0) a.S[hb] = Sacc;
    TSTAMP(0, 5, T0);
}

// ------------------------------------------------------------------------------------------
// panel_col_kernel : ONE launch per column for the tail of the reduction (order <= "trd_fuse_n").
//
// Up to n ~ 1300 (complex) a column's two kernels cost 4.3-5 us each whatever the work: launch ramp, one dependent memory
// round trip, a kernel boundary.  Here the row work of panel_row_kernel is folded into the mat-vec launch: a workgroup owns
// one 64x64 tile (I, J) of the upper triangle and FIRST derives, redundantly, the 2 x 64 entries x_I, x_J of the updated
// column it is about to multiply with (finishing W(:, c) for those rows on the way) from data of the previous launch --
// O((2 npo + nt) x 128) values from L2 -- while its tile is already in flight from HBM.  Diagonal tiles are the OWNERS of
// their 64 rows: only they store W(:, c), the finished reflector v_c, the raw new column and the per-block partial sums.
//
// What makes one launch per column possible is linearity: v = scale * xh + e_(n-1) (xh = raw column, last entry zeroed).
// scale needs ||xh|| -- a reduction over ALL rows, i.e. over all workgroups of the launch that produces xh -- so a launch
// never applies its own column's scalars: it multiplies the tile with the RAW xh and publishes raw partials
//     P^  = A xh (per tile stripe),  S^ = xh^H A xh,  D^ = xh^H A(:, n-1),  Z^ = [V W]^H xh (per owner block),  ||xh||^2,
// and the NEXT launch, which can sum the norm partials, reconstructs
//     y = scale P^ + A(:, n-1),   z = scale Z^ + [V W](n-1, :)^H,   v^H A v = |scale|^2 S^ + 2 Re(conj(scale) D^) + A(n-1, n-1)
// before it finishes W(:, c) = tau (y - V z2' - W z1') + alpha v_c exactly as panel_row_kernel does (zhetrd_gpu.F90:335-511,
// :750-879).  Partial-sum buffers are double (a launch reads the previous column's set and writes its own).
//   FIRST   : first column of a panel -- nothing to finish, x = A(:, i) as the trailing update left it;
//   FINONLY : after the last column of a panel -- only the owners run: W(:, c), v_c, e, tau; no new column, no mat-vec.
// ------------------------------------------------------------------------------------------
template <class T> struct ColArgs {
    T* A; int lda;
    T* W; int ldw;
    int np, nb, i;           // i = column generated by this launch (c = i + 1 is finished by it); FINONLY: i = c - 1
    double* d; double* e; T* tau;
    // set of the previous launch (describes column c) / set written by this launch (describes column i)
    const T* xprev; T* xnew;
    const T* Pp; T* Pn; int ldp;
    const T* Sp; T* Sn;
    const T* Dp; T* Dn;
    const double* NPp; double* NPn;
    const T* Zpp; T* Zpn;
    const T* alphap; T* alphan;
};
template <class T, int NB> struct ColBatch {
    ColArgs<T> p[NB];
};

constexpr int CTH = 320;    // panel_col_kernel: four row / tile waves + one scalar wave
constexpr int ZG = 17;      // Z^ partial blocks per lane without the tail loop (2 halves x 17 >= 33 blocks: order <= 2111)

template <class T, int NB, bool FIRST, bool FINONLY>
__global__ void __launch_bounds__(CTH) panel_col_kernel(ColBatch<T, NB> ab) {
    const ColArgs<T>& a = ab.p[NB == 1 ? 0 : blockIdx.y];
#if EIG_TRD_TIMING
    const long long CT0 = __builtin_readcyclecounter();
    if (!FIRST && !FINONLY && blockIdx.x < 2 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&g_trd_count[2 + blockIdx.x], 1ULL);
#endif
    const int i = a.i, c = i + 1;
    const int n = i;                                   // order of the mat-vec: v_i has rows 0 .. i-1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T zero = Tr<T>::zero();
    int I, J;
    if (FINONLY) { I = J = (int)blockIdx.x; }
    else tile_decode((int)blockIdx.x, I, J);
    const bool owner = (I == J);
    const int r0 = I * HT, c0 = J * HT;
    const int wbase = a.np - a.nb;
    const int npo = FIRST ? 0 : a.np - 1 - c;          // panel columns older than c
    const int ntc = c / HT + 1;                        // 64-row blocks of the previous launch (rows 0 .. c)
    const int nz = n - 1;                              // xh: rows >= nz are zero

    __shared__ T zs[2][2][NBMAX];                      // [which][half][kk] gathered partial sums of Z^ (raw)
    __shared__ T rowW[NBMAX + 1], rowV[NBMAX + 1];     // conj of row i of W / V (older columns), [npo] = column c itself
    __shared__ T s4[4];
    __shared__ T part[2][4][3][HT];                    // [block sel][sub][psum / Q1 / Q2][row]
    __shared__ T xs[2][HT];                            // raw new column: rows of block I / block J
    __shared__ T scal[4];                              // scale, tau, alpha (scalar wave -> everybody)
    __shared__ T redy[4][HT], redt[HT];
    __shared__ T vcs[HT], wcs[HT];                     // v_c, w_c of the owner's rows (for the Z^ partials)

    // row-work roles of waves 0-3.  Off-diagonal tile: waves 0,1 -> rows of block I, waves 2,3 -> rows of block J, the two
    // waves of a block split the panel columns / stripes by parity; diagonal tile (and FINONLY): all four on block I.
    const int bsel = owner ? 0 : ((wave >> 1) & 1);
    const int nsub = owner ? 4 : 2;
    const int sub = owner ? (wave & 3) : (wave & 1);
    const int rb0 = (bsel == 0) ? r0 : c0;
    const int r = rb0 + lane;
    const int rows = i + 1;                            // rows 0 .. i of the panel are live
    const bool active = r < rows;
    const size_t rc = (size_t)min(r, rows - 1);

    // =========================== the scalar wave ===========================
    // larfg scalars of column c, z totals, alpha, w_i: a serial chain (~2500 cycles) that runs beside the row work
    if (wave == 4) {
        if constexpr (!FIRST) {
            T l_ww = zero, l_wv = zero;
            if (lane < npo) {
                const int k = c + 1 + lane;
                l_ww = a.W[(size_t)i + (size_t)(k - wbase) * a.ldw];
                l_wv = a.A[(size_t)i + (size_t)k * a.lda];
            }
            const int ql = min(lane, ntc - 1);
            T l_p = a.Pp[(size_t)ql * a.ldp + i];
            double npv = a.NPp[ql];
            T dv = a.Dp[ql];
            const T alpha_e = *a.alphap;
            const double aLL = real_(a.A[(size_t)i + (size_t)i * a.lda]);
            l_p = sel(lane < ntc, l_p, zero); npv = lane < ntc ? npv : 0.0; dv = sel(lane < ntc, dv, zero);
            for (int q = lane + 64; q < ntc; q += 64) { l_p = l_p + a.Pp[(size_t)q * a.ldp + i]; npv += a.NPp[q]; dv = dv + a.Dp[q]; }
            if (lane < npo) { rowW[lane] = conj_(l_ww); rowV[lane] = conj_(l_wv); }
            __syncthreads();                           // #1: Z^ halves and S^ partials are in LDS, rowW / rowV published
            const double ss = wave_sum(npv);
            double beta;
            T tau, scale;
            larfg_scalars<T>(ss, alpha_e, beta, tau, scale);
            if (blockIdx.x == 0 && lane == 0) { a.e[c - 1] = beta; a.tau[c - 1] = tau; }
            const T Dsum = wave_sum(dv);
            const T Sraw = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            T cs = zero;
            fmac_(cs, scale, Dsum);
            // v^H A v = |scale|^2 S^ + 2 Re(conj(scale) D^) + A(n-1, n-1)
            const double S = abs2_(scale) * real_(Sraw) + 2.0 * real_(cs) + aLL;
            T z1l = zero, z2l = zero;
            if (lane < npo) {
                z1l = scale * (zs[0][0][lane] + zs[0][1][lane]) + conj_(l_wv);
                z2l = scale * (zs[1][0][lane] + zs[1][1][lane]) + conj_(l_ww);
            }
            T t = zero;
            fmac_(t, z1l, z2l);
            const double zz = wave_sum(real_(t));
            const T alpha = Tr<T>::make((-0.5 * abs2_(tau)) * (S - 2.0 * zz), 0.0);   // (v^H A v is real for Hermitian A)
            // w_i = W(i, c): row i of the column being finished;  y_i = scale P^(i) + A(i, i),  v_c(i) = 1
            const T yi = scale * wave_sum(l_p) + Tr<T>::make(aLL, 0.0);
            const T u = wave_sum(sel(lane == 0, yi, zero) - (l_ww * z1l + l_wv * z2l));
            const T wi = tau * u + alpha;
            if (lane == 0) { rowW[npo] = conj_(wi); rowV[npo] = Tr<T>::one(); scal[0] = scale; scal[1] = tau; scal[2] = alpha; }
        } else {
            __syncthreads();                           // #1
        }
        __syncthreads();                               // #2: scalars published, row-work partial sums published
        if constexpr (FINONLY) return;
        __syncthreads();                               // #3: xs published
        __syncthreads();                               // #4: tile partial sums published
        // finish the tile: raw partials of the new column
        const T yv = (redy[0][lane] + redy[1][lane]) + (redy[2][lane] + redy[3][lane]);
        const T tv = redt[lane];
        const T xI = sel(r0 + lane < nz, xs[0][lane], zero);
        const T xJ = sel(c0 + lane < nz, xs[1][lane], zero);
        T Sacc = zero;
        if (owner) {
            const T sm = yv + tv;
            a.Pn[(size_t)J * a.ldp + r0 + lane] = sm;
            fmac_(Sacc, xI, sm);
        } else {
            a.Pn[(size_t)J * a.ldp + r0 + lane] = yv;
            a.Pn[(size_t)I * a.ldp + c0 + lane] = tv;
            fmac_(Sacc, xI, yv);
            fmac_(Sacc, xJ, tv);
        }
        Sacc = wave_sum(Sacc);
        if (lane == 0) a.Sn[blockIdx.x] = Sacc;
        if (!FIRST) CSTAMP(6);
        return;
    }

    // =========================== waves 0-3 ===========================
    // ---- every load that depends on nothing: the tile, the gather of Z^ / S^, the first row-work loads ----
    T av[16];
    if constexpr (!FINONLY) {
        const size_t roff = (size_t)min(r0 + lane, max(n - 1, 0));
#pragma unroll
        for (int j = 0; j < 16; ++j) av[j] = a.A[roff + (size_t)min(c0 + wave * 16 + j, max(n - 1, 0)) * a.lda];
    }
    T xr_new = zero, wr = zero, vr = zero;
    if constexpr (FIRST) {
        T acur = a.A[rc + (size_t)i * a.lda];
        if (r == i) acur = Tr<T>::realpart(acur);
        xr_new = sel(active, acur, zero);
        __syncthreads();                               // #1
        __syncthreads();                               // #2
    } else {
        {
            // gather: Z^ (lane = panel column, waves = which x half of the blocks), S^ (all threads of waves 0-3)
            const int which = wave & 1, half = wave >> 1;
            const int kz = min(lane, max(npo - 1, 0));
            T zt[ZG];
#pragma unroll
            for (int u = 0; u < ZG; ++u) zt[u] = a.Zpp[(size_t)(min(half + 2 * u, ntc - 1) * 2 + which) * NBMAX + kz];
            const int ntl = ntc * (ntc + 1) / 2;
            T st[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) st[u] = a.Sp[min(tid + 256 * u, ntl - 1)];
            __builtin_amdgcn_sched_barrier(0);
            T zsum = zero, Ssum = zero;
#pragma unroll
            for (int u = 0; u < ZG; ++u) zsum = zsum + sel(half + 2 * u < ntc, zt[u], zero);
            for (int q = half + 2 * ZG; q < ntc; q += 2) zsum = zsum + a.Zpp[(size_t)(q * 2 + which) * NBMAX + kz];
#pragma unroll
            for (int u = 0; u < 3; ++u) Ssum = Ssum + sel(tid + 256 * u < ntl, st[u], zero);
            for (int q = tid + 768; q < ntl; q += 256) Ssum = Ssum + a.Sp[q];
            if (lane < npo) zs[which][half][lane] = zsum;
            Ssum = wave_sum(Ssum);
            if (lane == 0) s4[wave] = Ssum;
        }
        if (!FINONLY) CSTAMP(0);
        __syncthreads();                               // #1
        if (!FINONLY) CSTAMP(1);
        // ---- row sums that need no scalar:  psum = sum of P^ stripes,  Q1 = sum W z1^ + V z2^,  Q2 = sum V conj(W(i,:)) + W conj(V(i,:))
        //      (y = scale psum + A(:, i);  w = tau (y - scale Q1 - Q2) + alpha v_c;  column update = Q2 + v_c conj(w_i) + w) ----
        constexpr int KB = Tr<T>::cx ? 4 : 8;          // panel columns per load batch (two batches in flight)
        T psum = zero, q1 = zero, q2 = zero;
        {
            T pt[ZG];
#pragma unroll
            for (int u = 0; u < ZG; ++u) {
                const int q = sub + nsub * u;
                pt[u] = (u * nsub < 2 * ZG) ? a.Pp[(size_t)min(q, ntc - 1) * a.ldp + rc] : zero;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < ZG; ++u) psum = psum + sel(sub + nsub * u < ntc && u * nsub < 2 * ZG, pt[u], zero);
            for (int q = sub + nsub * ZG; q < ntc; q += nsub) psum = psum + a.Pp[(size_t)q * a.ldp + rc];
        }
        {
            T vb[2][KB], wb[2][KB];
            const int nmine = (npo - sub + nsub - 1) / nsub;       // my panel columns: kk = sub + nsub * m
            auto issue = [&](int buf, int m0) {
#pragma unroll
                for (int m = 0; m < KB; ++m) {
                    const int kk = min(sub + nsub * (m0 + m), max(npo - 1, 0));
                    const int k = c + 1 + kk;
                    vb[buf][m] = a.A[rc + (size_t)k * a.lda];
                    wb[buf][m] = a.W[rc + (size_t)(k - wbase) * a.ldw];
                }
            };
            auto consume = [&](int buf, int m0) {
#pragma unroll
                for (int m = 0; m < KB; ++m) {
                    const int kk = sub + nsub * (m0 + m);
                    if (m0 + m < nmine) {
                        const T z1r = zs[0][0][kk] + zs[0][1][kk], z2r = zs[1][0][kk] + zs[1][1][kk];
                        q1 = q1 + (wb[buf][m] * z1r + vb[buf][m] * z2r);
                        q2 = q2 + (vb[buf][m] * rowW[kk] + wb[buf][m] * rowV[kk]);
                    }
                }
            };
            if (nmine > 0) issue(0, 0);
            for (int m0 = 0; m0 < nmine; m0 += 2 * KB) {
                if (m0 + KB < nmine) issue(1, m0 + KB);
                consume(0, m0);
                if (m0 + 2 * KB < nmine) issue(0, m0 + 2 * KB);
                if (m0 + KB < nmine) consume(1, m0 + KB);
            }
        }
        part[bsel][sub][0][lane] = psum;
        part[bsel][sub][1][lane] = q1;
        part[bsel][sub][2][lane] = q2;
        const T xc_raw = a.xprev[rc];
        T acur = a.A[rc + (size_t)i * a.lda];          // column i: A(r, n_c - 1) for y AND the column to update
        if (r == i) acur = Tr<T>::realpart(acur);
        __syncthreads();                               // #2
        if (!FINONLY) CSTAMP(3);
        const T scale = scal[0], tau = scal[1], alpha = scal[2];
        T ps = part[bsel][0][0][lane] + part[bsel][1][0][lane];
        T a1 = part[bsel][0][1][lane] + part[bsel][1][1][lane];
        T a2 = part[bsel][0][2][lane] + part[bsel][1][2][lane];
        if (owner) {
            ps = ps + (part[0][2][0][lane] + part[0][3][0][lane]);
            a1 = a1 + (part[0][2][1][lane] + part[0][3][1][lane]);
            a2 = a2 + (part[0][2][2][lane] + part[0][3][2][lane]);
        }
        vr = (r < c - 1) ? scale * xc_raw : ((r == c - 1) ? Tr<T>::one() : zero);
        const T y = scale * ps + acur;
        wr = tau * (y - scale * a1 - a2) + alpha * vr;
        if constexpr (!FINONLY) {
            const T upd = a2 + (vr * rowW[npo] + wr * rowV[npo]);
            T anew = acur - upd;
            if (r == i) anew = Tr<T>::realpart(anew);
            xr_new = sel(active, anew, zero);
        }
        if (owner && wave == 0 && active) {            // owners publish the finished column c
            a.W[(size_t)r + (size_t)(c - wbase) * a.ldw] = wr;
            a.A[(size_t)r + (size_t)c * a.lda] = vr;
        }
    }
    if constexpr (FINONLY) return;

    // ---------------- publish the raw new column (LDS for this tile, global by the owners) ----------------
    if (owner) {
        if (wave == 0) { xs[0][lane] = xr_new; xs[1][lane] = xr_new; vcs[lane] = vr; wcs[lane] = wr; }
    } else if ((wave & 1) == 0) {
        xs[bsel][lane] = xr_new;
    }
    if (owner && wave == 0) {
        if (active) {
            a.xnew[r] = xr_new;
            if (r == i - 1) *a.alphan = xr_new;
            if (r == i) a.d[i] = real_(xr_new);        // (A(i,i) itself stays as it is: every workgroup of this launch reads it)
        }
        const T alast = a.A[(size_t)min(r, max(n - 1, 0)) + (size_t)max(n - 1, 0) * a.lda];   // A(r, n-1), used for r < nz
        double nrm = (active && r <= i - 2) ? abs2_(xr_new) : 0.0;
        T dd = zero;
        if (active && r < nz) fmac_(dd, xr_new, alast);
        nrm = wave_sum(nrm);
        dd = wave_sum(dd);
        if (lane == 0) { a.NPn[I] = nrm; a.Dn[I] = dd; }
    }
    __syncthreads();                                   // #3
    if (!FIRST) CSTAMP(4);

    // ---------------- the tile: y_I += A_IJ xh_J,  y_J += A_IJ^H xh_I  (as panel_mv_kernel, raw xh) ----------------
    {
        const bool diag = owner;
        const int rr = r0 + lane;
        const T xr = sel(rr < nz, xs[0][lane], zero);
        const bool interior = !diag && r0 + HT <= n && c0 + HT <= n && c0 + HT <= nz;
        T yI = zero;
        auto half = [&](int jb, T& w0, T& w1) {
            T tj[8];
            if (interior) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    fma_(yI, av[jb + j], xs[1][wave * 16 + jb + j]);
                    T p = zero;
                    fmac_(p, av[jb + j], xr);
                    tj[j] = p;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cc = c0 + wave * 16 + jb + j;
                    const bool ok = (rr < n) && (cc < n) && (!diag || rr <= cc);
                    const bool dg = diag && rr == cc;
                    T v = sel(ok, av[jb + j], zero);
                    v = sel(dg, Tr<T>::realpart(v), v);
                    fma_(yI, v, sel(cc < nz, xs[1][wave * 16 + jb + j], zero));
                    T p = zero;
                    fmac_(p, sel(dg, zero, v), xr);
                    tj[j] = p;
                }
            }
            transpose_reduce8_phase1<T>(tj, w0, w1);
        };
        T wa0, wa1, wb0, wb1;
        half(0, wa0, wa1);
        half(8, wb0, wb1);
        const T tval = transpose_reduce_phase2<T>(wa0, wa1, wb0, wb1, lane);
        redy[wave][lane] = yI;
        if ((lane & 3) == 0) redt[wave * 16 + transpose_col_of_lane(lane)] = tval;
    }
    __syncthreads();                                   // #4 (the scalar wave finishes the tile)
    if (!FIRST) CSTAMP(5);
    // ---------------- owners: Z^ partials of the new column over their 64 rows ([V W]^H xh for the columns older than i) ----
    if (owner) {
        const int npn = a.np - 1 - i;                  // panel columns older than i: k = c + kk, kk = 0 .. npn-1
        const int rr = r0 + lane;
        const T xh = sel(rr < nz, xs[0][lane], zero);
        const size_t rcl = (size_t)min(rr, max(n - 1, 0));
        const int nitem = 2 * npn;                     // items: [V columns kk = 0 .. npn-1 | W columns]; 16 per wave and pass
        for (int g0 = wave * 16; g0 < nitem; g0 += 64) {
            T src[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int it = min(g0 + j, nitem - 1);
                const bool isw = it >= npn;
                const int kk = isw ? it - npn : it;
                const T* base = isw ? a.W + (size_t)(c + kk - wbase) * a.ldw : a.A + (size_t)(c + kk) * a.lda;
                src[j] = base[rcl];
            }
            __builtin_amdgcn_sched_barrier(0);
            T wv[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                T tj[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int it = g0 + 8 * h + j;
                    T sv = src[8 * h + j];
                    if (it == 0) sv = vcs[lane];                 // column c itself: just finished by this workgroup
                    if (it == npn) sv = wcs[lane];
                    T p = zero;
                    fmac_(p, sel(rr < n && it < nitem, sv, zero), xh);
                    tj[j] = p;
                }
                transpose_reduce8_phase1<T>(tj, wv[h][0], wv[h][1]);
            }
            const T tval = transpose_reduce_phase2<T>(wv[0][0], wv[0][1], wv[1][0], wv[1][1], lane);
            if ((lane & 3) == 0) {
                const int it = g0 + transpose_col_of_lane(lane);
                if (it < nitem) {
                    const bool isw = it >= npn;
                    a.Zpn[(size_t)(I * 2 + (isw ? 1 : 0)) * NBMAX + (isw ? it - npn : it)] = tval;
                }
            }
        }
    }
    if (!FIRST) CSTAMP(7);
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
    long cap = c.hemv_blocks > 0 ? c.hemv_blocks : 2L * c.n_cu;  // = resident workgroups (2 per CU): one wave of blocks, no tail
    if (ntiles <= cap) return (int)ntiles;
    // Tile-round quantisation: workgroup b takes tiles b, b+G, b+2G, ...  With G = cap, 528 tiles (n = 2048) are one full
    // round plus 16 tiles that run alone.  Spreading the tiles evenly over the minimum number of rounds (G = 264 x 2 tiles)
    // was measured SLOWER (n=2048: 10.1 vs 9.0 us per launch; C3 tridiagonalization 71.2 vs 69.2 ms): two resident
    // workgroups per CU hide more latency than the balanced tail saves.  Kept as option "hemv_balance" (off).
    if (c.hemv_balance > 0) {   // k: balance when the tiles make at least k rounds
        long rounds = (ntiles + cap - 1) / cap;
        if (rounds >= c.hemv_balance) return (int)((ntiles + rounds - 1) / rounds);
    }
    return (int)cap;
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

// double-buffered partial-sum sets of the one-launch-per-column path (panel_col_kernel), sized for order nmax
template <class T> struct ColScratch {
    T* x[2]; T* P[2]; T* S[2]; T* D[2]; T* Zp[2]; T* alpha[2];
    double* NP[2];
    int ldp;
};
template <class T> static ColScratch<T> col_scratch(Ctx& c, int nmax, int prob = 0) {
    ColScratch<T> s;
    const int nt = nmax / HT + 1;
    s.ldp = nt * HT;
    const size_t per = (size_t)(nt * HT + 64) + (size_t)nt * s.ldp + (size_t)(nt * (nt + 1) / 2 + 64) + (size_t)(nt + 64) +
                       (size_t)(nt + 1) * 2 * NBMAX + 8;
    char nm[32];
    snprintf(nm, sizeof nm, prob == 0 ? "trd_col" : "trd_col#%d", prob);
    T* base = c.scratch<T>(nm, 2 * per);
    snprintf(nm, sizeof nm, prob == 0 ? "trd_colNP" : "trd_colNP#%d", prob);
    double* np = c.scratch<double>(nm, 2 * (size_t)(nt + 64));
    for (int b = 0; b < 2; ++b) {
        T* q = base + (size_t)b * per;
        s.x[b] = q; q += nt * HT + 64;
        s.P[b] = q; q += (size_t)nt * s.ldp;
        s.S[b] = q; q += nt * (nt + 1) / 2 + 64;
        s.D[b] = q; q += nt + 64;
        s.Zp[b] = q; q += (size_t)(nt + 1) * 2 * NBMAX;
        s.alpha[b] = q;
        s.NP[b] = np + (size_t)b * (nt + 64);
    }
    return s;
}

// one problem of a lockstep batch: its matrix, outputs, panel workspace and private scratch
template <class T> struct TrdProb {
    T* A; double* d; double* e; T* tau; T* W;
    TrdScratch<T> sc;
    ColScratch<T> cs;
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
        a.wt = c.p_wt;
    }
    auto set_all = [&](auto f) { for (int q = 0; q < NB; ++q) f(ab.p[q]); };
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
        hipLaunchKernelGGL((panel_mv_kernel<T, NB>), dim3(gh + gg, nprob), dim3(MVT), 0, st, ab, 0, gg);
        if (nlaunch) ++*nlaunch;
        if (algo_bytes) *algo_bytes += (double)sizeof(T) * (double)n * (double)(n + 1) * 0.5;
        gh_prev = gh; nchunk_prev = nchunk;
    }
    EIG_HIP(hipGetLastError());
}

// One panel through panel_col_kernel: nb launches (one per column) + the finish-only launch for the panel's leftmost column.
template <class T, int NB>
static void latrd_panel_fused(Ctx& c, hipStream_t st, int nprob, const TrdProb<T>* pr, int np, int nb, int lda, int ldw,
                              bool sweep_only = false, long* nlaunch = nullptr, double* algo_bytes = nullptr) {
    ColBatch<T, NB> ab;
    auto fill = [&](int i) {
        for (int q = 0; q < NB; ++q) {
            const TrdProb<T>& P = pr[q < nprob ? q : 0];
            ColArgs<T>& a = ab.p[q];
            const int nw = i & 1, pv = (i + 1) & 1;       // set written by this launch / set of the previous launch
            a.A = P.A; a.lda = lda; a.W = P.W; a.ldw = ldw; a.np = np; a.nb = nb; a.i = i; a.d = P.d; a.e = P.e; a.tau = P.tau;
            a.xprev = P.cs.x[pv]; a.xnew = P.cs.x[nw];
            a.Pp = P.cs.P[pv]; a.Pn = P.cs.P[nw]; a.ldp = P.cs.ldp;
            a.Sp = P.cs.S[pv]; a.Sn = P.cs.S[nw];
            a.Dp = P.cs.D[pv]; a.Dn = P.cs.D[nw];
            a.NPp = P.cs.NP[pv]; a.NPn = P.cs.NP[nw];
            a.Zpp = P.cs.Zp[pv]; a.Zpn = P.cs.Zp[nw];
            a.alphap = P.cs.alpha[pv]; a.alphan = P.cs.alpha[nw];
        }
    };
    for (int i = np - 1; i >= np - nb; --i) {
        fill(i);
        const int nt = i / HT + 1;                   // tiles over rows 0 .. i (row i has an owner even when i % 64 == 0)
        const dim3 grid(nt * (nt + 1) / 2, nprob);
        if (i == np - 1) hipLaunchKernelGGL((panel_col_kernel<T, NB, true, false>), grid, dim3(CTH), 0, st, ab);
        else hipLaunchKernelGGL((panel_col_kernel<T, NB, false, false>), grid, dim3(CTH), 0, st, ab);
        if (nlaunch) ++*nlaunch;
        if (algo_bytes) *algo_bytes += (double)sizeof(T) * (double)i * (double)(i + 1) * 0.5;
    }
    if (!sweep_only) {
        const int cfin = np - nb;                    // the panel's leftmost column: finished, nothing generated
        fill(cfin - 1);
        const int ntc = (cfin + HT - 1) / HT;
        hipLaunchKernelGGL((panel_col_kernel<T, NB, false, true>), dim3(ntc, nprob), dim3(CTH), 0, st, ab);
    }
    EIG_HIP(hipGetLastError());
}

// nprob problems of order N reduced in lockstep: every per-column launch carries all of them (blockIdx.y); the trailing
// rank-2nb updates and the final 32x32 blocks are launched per problem.  nprob = 1 is the plain zhetrd_gpu path.
// panels whose trailing order is <= this go through panel_col_kernel (one launch per column); option "trd_fuse"
template <class T> static int trd_fuse_order(const Ctx& c) {
    if (c.trd_fuse >= 0) return c.trd_fuse;
    return Tr<T>::cx ? kTrdFuseZ : kTrdFuseD;
}

template <class T, int NB>
static void hetrd_lockstep(Ctx& c, hipStream_t st, int N, int nprob, const TrdProb<T>* pr, int lda, int nb) {
    if (N <= 0) return;
    if (nb <= 0 || nb > NBMAX) nb = NBMAX;
    const int nx = TD;
    const int ldw = N;
    int np = N;
    auto trailing = [&](int npn, int nbn) {
        for (int q = 0; q < nprob; ++q)
            her2k_un<T>(c, st, npn - nbn, nbn, pr[q].A + (size_t)(npn - nbn) * lda, lda, pr[q].W, ldw, pr[q].A, lda);
    };
    const int fuse_n = trd_fuse_order<T>(c);
    int fused_from = 0;                               // columns below this were reduced by panel_col_kernel, which stores d itself
    auto panel = [&](int npn, int nbn) {
        if (npn <= fuse_n) {
            if (fused_from == 0) fused_from = npn;
            latrd_panel_fused<T, NB>(c, st, nprob, pr, npn, nbn, lda, ldw);
        } else {
            latrd_panel<T, NB>(c, st, nprob, pr, npn, nbn, lda, ldw);
        }
    };
    while (np - nb >= nx) {  // zhetrd_gpu.F90:60-71
        panel(np, nb);
        trailing(np, nb);
        np -= nb;
    }
    int nbr = np - nx;  // remainder panel, :73-83
    if (nbr > 0) {
        panel(np, nbr);
        trailing(np, nbr);
        np = nx;
    }
    int n0 = N < nx ? N : nx;
    for (int q = 0; q < nprob; ++q) {
        hipLaunchKernelGGL((hetd2_kernel<T>), dim3(1), dim3(256), 0, st, n0, pr[q].A, lda, pr[q].d, pr[q].e, pr[q].tau);
        const int x0 = fused_from > n0 ? fused_from : n0;
        if (N > x0)
            hipLaunchKernelGGL((diag_extract_kernel<T>), dim3((N - x0 + 255) / 256), dim3(256), 0, st, x0, N, (const T*)pr[q].A, lda,
                               pr[q].d);
    }
    EIG_HIP(hipGetLastError());
}

template <class T>
void hetrd_upper(Ctx& c, hipStream_t st, int N, T* A, int lda, double* d, double* e, T* tau, T* W, int nb) {
    if (N <= 0) return;
    TrdProb<T> pr{A, d, e, tau, W, trd_scratch<T>(c, N), col_scratch<T>(c, min(N, max(trd_fuse_order<T>(c), 64)))};
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
            pr[q] = TrdProb<T>{A[q0 + q], d[q0 + q], e[q0 + q], tau[q0 + q], W[q0 + q], trd_scratch<T>(c, N, q),
                               col_scratch<T>(c, min(N, max(trd_fuse_order<T>(c), 64)), q)};
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
    const int nx = TD;
    int np = N;
    double* dsink = c.scratch<double>("sweep_d", (size_t)N + 8);
    TrdProb<T> pr{A, dsink, e, tau, W, sc, col_scratch<T>(c, min(N, max(trd_fuse_order<T>(c), 64)))};
    const int fuse_n = trd_fuse_order<T>(c);
    auto panel = [&](int npn, int nbn) {
        // (below the fuse order the mat-vec IS the one-launch-per-column kernel: row work + tile, as the reduction runs it)
        if (npn <= fuse_n) latrd_panel_fused<T, 1>(c, st, 1, &pr, npn, nbn, lda, N, true, nlaunch, algo_bytes);
        else latrd_panel<T, 1>(c, st, 1, &pr, npn, nbn, lda, N, true, nlaunch, algo_bytes);
    };
    while (np - nb >= nx) {
        panel(np, nb);
        np -= nb;
    }
    int nbr = np - nx;
    if (nbr > 0) panel(np, nbr);
}

template <class T> void hemv_upper(Ctx& c, hipStream_t st, int n, const T* A, int lda, const T* x, T* y, bool gather) {
    if (n <= 0) return;
    TrdScratch<T> sc = trd_scratch<T>(c, n);
    PanelBatch<T, 1> ab;
    PanelArgs<T>& a = ab.p[0];
    a.A = const_cast<T*>(A); a.lda = lda; a.W = nullptr; a.ldw = 0; a.np = n + 1; a.nb = 1; a.i = n;
    a.e = nullptr; a.tau = nullptr; a.xbuf = const_cast<T*>(x); a.P = sc.P; a.ldp = sc.ldp; a.S = sc.S; a.Zp = sc.Zp;
    a.NP = sc.NP; a.alphaSlot = sc.alphaSlot; a.nblkA = 0; a.nchunk = 0; a.wt = c.p_wt;
    a.gh = hemv_grid(c, n);
    hipLaunchKernelGGL((panel_mv_kernel<T, 1>), dim3(a.gh), dim3(MVT), 0, st, ab, 1, 0);
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
