// blas3.hip -- fp64 MFMA tile engine for gfx950 and the blocked routines built on it.  Replaces every cuBLAS/cuSOLVER
// BLAS-3 call site of the reference: gemm, her2k/syr2k, herk/syrk, trmm, trsm, potrf, and the hegst blocking.
//
// Design (CDNA4-first, not a translation of any vendor kernel):
//  * ONE kernel template  gemm_fast_kernel<T, BM, BN, TA, TB, BK, MASKED>  with 4 wave64 waves in a 2x2 grid; operands are
//    staged global -> registers -> LDS (double-buffered, one barrier per K-slab, the loads of slab k+2 in flight during the
//    MFMAs of slab k), complex data split into re/im planes in LDS so every MFMA operand is one conflict-free ds_read_b64;
//  * LDS layouts are chosen per operand from how it sits in HBM ("idx-contiguous" or "k-contiguous") so global reads are
//    always coalesced and LDS reads bank-conflict-free: idx-contiguous slabs are XOR-swizzled (no padding), k-contiguous
//    rows padded by 2 doubles (MI355X_MICROARCH.md LDS table);
//  * the MFMA is issued as D[n][m] (B fragment first) so that the 16 lanes of a fragment row own 16 consecutive rows of C
//    -> 128/256-byte contiguous C read-modify-write;
//  * triangular / unit-trapezoid masks, conjugation, K-concatenation (her2k in one pass over C) and triangular-output
//    filtering are folded into the operand loader / epilogue, so no operand is ever physically modified (the reference
//    stashes/zeros/restores blocks of A);
//  * workgroup -> tile mapping is XCD-aware (tile_of): the workgroups that are resident on one XCD at a time cover a compact
//    super-tile of C, so the operand panels they share are fetched into that XCD's L2 once.
#include <type_traits>

#include "blas3.h"
#include "lanes.h"

// Real tiles use ONE v_mfma_f64_16x16x4 per 16x16x4 block product, complex tiles four v_mfma_f64_4x4x4 (round 5).  A bare stream of
// the 4x4x4 form sustains 72 TFLOP/s against 48 for 16x16x4 (profiles/r01_microbench3_mfma_variants.txt), and the complex tiles --
// 256 MFMAs per wave and K-slab, MFMA pipe 74 % busy -- live off that.  A real tile has a quarter of the MFMA work per staged
// element and is bound by everything around it: the 16x16x4 form needs one B-fragment read instead of four and a quarter of the
// issue slots: dgemm 4096^3 41.3 -> 45.5 TFLOP/s, 2048^3 37.9 -> 41.6, 8192^2 x 256 38.4 -> 42.2 (EIG_REAL_MFMA16=0 restores the
// 4x4x4 form for A/B runs; same lane -> (m, n) mapping of the result, same summation order over k: bit-identical).
#ifndef EIG_REAL_MFMA16
#define EIG_REAL_MFMA16 1
#endif

namespace eig {

typedef double d4 __attribute__((ext_vector_type(4)));

// workgroup -> tile maps (GemmArgs::map)
enum TileMapKind {
    TM_GRID = 0,      // tile = (blockIdx.x, blockIdx.y)
    TM_RECT = 1,      // 1-D grid over 64-slot super-tiles of a tile rectangle, dealt XCD by XCD (see tile_of)
    TM_FOLD = 2,      // the same over the stored triangle of a square tile grid folded into a rectangle
    TM_TRI = 3,       // 1-D grid over the tiles of the stored triangle, plain enumeration (small / batched grids)
};

// A lockstep group's launch (replay_group): `count` problems of ONE shape, problem q = blockIdx.z / zper with the pointers of
// entry q; everything else (extents, masks, strides, K-split, strided-batch descriptor) is shared.
constexpr int kGroupMax = 4;
template <class T> struct GemmGroup {
    int count = 0, zper = 1;
    const T* Ap[kGroupMax];
    const T* Ap2[kGroupMax];
    const T* Bp[kGroupMax];
    const T* Bp2[kGroupMax];
    T* Cp[kGroupMax];
    T* auxp[kGroupMax];
    T* Pp[kGroupMax];
};

template <class T> struct GemmArgs {
    int M, N, K;
    T alpha, beta;
    Operand<T> A, B;
    T* C;
    int ldc;
    Epi epi;
    int kchunk;      // >0: split-K, blockIdx.z owns [z*kchunk, (z+1)*kchunk)
    T* P;            // split-K partial output (M x N per split, ld = M)
    size_t pstride;
    int map;         // TileMapKind
    int mw, mh;      // TM_RECT / TM_FOLD: the tile rectangle
    int msw, msh;    // log2 of the super-tile's width / height (msw + msh = 6)
    int mnsx;        // super-tiles along x
    int mfull;       // slots below this index are dealt in XCD chunks, the rest round-robin
    int mnt;         // TM_FOLD: (even) order of the tile triangle
    GemmBatch bt;    // count > 0: blockIdx.z = batch entry * bt.splits + K-split
    GemmGroup<T> grp;   // count > 0: blockIdx.z = (problem * zper) + the z index of the problem's own launch
};

// Tile owned by this workgroup; false = a padding slot of the map (nothing to do).
//
// Hardware facts used (MI355X_MICROARCH.md, "Workgroup dispatch"; for SPEED only, any placement gives the same results):
// workgroup b of a launch runs on XCD b % 8, every XCD has a private 4 MB L2, and an XCD holds 64 of these workgroups at a
// time (32 CUs x 2).  With tile = (blockIdx.x, blockIdx.y) the 64 workgroups an XCD works on are every 8th tile of a tile
// column: they share ONE operand panel and read 64 different panels of the other operand -- zgemm 4096^3 fetched 8.3x its
// algorithmic bytes from HBM / Infinity Cache (profiles/r03_pmc_summary.txt).  Here the tile grid is cut into super-tiles of
// 64 tiles (8x8 unless the grid is narrow), and slot u of the 1-D launch maps to
//     super-tile  (u / 512) * 8 + u % 8,   position (u % 512) / 8 inside it,
// i.e. the k-th group of 64 workgroups that lands on XCD x is exactly one super-tile: 8 + 8 operand panels for 64 tiles.
// The last, partial round of super-tiles (slots >= mfull) is dealt tile by tile so that all XCDs finish together.
//
// Triangular outputs (epi.uplo) on a square tile grid: only the nt(nt+1)/2 tiles of the stored triangle are launched (a
// 2-D grid with an early exit left the XCDs unevenly loaded: 27 -> 4x TFLOP/s on an order-1984 update, round 2).  For the
// super-tile map the triangle is FOLDED into the dense rectangle (nt/2) x (nt+1) (nt even; odd orders are padded by one):
//     (i, j), j >  i  ->  tile (i, j - 1)                      rows 0 .. nt/2-1 of the triangle,
//     (i, j), j <= i  ->  tile (nt-1-i, nt-1-j)                rows nt/2 .. nt-1, point-reflected,
// both pieces of a super-tile are compact blocks of C.
template <class G> __device__ __forceinline__ bool tile_of(G& g, unsigned ux, unsigned uy, int& bx, int& by) {   // G: GemmArgs<T>, in any address space
    if (g.map == TM_GRID) { bx = (int)ux; by = (int)uy; return true; }
    int r, q;
    if (g.map == TM_TRI) {
        const int t = (int)ux;
        q = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((q + 1) * (q + 2) / 2 <= t) ++q;
        while (q * (q + 1) / 2 > t) --q;
        r = t - q * (q + 1) / 2;   // r <= q
    } else {
        const unsigned u = ux;
        unsigned p = u;
        if (u < (unsigned)g.mfull) p = ((((u >> 9) << 3) + (u & 7u)) << 6) + ((u & 511u) >> 3);
        const unsigned s = p >> 6, j = p & 63u;
        const unsigned sy = s / (unsigned)g.mnsx, sx = s - sy * (unsigned)g.mnsx;
        const int i = (int)((sx << g.msw) + (j & ((1u << g.msw) - 1u)));
        const int jj = (int)((sy << g.msh) + (j >> g.msw));
        if (i >= g.mw || jj >= g.mh) return false;
        if (g.map == TM_RECT) { bx = i; by = jj; return true; }
        if (jj > i) { r = i; q = jj - 1; } else { r = g.mnt - 1 - i; q = g.mnt - 1 - jj; }
    }
    if (g.epi.uplo == 1) { bx = r; by = q; } else { bx = q; by = r; }
    return true;
}

template <class T> __device__ __forceinline__ bool tile_of(const GemmArgs<T>& g, int& bx, int& by) { return tile_of(g, blockIdx.x, blockIdx.y, bx, by); }

// Restrict [kbeg,kend) to where a masked operand tile can be non-zero (skips the zero half of
// triangular operands: trmm/trsm/hemm-by-two-gemms cost no wasted MFMAs beyond the diagonal tiles).
template <class T>
__device__ __forceinline__ void trim_k(const Operand<T>& o, int i0, int bsz, int& kbeg, int& kend) {
    if (o.k1 != INT_MAX) return;
    int ilast = i0 + bsz - 1;
    if (o.mask == M_UPPER || o.mask == M_SUPPER) {
        if (o.trans == 0) kbeg = max(kbeg, i0); else kend = min(kend, ilast + 1);
    } else if (o.mask == M_LOWER) {
        if (o.trans == 0) kend = min(kend, ilast + 1); else kbeg = max(kbeg, i0);
    } else if (o.mask == M_UNITTRAP) {
        if (o.trans == 0) kbeg = max(kbeg, i0 - o.moff); else kend = min(kend, ilast + o.moff + 1);
    }
}

// K-slabs.  A slab of a complex 64x64 tile at BK = 16 is 256 MFMAs per wave (4096 cycles) between two barriers; the same
// slab of a real tile is 64 MFMAs (1024 cycles), so the per-slab cost (barrier, LDS write -> read turn-around) weighs four times
// as much: real tiles take slabs twice as deep (same LDS bytes per stage as the complex ones).
#ifndef EIG_BKL_REAL
#define EIG_BKL_REAL 16
#endif
#ifndef EIG_BKS_REAL
#define EIG_BKS_REAL 32
#endif
constexpr int BKL = 16;  // K-slab of the 64x64 tiles (complex)
constexpr int BKS = 32;  // K-slab of the 32x32 tiles (complex): K <= 64 (panel-sized products) is two stages
template <class T> constexpr int slab_k(bool small) { return Tr<T>::cx ? (small ? BKS : BKL) : (small ? EIG_BKS_REAL : EIG_BKL_REAL); }

// ------------------------------------------------------------------------------------------------
// gemm_fast_kernel
//  * per-thread element descriptors (stored coordinates, validity, LDS offset) are computed ONCE; a stage's global
//    loads are raw (clamped address only) and the mask / conjugation / zero-fill are applied one slab later, when
//    the registers are written to LDS -- so the loads really stay in flight across the MFMAs of a slab;
//  * K-concatenated operands (her2k) run as two segments over (p, ld) then (p2, ld2), each padded to whole slabs (the
//    segment boundary k1 may be anything: remainder panels of the tridiagonalization);
//  * LDS is double-buffered: the store of slab k+1 and the global loads of slab k+2 are issued
//    before the MFMAs of slab k, ONE barrier per stage.
//  * LDS per workgroup: idx-contiguous slabs carry no padding (XOR swizzle, below), so every instantiation stays below
//    80 KB and TWO workgroups share a CU (the padded layout of round 3 put the complex 32x32 forms at 84-98 KB:
//    one workgroup per CU, MFMA pipe 19-56 % busy on the solve's small products).
// ------------------------------------------------------------------------------------------------
// alpha * v (+ beta * c) with the multiply-adds spelled out: the engine has two kernels for the complex 64 x 64 tiles, and left to
// -ffp-contract the compiler fused `a.x * v.x - a.y * v.y` one way in one of them and the other way in the other (1 ulp apart for
// a complex alpha: found by the bit-identity check of the two staging paths, profiles/r06_experiments.txt section 2).
__device__ __forceinline__ double scal_(double al, double v) { return al * v; }
__device__ __forceinline__ cplx scal_(cplx al, cplx v) {
    return cplx{fma(al.x, v.x, -(al.y * v.y)), fma(al.x, v.y, al.y * v.x)};
}

template <class T, int BM, int BN, int TA, int TB, int BK, bool MASKED>
__global__ void __launch_bounds__(256, 2) gemm_fast_kernel(GemmArgs<T> g) {
    constexpr bool CX = Tr<T>::cx;
    constexpr int NPL = CX ? 2 : 1;
    static_assert(BM % 32 == 0 && BN % 32 == 0 && BK % 4 == 0, "tile shape");
    // idx-contiguous operand: slab row k holds the BM (BN) entries of that k, entry idx at position idx ^ (16 * (k & 1)).
    // A 16x4 fragment read (16 consecutive idx for 4 consecutive k; ds_read_b64 serves lanes 0-31 = two k's at once, 64
    // banks of 4 B = 32 doubles) then finds its even-k half and its odd-k half in different halves of the banks.
    // k-contiguous operand: row idx holds BK consecutive k, rows padded by 2 doubles (16 rows x 2 k's: distinct banks).
    constexpr int LDA = TA == 0 ? BM : BK + 2;
    constexpr int LDB = TB == 0 ? BN : BK + 2;
    constexpr int ASZ = TA == 0 ? BK * LDA : BM * LDA;
    constexpr int BSZ = TB == 0 ? BK * LDB : BN * LDB;
    constexpr int STG = NPL * (ASZ + BSZ);   // doubles per LDS stage
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
    static_assert(2 * STG * sizeof(double) <= 80 * 1024, "two workgroups per CU");
    __shared__ double sm[2 * STG];
    auto aoff = [](int idx, int k) -> int { return TA == 0 ? k * LDA + (idx ^ ((k & 1) << 4)) : idx * LDA + k; };
    auto boff = [](int idx, int k) -> int { return TB == 0 ? k * LDB + (idx ^ ((k & 1) << 4)) : idx * LDB + k; };

    // (wave index made wave-uniform for the compiler: the wave's tile origin and LDS offsets then live in SGPRs, 2-5 VGPRs less)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // strided batch: entry zb works on operands / output shifted by constant strides, with its own K, mask offsets and
    // (clipped) M, N -- all wave-uniform
    const int pld = g.M;          // leading dimension of the split-K partial blocks
    int bz = blockIdx.z;          // the z index of the problem's own launch
    if (g.grp.count > 0) {        // lockstep group: this workgroup's problem
        // (the pointer table is read through the kernel-argument segment -- g is the first argument --: indexed through the by-value
        //  copy `g`, which this kernel modifies, the whole argument block moved to scratch memory, 528-560 bytes per lane)
        typedef const __attribute__((address_space(4))) GemmArgs<T> KArgs;
        KArgs* kp = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        const int q = __builtin_amdgcn_readfirstlane(bz / g.grp.zper);
        bz -= q * g.grp.zper;
        g.A.p = kp->grp.Ap[q]; g.A.p2 = kp->grp.Ap2[q]; g.B.p = kp->grp.Bp[q]; g.B.p2 = kp->grp.Bp2[q];
        g.C = kp->grp.Cp[q]; g.epi.aux = kp->grp.auxp[q]; g.P = kp->grp.Pp[q];
    }
    int zs = bz;
    if (g.bt.count > 0) {
        const int zb = bz / g.bt.splits;
        zs = bz - zb * g.bt.splits;
        g.A.p += (long)zb * g.bt.sA; g.B.p += (long)zb * g.bt.sB; g.C += (long)zb * g.bt.sC;
        g.A.moff += zb * g.bt.dMoffA; g.B.moff += zb * g.bt.dMoffB;
        g.K += zb * g.bt.dK;
        if (g.bt.capK != INT_MAX) g.K = min(g.K, g.bt.capK - zb * g.bt.dcap);
        g.M = min(g.M, g.bt.capM - zb * g.bt.dcap);
        g.N = min(g.N, g.bt.capN - zb * g.bt.dcap);
    }
    int tbx, tby;
    if (!tile_of(g, tbx, tby)) return;
    const int i0 = tbx * BM, j0 = tby * BN;
    if (g.epi.uplo == 1 && i0 > j0 + BN - 1) return;
    if (g.epi.uplo == 2 && j0 > i0 + BM - 1) return;
    if (i0 >= g.M || j0 >= g.N) return;      // (clipped batch entries, the padding row / column of an odd folded triangle)

    int kbeg = 0, kend = g.K;
    if (g.kchunk > 0) {
        kbeg = zs * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    }
    trim_k(g.A, i0, BM, kbeg, kend);
    trim_k(g.B, j0, BN, kbeg, kend);
    // the K range as (up to) two segments, each walked in slabs of BK from its own start:
    //   segment 1 = logical [kbeg, min(kend, k1)) on (p, ld), segment 2 = logical [max(kbeg, k1), kend) on (p2, ld2) at k - k1
    const bool cat = g.A.k1 != INT_MAX;       // host guarantees A.k1 == B.k1
    const int k1 = cat ? g.A.k1 : INT_MAX;
    const int a1 = kbeg, b1 = min(kend, k1);
    const int a2 = cat ? max(kbeg, k1) - k1 : 0, b2 = cat ? kend - k1 : 0;
    const int nst1 = b1 > a1 ? (b1 - a1 + BK - 1) / BK : 0;
    const int nst2 = b2 > a2 ? (b2 - a2 + BK - 1) / BK : 0;
    const int nst = nst1 + nst2;

    const int wm0 = (wave & 1) * WM, wn0 = (wave >> 1) * WN;
    d4 acc[NPL][TM][TN];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[p][a][b] = d4{0.0, 0.0, 0.0, 0.0};

    // ---- element descriptors -----------------------------------------------------------------------
    // Per thread and element: global idx, k offset inside a slab, validity, LDS offset.  A slab's global loads are RAW
    // (clamped address, nothing else); mask / conjugation / zero-fill are applied when the registers go to LDS, one
    // slab later.  (With the select next to the load hipcc waits for the loads right where they are issued:
    // s_waitcnt vmcnt(0) in front of the MFMAs of EVERY slab, i.e. one exposed memory round trip per slab; round 3.)
    int sia[EA], kla[EA], sib[EB], klb[EB];
    bool oka[EA], okb[EB];
    int offa[EA], offb[EB];   // LDS offsets inside a stage
#pragma unroll
    for (int e = 0; e < EA; ++e) {
        int idx, k;
        if (TA == 0) { idx = tid % BM; k = tid / BM + e * (256 / BM); }
        else { k = tid % BK; idx = tid / BK + e * (256 / BK); }
        sia[e] = i0 + idx; kla[e] = k; oka[e] = (i0 + idx) < g.M;
        offa[e] = aoff(idx, k);
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        int idx, k;
        if (TB == 0) { idx = tid % BN; k = tid / BN + e * (256 / BN); }
        else { k = tid % BK; idx = tid / BK + e * (256 / BK); }
        sib[e] = j0 + idx; klb[e] = k; okb[e] = (j0 + idx) < g.N;
        offb[e] = NPL * ASZ + boff(idx, k);
    }

    // keep / unit-diagonal predicates of one element (kk = k inside its segment, ke = end of the segment).  `need` is
    // wave-uniform: false when the mask cannot touch any element of the current (tile, slab) -- the mask arithmetic is
    // then skipped altogether (measured: a unit-trapezoid operand evaluated per element costs 20 % of the launch, and
    // only the few tile-slabs on the diagonal band of a triangular / trapezoidal operand need it).
    auto pred = [&](const Operand<T>& o, int trans, int sidx, bool ok, int kk, int ke, bool need, bool& one) -> bool {
        bool keep = ok && kk < ke;
        one = false;
        if (MASKED && need) {
            const int sr = trans ? kk : sidx;
            const int sc = trans ? sidx : kk;
            const int d = sr - sc - o.moff;
            const bool km = (o.mask == M_UPPER) ? (sr <= sc) : (o.mask == M_SUPPER) ? (sr < sc) : (o.mask == M_LOWER) ? (sr >= sc) : (d < 0);
            one = keep && (o.mask == M_UNITTRAP) && (d == 0);
            keep = keep && km;
        }
        return keep;
    };
    // does the mask of operand o affect the slab [kl, kl + BK) of the tile rows/columns [x0, x0 + bs)?  (entries it zeroes
    // everywhere were removed by trim_k; here: is every entry kept unchanged?)
    auto mask_active = [&](const Operand<T>& o, int trans, int x0, int bs, int kl) -> bool {
        if (!MASKED || o.mask == M_NONE) return false;
        const int dmin = trans ? kl - (x0 + bs - 1) : x0 - (kl + BK - 1);      // min / max of (stored row - stored col)
        const int dmax = trans ? kl + BK - 1 - x0 : x0 + bs - 1 - kl;
        if (o.mask == M_UPPER) return dmax > 0;
        if (o.mask == M_SUPPER) return dmax >= 0;
        if (o.mask == M_LOWER) return dmin < 0;
        return dmax - o.moff >= 0;                                             // M_UNITTRAP
    };

    T ra[EA], rb[EB];
    int pkl = 0, pke = 0;     // segment-local slab origin and segment end of the slab held in ra / rb
    bool pna = false, pnb = false;   // mask_active of that slab
    auto gload = [&](int s_) {     // slab index 0 .. nst-1
        const bool s2 = s_ >= nst1;
        const T* pa = s2 ? g.A.p2 : g.A.p;
        const T* pb = s2 ? g.B.p2 : g.B.p;
        const long la = s2 ? g.A.ld2 : g.A.ld, lb = s2 ? g.B.ld2 : g.B.ld;
        pkl = s2 ? a2 + (s_ - nst1) * BK : a1 + s_ * BK;
        pke = s2 ? b2 : b1;
        pna = mask_active(g.A, TA, i0, BM, pkl);
        pnb = mask_active(g.B, TB, j0, BN, pkl);
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            bool one;
            const int kk = pkl + kla[e];
            const bool keep = pred(g.A, TA, sia[e], oka[e], kk, pke, pna, one);
            const T* addr = TA ? pa + (size_t)kk + (size_t)sia[e] * la : pa + (size_t)sia[e] + (size_t)kk * la;
            ra[e] = *(keep ? addr : pa);
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            bool one;
            const int kk = pkl + klb[e];
            const bool keep = pred(g.B, TB, sib[e], okb[e], kk, pke, pnb, one);
            const T* addr = TB ? pb + (size_t)kk + (size_t)sib[e] * lb : pb + (size_t)sib[e] + (size_t)kk * lb;
            rb[e] = *(keep ? addr : pb);
        }
    };
    auto lstore = [&](double* st) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            bool one;
            const bool keep = pred(g.A, TA, sia[e], oka[e], pkl + kla[e], pke, pna, one);
            T v = ra[e];
            if (g.A.conj) v = conj_(v);
            v = sel(keep, v, sel(one, Tr<T>::one(), Tr<T>::zero()));
            st[offa[e]] = real_(v);
            if (CX) st[ASZ + offa[e]] = imag_(v);
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            bool one;
            const bool keep = pred(g.B, TB, sib[e], okb[e], pkl + klb[e], pke, pnb, one);
            T v = rb[e];
            if (g.B.conj) v = conj_(v);
            v = sel(keep, v, sel(one, Tr<T>::one(), Tr<T>::zero()));
            st[offb[e]] = real_(v);
            if (CX) st[BSZ + offb[e]] = imag_(v);
        }
    };

    const int fi = lane & 15, fk = lane >> 4;
    if (nst > 0) {
        gload(0);
        lstore(sm);
        if (nst > 1) gload(1);
        __syncthreads();
    }
    // C of the tile (beta != 0): requested BEFORE the MFMAs of the last slab -- the operand staging registers are dead by then --
    // so that the read-modify-write epilogue does not start with an exposed memory round trip (at K = 64, four slabs per tile,
    // that round trip was ~20 % of a tile: rank-64 update of potrf 33.8 TFLOP/s against 43 for the same product with beta = 0)
    const bool use_c = g.kchunk == 0 && !(real_(g.beta) == 0.0 && imag_(g.beta) == 0.0);
    T cv[TM][TN][4];
    auto load_c = [&]() {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int gi = i0 + wm0 + a * 16 + (lane & 15);
                    int gj = j0 + wn0 + b * 16 + (lane >> 4) + 4 * r;
                    bool ok = gi < g.M && gj < g.N;
                    const T* cp = ok ? g.C + (size_t)gi + (size_t)gj * g.ldc : g.C;
                    cv[a][b][r] = *cp;
                }
    };
    // the MFMAs of one K-slab (As / Bs: the LDS stage holding it).
    // v_mfma_f64_4x4x4_4b_f64 sustains ~72 TFLOP/s on gfx950 where 16x16x4 saturates at ~48
    // (profiles/r01_microbench3_mfma_variants.txt).  Four of them, fed with the 4-row slices r = 0..3 of the B fragment
    // (replicated over lane bits 2-3) against the unchanged 16-wide A fragment, produce exactly one 16x16x4 product; result r
    // lands in component r of the accumulator: lane l holds C(m = l&15, n = 4r + (l>>4))
    // (layout measured in profiles/r01_probe_mfma_f64_4x4x4_layout.txt).  blgp bit 0 negates.
    constexpr bool M16 = !CX && EIG_REAL_MFMA16;
    // the complex 32x32 tiles (the <= 256-workgroup launches of hegst / trsm / the T factors: latency-bound, MFMA pipe 20-30 % busy) take
    // the 16x16x4 form too: gst 7.69 -> 7.54 ms at C3, C5 batch 108.8 -> 109.9 problems/s; the 64x64 complex tiles keep 4x4x4
#ifndef EIG_CPLX_SMALL_MFMA16
#define EIG_CPLX_SMALL_MFMA16 1
#endif
    // ... and so do the complex 64x64 tiles: on every shape of tools/gemm_shapes.py, fat ones included (zgemm 4096^3 58.3 -> 61.2
    // TFLOP/s, rank-64 update of order 4032 45.0 -> 47.8; in the C3 solve on one stream gst 7.49 -> 7.14, back-transformation
    // 4.56 -> 4.33, potrf 4.82 -> 4.70 ms).  The 48 TFLOP/s "ceiling" of round 1's bare 16x16x4 stream does not describe this kernel.
#ifndef EIG_CPLX_BIG_MFMA16
#define EIG_CPLX_BIG_MFMA16 1
#endif
    constexpr bool C16 = CX && ((EIG_CPLX_SMALL_MFMA16 && BM * BN <= 32 * 32) || (EIG_CPLX_BIG_MFMA16 && BM * BN > 32 * 32));
    auto mma_slab = [&](const double* As, const double* Bs) {
        if constexpr (C16) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                double ar[TM], ai[TM], br[TN], bi[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) { const int off = aoff(wm0 + a * 16 + fi, kk + fk); ar[a] = As[off]; ai[a] = As[ASZ + off]; }
#pragma unroll
                for (int b = 0; b < TN; ++b) { const int off = boff(wn0 + b * 16 + fi, kk + fk); br[b] = Bs[off]; bi[b] = Bs[BSZ + off]; }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        acc[0][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[b], ar[a], acc[0][a][b], 0, 0, 0);
                        acc[NPL - 1][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[b], ar[a], acc[NPL - 1][a][b], 0, 0, 0);
                    }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        acc[0][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[b], ai[a], acc[0][a][b], 0, 0, 1);
                        acc[NPL - 1][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[b], ai[a], acc[NPL - 1][a][b], 0, 0, 0);
                    }
            }
            return;
        }
        if constexpr (M16) {
            // D'[n][m] = sum_k Bt(n, k) A(m, k): first operand lane l = Bt(n = l & 15, k = l >> 4), second = A(m = l & 15, k = l >> 4);
            // result lane l, component r = C(m = l & 15, n = 4 r + (l >> 4)) -- the layout the four 4x4x4 products reproduce
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                double ar[TM], br[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) ar[a] = As[aoff(wm0 + a * 16 + fi, kk + fk)];
#pragma unroll
                for (int b = 0; b < TN; ++b) br[b] = Bs[boff(wn0 + b * 16 + fi, kk + fk)];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[0][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[b], ar[a], acc[0][a][b], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double ar[TM], ai[TM], br[TN][4], bi[TN][4];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int off = aoff(wm0 + a * 16 + fi, kk + fk);
                ar[a] = As[off];
                if (CX) ai[a] = As[ASZ + off];
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int off = boff(wn0 + b * 16 + 4 * r + (lane & 3), kk + fk);
                    br[b][r] = Bs[off];
                    if (CX) bi[b][r] = Bs[BSZ + off];
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][a][b][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(br[b][r], ar[a], acc[0][a][b][r], 0, 0, 0);
                        if (CX) acc[NPL - 1][a][b][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bi[b][r], ar[a], acc[NPL - 1][a][b][r], 0, 0, 0);
                    }
                }
            }
            if (CX) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[0][a][b][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bi[b][r], ai[a], acc[0][a][b][r], 0, 0, 1);
                            acc[NPL - 1][a][b][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(br[b][r], ai[a], acc[NPL - 1][a][b][r], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };
    // all slabs but the last: stage slab s+1, request slab s+2, multiply slab s
    for (int s_ = 0; s_ + 1 < nst; ++s_) {
        const double* As = sm + (s_ & 1) * STG;
        lstore(sm + ((s_ + 1) & 1) * STG);                 // slab s+1 (registers) -> other buffer
        if (s_ + 2 < nst) gload(s_ + 2);                   // slab s+2 in flight during the MFMAs
        mma_slab(As, As + NPL * ASZ);
        __syncthreads();
    }
    // the last slab, with C in flight (peeled: cv is live only here, so the kernel stays at two workgroups per CU)
    if (use_c) load_c();
    if (nst > 0) {
        const double* As = sm + ((nst - 1) & 1) * STG;
        mma_slab(As, As + NPL * ASZ);
    }

    // epilogue, branch-free on the load side: all C reads of the tile were issued together (clamped addresses, load_c above),
    // here they are combined and stored under a predicate.  (`if (valid) { load; store; }` per element serialises 16
    // dependent round trips -- the same hipcc pattern as in the operand loads.)
    T* const aux = reinterpret_cast<T*>(g.epi.aux);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int gi = i0 + wm0 + a * 16 + fi;
                int gj = j0 + wn0 + b * 16 + fk + 4 * r;
                bool ok = gi < g.M && gj < g.N;
                if (g.epi.uplo == 1 && gi > gj) ok = false;
                if (g.epi.uplo == 2 && gi < gj) ok = false;
                T v = Tr<T>::make(acc[0][a][b][r], CX ? acc[NPL - 1][a][b][r] : 0.0);
                if (g.kchunk > 0) {
                    if (ok) g.P[(size_t)bz * g.pstride + (size_t)gi + (size_t)gj * pld] = v;
                } else {
                    T out = scal_(g.alpha, v);
                    if (aux && ok) aux[(size_t)gi + (size_t)gj * g.epi.ldaux] = out;
                    if (use_c) fma_(out, g.beta, cv[a][b][r]);
                    if (g.epi.herm_diag && gi == gj) out = Tr<T>::realpart(out);
                    if (ok) g.C[(size_t)gi + (size_t)gj * g.ldc] = out;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// gemm_dma_kernel -- the same 64x64 complex tile with its K-slabs staged by LDS-DMA (round 6)
//
// global_load_lds_dwordx4 copies 64 x 16 B per wave instruction straight into LDS (lane-linear at the M0 base): no staging
// registers, no ds_write pass, no select / conjugation arithmetic between the load and the MFMAs.  What makes that possible here:
//  * 16 B = ONE complex number, so the LDS image is interleaved (re, im) and a fragment read is ONE ds_read_b128 per lane (the
//    register-staged kernel keeps re / im planes and reads each with ds_read_b64);
//  * the image is stored in FRAGMENT BLOCKS: block (kb, ib) = the 16 idx x 4 k entries one v_mfma_f64_16x16x4 consumes, 1 KB, in
//    lane order of the fragment (lane l = idx l & 15, k l >> 4) -- a fragment read is lane-linear, hence conflict-free without any
//    swizzle, and the block is exactly what one DMA instruction writes.  The SOURCE side is free per lane: an idx-contiguous
//    operand presents 4 runs of 256 B per instruction, a k-contiguous one 16 runs of 64 B -- the transposition happens in the
//    addresses, both kinds produce the same image (one code path, the operand kind is a pair of strides);
//  * masks, zero fill (K remainders, clipped edges) and the unit diagonal of the trapezoid are ADDRESS selects: a lane whose
//    element is dropped reads a 16-byte zero (or one) constant instead; only the slabs a mask or an edge touches take that path,
//    the others compute addresses from two strides;
//  * conjugation is a sign flip of the imaginary part after the fragment read (one v_xor per fragment value).
// Two LDS stages of 32 KB, two workgroups per CU; slab s+1 is requested right behind the barrier that opens slab s and has the
// whole slab (>= 4096 MFMA cycles) to land; the only wait is the wave's own vmcnt(0) in front of that barrier.  Same k order and
// the same lane -> (m, n) mapping as gemm_fast_kernel: results are bit-identical to it.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) double g_dma_zero[2] = {0.0, 0.0};
__device__ __attribute__((aligned(16))) double g_dma_one[2] = {1.0, 0.0};
typedef double d2 __attribute__((ext_vector_type(2)));

// Work item w of a launch = slot (w % (gx * gy)) of the tile map, K-split / batch entry w / (gx * gy): the dimensions the plain
// kernel gets as its grid.  The kernel is PERSISTENT: gridDim.x workgroups (two per CU) walk the items w = blockIdx.x, + gridDim.x,
// ... and the slab pipeline runs straight across tile boundaries -- while a tile's last slab is multiplied the first slab of the
// workgroup's NEXT tile is already on its way into the other LDS stage (the DMA needs no registers, so nothing has to be free for
// it), and the C read-modify-write of the finished tile overlaps that transfer.  The register-staged kernel pays first-slab
// latency + epilogue per tile round (7.7 us at two workgroups per CU: 30 % of a K = 64 update).
struct DmaTile {
    int ok, w;
    int i0, j0, z, M, N;
    int a1, b1, a2, b2, nst1, nst;
    const cplx *pa, *pb;
    cplx* pc;
    int moffA, moffB;
};

// LEAN form (BK = 8, WPC = 3): slabs half as deep -- 16 KB per stage, 32 KB per workgroup -- and the C tile fetched in the epilogue,
// one 16 x 16 block at a time, instead of being held beside the accumulators: <= 168 VGPRs, THREE workgroups per CU.  For the
// short-K updates (her2k / the rank-128 updates of the factorization: 4-8 slabs of 16 per tile) the fixed part of a tile -- first
// slab latency, C read-modify-write: 7.7 us of a 25 us tile round at K = 64 -- is then covered by the MFMAs of two other workgroups
// instead of one.  Same k order, same lane mapping: bit-identical to the other forms.
template <bool MASKED, int BK, int WPC>
__global__ void __launch_bounds__(256, WPC) gemm_dma_kernel(GemmArgs<cplx> g, int gx, int gy, int gz, int persistent) {
    using T = cplx;
    constexpr int BM = 64, BN = 64, NPL = 2;
    constexpr bool LEAN = WPC > 2;
    constexpr int BPW = BK / 4;           // fragment blocks per operand, slab and wave
    constexpr int OPB = BK * 64 * 16;     // bytes of one operand's slab: BK fragment blocks
    constexpr int STG = 2 * OPB;
    constexpr int WM = 32, WN = 32, TM = 2, TN = 2;
    __shared__ __attribute__((aligned(1024))) unsigned char sm[2 * STG];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slots = gx * gy, total = slots * gz;
    // Register budget.  The slab loop needs the two operand descriptors and the current tile (~45 SGPRs); the tile map, the batch
    // strides, alpha / beta / C / the epilogue options are needed only at tile boundaries.  Referenced through `g` they would all be
    // loaded once and kept alive across the loop -- 106 SGPRs, 175 v_readlane / v_writelane spill moves per slab and 256 VGPRs in
    // the first version (persistent form 56 TFLOP/s against 63 for the one-tile form).  cold() hands out the kernel-argument block
    // through a pointer the compiler cannot trace: fields read through it are re-loaded (scalar cache) where they are used.
    // (The pointer keeps the CONSTANT address space: through a generic pointer every field became a per-lane flat load -- ~40 of
    //  them in front of a workgroup's first request -- and a K = 64 update ran 38 % slower.)
    typedef const __attribute__((address_space(4))) GemmArgs<cplx> ColdArgs;
    auto cold = [&]() -> ColdArgs& {
        ColdArgs* kp = (ColdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return *kp;
    };
    Operand<T> opA = g.A, opB = g.B;
    int gq = 0;                                  // lockstep group (one-item form only): this workgroup's problem
    if (g.grp.count > 0) {
        ColdArgs& gc = cold();
        gq = __builtin_amdgcn_readfirstlane((int)blockIdx.z / gc.grp.zper);
        opA.p2 = gc.grp.Ap2[gq]; opB.p2 = gc.grp.Bp2[gq];
    }
    const bool cat = opA.k1 != INT_MAX;
    const int k1 = cat ? opA.k1 : INT_MAX;

    // the next valid work item at or after w0 (invalid: padding slots of the map, tiles outside the stored triangle / a clipped batch entry)
    auto decode = [&](int w0) -> DmaTile {
        DmaTile t;
        t.ok = 0;
        ColdArgs& g = cold();     // (shadows the kernel argument on purpose)
        for (int w = w0; w < total; w += (int)gridDim.x) {
            // (one item per workgroup: the launch has the plain kernel's 3-D grid and the item is the block index -- the four integer
            //  divisions below cost a short-K tile 10-30 % of its time, profiles/r06_experiments.txt section 2)
            const int u = persistent ? w % slots : 0, z = persistent ? w / slots : (int)blockIdx.z;
            const unsigned ux = persistent ? (unsigned)(u % gx) : blockIdx.x, uy = persistent ? (unsigned)(u / gx) : blockIdx.y;
            int M = g.M, N = g.N, K = g.K;
            const cplx *pa = g.A.p, *pb = g.B.p;
            cplx* pc = g.C;
            int zl = z;                              // the z index of the problem's own launch
            if (g.grp.count > 0) {
                zl = z - gq * g.grp.zper;
                pa = g.grp.Ap[gq]; pb = g.grp.Bp[gq]; pc = g.grp.Cp[gq];
            }
            int zs = zl;
            int moffA = g.A.moff, moffB = g.B.moff;
            if (g.bt.count > 0) {
                const int zb = zl / g.bt.splits;
                zs = zl - zb * g.bt.splits;
                pa += (long)zb * g.bt.sA; pb += (long)zb * g.bt.sB; pc += (long)zb * g.bt.sC;
                moffA += zb * g.bt.dMoffA; moffB += zb * g.bt.dMoffB;
                K += zb * g.bt.dK;
                if (g.bt.capK != INT_MAX) K = min(K, g.bt.capK - zb * g.bt.dcap);
                M = min(M, g.bt.capM - zb * g.bt.dcap);
                N = min(N, g.bt.capN - zb * g.bt.dcap);
            }
            int tbx, tby;
            if (!tile_of(g, ux, uy, tbx, tby)) continue;
            const int i0 = tbx * BM, j0 = tby * BN;
            if (g.epi.uplo == 1 && i0 > j0 + BN - 1) continue;
            if (g.epi.uplo == 2 && j0 > i0 + BM - 1) continue;
            if (i0 >= M || j0 >= N) continue;
            int kbeg = 0, kend = K;
            if (g.kchunk > 0) {
                kbeg = zs * g.kchunk;
                kend = min(K, kbeg + g.kchunk);
            }
            Operand<T> oa = opA, ob = opB;
            oa.moff = moffA; ob.moff = moffB;
            trim_k(oa, i0, BM, kbeg, kend);
            trim_k(ob, j0, BN, kbeg, kend);
            t.a1 = kbeg; t.b1 = min(kend, k1);
            t.a2 = cat ? max(kbeg, k1) - k1 : 0; t.b2 = cat ? kend - k1 : 0;
            t.nst1 = t.b1 > t.a1 ? (t.b1 - t.a1 + BK - 1) / BK : 0;
            const int nst2 = t.b2 > t.a2 ? (t.b2 - t.a2 + BK - 1) / BK : 0;
            t.nst = t.nst1 + nst2;
            // (everything here is wave-uniform; said explicitly so that the two tile descriptors live in SGPRs -- the map's
            //  square root runs on the vector unit and would drag them into VGPRs: 256 VGPRs + scratch without this)
            auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
            auto unip = [&](const void* q) -> const void* {
                const unsigned long long x = (unsigned long long)q;
                return (const void*)(((unsigned long long)(unsigned)uni((int)(x >> 32)) << 32) | (unsigned)uni((int)(unsigned)x));
            };
            t.a1 = uni(t.a1); t.b1 = uni(t.b1); t.a2 = uni(t.a2); t.b2 = uni(t.b2); t.nst1 = uni(t.nst1); t.nst = uni(t.nst);
            t.ok = 1; t.w = uni(w); t.i0 = uni(i0); t.j0 = uni(j0); t.z = uni(zl); t.M = uni(M); t.N = uni(N);
            t.pa = (const cplx*)unip(pa); t.pb = (const cplx*)unip(pb); t.pc = (cplx*)unip(pc); t.moffA = uni(moffA); t.moffB = uni(moffB);
            return t;
        }
        return t;
    };

    const int wm0 = (wave & 1) * WM, wn0 = (wave >> 1) * WN;
    const int fi = lane & 15, fk = lane >> 4;
    const unsigned lds0 = lds_offset_of(sm);

    // keep / unit-diagonal predicate of one element, as in gemm_fast_kernel
    auto pred = [&](const Operand<T>& o, int moff, int sidx, bool ok, int kk, int ke, bool need, bool& one) -> bool {
        bool keep = ok && kk < ke;
        one = false;
        if (MASKED && need) {
            const int sr = o.trans ? kk : sidx;
            const int sc = o.trans ? sidx : kk;
            const int d = sr - sc - moff;
            const bool km = (o.mask == M_UPPER) ? (sr <= sc) : (o.mask == M_SUPPER) ? (sr < sc) : (o.mask == M_LOWER) ? (sr >= sc) : (d < 0);
            one = keep && (o.mask == M_UNITTRAP) && (d == 0);
            keep = keep && km;
        }
        return keep;
    };
    auto mask_active = [&](const Operand<T>& o, int moff, int x0, int bs, int kl) -> bool {
        if (!MASKED || o.mask == M_NONE) return false;
        const int dmin = o.trans ? kl - (x0 + bs - 1) : x0 - (kl + BK - 1);
        const int dmax = o.trans ? kl + BK - 1 - x0 : x0 + bs - 1 - kl;
        if (o.mask == M_UPPER) return dmax > 0;
        if (o.mask == M_SUPPER) return dmax >= 0;
        if (o.mask == M_LOWER) return dmin < 0;
        return dmax - moff >= 0;
    };
    // wave w requests the fragment blocks b = w * BPW .. (kb = b / 4, ib = b % 4) of both operands: 2 * BPW instructions per slab and wave
    // (BK = 16: kb = w, ib = 0..3)
    auto request_operand = [&](const Operand<T>& o, int moff, const T* p, long ld, int x0, int xmax, int kl, int ke, unsigned dst) {
        const long sidx = o.trans ? ld : 1, sk = o.trans ? 1 : ld;
        const int b0 = wave * BPW;
        const int kk = kl + (b0 >> 2) * 4 + fk;                              // this lane's k inside the segment
        const bool need = mask_active(o, moff, x0, 64, kl);
        if (!need && x0 + 64 <= xmax && kl + BK <= ke) {                    // interior slab: two strides
            const T* src = p + (size_t)(x0 + 16 * (b0 & 3) + fi) * sidx + (size_t)kk * sk;
#pragma unroll
            for (int t = 0; t < BPW; ++t) lds_dma16(src + (size_t)(16 * t) * sidx, dst + (unsigned)((b0 + t) * 1024));
        } else {
#pragma unroll
            for (int t = 0; t < BPW; ++t) {
                const int sx = x0 + 16 * ((b0 & 3) + t) + fi;
                bool one;
                const bool keep = pred(o, moff, sx, sx < xmax, kk, ke, need, one);
                const T* src = p + (size_t)sx * sidx + (size_t)kk * sk;
                const T* alt = reinterpret_cast<const T*>(one ? g_dma_one : g_dma_zero);
                lds_dma16(keep ? src : alt, dst + (unsigned)((b0 + t) * 1024));
            }
        }
    };
    auto request = [&](const DmaTile& t, int s_, unsigned fs) {          // slab s_ of tile t into LDS stage fs & 1
        const bool s2 = s_ >= t.nst1;
        const int kl = s2 ? t.a2 + (s_ - t.nst1) * BK : t.a1 + s_ * BK;
        const int ke = s2 ? t.b2 : t.b1;
        const unsigned stage = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (fs & 1u) * (unsigned)STG));
        request_operand(opA, t.moffA, s2 ? opA.p2 : t.pa, s2 ? opA.ld2 : opA.ld, t.i0, t.M, kl, ke, stage);
        request_operand(opB, t.moffB, s2 ? opB.p2 : t.pb, s2 ? opB.ld2 : opB.ld, t.j0, t.N, kl, ke, stage + OPB);
    };

    const bool use_c = g.kchunk == 0 && !(real_(g.beta) == 0.0 && imag_(g.beta) == 0.0);
    const double sgnA = opA.conj ? -1.0 : 1.0, sgnB = opB.conj ? -1.0 : 1.0;

    DmaTile cur = decode(persistent ? (int)blockIdx.x : (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)));
    if (!cur.ok) return;
    unsigned fs = 0;                       // slabs requested so far (LDS stage = parity)
    if (cur.nst > 0) { request(cur, 0, fs); wait_vmcnt<0>(); }
    while (cur.ok) {
        // (the next tile is decoded twice -- once for the request of its first slab, once when it becomes the current one: two
        //  descriptors kept alive across the slab loop overflow the 102 SGPRs and spill into VGPR lanes and scratch)
        d4 acc[NPL][TM][TN];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[p][a][b] = d4{0.0, 0.0, 0.0, 0.0};
        T cv[LEAN ? 1 : TM][LEAN ? 1 : TN][4];
        auto load_c = [&]() {
            if constexpr (LEAN) return;
            const int ldc_ = cold().ldc;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int gi = cur.i0 + wm0 + a * 16 + (lane & 15);
                        int gj = cur.j0 + wn0 + b * 16 + (lane >> 4) + 4 * r;
                        bool ok = gi < cur.M && gj < cur.N;
                        const T* cp = ok ? cur.pc + (size_t)gi + (size_t)gj * ldc_ : cur.pc;
                        cv[LEAN ? 0 : a][LEAN ? 0 : b][r] = *cp;
                    }
        };
        auto mma_slab = [&](unsigned stage_) {
            const unsigned char* As = sm + (stage_ & 1u) * STG;
        const unsigned char* Bs = As + OPB;
#pragma unroll
        for (int kb = 0; kb < BK / 4; ++kb) {
            double ar[TM], ai[TM], br[TN], bi[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const d2 v = *reinterpret_cast<const d2*>(As + (kb * 4 + (wm0 >> 4) + a) * 1024 + lane * 16);
                ar[a] = v.x; ai[a] = sgnA * v.y;
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const d2 v = *reinterpret_cast<const d2*>(Bs + (kb * 4 + (wn0 >> 4) + b) * 1024 + lane * 16);
                br[b] = v.x; bi[b] = sgnB * v.y;
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[0][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[b], ar[a], acc[0][a][b], 0, 0, 0);
                    acc[1][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[b], ar[a], acc[1][a][b], 0, 0, 0);
                }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[0][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[b], ai[a], acc[0][a][b], 0, 0, 1);
                    acc[1][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[b], ai[a], acc[1][a][b], 0, 0, 0);
                }
        }
        };
        // all slabs of the tile but the last: request the next one, multiply this one
        for (int s_ = 0; s_ + 1 < cur.nst; ++s_, ++fs) {
            __syncthreads();                      // every wave's pieces of this slab have landed (each waited for its own below), and
                                                  // every wave is done reading the slab before: its stage may be overwritten
            request(cur, s_ + 1, fs + 1);
            mma_slab(fs);
            wait_vmcnt<0>();                      // the slab requested above has landed (this wave's pieces; the barrier covers the rest)
        }
        // the last slab (peeled: C of the tile is live only from here on): the next tile's first slab and C fly during its MFMAs
        if (cur.nst > 0) {
            __syncthreads();
            DmaTile nxt;
            nxt.ok = 0;
            if (persistent) nxt = decode(cur.w + (int)gridDim.x);
            if (nxt.ok && nxt.nst > 0) request(nxt, 0, fs + 1);
            if (use_c) load_c();
            mma_slab(fs);
            wait_vmcnt<0>();
            ++fs;
        }
        if (cur.nst == 0) {                       // (empty K range: C <- beta C)
            if (use_c) load_c();
            DmaTile nxt;
            nxt.ok = 0;
            if (persistent) nxt = decode(cur.w + (int)gridDim.x);
            if (nxt.ok && nxt.nst > 0) { request(nxt, 0, fs); wait_vmcnt<0>(); }
        }
        // epilogue of the finished tile: its stores are in flight while the next tile starts
        ColdArgs& ge = cold();
        const T al_ = cplx{ge.alpha.x, ge.alpha.y}, be_ = cplx{ge.beta.x, ge.beta.y};
        T* const aux = ge.grp.count > 0 ? ge.grp.auxp[gq] : reinterpret_cast<T*>(ge.epi.aux);
        T* const Pq = ge.grp.count > 0 ? ge.grp.Pp[gq] : ge.P;
        const int pld = ge.M;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if constexpr (LEAN) {          // C of this 16 x 16 block: fetched here (the other workgroups of the CU cover the round trip)
                    if (use_c) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gi = cur.i0 + wm0 + a * 16 + fi, gj = cur.j0 + wn0 + b * 16 + fk + 4 * r;
                            const bool ok = gi < cur.M && gj < cur.N;
                            const T* cp = ok ? cur.pc + (size_t)gi + (size_t)gj * ge.ldc : cur.pc;
                            cv[0][0][r] = *cp;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int gi = cur.i0 + wm0 + a * 16 + fi;
                    int gj = cur.j0 + wn0 + b * 16 + fk + 4 * r;
                    bool ok = gi < cur.M && gj < cur.N;
                    if (ge.epi.uplo == 1 && gi > gj) ok = false;
                    if (ge.epi.uplo == 2 && gi < gj) ok = false;
                    T v = Tr<T>::make(acc[0][a][b][r], acc[1][a][b][r]);
                    if (ge.kchunk > 0) {
                        if (ok) Pq[(size_t)cur.z * ge.pstride + (size_t)gi + (size_t)gj * pld] = v;
                    } else {
                        T out = scal_(al_, v);
                        if (aux && ok) aux[(size_t)gi + (size_t)gj * ge.epi.ldaux] = out;
                        if (use_c) fma_(out, be_, cv[LEAN ? 0 : a][LEAN ? 0 : b][r]);
                        if (ge.epi.herm_diag && gi == gj) out = Tr<T>::realpart(out);
                        if (ok) cur.pc[(size_t)gi + (size_t)gj * ge.ldc] = out;
                    }
                }
            }
        }
        if (!persistent) break;
        cur = decode(cur.w + (int)gridDim.x);
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_wide_kernel -- complex 32 x 32 tiles on a workgroup of KG x 4 waves with K split INSIDE the workgroup (round 6)
//
// The products with fewer than one 64 x 64 tile per CU (the base cases and small updates of the triangular solves in hegst / trsm,
// W T^H of the back-transformation, the merges of inverse blocks: ~190 launches of a C3 solve) run on 32 x 32 tiles so that a
// launch still spans the chip.  On gemm_fast_kernel that is ONE four-wave workgroup per CU -- one wave per SIMD, every K-slab an
// exposed global -> register -> LDS round trip between two bursts of 32 MFMAs: 1.77 us per slab of 32 for 0.85 us of MFMA work
// (profiles/r05_pmc_summary.txt: MFMA pipe 22-28 % busy).  Here the tile gets the WHOLE CU: KG groups of four waves (2 x 2 blocks
// of 16 x 16), group q owns the q-th part of the tile's K-slabs (contiguous, in units of slabs of 16), stages them by LDS-DMA into
// two stages of its own (fragment-block image as in gemm_dma_kernel: 16 KB per stage), and accumulates its partial tile in
// registers (16 VGPRs).  With KG = 4 every SIMD holds four waves whose MFMA bursts and DMA waits interleave; all of a tile's
// K (up to 2 x KG slabs) is in flight at once.  The partial tiles are summed through LDS in the fixed order 0, 1, .., KG - 1
// (deterministic; no partial sums in HBM, no reduce launch) and group 0 runs the epilogue.
// The order of summation over k differs from gemm_fast_kernel's (KG chains instead of one): which kernel serves a product is a
// function of the product's shape alone (dispatch_gemm_now), never of the execution mode, so batch / single / overlapped forms of
// a solve stay bit-identical to each other.
// ------------------------------------------------------------------------------------------------
#ifndef EIG_WIDE_MFMA16
#define EIG_WIDE_MFMA16 0
#endif
template <bool MASKED, int KG>
__global__ void __launch_bounds__(256 * KG) gemm_wide_kernel(GemmArgs<cplx> g) {
    using T = cplx;
    constexpr int BM = 32, BN = 32, BK = BKL;
    constexpr int OPB = BK * BM * 16;       // bytes of one operand's slab: 8 fragment blocks of 1 KB
    constexpr int STG = 2 * OPB;            // one stage: A slab, B slab
    constexpr int GRP = 2 * STG;            // one group's two stages
    __shared__ __attribute__((aligned(1024))) unsigned char sm[KG * GRP];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    typedef const __attribute__((address_space(4))) GemmArgs<cplx> ColdArgs;
    auto cold = [&]() -> ColdArgs& {
        ColdArgs* kp = (ColdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return *kp;
    };
    Operand<T> opA = g.A, opB = g.B;
    const T *pa = g.A.p, *pb = g.B.p;
    T* pc = g.C;
    int gq = 0, bz = (int)blockIdx.z;
    if (g.grp.count > 0) {                  // lockstep group: this workgroup's problem
        ColdArgs& gc = cold();
        gq = __builtin_amdgcn_readfirstlane(bz / gc.grp.zper);
        bz -= gq * gc.grp.zper;
        pa = gc.grp.Ap[gq]; pb = gc.grp.Bp[gq]; pc = gc.grp.Cp[gq];
        opA.p2 = gc.grp.Ap2[gq]; opB.p2 = gc.grp.Bp2[gq];
    }
    int M = g.M, N = g.N, K = g.K, zs = bz;
    int moffA = g.A.moff, moffB = g.B.moff;
    if (g.bt.count > 0) {
        const int zb = bz / g.bt.splits;
        zs = bz - zb * g.bt.splits;
        pa += (long)zb * g.bt.sA; pb += (long)zb * g.bt.sB; pc += (long)zb * g.bt.sC;
        moffA += zb * g.bt.dMoffA; moffB += zb * g.bt.dMoffB;
        K += zb * g.bt.dK;
        if (g.bt.capK != INT_MAX) K = min(K, g.bt.capK - zb * g.bt.dcap);
        M = min(M, g.bt.capM - zb * g.bt.dcap);
        N = min(N, g.bt.capN - zb * g.bt.dcap);
    }
    int tbx, tby;
    if (!tile_of(g, tbx, tby)) return;
    const int i0 = tbx * BM, j0 = tby * BN;
    if (g.epi.uplo == 1 && i0 > j0 + BN - 1) return;
    if (g.epi.uplo == 2 && j0 > i0 + BM - 1) return;
    if (i0 >= M || j0 >= N) return;
    int kbeg = 0, kend = K;
    if (g.kchunk > 0) {
        kbeg = zs * g.kchunk;
        kend = min(K, kbeg + g.kchunk);
    }
    opA.moff = moffA; opB.moff = moffB;
    trim_k(opA, i0, BM, kbeg, kend);
    trim_k(opB, j0, BN, kbeg, kend);
    const bool cat = opA.k1 != INT_MAX;
    const int k1 = cat ? opA.k1 : INT_MAX;
    const int a1 = kbeg, b1 = min(kend, k1);
    const int a2 = cat ? max(kbeg, k1) - k1 : 0, b2 = cat ? kend - k1 : 0;
    const int nst1 = b1 > a1 ? (b1 - a1 + BK - 1) / BK : 0;
    const int nst2 = b2 > a2 ? (b2 - a2 + BK - 1) / BK : 0;
    const int nst = nst1 + nst2;
    const int per = (nst + KG - 1) / KG;                    // slabs per group (the last groups may have fewer, or none)
    const int s0 = grp * per;
    const int mine = max(0, min(per, nst - s0));            // this group's slabs: s0 .. s0 + mine - 1

    const int wm0 = (wq & 1) * 16, wn0 = (wq >> 1) * 16;
    const int fi = lane & 15, fk = lane >> 4;
    const unsigned lds0 = lds_offset_of(sm) + (unsigned)(grp * GRP);

    auto pred = [&](const Operand<T>& o, int sidx, bool ok, int kk, int ke, bool need, bool& one) -> bool {
        bool keep = ok && kk < ke;
        one = false;
        if (MASKED && need) {
            const int sr = o.trans ? kk : sidx;
            const int sc = o.trans ? sidx : kk;
            const int d = sr - sc - o.moff;
            const bool km = (o.mask == M_UPPER) ? (sr <= sc) : (o.mask == M_SUPPER) ? (sr < sc) : (o.mask == M_LOWER) ? (sr >= sc) : (d < 0);
            one = keep && (o.mask == M_UNITTRAP) && (d == 0);
            keep = keep && km;
        }
        return keep;
    };
    auto mask_active = [&](const Operand<T>& o, int x0, int bs, int kl) -> bool {
        if (!MASKED || o.mask == M_NONE) return false;
        const int dmin = o.trans ? kl - (x0 + bs - 1) : x0 - (kl + BK - 1);
        const int dmax = o.trans ? kl + BK - 1 - x0 : x0 + bs - 1 - kl;
        if (o.mask == M_UPPER) return dmax > 0;
        if (o.mask == M_SUPPER) return dmax >= 0;
        if (o.mask == M_LOWER) return dmin < 0;
        return dmax - o.moff >= 0;
    };
    // wave wq of a group requests the fragment blocks (kb = wq, ib = 0, 1) of both operands: 4 instructions per slab and wave
    auto request_operand = [&](const Operand<T>& o, const T* p, long ld, int x0, int xmax, int kl, int ke, unsigned dst) {
        const long sidx = o.trans ? ld : 1, sk = o.trans ? 1 : ld;
        const int kk = kl + wq * 4 + fk;
        const bool need = mask_active(o, x0, 32, kl);
        if (!need && x0 + 32 <= xmax && kl + BK <= ke) {
            const T* src = p + (size_t)(x0 + fi) * sidx + (size_t)kk * sk;
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) lds_dma16(src + (size_t)(16 * ib) * sidx, dst + (unsigned)((wq * 2 + ib) * 1024));
        } else {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const int sx = x0 + 16 * ib + fi;
                bool one;
                const bool keep = pred(o, sx, sx < xmax, kk, ke, need, one);
                const T* src = p + (size_t)sx * sidx + (size_t)kk * sk;
                const T* alt = reinterpret_cast<const T*>(one ? g_dma_one : g_dma_zero);
                lds_dma16(keep ? src : alt, dst + (unsigned)((wq * 2 + ib) * 1024));
            }
        }
    };
    auto request = [&](int s_, int st_) {                  // slab s_ (index in the tile's slab list) into this group's stage st_
        const bool s2 = s_ >= nst1;
        const int kl = s2 ? a2 + (s_ - nst1) * BK : a1 + s_ * BK;
        const int ke = s2 ? b2 : b1;
        const unsigned stage = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(st_ * STG)));
        request_operand(opA, s2 ? opA.p2 : pa, s2 ? opA.ld2 : opA.ld, i0, M, kl, ke, stage);
        request_operand(opB, s2 ? opB.p2 : pb, s2 ? opB.ld2 : opB.ld, j0, N, kl, ke, stage + OPB);
    };

    const bool use_c = g.kchunk == 0 && !(real_(g.beta) == 0.0 && imag_(g.beta) == 0.0);
    const double sgnA = opA.conj ? -1.0 : 1.0, sgnB = opB.conj ? -1.0 : 1.0;
    d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
    auto mma_slab = [&](int st_) {
        const unsigned char* As = sm + grp * GRP + st_ * STG;
        const unsigned char* Bs = As + OPB;
#pragma unroll
        for (int kb = 0; kb < BK / 4; ++kb) {
            const d2 va = *reinterpret_cast<const d2*>(As + (kb * 2 + (wm0 >> 4)) * 1024 + lane * 16);
            const double ar = va.x, ai = sgnA * va.y;
#if EIG_WIDE_MFMA16
            const d2 vb = *reinterpret_cast<const d2*>(Bs + (kb * 2 + (wn0 >> 4)) * 1024 + lane * 16);
            const double br = vb.x, bi = sgnB * vb.y;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(br, ar, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bi, ar, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(bi, ai, acc0, 0, 0, 1);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(br, ai, acc1, 0, 0, 0);
#else
            // four v_mfma_f64_4x4x4 per block product (gemm_fast_kernel's round-1 form: the 4-row slices r of the B fragment,
            // replicated over lane bits 2-3 -- here four broadcast reads of the fragment block --, against the unchanged A fragment;
            // result r lands in component r): with four waves per SIMD the kernel is bound by the MFMA pipe, and the pipe runs this
            // form at 72 TFLOP/s against 48 for 16x16x4 (profiles/r01_microbench3_mfma_variants.txt); same sums, same bits
            double br[4], bi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const d2 vb = *reinterpret_cast<const d2*>(Bs + (kb * 2 + (wn0 >> 4)) * 1024 + ((4 * r + (lane & 3)) + 16 * fk) * 16);
                br[r] = vb.x; bi[r] = sgnB * vb.y;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(br[r], ar, acc0[r], 0, 0, 0);
                acc1[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bi[r], ar, acc1[r], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bi[r], ai, acc0[r], 0, 0, 1);
                acc1[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(br[r], ai, acc1[r], 0, 0, 0);
            }
#endif
        }
    };

    // C of the tile (group 0 only): requested first, consumed last
    T cv[4];
    if (use_c && grp == 0) {
        const int ldc_ = cold().ldc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = i0 + wm0 + fi, gj = j0 + wn0 + fk + 4 * r;
            const bool ok = gi < M && gj < N;
            const T* cp = ok ? pc + (size_t)gi + (size_t)gj * ldc_ : pc;
            cv[r] = *cp;
        }
    }
    // the group's first TWO slabs go out at once (both stages are free); from then on slab s + 2 follows the MFMAs of slab s
    if (mine > 0) request(s0, 0);
    if (mine > 1) request(s0 + 1, 1);
    for (int s_ = 0; s_ < per; ++s_) {
        // this wave's pieces of slab s_ have landed (the pieces of slab s_ + 1, requested later, may still fly: 4 instructions)
        if (s_ + 1 < mine) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __syncthreads();                                       // ... and so have the other waves' pieces
        if (s_ < mine) mma_slab(s_ & 1);
        if (s_ + 2 < per) {                                    // (a workgroup-uniform condition: the groups' slab counts differ)
            __syncthreads();                                   // every wave of the workgroup is done reading stage s_ & 1
            if (s_ + 2 < mine) request(s0 + s_ + 2, s_ & 1);
        }
    }
    // partial tiles -> LDS (groups 1 .. KG - 1), summed by group 0 in the order of the groups
    __syncthreads();
    double* red = reinterpret_cast<double*>(sm);
    if (grp > 0) {
        double* dst = red + ((grp - 1) * 4 + wq) * 512 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dst[r * 64] = acc0[r]; dst[(4 + r) * 64] = acc1[r]; }
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int q = 1; q < KG; ++q) {
        const double* src = red + ((q - 1) * 4 + wq) * 512 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[r] += src[r * 64]; acc1[r] += src[(4 + r) * 64]; }
    }
    ColdArgs& ge = cold();
    const T al_ = cplx{ge.alpha.x, ge.alpha.y}, be_ = cplx{ge.beta.x, ge.beta.y};
    T* const aux = ge.grp.count > 0 ? ge.grp.auxp[gq] : reinterpret_cast<T*>(ge.epi.aux);
    T* const Pq = ge.grp.count > 0 ? ge.grp.Pp[gq] : ge.P;
    const int pld = ge.M;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wm0 + fi, gj = j0 + wn0 + fk + 4 * r;
        bool ok = gi < M && gj < N;
        if (ge.epi.uplo == 1 && gi > gj) ok = false;
        if (ge.epi.uplo == 2 && gi < gj) ok = false;
        const T v = Tr<T>::make(acc0[r], acc1[r]);
        if (ge.kchunk > 0) {
            if (ok) Pq[(size_t)bz * ge.pstride + (size_t)gi + (size_t)gj * pld] = v;
        } else {
            T out = scal_(al_, v);
            if (aux && ok) aux[(size_t)gi + (size_t)gj * ge.epi.ldaux] = out;
            if (use_c) fma_(out, be_, cv[r]);
            if (ge.epi.herm_diag && gi == gj) out = Tr<T>::realpart(out);
            if (ok) pc[(size_t)gi + (size_t)gj * ge.ldc] = out;
        }
    }
}

template <class T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(int M, int N, int splits, const T* P, size_t pstride, T alpha,
                                                            T beta, T* C, int ldc, Epi epi, GemmBatch bt) {
    size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)M * N) return;
    int i = (int)(id % M), j = (int)(id / M);
    if (bt.count > 0) {           // batch entry blockIdx.y: its partial blocks, its output, its clipped extent
        const int zb = blockIdx.y;
        P += (size_t)zb * splits * pstride;
        C += (long)zb * bt.sC;
        if (i >= bt.capM - zb * bt.dcap || j >= bt.capN - zb * bt.dcap) return;
    }
    if (epi.uplo == 1 && i > j) return;
    if (epi.uplo == 2 && i < j) return;
    T s = Tr<T>::zero();
    for (int z = 0; z < splits; ++z) s = s + P[(size_t)z * pstride + id];
    T* cp = C + (size_t)i + (size_t)j * ldc;
    T out = scal_(alpha, s);
    if (epi.aux) reinterpret_cast<T*>(epi.aux)[(size_t)i + (size_t)j * epi.ldaux] = out;
    if (!(real_(beta) == 0.0 && imag_(beta) == 0.0)) fma_(out, beta, *cp);
    if (epi.herm_diag && i == j) out = Tr<T>::realpart(out);
    *cp = out;
}

// Split-K partial sums live in a per-stream scratch slot: with the two-stream overlap options gemms on c.s1 and c.s2
// may both take the split path at the same time.
static const char* splitk_slot(const Ctx& c, hipStream_t st) { return (c.s2 && st == c.s2) ? "splitk_s2" : ((c.s3 && st == c.s3) ? "splitk_s3" : "splitk"); }

// Host side of tile_of: the map for a tm x tn tile grid (tri: only the stored triangle of a square grid is launched).
// Super-tile shapes of 64 tiles; the one that pads the grid least wins, squarer shapes preferred (2 % per factor of two
// of aspect ratio); grids of fewer than 128 tiles, batched launches and grids that would be padded by more than 15 %
// keep the plain enumerations.
template <class T> static dim3 choose_map(GemmArgs<T>& g, int tm, int tn, bool tri, int zdim, bool use_map) {
    g.map = tri ? TM_TRI : TM_GRID;
    g.mw = g.mh = g.msw = g.msh = g.mnsx = g.mfull = g.mnt = 0;
    const long active = tri ? (long)tm * (tm + 1) / 2 : (long)tm * tn;
    dim3 plain = tri ? dim3((unsigned)active, 1, zdim) : dim3(tm, tn, zdim);
    if (!use_map || g.bt.count > 0 || active < 128) return plain;   // (option "tile_map" = 0: A/B measurements of the map)
    int W = tm, H = tn, nte = 0;
    if (tri) { nte = tm + (tm & 1); W = nte / 2; H = nte + 1; }
    double bestc = 0.0;
    int bw = -1;
    long bpad = 0;
    for (int lw = 0; lw <= 6; ++lw) {
        const int lh = 6 - lw;
        const long nsx = (W + (1 << lw) - 1) >> lw, nsy = (H + (1 << lh) - 1) >> lh;
        const long pad = nsx * nsy * 64;
        const double cost = (double)pad * (1.0 + 0.02 * abs(lw - lh));
        if (bw < 0 || cost < bestc) { bestc = cost; bw = lw; bpad = pad; }
    }
    if ((double)bpad > 1.15 * (double)active) return plain;
    g.map = tri ? TM_FOLD : TM_RECT;
    g.mw = W; g.mh = H; g.msw = bw; g.msh = 6 - bw; g.mnsx = (W + (1 << bw) - 1) >> bw;
    g.mfull = (int)((bpad / 512) * 512);
    g.mnt = nte;
    return dim3((unsigned)bpad, 1, zdim);
}

template <class T, int BM, int BN>
static void launch_gemm(hipStream_t st, const GemmArgs<T>& g_in, int splits, bool use_map, int dma = 0, bool lean = false) {
    constexpr int BK = slab_k<T>(BM * BN <= 32 * 32);
    GemmArgs<T> g = g_in;
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const bool tri = g.epi.uplo != 0 && BM == BN && tm == tn;
    const dim3 grid = choose_map(g, tm, tn, tri, splits, use_map);
    const dim3 block(256);
    const int ta = g.A.trans, tb = g.B.trans;
    const bool masked = g.A.mask != M_NONE || g.B.mask != M_NONE;
    if constexpr (Tr<T>::cx && BM == 64 && BN == 64) {
        if (dma > 0) {     // option "gemm_dma": K-slabs staged by LDS-DMA (the operand kind is a pair of strides there: no TA / TB forms);
                           // dma = number of workgroups the work items are dealt to: INT_MAX = one item each (option value 1), two
                           // per CU = the persistent form with the slab pipeline running across tile boundaries (option value 2)
            const long total = (long)grid.x * grid.y * grid.z;
            const int pers = total > dma;
            const dim3 pg = pers ? dim3((unsigned)dma) : grid;
            if (lean) {
                if (masked) hipLaunchKernelGGL((gemm_dma_kernel<true, 8, 3>), grid, block, 0, st, g, (int)grid.x, (int)grid.y, (int)grid.z, 0);
                else hipLaunchKernelGGL((gemm_dma_kernel<false, 8, 3>), grid, block, 0, st, g, (int)grid.x, (int)grid.y, (int)grid.z, 0);
            } else if (masked) hipLaunchKernelGGL((gemm_dma_kernel<true, BKL, 2>), pg, block, 0, st, g, (int)grid.x, (int)grid.y, (int)grid.z, pers);
            else hipLaunchKernelGGL((gemm_dma_kernel<false, BKL, 2>), pg, block, 0, st, g, (int)grid.x, (int)grid.y, (int)grid.z, pers);
            EIG_HIP(hipGetLastError());
            return;
        }
    }
#define EIG_LAUNCH_FAST(TA_, TB_)                                                                                      \
    do {                                                                                                               \
        if (masked) hipLaunchKernelGGL((gemm_fast_kernel<T, BM, BN, TA_, TB_, BK, true>), grid, block, 0, st, g);      \
        else hipLaunchKernelGGL((gemm_fast_kernel<T, BM, BN, TA_, TB_, BK, false>), grid, block, 0, st, g);            \
    } while (0)
    if (ta == 0 && tb == 0) EIG_LAUNCH_FAST(0, 0);
    else if (ta == 0 && tb == 1) EIG_LAUNCH_FAST(0, 1);
    else if (ta == 1 && tb == 0) EIG_LAUNCH_FAST(1, 0);
    else EIG_LAUNCH_FAST(1, 1);
#undef EIG_LAUNCH_FAST
    EIG_HIP(hipGetLastError());
}

static void launch_gemm_wide(hipStream_t st, const GemmArgs<cplx>& g_in, int zdim, bool use_map, int kg) {
    GemmArgs<cplx> g = g_in;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    const bool tri = g.epi.uplo != 0 && tm == tn;
    const dim3 grid = choose_map(g, tm, tn, tri, zdim, use_map);
    const bool masked = g.A.mask != M_NONE || g.B.mask != M_NONE;
    if (kg == 4) {
        if (masked) hipLaunchKernelGGL((gemm_wide_kernel<true, 4>), grid, dim3(1024), 0, st, g);
        else hipLaunchKernelGGL((gemm_wide_kernel<false, 4>), grid, dim3(1024), 0, st, g);
    } else {
        if (masked) hipLaunchKernelGGL((gemm_wide_kernel<true, 2>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_wide_kernel<false, 2>), grid, dim3(512), 0, st, g);
    }
    EIG_HIP(hipGetLastError());
}
static void launch_gemm_wide(hipStream_t, const GemmArgs<double>&, int, bool, int) {}

template <class T> static void dispatch_gemm_now(Ctx& c, hipStream_t st, const GemmArgs<T>& g, int splits, int ngroup);

// one recorded product: its argument block and the z extent of its own launch
template <class T> struct GemmRec {
    GemmArgs<T> g;
    int splits;
};
// may b ride in a's launch?  (everything but the pointers must agree)
template <class T> static bool same_operand_shape(const Operand<T>& a, const Operand<T>& b) {
    return a.ld == b.ld && a.trans == b.trans && a.conj == b.conj && a.mask == b.mask && a.moff == b.moff && a.ld2 == b.ld2 && a.k1 == b.k1 &&
           (a.p2 == nullptr) == (b.p2 == nullptr);
}
template <class T> static bool same_batch(const GemmBatch& a, const GemmBatch& b) {
    return a.count == b.count && a.splits == b.splits && a.sA == b.sA && a.sB == b.sB && a.sC == b.sC && a.dMoffA == b.dMoffA &&
           a.dMoffB == b.dMoffB && a.dK == b.dK && a.capM == b.capM && a.capN == b.capN && a.capK == b.capK && a.dcap == b.dcap;
}
template <class T> static bool same_product_shape(const GemmRec<T>& x, const GemmRec<T>& y) {
    const GemmArgs<T>&a = x.g, &b = y.g;
    return x.splits == y.splits && a.M == b.M && a.N == b.N && a.K == b.K && real_(a.alpha) == real_(b.alpha) && imag_(a.alpha) == imag_(b.alpha) &&
           real_(a.beta) == real_(b.beta) && imag_(a.beta) == imag_(b.beta) && same_operand_shape(a.A, b.A) && same_operand_shape(a.B, b.B) &&
           a.ldc == b.ldc && a.epi.uplo == b.epi.uplo && a.epi.herm_diag == b.epi.herm_diag && a.epi.ldaux == b.epi.ldaux &&
           (a.epi.aux == nullptr) == (b.epi.aux == nullptr) && a.kchunk == b.kchunk && a.pstride == b.pstride && (a.P == nullptr) == (b.P == nullptr) &&
           same_batch<T>(a.bt, b.bt);
}

template <class T> static void dispatch_gemm(Ctx& c, hipStream_t st, const GemmArgs<T>& g, int splits) {
    if (g.M <= 0 || g.N <= 0) return;
    if (c.rec) {      // a problem of a lockstep group: the launch is recorded, replay_group() issues it (blas3.h)
        constexpr int type = Tr<T>::cx ? 2 : 1;
        auto rec = std::make_shared<GemmRec<T>>(GemmRec<T>{g, splits});
        Ctx* cp = &c;
        LaunchRec r;
        r.kind = 1;
        r.gemm_type = type;
        r.gemm_args = rec;
        r.run = [cp, rec](hipStream_t s) { dispatch_gemm_now<T>(*cp, s, rec->g, rec->splits, 1); };
        r.run_group = [cp, rec](hipStream_t s, const LaunchRec* const* peers, int n) -> bool {
            if (n < 2 || n > kGroupMax) return false;
            GemmArgs<T> g0 = rec->g;
            g0.grp.count = n; g0.grp.zper = rec->splits;
            for (int q = 0; q < n; ++q) {
                if (peers[q]->kind != 1 || peers[q]->gemm_type != type) return false;
                const GemmRec<T>& pq = *static_cast<const GemmRec<T>*>(peers[q]->gemm_args.get());
                if (!same_product_shape(*rec, pq)) return false;
                g0.grp.Ap[q] = pq.g.A.p; g0.grp.Ap2[q] = pq.g.A.p2; g0.grp.Bp[q] = pq.g.B.p; g0.grp.Bp2[q] = pq.g.B.p2;
                g0.grp.Cp[q] = pq.g.C; g0.grp.auxp[q] = reinterpret_cast<T*>(pq.g.epi.aux); g0.grp.Pp[q] = pq.g.P;
            }
            dispatch_gemm_now<T>(*cp, s, g0, rec->splits, n);
            return true;
        };
        c.rec->seq.push_back(std::move(r));
        return;
    }
    dispatch_gemm_now<T>(c, st, g, splits, 1);
}

void replay_group(hipStream_t st, GroupRecorder* recs, int n) {
    bool aligned = n >= 2;
    for (int q = 1; q < n && aligned; ++q) {
        aligned = recs[q].seq.size() == recs[0].seq.size();
        for (size_t i = 0; aligned && i < recs[0].seq.size(); ++i) aligned = recs[q].seq[i].kind == recs[0].seq[i].kind;
    }
    if (!aligned) {
        for (int q = 0; q < n; ++q)
            for (LaunchRec& r : recs[q].seq) r.run(st);
        return;
    }
    const LaunchRec* peers[kGroupMax];
    for (size_t i = 0; i < recs[0].seq.size(); ++i) {
        bool merged = false;
        if (recs[0].seq[i].kind == 1 && n <= kGroupMax) {
            for (int q = 0; q < n; ++q) peers[q] = &recs[q].seq[i];
            merged = recs[0].seq[i].run_group(st, peers, n);
        }
        if (!merged)
            for (int q = 0; q < n; ++q) recs[q].seq[i].run(st);
    }
    EIG_HIP(hipGetLastError());
}

// ngroup > 1: g carries a lockstep group (g.grp): ngroup times the z extent, the tile shape and data path of ONE problem's launch
template <class T> static void dispatch_gemm_now(Ctx& c, hipStream_t st, const GemmArgs<T>& g, int splits, int ngroup) {
    // pick the largest tile that still yields about one workgroup per CU: a 64x64x64 complex
    // tile is ~256 dependent MFMAs per wave (~15 us), so small problems want many small tiles.
    // (128x128 and 128x64 real tiles were measured in rounds 2-3 and lost inside the solver; they are gone.)
    const long tiles64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * splits;
    if (tiles64 >= c.n_cu) {
        // staging path (option "gemm_dma"): LDS-DMA pays from ~6 slabs per work item on
        const int kitem = g.kchunk > 0 ? g.kchunk : g.K + (g.bt.count > 0 && g.bt.dK > 0 ? (g.bt.count - 1) * g.bt.dK : 0);
        int dma = c.gemm_dma == 0 ? 0 : ((c.gemm_dma == 2 && ngroup == 1) ? 2 * c.n_cu : INT_MAX);
        if (c.gemm_dma == 3 && kitem < kGemmDmaMinK) dma = 0;
        // short-K items: the lean LDS-DMA form (three workgroups per CU), option "gemm_lean" = the largest K it serves
        // (only where the launch is several rounds of tiles: with <= 2 rounds the lean form's later C fetch is exposed -- her2k n = 1024
        //  16.0 -> 18.6 us, n = 2048 unchanged, n >= 3000 -5..7 %)
        const long tm64 = (g.M + 63) / 64, tn64 = (g.N + 63) / 64;
        const long active = (g.epi.uplo != 0 && tm64 == tn64 ? tm64 * (tm64 + 1) / 2 : tm64 * tn64) * splits * ngroup;
        const bool lean = Tr<T>::cx && c.gemm_lean > 0 && kitem <= c.gemm_lean && active >= 4L * c.n_cu;
        if (lean) dma = INT_MAX;
        launch_gemm<T, 64, 64>(st, g, splits * ngroup, c.tile_map != 0, dma, lean);
    }
    else {
        if constexpr (Tr<T>::cx) {
            // whole-CU workgroups (gemm_wide_kernel): chosen from the problem's OWN tile count (not the group's) -- see the kernel
            const long items = (long)((g.M + 31) / 32) * ((g.N + 31) / 32) * splits;
            const int kitem = g.kchunk > 0 ? g.kchunk : g.K;
            int kg = 0;
            if (c.gemm_wide >= 1 && items <= c.n_cu) kg = 4;
            else if (c.gemm_wide >= 2 && items <= 2L * c.n_cu) kg = 2;
            if (kg > 0 && kitem >= BKL * kg) {
                launch_gemm_wide(st, g, splits * ngroup, c.tile_map != 0, kg);
                return;
            }
        }
        launch_gemm<T, 32, 32>(st, g, splits * ngroup, c.tile_map != 0);
    }
}

template <class T>
void gemm(Ctx& c, hipStream_t st, int M, int N, int K, T alpha, const Operand<T>& A, const Operand<T>& Bt, T beta, T* C,
          int ldc, Epi epi) {
    GemmArgs<T> g;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.A = A; g.B = Bt; g.C = C; g.ldc = ldc; g.epi = epi;
    g.kchunk = 0; g.P = nullptr; g.pstride = 0; g.bt = GemmBatch();
    // Tile quantisation: a grid whose 64x64 tiles do not fill the resident workgroups (2 per CU) a whole number of
    // times leaves most of the chip idle in the last round (an upper-triangle update of order 2048 is 528 tiles on
    // 512 slots: 2 rounds for 1.03 rounds of work).  When K is long enough, split it so that the work items are
    // short and many; the partial sums cost one pass over C and one more launch.
    if constexpr (Tr<T>::cx) {
        const bool masked = A.mask != M_NONE || Bt.mask != M_NONE;
        if (!masked && K >= 1024) {
            const long tm = (M + 63) / 64, tn = (N + 63) / 64, tmin = tm < tn ? tm : tn;
            long tiles = tm * tn;
            if (epi.uplo != 0) tiles = tmin * (tmin + 1) / 2 + (epi.uplo == 1 ? (tn - tmin) * tm : (tm - tmin) * tn);
            const long slots = 2L * c.n_cu;
            if (tiles >= c.n_cu / 2) {
                const double u = 0.305;   // us per unit of K for one 64x64 complex tile with 2 workgroups per CU
                int best = 1;
                double bestc = (double)((tiles + slots - 1) / slots) * K * u;
                const int cand[] = {2, 3, 4, 6, 8};
                for (int sp : cand) {
                    if (K / sp < 256) break;
                    if ((double)M * N * sp * sizeof(T) > 512e6) break;
                    double cost = (double)((tiles * sp + slots - 1) / slots) * ((double)K / sp) * u +
                                  2.0 * tiles * 4096.0 * sizeof(T) * sp / 5.0e6 + 6.0;
                    if (cost < 0.9 * bestc) { bestc = cost; best = sp; }
                }
                if (best > 1) {
                    int kchunk = ((K + best - 1) / best + 63) & ~63;
                    g.kchunk = kchunk;
                    int splits = (K + kchunk - 1) / kchunk;
                    g.pstride = (size_t)M * N;
                    g.P = c.scratch<T>(splitk_slot(c, st), g.pstride * splits);
                    dispatch_gemm(c, st, g, splits);
                    size_t total = (size_t)M * N;
                    klaunch(c, st, (splitk_reduce_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), M, N,
                            splits, (const T*)g.P, g.pstride, alpha, beta, C, ldc, epi, GemmBatch());
                    EIG_HIP(hipGetLastError());
                    return;
                }
            }
        }
    }
    dispatch_gemm(c, st, g, 1);
}

template <class T>
void gemm_splitk(Ctx& c, hipStream_t st, int M, int N, int K, T alpha, const Operand<T>& A, const Operand<T>& Bt, T beta,
                 T* C, int ldc, int kchunk, Epi epi) {
    if (M <= 0 || N <= 0) return;
    kchunk = ((kchunk + BKS - 1) / BKS) * BKS;
    int splits = (K + kchunk - 1) / kchunk;
    if (splits <= 1) {
        gemm(c, st, M, N, K, alpha, A, Bt, beta, C, ldc, epi);
        return;
    }
    GemmArgs<T> g;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.A = A; g.B = Bt; g.C = C; g.ldc = ldc; g.epi = epi;
    g.kchunk = kchunk; g.bt = GemmBatch();
    g.pstride = (size_t)M * N;
    g.P = c.scratch<T>(splitk_slot(c, st), g.pstride * splits);
    dispatch_gemm(c, st, g, splits);
    size_t total = (size_t)M * N;
    klaunch(c, st, (splitk_reduce_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), M, N, splits,
            (const T*)g.P, g.pstride, alpha, beta, C, ldc, epi, GemmBatch());
    EIG_HIP(hipGetLastError());
}

// Strided batch of `bt.count` products of one shape in ONE launch (optionally split along K): the many small independent
// products of the back-transformation's T factors (zheevd_gpu.F90:136-176 builds them one reflector block at a time).
template <class T>
void gemm_batched(Ctx& c, hipStream_t st, int M, int N, int K, T alpha, const Operand<T>& A, const Operand<T>& Bt, T beta,
                  T* C, int ldc, Epi epi, GemmBatch bt, int kchunk) {
    if (M <= 0 || N <= 0 || bt.count <= 0) return;
    const int Kmax = K + (bt.dK > 0 ? (bt.count - 1) * bt.dK : 0);
    int splits = 1;
    if (kchunk > 0) {
        kchunk = ((kchunk + BKS - 1) / BKS) * BKS;
        splits = (Kmax + kchunk - 1) / kchunk;
    }
    GemmArgs<T> g;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.A = A; g.B = Bt; g.C = C; g.ldc = ldc; g.epi = epi;
    g.kchunk = splits > 1 ? kchunk : 0; g.P = nullptr; g.pstride = 0;
    bt.splits = splits;
    g.bt = bt;
    if (splits > 1) {
        g.pstride = (size_t)M * N;
        g.P = c.scratch<T>(splitk_slot(c, st), g.pstride * splits * bt.count);
    }
    dispatch_gemm(c, st, g, splits * bt.count);
    if (splits > 1) {
        size_t total = (size_t)M * N;
        klaunch(c, st, (splitk_reduce_kernel<T>), dim3((unsigned)((total + 255) / 256), bt.count), dim3(256), M, N,
                splits, (const T*)g.P, g.pstride, alpha, beta, C, ldc, epi, bt);
        EIG_HIP(hipGetLastError());
    }
}


template <class T> void her2k_un(Ctx& c, hipStream_t st, int n, int k, const T* V, int ldv, const T* W, int ldw, T* C, int ldc) {
    if (n <= 0 || k <= 0) return;
    Operand<T> A, B;
    A.p = V; A.ld = ldv; A.trans = 0; A.conj = 0; A.k1 = k; A.p2 = W; A.ld2 = ldw;
    B.p = W; B.ld = ldw; B.trans = 0; B.conj = 1; B.k1 = k; B.p2 = V; B.ld2 = ldv;
    Epi e; e.uplo = 1; e.herm_diag = 1;
    gemm<T>(c, st, n, n, 2 * k, Tr<T>::make(-1.0, 0.0), A, B, Tr<T>::one(), C, ldc, e);
}

// ------------------------------------------------------------------------------------------
// 64x64 base kernels (one workgroup, matrix resident in LDS)
// ------------------------------------------------------------------------------------------
constexpr int DB = kDiagBlk;
constexpr int DBL = DB + 1;  // LDS leading dimension
constexpr int PB = 2;            // diag_block_kernel: each thread owns a PB x PB block ...
constexpr int NTD = DB / PB;     // ... of the 64x64 matrix: NTD x NTD threads
constexpr int DGT = NTD * NTD;   // = 1024: four waves per SIMD, to overlap the 11-cycle latency of dependent fp64 FMAs

// 64x64 upper Cholesky (optional) followed by the inverse of the factor, one workgroup.
// Register-resident: thread (tr, tc) = (tid/NTD, tid%NTD) owns the PB x PB block rows PB*tr.., cols PB*tc..
// of U (and of X = U^-1); PB = 2 -> 1024 threads = four waves per SIMD (a dependent fp64 FMA has 11 cycles of latency
// against 4 of issue, and the elimination is one long dependency chain: one wave per SIMD leaves the pipe idle).  Each of the 64 elimination steps broadcasts one row (and for the inverse
// one column) through a double-buffered LDS line and costs a single barrier.  The step loops are
// unrolled over the position inside the PB x PB block so that every register index is static; the pivot
// is a Newton-refined v_rsq_f64 (no IEEE sqrt / division on the 64-step chain) and its reciprocal is
// kept for the inversion, which then has no division at all.  This kernel sits on the critical path
// of potrf N/64 times.
//   do_chol = 1: block <- chol(block) (upper), written back; info <- first bad pivot (1-based, global)
//   inverse written to invU (DB x DB, ld DB, identity-padded, zero below the diagonal).
template <class T>
__global__ void __launch_bounds__(DGT) diag_block_kernel(int n_total, T* Umat, int ldu, T* invU, int do_chol, int k0_single,
                                                         int* info, int blk0) {
    __shared__ T rowb[2][DB];
    __shared__ T colb[2][DB];
    __shared__ T dinvs[DB];   // reciprocals of the diagonal of U
    const int tid = threadIdx.x;
    const int tr = tid / NTD, tc = tid % NTD;
    const int blk = (k0_single >= 0) ? k0_single / DB : blk0 + (int)blockIdx.x;
    const int k0 = blk * DB;
    const int nb = min(DB, n_total - k0);
    T* Ublk = Umat + (size_t)k0 + (size_t)k0 * ldu;
    T* inv = invU + (size_t)blk * DB * DB;

    T u[PB][PB], x[PB][PB];
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            int r = PB * tr + i, cc = PB * tc + j;
            const bool in = r < nb && cc < nb && r <= cc;
            T v = Ublk[(size_t)min(r, nb - 1) + (size_t)min(cc, nb - 1) * ldu];
            u[i][j] = sel(in, v, sel(r == cc, Tr<T>::one(), Tr<T>::zero()));
            x[i][j] = (r == cc) ? Tr<T>::one() : Tr<T>::zero();
        }

    const bool upper_blk = tc >= tr;   // only blocks on or above the block diagonal carry data
    const bool diag_blk = tc == tr;
    if (do_chol) {
        for (int jb = 0; jb < DB / PB; ++jb) {
#pragma unroll
            for (int jj = 0; jj < PB; ++jj) {
                const int j = PB * jb + jj, buf = jj & 1;
                if (tr == jb) {
#pragma unroll
                    for (int q = 0; q < PB; ++q) rowb[buf][PB * tc + q] = u[jj][q];
                }
                __syncthreads();
                double d = real_(rowb[buf][j]);
                if (!(d > 0.0)) {
                    if (tid == 0 && j < nb) atomicCAS(info, 0, k0 + j + 1);
                    d = 1.0;
                }
                if (upper_blk && tr >= jb) {   // rows >= j only; roles are static inside a block phase
                    const double ipiv = fast_rsqrt(d);
                    double piv = d * ipiv;
                    piv = fma(fma(-piv, piv, d), 0.5 * ipiv, piv);
                    T uc[PB], ur[PB];
#pragma unroll
                    for (int q = 0; q < PB; ++q) {
                        uc[q] = rowb[buf][PB * tc + q] * ipiv;   // u(j, c) for my columns
                        ur[q] = rowb[buf][PB * tr + q] * ipiv;   // u(j, r) for my rows
                    }
                    if (tr > jb) {
#pragma unroll
                        for (int i = 0; i < PB; ++i)
#pragma unroll
                            for (int q = 0; q < PB; ++q) {
                                T t = Tr<T>::zero();
                                fmac_(t, ur[i], uc[q]);
                                u[i][q] = u[i][q] - t;    // (entries below the diagonal of a diagonal block are never read)
                            }
                    } else {
                        if (diag_blk) dinvs[j] = Tr<T>::make(ipiv, 0.0);
#pragma unroll
                        for (int q = 0; q < PB; ++q) {
                            if (!diag_blk || q > jj) u[jj][q] = uc[q];
                            else if (q == jj) u[jj][q] = Tr<T>::make(piv, 0.0);
                        }
#pragma unroll
                        for (int i = jj + 1; i < PB; ++i)
#pragma unroll
                            for (int q = 0; q < PB; ++q) {
                                T t = Tr<T>::zero();
                                fmac_(t, ur[i], uc[q]);
                                u[i][q] = u[i][q] - t;
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                int r = PB * tr + i, cc = PB * tc + q;
                if (r < nb && cc < nb && r <= cc) Ublk[(size_t)r + (size_t)cc * ldu] = u[i][q];
            }
    } else if (diag_blk) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const T dgn = u[i][i];
            dinvs[PB * tr + i] = conj_(dgn) * (1.0 / abs2_(dgn));
        }
    }
    __syncthreads();

    // X = U^-1 by right-looking back substitution on the rows, bottom up
    for (int ib = DB / PB - 1; ib >= 0; --ib) {
#pragma unroll
        for (int ii = PB - 1; ii >= 0; --ii) {
            const int i2 = PB * ib + ii, buf = ii & 1;
            if (tr == ib) {
#pragma unroll
                for (int q = 0; q < PB; ++q) rowb[buf][PB * tc + q] = x[ii][q];
            }
            if (tc == ib) {
#pragma unroll
                for (int i = 0; i < PB; ++i) colb[buf][PB * tr + i] = u[i][ii];
            }
            __syncthreads();
            if (upper_blk && tr <= ib && tc >= ib) {   // rows <= i2, columns >= i2
                const T dinv = dinvs[i2];
                T xr[PB], uc2[PB];
#pragma unroll
                for (int q = 0; q < PB; ++q) {
                    xr[q] = rowb[buf][PB * tc + q] * dinv;   // final x(i2, c)   (zero for c < i2)
                    uc2[q] = colb[buf][PB * tr + q];         // u(r, i2)
                }
                if (tr < ib) {
#pragma unroll
                    for (int i = 0; i < PB; ++i)
#pragma unroll
                        for (int q = 0; q < PB; ++q) {
                            T t = Tr<T>::zero();
                            fma_(t, uc2[i], xr[q]);
                            x[i][q] = x[i][q] - t;
                        }
                } else {
#pragma unroll
                    for (int q = 0; q < PB; ++q) x[ii][q] = xr[q];
#pragma unroll
                    for (int i = 0; i < ii; ++i)
#pragma unroll
                        for (int q = 0; q < PB; ++q) {
                            T t = Tr<T>::zero();
                            fma_(t, uc2[i], xr[q]);
                            x[i][q] = x[i][q] - t;
                        }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            int r = PB * tr + i, cc = PB * tc + q;
            inv[r + cc * DB] = (r <= cc) ? x[i][q] : Tr<T>::zero();
        }
}

// ------------------------------------------------------------------------------------------------
// chol_row_kernel: one block row of the right-looking blocked Cholesky (upper), WITHOUT any inverse:
//     U_kk = chol(B_kk),   U(k, chunk c) = U_kk^-H B(k, chunk c)   for every 64-column chunk c of the block row.
// Workgroup c owns chunk c (chunk 0 = the diagonal block itself) and carries it through the 64 elimination steps of the
// diagonal block, which every workgroup repeats on its own copy (64^3/3 multiply-adds: nothing next to a launch and
// a dependent 64x64 triangular solve).  Row operations applied to [B_kk | B_kc] produce U_kc directly -- the same
// arithmetic as LAPACK's potf2 on the block row, no explicit inverse of U_kk on the factorization's critical path.
// Layout as in diag_block_kernel: thread (tr, tc) owns a 2x2 block of the diagonal block and the 2x2 block at the same
// position of its chunk; one double-buffered LDS broadcast of the pivot row and one barrier per elimination step.
// ------------------------------------------------------------------------------------------------
// The diagonal block is factored IN PLACE by workgroup 0 while the other workgroups read the original block: workgroup 0
// only stores U_kk once every other workgroup has announced (one agent-scope atomic on `loaded`, a cumulative counter
// of the current factorization) that its copy sits in registers.  Nobody waits for workgroup 0, so the spin cannot deadlock.
// (6 waves per SIMD = 80 VGPRs.  Round 3 forced 7 waves / 72 VGPRs so that the sixteen waves of a block-row workgroup fit on a CU
// beside ONE 224-VGPR product workgroup of the overlapped hegst chain; the complex instantiation then spilled 8 bytes to scratch.
// Re-measured in round 4: potrf || hegst 12.25 vs 12.26 ms, one stream 5.60 vs 5.62 -- no difference, no spill.)
// up to four factorizations of one order side by side (blockIdx.y = problem: the lockstep groups of a batch call); every
// problem has its own info word and its own announce counter
constexpr int CHOL_MAXB = 4;
template <class T> struct CholBatch { T* B[CHOL_MAXB]; };
template <class T>
__global__ void __launch_bounds__(DGT) __attribute__((amdgpu_waves_per_eu(6, 6)))
chol_row_kernel(int n_total, CholBatch<T> cb, int ldb, int k0, int* info0, unsigned* loaded0, unsigned expect) {
    __shared__ T rowb[2][DB];   // pivot row of the diagonal block
    __shared__ T rowp[2][DB];   // pivot row of this workgroup's chunk
    T* const Bm = cb.B[blockIdx.y];
    int* const info = info0 + blockIdx.y;
    unsigned* const loaded = loaded0 + blockIdx.y;
    const int tid = threadIdx.x;
    const int tr = tid / NTD, tc = tid % NTD;
    const int chunk = blockIdx.x;
    const bool has_p = chunk > 0;
    const int nb = min(DB, n_total - k0);
    const int c0 = k0 + chunk * DB;                 // first column of the chunk
    const int pc = min(DB, n_total - c0);
    T* Dblk = Bm + (size_t)k0 + (size_t)k0 * ldb;
    T* Pblk = Bm + (size_t)k0 + (size_t)c0 * ldb;

    T u[PB][PB], p[PB][PB];
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int r = PB * tr + i, cc = PB * tc + j;
            const bool in = r < nb && cc < nb && r <= cc;
            const T v = Dblk[(size_t)min(r, nb - 1) + (size_t)min(cc, nb - 1) * ldb];
            u[i][j] = sel(in, v, sel(r == cc, Tr<T>::one(), Tr<T>::zero()));
            const T w = Pblk[(size_t)min(r, nb - 1) + (size_t)min(cc, pc - 1) * ldb];
            p[i][j] = sel(has_p && r < nb && cc < pc, w, Tr<T>::zero());
        }
    if (has_p) {
        // the loads above have landed once their values are consumed; make that explicit, then announce
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(loaded, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const bool upper_blk = tc >= tr;
    const bool diag_blk = tc == tr;
    static_assert(PB == 2, "the look-ahead schedule below is written for 2x2 register blocks");
    // Row j is scaled by its owners (the 32 threads of block row j/2: one half wave, the pivot comes from the diagonal
    // thread by v_readlane) BEFORE it is broadcast, and it is published one step ahead: right after the barrier of step j
    // the owners of row j+1 update only that row, take the reciprocal square root, scale and write the row to the other
    // LDS buffer, and only then do their share of the bulk update -- the rsqrt chain of the next pivot overlaps with the
    // rank-1 update of everybody else, and nobody but the owners evaluates it.
    auto publish = [&](auto JJ, int jbn, int bufn) {
        constexpr int jj = decltype(JJ)::value;
        double dd = read_lane(real_(u[jj][jj]), (jbn & 1) * 32 + jbn);   // u(j,j) of the diagonal thread (tc == tr == jbn)
        const int j = PB * jbn + jj;
        if (!(dd > 0.0)) {
            if (diag_blk && chunk == 0 && j < nb) atomicCAS(info, 0, k0 + j + 1);
            dd = 1.0;
        }
        const double ipiv = fast_rsqrt(dd);
        double piv = dd * ipiv;
        piv = fma(fma(-piv, piv, dd), 0.5 * ipiv, piv);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            u[jj][q] = (diag_blk && q == jj) ? Tr<T>::make(piv, 0.0) : u[jj][q] * ipiv;
            p[jj][q] = p[jj][q] * ipiv;
            rowb[bufn][PB * tc + q] = u[jj][q];
            rowp[bufn][PB * tc + q] = p[jj][q];
        }
    };
    // rank-1 update of local row i with the scaled pivot row (ur = u(j, my row i), uc / pr = u(j, my columns))
    auto upd_row = [&](auto II, const T (&ur)[PB], const T (&uc)[PB], const T (&pr)[PB]) {
        constexpr int i = decltype(II)::value;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            if (upper_blk) {
                T t = Tr<T>::zero();
                fmac_(t, ur[i], uc[q]);
                u[i][q] = u[i][q] - t;
            }
            if (has_p) {
                T t = Tr<T>::zero();
                fmac_(t, ur[i], pr[q]);
                p[i][q] = p[i][q] - t;
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (tr == 0) publish(I0{}, 0, 0);
    for (int jb = 0; jb < DB / PB; ++jb) {
        // ---- step j = 2 jb (local row 0 of block row jb is the pivot row) ----
        __syncthreads();
        if (tr >= jb) {
            T ur[PB], uc[PB], pr[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                ur[q] = rowb[0][PB * tr + q];
                uc[q] = rowb[0][PB * tc + q];
                pr[q] = rowp[0][PB * tc + q];
            }
            if (tr == jb) {
                upd_row(I1{}, ur, uc, pr);       // the next pivot row first ...
                publish(I1{}, jb, 1);            // ... scaled and published one step ahead
            } else {
                upd_row(I0{}, ur, uc, pr);
                upd_row(I1{}, ur, uc, pr);
            }
        }
        // ---- step j = 2 jb + 1 (local row 1 is the pivot row; the next one is local row 0 of block row jb + 1) ----
        __syncthreads();
        if (tr > jb) {
            T ur[PB], uc[PB], pr[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                ur[q] = rowb[1][PB * tr + q];
                uc[q] = rowb[1][PB * tc + q];
                pr[q] = rowp[1][PB * tc + q];
            }
            upd_row(I0{}, ur, uc, pr);
            if (tr == jb + 1) publish(I0{}, jb + 1, 0);
            upd_row(I1{}, ur, uc, pr);
        }
    }
    if (!has_p) {
        if (tid == 0) {
            // Bounded: the other workgroups of this grid have nothing to wait for, so they announce themselves as soon as they
            // are dispatched -- but if the environment serialises workgroup dispatch (one-CU HSA_CU_MASK, a debugger), this
            // workgroup must not spin forever: after ~0.1 s it gives up and flags the factorization as failed (info = -4096 - k0,
            // reported like a bad pivot) instead of hanging the queue; option "potrf" = 0 has no intra-grid dependency.
            long spins = 0;
            while ((int)(__hip_atomic_load(loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1L << 22)) { atomicCAS(info, 0, -4096 - k0); break; }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int r = PB * tr + i, cc = PB * tc + q;
            if (!has_p) {
                if (r < nb && cc < nb && r <= cc) Dblk[(size_t)r + (size_t)cc * ldb] = u[i][q];
            } else {
                if (r < nb && cc < pc) Pblk[(size_t)r + (size_t)cc * ldb] = p[i][q];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// chol_row2_kernel (round 5): the same block row -- U_kk = chol(B_kk), U(k, chunk c) = U_kk^-H B(k, chunk c), one workgroup per
// 64-column chunk, every workgroup repeating the diagonal block -- with the elimination BLOCKED by 16 and everything but the
// 16x16 diagonal blocks on MFMA.  chol_row_kernel walks 64 dependent elimination steps with sixteen waves and a workgroup
// barrier per step (0.65 us per step, 42 us per block row, 2.7 ms of a C3 factorization with the chip idle); here a block row
// is 4 panels, a panel = ONE wave factoring and inverting a 16x16 block (16 steps, no workgroup barrier inside) + two
// barrier-separated MFMA stages for everybody.
//
// Layout.  The workgroup (4 waves) holds the TRANSPOSED block row L' = [B_kk | B_kc]^H (128 x 64: lower Cholesky L = U^H of the
// diagonal block on top, the chunk below) in MFMA accumulator layout: wave w owns the 16-row blocks rb = w (diagonal part) and
// rb = w + 4 (chunk part), four 16x16 column blocks each; lane l, component r of block (rb, cb) is
// L'(16 rb + (l & 15), 16 cb + 4 r + (l >> 4)).  In this form an accumulator block IS the A fragment sequence of a product that
// contracts over its column index (component r = k-step r), which is what both panel operations do:
//     strip      L(rb, p)  =  A(rb, p) * conj(Linv_pp)^T        (A operand: the block itself, B operand: Linv_pp from LDS)
//     trailing   A(rb, cb) -= L(rb, p) * L(cb, p)^H             (A operand: the strip block just formed, B operand: L(cb, p) from LDS)
// so no accumulator ever goes through LDS to become an operand; only the three diagonal-part strip blocks and the 16x16 inverse
// are published (2 workgroup barriers per panel, 8 per block row instead of 64).
// The 16x16 factorization: Gaussian elimination on [A_pp | I] -> [L^H | L^-1], lane (i, g) = (l & 15, l >> 4) holding row i,
// columns 4g..4g+3 of both halves; the pivot row and column travel through LDS inside the one wave (ds ops of a wave execute in
// order: no barrier), pivots by Newton-refined v_rsq_f64 as in chol_row_kernel.
// ------------------------------------------------------------------------------------------------
#ifndef EIG_CHOL2_SKIP
#define EIG_CHOL2_SKIP 0   // (timing variants: bit 0 no elimination, 1 no strip products, 2 no trailing products, 3 no stores, 4 no loads, 5 no announce)
#endif
constexpr int PW = 16;         // panel width of the blocked elimination
constexpr int PLD = PW + 1;    // leading dimension of the 16x16 LDS blocks

// c(m, n) +/-= sum_k a(m, k) conj(b(n, k)) for one 16x16x16 block: a = an accumulator block (component ks = k-step ks),
// bf[ks][r] = b(4r + (lane & 3), 4ks + (lane >> 4))
template <class T, bool NEG>
__device__ __forceinline__ void mma16_conjb(double (&cr)[4], double (&ci)[4], const double (&xr)[4], const double (&xi)[4],
                                            const T (&bf)[4][4]) {
    constexpr bool CX = Tr<T>::cx;
    constexpr int NG = NEG ? 1 : 0;
    // (consecutive MFMAs write different accumulators: a dependent v_mfma_f64_4x4x4 waits for the whole pipeline)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cr[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(real_(bf[ks][r]), xr[ks], cr[r], 0, 0, NG);
        if (CX) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ci[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(real_(bf[ks][r]), xi[ks], ci[r], 0, 0, NG);
#pragma unroll
            for (int r = 0; r < 4; ++r) cr[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(imag_(bf[ks][r]), xi[ks], cr[r], 0, 0, NG);
#pragma unroll
            for (int r = 0; r < 4; ++r) ci[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(imag_(bf[ks][r]), xr[ks], ci[r], 0, 0, 1 - NG);
        }
    }
}

template <class T>
__global__ void __launch_bounds__(256) chol_row2_kernel(int n_total, CholBatch<T> cb, int ldb, int k0, int* info0,
                                                        unsigned* loaded0, unsigned expect) {
    constexpr bool CX = Tr<T>::cx;
    constexpr int BLK = PW * PLD;
    __shared__ T s_A[BLK];            // diagonal 16x16 block: staging, then L^H (scaled rows)
    __shared__ T s_I[3][BLK];         // L_pp^-1 (lower), [n][k]; slot p % 3 (the waves that run ahead / catch up still read older panels)
    __shared__ T s_L[3][3][BLK];      // L(rb, p), rb = 1..3 of the diagonal part, [n][k]; slot p % 3
    __shared__ T s_col[PW * PW];      // the pivot columns (unscaled) of the 16 elimination steps, [k][i]
    __shared__ double s_piv[PW];      // l(i, i)
    T* const Bm = cb.B[blockIdx.y];
    int* const info = info0 + blockIdx.y;
    unsigned* const loaded = loaded0 + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.x;
    const bool has_p = chunk > 0;
    const int nb = min(DB, n_total - k0);
    const int c0 = k0 + chunk * DB;
    const int pc = min(DB, n_total - c0);
    const int fm = lane & 15, fq = lane >> 4;
    T* const Dblk = Bm + (size_t)k0 + (size_t)k0 * ldb;
    T* const Pblk = Bm + (size_t)k0 + (size_t)c0 * ldb;

    double ar_[2][4][4], ai_[2][4][4];
    const int a_ = 16 * w + fm;                       // column of B inside the 64-block = row of L'
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b_ = 16 * q + 4 * r + fq;   // row of B inside the block row = column of L'
                T v = Tr<T>::zero();
                if (EIG_CHOL2_SKIP & 16) {
                } else if (h == 0) {
                    if (q <= w) {
                        const T x = Dblk[(size_t)min(b_, nb - 1) + (size_t)min(a_, nb - 1) * ldb];
                        const bool in = b_ <= a_ && a_ < nb;
                        v = sel(in, conj_(x), sel(a_ == b_, Tr<T>::one(), Tr<T>::zero()));
                        if (a_ == b_) v = Tr<T>::realpart(v);
                    }
                } else if (has_p) {
                    const T x = Pblk[(size_t)min(b_, nb - 1) + (size_t)min(a_, pc - 1) * ldb];
                    v = sel(b_ < nb && a_ < pc, conj_(x), Tr<T>::zero());
                }
                ar_[h][q][r] = real_(v);
                ai_[h][q][r] = imag_(v);
            }

    auto frag = [&](const T* blk, T (&bf)[4][4]) {     // B fragments of a 16x16 LDS block [n][k]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) bf[ks][r] = blk[(4 * r + (lane & 3)) * PLD + 4 * ks + fq];
    };
    // block (h, P) <- block (h, P) * conj(Linv)^T
    auto strip = [&](auto H_, auto P_, const T (&bI)[4][4]) {
        constexpr int h = decltype(H_)::value, p = decltype(P_)::value;
        if (EIG_CHOL2_SKIP & 2) return;
        double xr[4] = {0.0, 0.0, 0.0, 0.0}, xi[4] = {0.0, 0.0, 0.0, 0.0};
        mma16_conjb<T, false>(xr, xi, ar_[h][p], ai_[h][p], bI);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ar_[h][p][r] = xr[r]; ai_[h][p][r] = CX ? xi[r] : 0.0; }
    };
    // block (h, Q) -= block (h, P) * L(Q, P)^H
    auto trail = [&](auto H_, auto Q_, auto P_, const T (&bL)[4][4]) {
        constexpr int h = decltype(H_)::value, q = decltype(Q_)::value, p = decltype(P_)::value;
        if (EIG_CHOL2_SKIP & 4) return;
        mma16_conjb<T, true>(ar_[h][q], ai_[h][q], ar_[h][p], ai_[h][p], bL);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- the 16x16 diagonal block of panel p (one wave): A_pp -> L_pp (into the accumulator block), L_pp^-1 (into s_I) ----
    auto factor = [&](auto P_) {
        constexpr int p = decltype(P_)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) s_A[fm * PLD + 4 * r + fq] = Tr<T>::make(ar_[0][p][r], ai_[0][p][r]);
        __builtin_amdgcn_wave_barrier();
        const int i = fm, g = fq;
        T left[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 4 * g + c;
            const T lo = s_A[i * PLD + j], up = s_A[j * PLD + i];
            T v = sel(i >= j, lo, conj_(up));
            if (i == j) v = Tr<T>::realpart(v);
            left[c] = v;
        }
        __builtin_amdgcn_wave_barrier();
        // Gaussian elimination on the (full, Hermitian) block, lane (i, g) = row i, columns 4g..4g+3.  Row k is NOT scaled when
        // it becomes the pivot row: rows below take  row_i -= (a_ik / d) row_k  (one reciprocal on the chain), every row is
        // scaled by its own 1 / sqrt(d_i) after the loop.  The pivot row is the conjugate of the pivot COLUMN (the Schur
        // complement is Hermitian), so one LDS line per step carries everything, and the only thing the next pivot waits for is
        // the next column: one multiply-add per lane, published before the other three are issued.  A bad pivot is recorded
        // without a branch.  (First version: scaled pivot row, the inverse carried along as a right-hand side, per-value
        // selects: 150 instructions + 18 wide LDS operations per step, 0.48 us per step -- one wave gets a fraction of the LDS
        // rates, MI355X_MICROARCH.md "LDS".)
        //
        // L^-1 rides along as a second, independent dependency chain in the same loop (forward substitution, column j = l & 15 in
        // all four 16-lane rows).  With c_k = the unscaled pivot column of step k (all 16 stay in LDS), l_ik = c_k[i] / sqrt(d_k),
        // the substitution  x_i = -(sum_{k<i} l_ik x_k) / l_ii,  x_j = 1 / l_jj  becomes, for z_k = x_k / sqrt(d_k),
        //     z_i = -(sum_{k<i} c_k[i] z_k) / d_i,   z_j = 1 / d_j      (x_k = 0 above the diagonal comes out by itself)
        // -- only the reciprocal the elimination computes anyway; x_i = z_i sqrt(d_i) after the loop.  The four lanes of a column
        // split the sum (lane row fq takes k = 4t + fq and keeps exactly those z_k) and add the partial sums with two half / row
        // exchanges.  As a loop of its own behind the elimination this substitution cost as much as the elimination (two latency
        // chains in a row); interleaved, each hides in the other's stalls.
        double myd = 1.0;
        int badk = PW;
        T zq[4] = {Tr<T>::zero(), Tr<T>::zero(), Tr<T>::zero(), Tr<T>::zero()};
        const int j = fm;
        if (g == 0) s_col[i] = left[0];
#pragma unroll
        for (int k = 0; k < ((EIG_CHOL2_SKIP & 1) ? 0 : PW); ++k) {
            T* const colk = s_col + k * PW;
            // (substitution, row k: the columns it reads were published in earlier steps)
            T sum = Tr<T>::zero();
#pragma unroll
            for (int t = 0; 4 * t < k; ++t)      // (columns k' >= k are not there yet: whatever the slot holds, possibly NaN, must not meet z = 0)
                fma_(sum, sel(4 * t + fq < k, s_col[(4 * t + fq) * PW + k], Tr<T>::zero()), zq[t]);
            __builtin_amdgcn_wave_barrier();
            double d = real_(colk[k]);
            const T ci_ = colk[i];
            T cj[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) cj[c] = conj_(colk[4 * g + c]);
            __builtin_amdgcn_wave_barrier();
            const bool ok = d > 0.0;
            badk = (!ok && badk == PW) ? k : badk;
            d = ok ? d : 1.0;
            const double invd = fast_rcp(d);
            const T f = sel(i > k, ci_ * invd, Tr<T>::zero());     // a(i, k) / d for the rows below the pivot row, else 0
            const int k1 = k + 1, kc1 = k1 & 3;
            if (k1 < PW) {
                fms_(left[kc1], f, cj[kc1]);                                  // the next pivot column first ...
                if (g == (k1 >> 2)) s_col[k1 * PW + i] = left[kc1];           // ... and published at once
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (!(k1 < PW && c == kc1)) fms_(left[c], f, cj[c]);
            myd = (i == k) ? d : myd;
            if (k > 0) {
                T o = sum;
                swap_halves(sum, o);
                sum = sum + o;
                o = sum;
                swap_rows(sum, o);
                sum = sum + o;
            }
            const T zk = (k == j) ? Tr<T>::make(invd, 0.0) : sum * (-invd);
            zq[k >> 2] = sel((k & 3) == fq, zk, zq[k >> 2]);
        }
        if (badk < PW && lane == 0 && chunk == 0 && 16 * p + badk < nb) atomicCAS(info, 0, k0 + 16 * p + badk + 1);
        {
            const double ip = fast_rsqrt(myd);
            double piv = myd * ip;
            piv = fma(fma(-piv, piv, myd), 0.5 * ip, piv);
#pragma unroll
            for (int c = 0; c < 4; ++c) s_A[i * PLD + 4 * g + c] = (4 * g + c == i) ? Tr<T>::make(piv, 0.0) : left[c] * ip;   // L^H(i, j), j >= i
            if (g == 0) s_piv[i] = piv;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {               // L(m, n) = conj(L^H(n, m)), n <= m, back into the accumulator block
            const int n = 4 * r + fq;
            const T v = conj_(s_A[n * PLD + fm]);
            ar_[0][p][r] = n <= fm ? real_(v) : 0.0;
            ai_[0][p][r] = n <= fm ? imag_(v) : 0.0;
        }
        {
            T* const sI = s_I[p % 3];               // L^-1(i, j) = z_i sqrt(d_i), rows i = 4t + fq of column j from this lane
#pragma unroll
            for (int t = 0; t < 4; ++t) sI[(4 * t + fq) * PLD + j] = zq[t] * s_piv[4 * t + fq];
        }
    };

    // One panel (p a compile-time constant: every accumulator index is static).  Only the diagonal part feeds the chain of
    // factorizations; the schedule keeps everything else off it:
    //  * wave p+1 runs AHEAD: as soon as its own strip block of column p exists it updates the next diagonal block (the only
    //    operand is that block itself) and, after the barrier that publishes the strips, factors it while the other waves do the
    //    panel's trailing products;
    //  * the wave that has just factored (wave p) does all of its chunk-part work for panels p-1 (what it skipped while it was
    //    factoring) and p AFTER that barrier, beside the next factorization -- nobody waits for it;
    //  so a panel costs the chain: one 16x16 factorization + two block products + two barriers.  LDS blocks live in slot p % 3
    //  (a wave that catches up reads panel p-1's inverse and strips while panel p+1's are being written).
    auto panel = [&](auto P_) {
        constexpr int p = decltype(P_)::value;
        using PP = std::integral_constant<int, p>;
        if (w == p) factor(PP{});
        // (every load has landed = my copy of the diagonal part is in registers; free: three waves wait here for the first 16x16 block anyway)
        if (p == 0 && has_p && !(EIG_CHOL2_SKIP & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                                   // B1(p): L_pp^-1 is in s_I[p % 3]
        if (p == 0 && has_p && !(EIG_CHOL2_SKIP & 32) && tid == 0)
            __hip_atomic_fetch_add(loaded, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
            T bI[4][4];
            frag(s_I[p % 3], bI);
            if (w > p) {
                strip(I0{}, PP{}, bI);
#pragma unroll
                for (int r = 0; r < 4; ++r) s_L[p % 3][w - 1][fm * PLD + 4 * r + fq] = Tr<T>::make(ar_[0][p][r], ai_[0][p][r]);
            }
            if constexpr (p < 3) {
                if (w == p + 1) {
                    __builtin_amdgcn_wave_barrier();
                    T bL[4][4];
                    frag(s_L[p % 3][p], bL);
                    trail(I0{}, std::integral_constant<int, p + 1>{}, PP{}, bL);
                }
            }
            if (has_p && w != p + 1 && w != p) strip(I1{}, PP{}, bI);
        }
        __syncthreads();                                                                   // B2(p): the strips of column p are in s_L[p % 3]
        if (w == p) {
            if (has_p) {
                T bI[4][4], bL[4][4];
                if constexpr (p > 0) {       // panel p-1, skipped while this wave was factoring: the chunk-part blocks
                    using PM = std::integral_constant<int, p - 1>;
                    frag(s_I[(p - 1) % 3], bI);
                    strip(I1{}, PM{}, bI);
                    frag(s_L[(p - 1) % 3][p - 1], bL); trail(I1{}, PP{}, PM{}, bL);
                    if constexpr (p + 1 < 4) { frag(s_L[(p - 1) % 3][p], bL); trail(I1{}, std::integral_constant<int, p + 1>{}, PM{}, bL); }
                    if constexpr (p + 2 < 4) { frag(s_L[(p - 1) % 3][p + 1], bL); trail(I1{}, std::integral_constant<int, p + 2>{}, PM{}, bL); }
                }
                frag(s_I[p % 3], bI);
                strip(I1{}, PP{}, bI);
                if constexpr (p + 1 < 4) { frag(s_L[p % 3][p], bL); trail(I1{}, std::integral_constant<int, p + 1>{}, PP{}, bL); }
                if constexpr (p + 2 < 4) { frag(s_L[p % 3][p + 1], bL); trail(I1{}, std::integral_constant<int, p + 2>{}, PP{}, bL); }
                if constexpr (p + 3 < 4) { frag(s_L[p % 3][p + 2], bL); trail(I1{}, std::integral_constant<int, p + 3>{}, PP{}, bL); }
            }
        } else if (w != p + 1) {
            T bL[4][4];
            if constexpr (p + 1 < 4) {
                using Q = std::integral_constant<int, p + 1>;
                frag(s_L[p % 3][p], bL);
                if (w >= p + 1) trail(I0{}, Q{}, PP{}, bL);
                if (has_p) trail(I1{}, Q{}, PP{}, bL);
            }
            if constexpr (p + 2 < 4) {
                using Q = std::integral_constant<int, p + 2>;
                frag(s_L[p % 3][p + 1], bL);
                if (w >= p + 2) trail(I0{}, Q{}, PP{}, bL);
                if (has_p) trail(I1{}, Q{}, PP{}, bL);
            }
            if constexpr (p + 3 < 4) {
                using Q = std::integral_constant<int, p + 3>;
                frag(s_L[p % 3][p + 2], bL);
                if (w >= p + 3) trail(I0{}, Q{}, PP{}, bL);
                if (has_p) trail(I1{}, Q{}, PP{}, bL);
            }
        }
    };
    panel(std::integral_constant<int, 0>{});
    panel(std::integral_constant<int, 1>{});
    panel(std::integral_constant<int, 2>{});
    panel(std::integral_constant<int, 3>{});

    if (!has_p && !(EIG_CHOL2_SKIP & 32)) {
        if (tid == 0) {      // (bounded spin, see chol_row_kernel)
            long spins = 0;
            while ((int)(__hip_atomic_load(loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1L << 22)) { atomicCAS(info, 0, -4096 - k0); break; }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b_ = 16 * q + 4 * r + fq;
            if (EIG_CHOL2_SKIP & 8) {
            } else if (!has_p) {
                if (q <= w && b_ <= a_ && a_ < nb)
                    Dblk[(size_t)b_ + (size_t)a_ * ldb] = conj_(Tr<T>::make(ar_[0][q][r], ai_[0][q][r]));
            } else {
                if (b_ < nb && a_ < pc) Pblk[(size_t)b_ + (size_t)a_ * ldb] = conj_(Tr<T>::make(ar_[1][q][r], ai_[1][q][r]));
            }
        }
}

// A_kk <- invU^H * Herm(A_kk) * invU for one diagonal block (upper triangle in/out, real diagonal).
template <class T>
__global__ void __launch_bounds__(256) hegs2_block_kernel(int nb, T* Ablk, int lda, const T* inv) {
    __shared__ T h[DB * DBL];
    __shared__ T x[DB * DBL];
    const int tid = threadIdx.x;
    for (int e = tid; e < DB * DB; e += 256) {
        int r = e % DB, cc = e / DB;
        T v = Tr<T>::zero();
        if (r < nb && cc < nb) {
            if (r < cc) v = Ablk[(size_t)r + (size_t)cc * lda];
            else if (r == cc) v = Tr<T>::realpart(Ablk[(size_t)r + (size_t)cc * lda]);
            else v = conj_(Ablk[(size_t)cc + (size_t)r * lda]);
        }
        h[r + cc * DBL] = v;
        x[r + cc * DBL] = inv[r + cc * DB];
    }
    __syncthreads();
    // tmp = H * X   (X upper: sum over k <= col)
    T t[DB * DB / 256];
#pragma unroll
    for (int q = 0; q < DB * DB / 256; ++q) {
        int e = tid + q * 256;
        int r = e % DB, cc = e / DB;
        T s = Tr<T>::zero();
        for (int k = 0; k <= cc; ++k) fma_(s, h[r + k * DBL], x[k + cc * DBL]);
        t[q] = s;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DB * DB / 256; ++q) {
        int e = tid + q * 256;
        h[(e % DB) + (e / DB) * DBL] = t[q];
    }
    __syncthreads();
    // out = X^H * tmp  (X^H lower: sum over k <= row); write upper part only
#pragma unroll
    for (int q = 0; q < DB * DB / 256; ++q) {
        int e = tid + q * 256;
        int r = e % DB, cc = e / DB;
        if (r < nb && cc < nb && r <= cc) {
            T s = Tr<T>::zero();
            for (int k = 0; k <= r; ++k) fmac_(s, x[k + r * DBL], h[k + cc * DBL]);
            if (r == cc) s = Tr<T>::realpart(s);
            Ablk[(size_t)r + (size_t)cc * lda] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Blocked (recursive) routines
// ------------------------------------------------------------------------------------------
static inline int split_n1(int n, int gran = DB) {
    int nblk = (n + gran - 1) / gran;
    return ((nblk + 1) / 2) * gran;
}

template <class T> static Operand<T> op_inv(const T* inv, int trans, int conj) {
    Operand<T> o;
    o.p = inv; o.ld = DB; o.trans = trans; o.conj = conj; o.mask = M_UPPER;
    return o;
}

// ---- merged 256x256 inverse diagonal blocks (solves outside potrf) --------------------------------------------
// The solves recurse down to inverted diagonal blocks.  With 64x64 blocks a triangular solve of order 2048 is 63
// launches, half of them 64-row products that fill a quarter of the chip for ~14 us each (30 % of hegst's time at
// C3).  After the factorization the 64-block inverses are merged pairwise, twice,
//     inv([[U0, M], [0, U1]]) = [[I0, -I0 M I1], [0, I1]],
// into inverses of the 256x256 diagonal blocks (two small launches for the whole matrix); the solves then stop
// at 256 (15 launches for order 2048).  Same flops: the inverse is used through its stored triangle only.
constexpr int BB = 256;

template <class T> __global__ void __launch_bounds__(256) place_inv64_kernel(int nblk64, const T* inv64, T* inv256, int g0) {
    const int b = g0 * 4 + blockIdx.x;        // 64-block slot, 4 per 256-group
    T* G = inv256 + (size_t)(b / 4) * BB * BB + (size_t)(b % 4) * DB * (1 + BB);
    for (int e = threadIdx.x; e < DB * DB; e += 256) {
        const int r = e % DB, cc = e / DB;
        T v = (r == cc) ? Tr<T>::one() : Tr<T>::zero();
        if (b < nblk64) v = inv64[(size_t)b * DB * DB + e];
        G[(size_t)r + (size_t)cc * BB] = v;
    }
}

// groups g0 .. g0+ng-1 (their 64-block inverses must be complete).  The merges P = -L M R are two strided-batch MFMA products
// per off-diagonal block position (X = M R, P = -L X over all groups at once): 6 small launches, ~50 us for N = 4096, where
// the round-1 one-lane-per-row kernel took 2 x 150 us.
template <class T> static void build_inv256_groups(Ctx& c, hipStream_t st, int N, const T* U, int ldu, int g0, int ng) {
    const int nblk64 = (N + DB - 1) / DB, ngall = (N + BB - 1) / BB;
    if (ng <= 0) return;
    const T* inv64 = c.scratch<T>("invU", 0);
    T* inv256 = c.scratch<T>("invU256", (size_t)ngall * BB * BB);
    kmemset(c, st, inv256 + (size_t)g0 * BB * BB, 0, sizeof(T) * (size_t)ng * BB * BB);
    klaunch(c, st, (place_inv64_kernel<T>), dim3(ng * 4), dim3(256), nblk64, inv64, inv256, g0);
    EIG_HIP(hipGetLastError());
    T* X = c.scratch<T>("inv_X", (size_t)ngall * 128 * 128);
    T* G0 = inv256 + (size_t)g0 * BB * BB;
    const size_t k00 = (size_t)g0 * BB;
    auto merge = [&](int s_, int r0) {
        const int c0 = r0 + s_;
        kmemset(c, st, X, 0, sizeof(T) * (size_t)ng * s_ * s_);     // rows clipped at the matrix end stay zero
        GemmBatch b1;                                                              // X = M R
        b1.count = ng; b1.sA = (long)BB * (ldu + 1); b1.sB = (long)BB * BB; b1.sC = (long)s_ * s_; b1.dcap = BB;
        b1.capM = N - (int)k00 - r0; b1.capK = N - (int)k00 - c0;
        Operand<T> M = opA('N', U + (k00 + r0) + (k00 + c0) * (size_t)ldu, ldu);
        Operand<T> R = op_plain((const T*)(G0 + (size_t)c0 * (1 + BB)), BB, 1, 0);
        R.mask = M_UPPER;
        gemm_batched<T>(c, st, s_, s_, s_, Tr<T>::one(), M, R, Tr<T>::zero(), X, s_, Epi(), b1);
        GemmBatch b2;                                                              // P = -L X
        b2.count = ng; b2.sA = (long)BB * BB; b2.sB = (long)s_ * s_; b2.sC = (long)BB * BB;
        Operand<T> L = op_plain((const T*)(G0 + (size_t)r0 * (1 + BB)), BB, 0, 0);
        L.mask = M_UPPER;
        gemm_batched<T>(c, st, s_, s_, s_, Tr<T>::make(-1.0, 0.0), L, opB('N', (const T*)X, s_), Tr<T>::zero(),
                        G0 + (size_t)r0 + (size_t)c0 * BB, BB, Epi(), b2);
    };
    merge(64, 0);
    merge(64, 128);
    merge(128, 0);
}
template <class T> void build_inv256(Ctx& c, hipStream_t st, int N, const T* U, int ldu) {
    build_inv256_groups(c, st, N, U, ldu, 0, (N + BB - 1) / BB);
}

template <class T> static Operand<T> op_inv256(Ctx& c, int k0, int trans, int conj) {
    Operand<T> o;
    o.p = c.scratch<T>("invU256", 0) + (size_t)(k0 / BB) * BB * BB; o.ld = BB; o.trans = trans; o.conj = conj; o.mask = M_UPPER;
    return o;
}

// ---- inverse diagonal blocks of order 512 / 1024 (option "trsm_base") ------------------------------------------------
// One more level (or two) of the same merge, inv([[U0, M], [0, U1]]) = [[I0, -I0 M I1], [0, I1]], this time as MFMA
// products (K-trimmed triangular operands): the solves of hegst and the final trsm then stop at 512/1024-blocks -- a
// quarter / a sixteenth of the dependent launches of the 256-block form.  tools/inverse_vs_substitution.py: residual and
// B-orthonormality on the reference recipe (cond(B) 1.5e10) are the same for substitution and for inverse blocks of any
// order up to the whole factor.
static inline int norm_base(int base) { return base >= 1024 ? 1024 : (base >= 512 ? 512 : (base >= 256 ? 256 : 64)); }
static const char* big_slot(int base) { return base == 1024 ? "invU1024" : "invU512"; }

template <class T> static Operand<T> op_invbig(Ctx& c, int base, int k0, int trans, int conj) {
    if (base == BB) return op_inv256<T>(c, k0, trans, conj);
    Operand<T> o;
    o.p = c.scratch<T>(big_slot(base), 0) + (size_t)(k0 / base) * base * base; o.ld = base; o.trans = trans; o.conj = conj;
    o.mask = M_UPPER;
    return o;
}

// diagonal sub-blocks of order `sub` (from the slot of that order, ld = sub) -> diagonal positions of the groups of order
// `big`; identity where the matrix has ended
template <class T>
__global__ void __launch_bounds__(256) place_inv_kernel(int N, int sub, int big, const T* src, T* dst) {
    const int per = big / sub, slot = blockIdx.y;                 // sub-block index over the whole matrix
    T* G = dst + (size_t)(slot / per) * big * big + (size_t)(slot % per) * sub * (1 + big);
    const bool have = slot * sub < N;
    const T* S = src + (size_t)slot * sub * sub;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)sub * sub; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e % sub), cc = (int)(e / sub);
        G[(size_t)r + (size_t)cc * big] = have ? S[e] : ((r == cc) ? Tr<T>::one() : Tr<T>::zero());
    }
}

// inv blocks of order `big` from those of order big/2 (which must exist: 256 from build_inv256, 512 from this routine)
template <class T> static void build_inv_level(Ctx& c, hipStream_t st, int N, const T* U, int ldu, int big) {
    const int sub = big / 2, ng = (N + big - 1) / big;
    const T* src = sub == BB ? c.scratch<T>("invU256", 0) : c.scratch<T>(big_slot(sub), 0);
    T* dst = c.scratch<T>(big_slot(big), (size_t)ng * big * big);
    kmemset(c, st, dst, 0, sizeof(T) * (size_t)ng * big * big);
    klaunch(c, st, (place_inv_kernel<T>), dim3(64, ng * 2), dim3(256), N, sub, big, src, dst);
    T* tmp = c.scratch<T>("inv_tmp", (size_t)sub * sub);
    for (int g = 0; g < ng; ++g) {
        const int k = g * big;
        const int n0 = min(sub, N - k), n1 = min(sub, N - (k + sub));
        if (n1 <= 0) continue;
        T* G = dst + (size_t)g * big * big;
        Operand<T> I0 = op_plain((const T*)G, big, 0, 0);                                   // A operand (rows x k)
        I0.mask = M_UPPER;
        Operand<T> I1 = op_plain((const T*)(G + (size_t)sub * (1 + big)), big, 1, 0);        // B operand 'N': Bt(j,k) = I1(k,j)
        I1.mask = M_UPPER;
        const T* M = U + (size_t)k + (size_t)(k + sub) * ldu;
        gemm<T>(c, st, n0, n1, n1, Tr<T>::one(), opA('N', M, ldu), I1, Tr<T>::zero(), tmp, sub);             // tmp = M I1
        gemm<T>(c, st, n0, n1, n0, Tr<T>::make(-1.0, 0.0), I0, opB('N', (const T*)tmp, sub), Tr<T>::zero(),
                G + (size_t)sub * big, big);                                                                   // P = -I0 tmp
    }
    EIG_HIP(hipGetLastError());
}

// everything the solves outside potrf need beyond the 64-block inverses, per the "trsm_base" option
template <class T> void build_inv_blocks(Ctx& c, hipStream_t st, int N, const T* U, int ldu) {
    const int base = norm_base(c.trsm_base);
    if (base >= BB) build_inv256<T>(c, st, N, U, ldu);
    if (base >= 512) build_inv_level<T>(c, st, N, U, ldu, 512);
    if (base >= 1024) build_inv_level<T>(c, st, N, U, ldu, 1024);
}
// ---- triangular solves, out of place --------------------------------------------------------------------------------------
// Y = op(U)^-1 X  /  Y = X U^-1 by recursive halving down to the inverted diagonal blocks.  The result goes to Y and X is the
// workspace (its not yet solved blocks receive the updates): base-case products read X and write Y, updates read Y and
// modify X.  The in-place form of rounds 1-3 had to stage every base-case product through scratch and copy it back (row
// blocks of the result are other workgroups' operands): 80 rectangular copies of ~6 us per C3 solve.
template <class T> static Operand<T> op_invbase(Ctx& c, int base, int k0, int trans, int conj) {
    if (base == DB) return op_inv(c.scratch<T>("invU", 0) + (size_t)(k0 / DB) * DB * DB, trans, conj);
    return op_invbig<T>(c, base, k0, trans, conj);
}

template <class T>
void trsm_LUN(Ctx& c, hipStream_t st, int n, int m, const T* U, int ldu, int k0, T* X, int ldx, T* Y, int ldy, int base,
              const std::function<void(int, int)>* rows_done, int hook_depth, int row0) {
    if (n <= 0 || m <= 0) return;
    base = norm_base(base);
    if (n <= base) {
        gemm<T>(c, st, n, m, n, Tr<T>::one(), op_invbase<T>(c, base, k0, 0, 0), opB('N', (const T*)X, ldx), Tr<T>::zero(), Y, ldy);
        if (rows_done && hook_depth >= 0) (*rows_done)(row0, n);
        return;
    }
    int n1 = split_n1(n, base), n2 = n - n1;
    // (hook: a block at depth hook_depth reports as a whole; below that depth nothing reports)
    const std::function<void(int, int)>* sub = (rows_done && hook_depth > 0) ? rows_done : nullptr;
    trsm_LUN(c, st, n2, m, U, ldu, k0 + n1, X + n1, ldx, Y + n1, ldy, base, sub, hook_depth - 1, row0 + n1);   // Y2 = U22^-1 X2
    gemm<T>(c, st, n1, m, n2, Tr<T>::make(-1.0, 0.0), opA('N', U + (size_t)k0 + (size_t)(k0 + n1) * ldu, ldu),
            opB('N', (const T*)(Y + n1), ldy), Tr<T>::one(), X, ldx);                          // X1 -= U12 Y2
    trsm_LUN(c, st, n1, m, U, ldu, k0, X, ldx, Y, ldy, base, sub, hook_depth - 1, row0);       // Y1 = U11^-1 X1
    if (rows_done && hook_depth == 0) (*rows_done)(row0, n);
}

template <class T>
void trsm_LUC(Ctx& c, hipStream_t st, int n, int m, const T* U, int ldu, int k0, T* X, int ldx, T* Y, int ldy, int base) {
    if (n <= 0 || m <= 0) return;
    base = norm_base(base);
    if (n <= base) {
        gemm<T>(c, st, n, m, n, Tr<T>::one(), op_invbase<T>(c, base, k0, 1, 1), opB('N', (const T*)X, ldx), Tr<T>::zero(), Y, ldy);
        return;
    }
    int n1 = split_n1(n, base), n2 = n - n1;
    trsm_LUC(c, st, n1, m, U, ldu, k0, X, ldx, Y, ldy, base);                                 // Y1 = U11^-H X1
    gemm<T>(c, st, n2, m, n1, Tr<T>::make(-1.0, 0.0), opA('C', U + (size_t)k0 + (size_t)(k0 + n1) * ldu, ldu),
            opB('N', (const T*)Y, ldy), Tr<T>::one(), X + n1, ldx);                            // X2 -= U12^H Y1
    trsm_LUC(c, st, n2, m, U, ldu, k0 + n1, X + n1, ldx, Y + n1, ldy, base);                  // Y2 = U22^-H X2
}

// X, Y are m x n (m rows): Y = X U^-1 with U = U(k0:k0+n, k0:k0+n)
template <class T>
void trsm_RUN(Ctx& c, hipStream_t st, int n, int m, const T* U, int ldu, int k0, T* X, int ldx, T* Y, int ldy, int base) {
    if (n <= 0 || m <= 0) return;
    base = norm_base(base);
    if (n <= base) {
        gemm<T>(c, st, m, n, n, Tr<T>::one(), opA('N', (const T*)X, ldx), op_invbase<T>(c, base, k0, 1, 0), Tr<T>::zero(), Y, ldy);
        return;
    }
    int n1 = split_n1(n, base), n2 = n - n1;
    trsm_RUN(c, st, n1, m, U, ldu, k0, X, ldx, Y, ldy, base);                                 // Y1 = X1 U11^-1
    gemm<T>(c, st, m, n2, n1, Tr<T>::make(-1.0, 0.0), opA('N', (const T*)Y, ldy),
            opB('N', U + (size_t)k0 + (size_t)(k0 + n1) * ldu, ldu), Tr<T>::one(), X + (size_t)n1 * ldx, ldx);   // X2 -= Y1 U12
    trsm_RUN(c, st, n2, m, U, ldu, k0 + n1, X + (size_t)n1 * ldx, ldx, Y + (size_t)n1 * ldy, ldy, base);          // Y2 = X2 U22^-1
}

// use256: block boundaries above 256 fall on multiples of 256, every finished 256-block gets its merged inverse at
// once (3 small launches) and the panel solves of the larger levels stop at it.
template <class T>
static void potrf_rec(Ctx& c, hipStream_t st, int Ntot, int n, int k0, T* B, int ldb, T* invU, bool use256 = false, bool block_root = false) {
    if (n <= 0) return;
    if (n <= DB) {
        hipLaunchKernelGGL((diag_block_kernel<T>), dim3(1), dim3(DGT), 0, st, Ntot, B, ldb, invU, 1, k0, c.d_info, 0);
        EIG_HIP(hipGetLastError());
        if (use256 && block_root) build_inv256_groups<T>(c, st, Ntot, (const T*)B, ldb, k0 / BB, 1);
        return;
    }
    const bool big = use256 && n > BB;
    int n1 = split_n1(n, big ? BB : DB), n2 = n - n1;
    potrf_rec(c, st, Ntot, n1, k0, B, ldb, invU, use256, big && n1 <= BB);
    T* B12 = B + (size_t)k0 + (size_t)(k0 + n1) * ldb;
    T* B22 = B + (size_t)(k0 + n1) + (size_t)(k0 + n1) * ldb;
    {   // B12 <- U11^-H B12 (the recursive form is the fallback without an intra-grid dependency, not the fast path: one staging copy)
        T* tmp = c.scratch<T>("potrf_tmp", (size_t)Ntot * Ntot / 4 + 64);
        trsm_LUC(c, st, n1, n2, B, ldb, k0, B12, ldb, tmp, n1, big ? BB : DB);
        EIG_HIP(hipMemcpy2DAsync(B12, sizeof(T) * ldb, tmp, sizeof(T) * n1, sizeof(T) * n1, n2, hipMemcpyDeviceToDevice, st));
    }
    Epi e; e.uplo = 1; e.herm_diag = 1;
    gemm<T>(c, st, n2, n2, n1, Tr<T>::make(-1.0, 0.0), opA('C', (const T*)B12, ldb), opB('N', (const T*)B12, ldb),
            Tr<T>::one(), B22, ldb, e);
    potrf_rec(c, st, Ntot, n2, k0 + n1, B, ldb, invU, use256, big && n2 <= BB);
    if (use256 && block_root) build_inv256_groups<T>(c, st, Ntot, (const T*)B, ldb, k0 / BB, 1);
}

template <class T> void build_invU(Ctx& c, hipStream_t st, int N, const T* U, int ldu);

// Right-looking blocked Cholesky with block rows of 64: per block row ONE chol_row_kernel launch (diagonal block factored,
// the whole block row solved, one workgroup per 64-column chunk) and ONE rank-64 MFMA update of the trailing triangle.
// 2 launches per block row, no inverse and no 64-wide product on the critical path (the recursive form -- potrf_rec,
// kept for the two-stream option -- spent its time in a chain of 64 one-workgroup factor+invert kernels and ~190 small
// dependent products: 10.9 ms at C3).  The inverted diagonal blocks the later solves use are formed afterwards, all blocks
// in parallel.  EIGSOLVE_POTRF=rec / option "potrf" = 0 restores the recursive form.
// The block-row kernel is latency-bound on its 64 dependent elimination steps (~0.45 us each: reciprocal square root, LDS
// broadcast, barrier): a 4x4-cyclic 256-thread layout, a scaled look-ahead broadcast and a blocked-by-16 elimination (rank-16
// updates, 16x fewer barriers per multiply-add) all measured 40-44 us per launch like this one (profiles/r02_experiments.txt;
// kernels in the git history of round 2).
// block rows kb0 .. kb1-1 of the right-looking factorization of nprob (<= 4) matrices of one order in lockstep: ONE block-row
// launch per block row for all of them (blockIdx.y = problem), one rank-64 update per problem.  `expect`: the cumulative
// announce count of chol_row_kernel; info0 / loaded0: nprob consecutive words each, zeroed by the caller.
// Look-ahead (option "overlap" bit 2, round 6; one problem that has the device to itself).  The rank-128 update behind a pair of
// block rows is split into the two block rows the NEXT pair factors (a 128-row strip, queued on the chain's stream) and the rest
// (queued on a third stream behind an event): the next pair's latency-bound block-row kernels (2 x 33 us with <= 64 workgroups
// resident) run beside the rest update of this pair instead of behind it.  Dependencies: the strip of pair k+1 is also touched by
// the rest update of pair k-1..0, so the chain waits for the previous rest update before it applies the strip update (evU); the
// rest update reads the pair's finished block rows (evC).  Every element still receives its updates in the order of the pairs and
// each update is the same K = 128 sum: bit-identical to the one-stream form.
struct LookAhead {
    hipStream_t sB = nullptr;
    hipEvent_t evC = nullptr, evU = nullptr;
    bool pending = false;        // a rest update is in flight on sB (evU recorded behind it)
};
constexpr int kLookAheadMinRest = 1024;    // order of the rest update below which the split does not pay
static void lookahead_join(LookAhead* la, hipStream_t st) {
    if (la && la->pending) { EIG_HIP(hipStreamWaitEvent(st, la->evU, 0)); la->pending = false; }
}

template <class T>
static void potrf_block_rows(Ctx& c, hipStream_t st, int N, int nprob, T* const* B, int ldb, int kb0, int kb1, unsigned& expect,
                             int* info0, unsigned* loaded0, LookAhead* la = nullptr) {
    CholBatch<T> cb;
    for (int q = 0; q < CHOL_MAXB; ++q) cb.B[q] = B[q < nprob ? q : 0];
    if (c.potrf_mode >= 2) {
        // Round 5: block rows in PAIRS.  Per pair: block row k (chol_row2_kernel), the rank-64 update of block row k+1 alone (a
        // 64 x rem product: <= 252 small tiles), block row k+1, then ONE rank-128 update of everything below the pair -- half the
        // passes over the trailing matrix at twice the K (the engine does 47 instead of 34 TFLOP/s there).
        Epi e; e.uplo = 1; e.herm_diag = 1;
        for (int kb = kb0; kb < kb1; kb += 2) {
            const int k0 = kb * DB;
            const int nb = min(DB, N - k0), rem = N - k0 - nb;
            int chunks = (rem + DB - 1) / DB;
            expect += (unsigned)chunks;
            hipLaunchKernelGGL((chol_row2_kernel<T>), dim3(1 + chunks, nprob), dim3(256), 0, st, N, cb, ldb, k0, info0, loaded0, expect);
            if (rem <= 0) break;
            const int k1 = k0 + nb;
            if (kb + 1 >= kb1) {       // (odd number of block rows in this range: plain rank-64 update)
                lookahead_join(la, st);
                for (int q = 0; q < nprob; ++q) {
                    const T* B12 = B[q] + (size_t)k0 + (size_t)k1 * ldb;
                    gemm<T>(c, st, rem, rem, nb, Tr<T>::make(-1.0, 0.0), opA('C', B12, ldb), opB('N', B12, ldb), Tr<T>::one(),
                            B[q] + (size_t)k1 + (size_t)k1 * ldb, ldb, e);
                }
                break;
            }
            const int nb1 = min(DB, rem), rem1 = rem - nb1;
            for (int q = 0; q < nprob; ++q) {
                const T* B12 = B[q] + (size_t)k0 + (size_t)k1 * ldb;
                gemm<T>(c, st, nb1, rem, nb, Tr<T>::make(-1.0, 0.0), opA('C', B12, ldb), opB('N', B12, ldb), Tr<T>::one(),
                        B[q] + (size_t)k1 + (size_t)k1 * ldb, ldb, e);
            }
            chunks = (rem1 + DB - 1) / DB;
            expect += (unsigned)chunks;
            hipLaunchKernelGGL((chol_row2_kernel<T>), dim3(1 + chunks, nprob), dim3(256), 0, st, N, cb, ldb, k1, info0, loaded0, expect);
            if (rem1 > 0) {
                const int k2 = k1 + nb1;
                const int nbn = min(2 * DB, rem1);      // the block rows of the next pair
                if (la && nprob == 1 && rem1 - nbn >= kLookAheadMinRest) {
                    const T* B13 = B[0] + (size_t)k0 + (size_t)k2 * ldb;
                    EIG_HIP(hipEventRecord(la->evC, st));                                  // block rows k0, k1 are final
                    lookahead_join(la, st);                                                // (the previous rest update reached into the strip)
                    gemm<T>(c, st, nbn, rem1, nb + nb1, Tr<T>::make(-1.0, 0.0), opA('C', B13, ldb), opB('N', B13, ldb), Tr<T>::one(),
                            B[0] + (size_t)k2 + (size_t)k2 * ldb, ldb, e);                 // the strip: rows k2 .. k2 + nbn - 1
                    EIG_HIP(hipStreamWaitEvent(la->sB, la->evC, 0));
                    const T* B13r = B13 + (size_t)nbn * ldb;
                    gemm<T>(c, la->sB, rem1 - nbn, rem1 - nbn, nb + nb1, Tr<T>::make(-1.0, 0.0), opA('C', B13r, ldb), opB('N', B13r, ldb),
                            Tr<T>::one(), B[0] + (size_t)(k2 + nbn) + (size_t)(k2 + nbn) * ldb, ldb, e);   // everything below the strip
                    EIG_HIP(hipEventRecord(la->evU, la->sB));
                    la->pending = true;
                } else {
                    lookahead_join(la, st);
                    for (int q = 0; q < nprob; ++q) {
                        const T* B13 = B[q] + (size_t)k0 + (size_t)k2 * ldb;
                        gemm<T>(c, st, rem1, rem1, nb + nb1, Tr<T>::make(-1.0, 0.0), opA('C', B13, ldb), opB('N', B13, ldb), Tr<T>::one(),
                                B[q] + (size_t)k2 + (size_t)k2 * ldb, ldb, e);
                    }
                }
            }
        }
        EIG_HIP(hipGetLastError());
        return;
    }
    for (int kb = kb0; kb < kb1; ++kb) {
        const int k0 = kb * DB;
        const int nb = min(DB, N - k0), rem = N - k0 - nb;
        const int chunks = (rem + DB - 1) / DB;
        expect += (unsigned)chunks;
        hipLaunchKernelGGL((chol_row_kernel<T>), dim3(1 + chunks, nprob), dim3(DGT), 0, st, N, cb, ldb, k0, info0, loaded0, expect);
        if (rem > 0) {
            for (int q = 0; q < nprob; ++q) {
                const T* B12 = B[q] + (size_t)k0 + (size_t)(k0 + nb) * ldb;
                Epi e; e.uplo = 1; e.herm_diag = 1;
                gemm<T>(c, st, rem, rem, nb, Tr<T>::make(-1.0, 0.0), opA('C', B12, ldb), opB('N', B12, ldb), Tr<T>::one(),
                        B[q] + (size_t)(k0 + nb) + (size_t)(k0 + nb) * ldb, ldb, e);
            }
        }
    }
    EIG_HIP(hipGetLastError());
}
template <class T> static void potrf_block_rows(Ctx& c, hipStream_t st, int N, T* B, int ldb, int kb0, int kb1, unsigned& expect,
                                                LookAhead* la = nullptr) {
    T* one[1] = {B};
    potrf_block_rows<T>(c, st, N, 1, one, ldb, kb0, kb1, expect, c.d_info, reinterpret_cast<unsigned*>(c.d_info) + 2, la);
}
// the look-ahead state of a factorization on stream st, or "none" (la.sB stays null): one problem, alone on the device, block-row
// pairs, large enough for at least one split update
static bool lookahead_begin(Ctx& c, int N, LookAhead& la) {
    if (!(c.overlap & 4) || c.in_batch || c.rec || c.potrf_mode < 2 || N < kLookAheadMinRest + 4 * DB) return false;
    if (streams_in_use(c.dev) > c.own_streams()) return false;
    for (auto& e : c.evLA)
        if (!e) EIG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    la.sB = c.third_stream();
    la.evC = c.evLA[0]; la.evU = c.evLA[1];
    la.pending = false;
    return true;
}

// The factorizations of a lockstep group (batch calls): block rows only -- the inverse diagonal blocks live in ONE set of scratch
// slots per context and are built per problem by the caller right before they are used.  info of problem q: c.d_info[4 + q].
template <class T> void potrf_upper_group(Ctx& c, hipStream_t st, int N, int nprob, T* const* B, int ldb) {
    EIG_HIP(hipMemsetAsync(c.d_info, 0, 16 * sizeof(int), st));
    unsigned expect = 0;
    potrf_block_rows<T>(c, st, N, nprob, B, ldb, 0, (N + DB - 1) / DB, expect, c.d_info + 4, reinterpret_cast<unsigned*>(c.d_info) + 8);
}

// inverses of the 64x64 diagonal blocks blk0 .. blk0+nb-1 of a finished factor
template <class T> static void build_invU_range(Ctx& c, hipStream_t st, int N, const T* U, int ldu, int blk0, int nb) {
    const int nblk = (N + DB - 1) / DB;
    if (nb <= 0) return;
    T* invU = c.scratch<T>("invU", (size_t)nblk * DB * DB);
    klaunch(c, st, (diag_block_kernel<T>), dim3(nb), dim3(DGT), N, const_cast<T*>(U), ldu, invU, 0, -1, c.d_info, blk0);
    EIG_HIP(hipGetLastError());
}

template <class T> void potrf_upper(Ctx& c, hipStream_t st, int N, T* B, int ldb) {
    int nblk = (N + DB - 1) / DB;
    T* invU = c.scratch<T>("invU", (size_t)(nblk > 0 ? nblk : 1) * DB * DB);
    EIG_HIP(hipMemsetAsync(c.d_info, 0, 4 * sizeof(int), st));
    if (c.potrf_mode == 0) {
        potrf_rec(c, st, N, N, 0, B, ldb, invU);
    } else {
        unsigned expect = 0;
        LookAhead la;
        const bool use_la = lookahead_begin(c, N, la);
        potrf_block_rows<T>(c, st, N, B, ldb, 0, nblk, expect, use_la ? &la : nullptr);
        lookahead_join(use_la ? &la : nullptr, st);
        build_invU<T>(c, st, N, (const T*)B, ldb);
    }
    build_inv_blocks<T>(c, st, N, (const T*)B, ldb);
}

template <class T> void build_invU(Ctx& c, hipStream_t st, int N, const T* U, int ldu) {
    build_invU_range<T>(c, st, N, U, ldu, 0, (N + DB - 1) / DB);
}

// ---- reduction to standard form --------------------------------------------------------------------------------------------
// hegst on a stream of its own, released stage by stage as block rows of the factor complete on the factorization's stream
// (potrf || hegst pipeline below): need(r) = "the next operation reads rows < r of U (and their inverse diagonal blocks)".
constexpr int kStageRows = 1024;
struct UGate {
    hipStream_t st;
    hipEvent_t* ev;
    int have;
    void need(int rows) {
        const int s_ = (rows + kStageRows - 1) / kStageRows;
        while (have < s_) EIG_HIP(hipStreamWaitEvent(st, ev[have++], 0));
    }
};

template <class T> __global__ void __launch_bounds__(256) herm_complete_kernel(int n, const T* A, int lda, T* F, int ldf) {
    __shared__ T tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;   // tile (bx = row block, by = col block), only bx <= by launched work
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int q = ty; q < 32; q += 8) {
        int r = bx * 32 + tx, cc = by * 32 + q;
        T v = Tr<T>::zero();
        if (r < n && cc < n && r <= cc) v = A[(size_t)r + (size_t)cc * lda];
        if (r == cc) v = Tr<T>::realpart(v);
        tile[q][tx] = v;
        if (r < n && cc < n && r <= cc) F[(size_t)r + (size_t)cc * ldf] = v;
    }
    __syncthreads();
    // mirrored block: F(c, r) = conj(A(r, c)) for r < c
    for (int q = ty; q < 32; q += 8) {
        int cc = by * 32 + tx, r = bx * 32 + q;     // write F(cc, r): consecutive tx -> consecutive rows cc
        if (r < n && cc < n && r < cc) F[(size_t)cc + (size_t)r * ldf] = conj_(tile[tx][q]);
    }
}
template <class T> __global__ void __launch_bounds__(256) copy_upper_kernel(int n, const T* F, int ldf, T* A, int lda) {
    size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)n * n) return;
    int r = (int)(id % n), cc = (int)(id / n);
    if (r > cc) return;
    T v = F[(size_t)r + (size_t)cc * ldf];
    if (r == cc) v = Tr<T>::realpart(v);
    A[(size_t)r + (size_t)cc * lda] = v;
}
// Y(rows x cols, ldy) += X(rows x cols, ldx)
template <class T> __global__ void __launch_bounds__(256) add_block_kernel(int rows, int cols, const T* X, int ldx, T* Y, int ldy) {
    size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)rows * cols) return;
    const int r = (int)(id % rows), cc = (int)(id / rows);
    Y[(size_t)r + (size_t)cc * ldy] = Y[(size_t)r + (size_t)cc * ldy] + X[(size_t)r + (size_t)cc * ldx];
}

// ---- hegst as two full triangular solves ------------------------------------------------------------
// F = Herm(A) (completed copy), G = U^-H F, F = G U^-1, upper(A) <- upper(F).  2x the flops of the
// symmetric algorithm (zhegst_gpu.F90:51-107) but ~4N/64 large launches instead of ~16N/64 small ones:
// on MI355X the small-launch chain, not the flops, dominates the symmetric form.
template <class T> static void hegst_two_solves_at(Ctx& c, hipStream_t st, int N, int k0, T* A, int lda, const T* U, int ldu) {
    if (N <= 0) return;
    T* Ablk = A + (size_t)k0 + (size_t)k0 * lda;
    T* F = c.scratch<T>(Tr<T>::cx ? "gst_Fz" : "gst_Fd", (size_t)N * N);
    T* G = c.scratch<T>(Tr<T>::cx ? "gst_Gz" : "gst_Gd", (size_t)N * N);
    const int nb32 = (N + 31) / 32;
    klaunch(c, st, (herm_complete_kernel<T>), dim3(nb32, nb32), dim3(256), N, (const T*)Ablk, lda, F, N);
    trsm_LUC(c, st, N, N, U, ldu, k0, F, N, G, N, c.trsm_base);   // G = U(k0.., k0..)^-H F
    trsm_RUN(c, st, N, N, U, ldu, k0, G, N, F, N, c.trsm_base);   // F = G U(k0.., k0..)^-1
    size_t tot = (size_t)N * N;
    klaunch(c, st, (copy_upper_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), N, (const T*)F, N, Ablk, lda);
    EIG_HIP(hipGetLastError());
}
template <class T> void hegst_two_solves(Ctx& c, hipStream_t st, int N, T* A, int lda, const T* U, int ldu) {
    hegst_two_solves_at(c, st, N, 0, A, lda, U, ldu);
}

// The block step of the symmetric algorithm (zhegst_gpu.F90:85-104) for the leading block [k0, k0+n1) against the n2
// columns to its right, A11 already reduced:
//     A12 <- U11^-H A12                      (:87-88)
//     A12 -= 1/2 Herm(A11) U12               (:93-94)
//     A22 -= A12^H U12 + U12^H A12 (upper)   (:95-96)
//     A12 -= 1/2 Herm(A11) U12               (:100-101)
//     A12 <- A12 U22^-1                      (:103-104)
// The reference (like LAPACK's zhegst) forms the product Herm(A11) U12 twice to save workspace; here it is formed ONCE --
// the first application also stores -1/2 Herm(A11) U12 (Epi::aux), the second is an addition -- which removes a sixth of the
// multiply-adds of hegst.  The two solves are out of place (A12 -> T -> A12), so nothing is staged or copied.
template <class T>
static void hegst_block_step(Ctx& c, hipStream_t st, int n1, int n2, int k0, T* A, int lda, const T* U, int ldu, int base, UGate* gate) {
    T* A11 = A + (size_t)k0 + (size_t)k0 * lda;
    T* A12 = A + (size_t)k0 + (size_t)(k0 + n1) * lda;
    T* A22 = A + (size_t)(k0 + n1) + (size_t)(k0 + n1) * lda;
    const T* U12 = U + (size_t)k0 + (size_t)(k0 + n1) * ldu;
    T* Tm = c.scratch<T>(Tr<T>::cx ? "gst_Tz" : "gst_Td", (size_t)n1 * n2);
    T* Xh = c.scratch<T>(Tr<T>::cx ? "gst_Xz" : "gst_Xd", (size_t)n1 * n2);
    T* H = c.scratch<T>(Tr<T>::cx ? "gst_Hz" : "gst_Hd", (size_t)n1 * n1);
    if (gate) gate->need(k0 + n1);                                     // the next four steps read U(k0 : k0+n1, :) only
    trsm_LUC(c, st, n1, n2, U, ldu, k0, A12, lda, Tm, n1, base);      // T = U11^-H A12
    {
        const int nb32 = (n1 + 31) / 32;                               // Herm(A11) completed once: the product is a plain full-rate gemm
        klaunch(c, st, (herm_complete_kernel<T>), dim3(nb32, nb32), dim3(256), n1, (const T*)A11, lda, H, n1);
    }
    {
        Epi e; e.aux = Xh; e.ldaux = n1;                               // T -= 1/2 Herm(A11) U12,  Xh = -1/2 Herm(A11) U12
        gemm<T>(c, st, n1, n2, n1, Tr<T>::make(-0.5, 0.0), opA('N', (const T*)H, n1), opB('N', U12, ldu), Tr<T>::one(), Tm, n1, e);
    }
    {
        Operand<T> Ao, Bo;                                             // A22 -= T^H U12 + U12^H T (upper), one pass over A22
        Ao.p = Tm; Ao.ld = n1; Ao.trans = 1; Ao.conj = 1; Ao.k1 = n1; Ao.p2 = U12; Ao.ld2 = ldu;
        Bo.p = U12; Bo.ld = ldu; Bo.trans = 1; Bo.conj = 0; Bo.k1 = n1; Bo.p2 = Tm; Bo.ld2 = n1;
        Epi e; e.uplo = 1; e.herm_diag = 1;
        gemm<T>(c, st, n2, n2, 2 * n1, Tr<T>::make(-1.0, 0.0), Ao, Bo, Tr<T>::one(), A22, lda, e);
    }
    {
        const size_t tot = (size_t)n1 * n2;                            // T += Xh
        klaunch(c, st, (add_block_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), n1, n2, (const T*)Xh, n1, Tm, n1);
    }
    if (gate) gate->need(k0 + n1 + n2);                                // U22 from here on
    trsm_RUN(c, st, n2, n1, U, ldu, k0 + n1, Tm, n1, A12, lda, base); // A12 = T U22^-1
    EIG_HIP(hipGetLastError());
}

// symmetric recursion down to the 64x64 blocks (gst = 0, and every order below 256)
template <class T> static void hegst_rec(Ctx& c, hipStream_t st, int n, int k0, T* A, int lda, const T* U, int ldu) {
    if (n <= 0) return;
    const T* invU = c.scratch<T>("invU", 0);
    if (n <= DB) {
        klaunch(c, st, (hegs2_block_kernel<T>), dim3(1), dim3(256), n, A + (size_t)k0 + (size_t)k0 * lda, lda,
                invU + (size_t)(k0 / DB) * DB * DB);
        EIG_HIP(hipGetLastError());
        return;
    }
    int n1 = split_n1(n), n2 = n - n1;
    hegst_rec(c, st, n1, k0, A, lda, U, ldu);
    hegst_block_step<T>(c, st, n1, n2, k0, A, lda, U, ldu, DB, nullptr);
    hegst_rec(c, st, n2, k0 + n1, A, lda, U, ldu);
}

template <class T>
static void hegst_hybrid(Ctx& c, hipStream_t st, int n, int k0, T* A, int lda, const T* U, int ldu, int thr, UGate* gate = nullptr);
template <class T> static void hegst_blocked(Ctx& c, hipStream_t st, int N, T* A, int lda, const T* U, int ldu);


// Largest sizes the scratch slots of hegst take inside the hybrid recursion, allocated BEFORE anything of hegst is queued: a slot
// that grows inside the recursion synchronises the context's streams (Ctx::scratch_bytes) -- the potrf || hegst pipeline then
// loses its overlap on the first solve of a size.  Block steps: n1 = split_n1(n, gran) rounds UP, so T / Xh are n1 x (n - n1) and H
// is n1 x n1 with n1 possibly above n / 2 (N = 1300, gran 256: n1 = 768); two-solve leaves: F, G of the largest leaf order.
template <class T> static void hegst_pregrow(Ctx& c, int N) {
    if (N <= 1) return;
    const int gran = norm_base(c.trsm_base), thr = c.gst_thr;
    size_t qT = 64, qH = 64, qF = 64;
    if (c.gst_mode == 2 && N > thr && N >= 256) {
        std::function<void(int)> walk = [&](int n) {
            if (n <= thr || n <= gran) { qF = std::max(qF, (size_t)n * n); return; }
            const int n1 = split_n1(n, gran), n2 = n - n1;
            qT = std::max(qT, (size_t)n1 * n2);
            qH = std::max(qH, (size_t)n1 * n1);
            walk(n1);
            walk(n2);
        };
        walk(N);
    } else if (c.gst_mode == 3 && N >= 256) {
        const int nb = gran < 256 ? 256 : gran;
        qT = (size_t)nb * N; qH = (size_t)nb * nb; qF = (size_t)nb * nb;
    } else if (c.gst_mode == 0 || N < 256) {
        // (orders up to 64 are one block: no block step.  Every level of hegst_rec is walked: N = 129 splits 128 + 1 at the top
        //  -- T is 128 x 1 -- and 64 + 64 one level down, where T is 64 x 64)
        std::function<void(int)> walk = [&](int n) {
            if (n <= DB) return;
            const int n1 = split_n1(n), n2 = n - n1;
            qT = std::max(qT, (size_t)n1 * n2 + 64);
            qH = std::max(qH, (size_t)n1 * n1);
            walk(n1);
            walk(n2);
        };
        walk(N);
    } else {
        qF = (size_t)N * N;
    }
    (void)c.scratch<T>(Tr<T>::cx ? "gst_Tz" : "gst_Td", qT);
    (void)c.scratch<T>(Tr<T>::cx ? "gst_Xz" : "gst_Xd", qT);
    (void)c.scratch<T>(Tr<T>::cx ? "gst_Hz" : "gst_Hd", qH);
    if (qF > 64) {
        (void)c.scratch<T>(Tr<T>::cx ? "gst_Fz" : "gst_Fd", qF);
        (void)c.scratch<T>(Tr<T>::cx ? "gst_Gz" : "gst_Gd", qF);
    }
}

template <class T> void hegst_upper(Ctx& c, hipStream_t st, int N, T* A, int lda, const T* U, int ldu) {
    // option "gst": 0 = symmetric recursion down to 64x64 blocks (~16N/64 small launches),
    //               1 = two full triangular solves (N^3 multiply-adds, ~4N/64 large launches),
    //               2 = hybrid (default): the symmetric algorithm (zhegst_gpu.F90:51-107) on the large levels, where
    //                   every operation is a chip-filling MFMA launch, two solves on diagonal blocks of order
    //                   <= "gst_thr" (1024),
    //               3 = the reference's loop with nb = the order of the inverse diagonal blocks.
    const int mode = c.gst_mode, thr = c.gst_thr;
    hegst_pregrow<T>(c, N);   // (the recursion asks for the small blocks first)
    if (N < 256 || mode == 0) hegst_rec(c, st, N, 0, A, lda, U, ldu);
    else if (mode == 3) hegst_blocked(c, st, N, A, lda, U, ldu);
    else if (mode == 1 || N <= thr) hegst_two_solves(c, st, N, A, lda, U, ldu);
    else hegst_hybrid(c, st, N, 0, A, lda, U, ldu, thr);
}

// One level of the symmetric algorithm (zhegst_gpu.F90:51-107 with the block size = half the matrix): every
// operation is a large MFMA launch.  Diagonal blocks of order <= thr fall back to the two-solve form.
template <class T>
static void hegst_hybrid(Ctx& c, hipStream_t st, int n, int k0, T* A, int lda, const T* U, int ldu, int thr, UGate* gate) {
    if (n <= 0) return;
    const int gran = norm_base(c.trsm_base);
    if (n <= thr || n <= gran) {
        if (gate) gate->need(k0 + n);
        hegst_two_solves_at(c, st, n, k0, A, lda, U, ldu);
        return;
    }
    int n1 = split_n1(n, gran), n2 = n - n1;   // block boundaries must match the inverse diagonal blocks
    hegst_hybrid(c, st, n1, k0, A, lda, U, ldu, thr, gate);
    hegst_block_step<T>(c, st, n1, n2, k0, A, lda, U, ldu, c.trsm_base, gate);
    hegst_hybrid(c, st, n2, k0 + n1, A, lda, U, ldu, thr, gate);
}

// The reference's own loop (zhegst_gpu.F90:51-107) with block size nb = the order of the inverse diagonal blocks
// ("trsm_base" 512 / 1024): per block step the diagonal block by two products with its inverse, then the block step above on
// the block row -- (1/2 + O(nb/N)) N^3 multiply-adds instead of the ~2/3 N^3 of the half-split recursion.
template <class T> static void hegst_blocked(Ctx& c, hipStream_t st, int N, T* A, int lda, const T* U, int ldu) {
    const int base = norm_base(c.trsm_base), nb = base < 256 ? 256 : base;
    for (int k0 = 0; k0 < N; k0 += nb) {
        const int kb = min(nb, N - k0), rest = N - k0 - kb;
        hegst_two_solves_at(c, st, kb, k0, A, lda, U, ldu);                          // :57-83
        if (rest <= 0) break;
        hegst_block_step<T>(c, st, kb, rest, k0, A, lda, U, ldu, base, nullptr);      // :87-104
    }
    EIG_HIP(hipGetLastError());
}

// ---- potrf || hegst pipeline (option "overlap" bit 0) ----------------------------------------------------------------
// The block-row Cholesky is a chain of launches that leave most of the chip idle (64 x 42 us block-row kernels of <= 64
// workgroups; short rank-64 updates in its second half), and hegst needs the factor only progressively: with the half-split
// recursion every step reads U(0 : r, :) for a bound r that grows as the recursion moves down the diagonal (hegst_hybrid's
// gate->need(r) calls).  So: the factorization on the call's stream, in stages of 1024 rows (block rows + the inverse diagonal
// blocks of the stage, then an event); hegst as a whole on the second stream, each step waiting for the stage event it needs.
// Same kernels, same operands, same order of operations on every block as the one-stream path: results are bit-identical.
// Only for a solve that has the device to itself (the caller checks): with several solves in flight the chip is full anyway and
// every additional active queue costs the others (profiles/r03_experiments.txt 5, 10, 14).
template <class T> bool pipeline_applicable(const Ctx& c, int N) {
    return c.gst_mode == 2 && c.potrf_mode != 0 && norm_base(c.trsm_base) == BB && N > c.gst_thr && N >= 2 * BB &&
           (N + kStageRows - 1) / kStageRows <= 16;
}

// Enqueues the whole factorization on c.s1 and the whole of hegst, gated, on the second stream.  The caller synchronises c.s1
// (potrf's info -- every stage event is recorded whether or not the factorization fails), then calls hegst_pipelined_finish,
// or, on failure, synchronises the second stream.
template <class T> void potrf_hegst_pipelined_begin(Ctx& c, int N, T* A, int lda, T* B, int ldb) {
    hipStream_t s1 = c.s1;
    const int nblk = (N + DB - 1) / DB, ngall = (N + BB - 1) / BB;
    const int nstage = (N + kStageRows - 1) / kStageRows;
    (void)c.scratch<T>("invU", (size_t)nblk * DB * DB);
    for (int s_ = 0; s_ < nstage; ++s_)
        if (!c.evStage[s_]) EIG_HIP(hipEventCreateWithFlags(&c.evStage[s_], hipEventDisableTiming));
    EIG_HIP(hipMemsetAsync(c.d_info, 0, 4 * sizeof(int), s1));
    unsigned expect = 0;
    constexpr int SB = kStageRows / DB, SG = kStageRows / BB;
    LookAhead la;
    const bool use_la = lookahead_begin(c, N, la);     // (the rest updates only touch rows below the stage being factored)
    for (int s_ = 0; s_ < nstage; ++s_) {
        const int kb0 = s_ * SB, kb1 = min(nblk, kb0 + SB), g0 = s_ * SG, g1 = min(ngall, g0 + SG);
        potrf_block_rows<T>(c, s1, N, B, ldb, kb0, kb1, expect, use_la ? &la : nullptr);
        build_invU_range<T>(c, s1, N, (const T*)B, ldb, kb0, kb1 - kb0);
        build_inv256_groups<T>(c, s1, N, (const T*)B, ldb, g0, g1 - g0);
        EIG_HIP(hipEventRecord(c.evStage[s_], s1));   // rows < (s_+1) * 1024 of U and their inverse diagonal blocks are final
    }
    lookahead_join(use_la ? &la : nullptr, s1);
    // (everything above is queued before the first wait below is: a wait on an event that has not been recorded yet is a no-op)
    hipStream_t s2 = c.second_stream();
    hegst_pregrow<T>(c, N);
    UGate gate{s2, c.evStage, 0};
    hegst_hybrid<T>(c, s2, N, 0, A, lda, (const T*)B, ldb, c.gst_thr, &gate);
    EIG_HIP(hipEventRecord(c.evB, s2));
}

template <class T> void hegst_pipelined_finish(Ctx& c, int N, T* A, int lda, const T* U, int ldu) {
    EIG_HIP(hipStreamWaitEvent(c.s1, c.evB, 0));
}

#ifdef EIG_TOOLS   // experiment code: only in the tools-side build (make tools), never in libeigsolve_gpu.so
// ---- two-stage reduction, stage 1 (full -> band of width b = 64): LAUNCH SKELETON for the go / no-go measurement ----------------
// VERDICT r4 item 3 asks for a stage-1 spike with a kill criterion (C3 > 15 ms or C4 > 100 ms => stop).  Before building the
// numerics this routine issues the complete launch sequence stage 1 would consist of -- every product with its true shape, operand
// masks and K on the MFMA engine, every 64 x 64 serial step by the library's fastest 64 x 64 one-workgroup kernel -- on whatever
// data the buffers hold (no result is meaningful), so that the phase can be TIMED: a real implementation cannot be faster than its
// own launch sequence.  Per panel k (columns k0..k0+b-1, m = N - k0 - b rows below the band):
//   panel (CholeskyQR2 + Householder reconstruction):  G = P^H P (split-K) . chol(G), inverse . Q1 = P R1^-1 . the same again
//     (second pass) . R = R2 R1 . LU-with-signs of the top block / T factor (stand-in: the 64 x 64 factor-and-invert kernel)
//     . V_bottom = Q_bottom U^-1
//   trailing matrix:  W = A22 V as two triangular-masked products (lower part, strict lower part^H) . W = W T . S = V^H W (split-K)
//     . X = W - 1/2 V (T^H S) . A22 -= X V^H + V X^H on the lower triangle (one K-concatenated rank-2b update)
// what: 0 = everything, 1 = the panel chains only, 2 = the trailing-matrix products only.
template <class T> __global__ void __launch_bounds__(256) fill_pseudo_random_kernel(size_t count, T* x) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= count) return;
    unsigned long long z = (id + 0x9E3779B97F4A7C15ULL) * 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 31; z *= 0x94D049BB133111EBULL; z ^= z >> 29;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5, v = (double)((z * 0x2545F4914F6CDD1DULL) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    x[id] = Tr<T>::make(u, v);
}
template <class T> void two_stage_stage1_skeleton(Ctx& c, hipStream_t st, int N, int what) {
    constexpr int b = 64;
    T* F = c.scratch<T>("ts_F", (size_t)N * N);
    T* V = c.scratch<T>("ts_V", (size_t)N * b);
    T* W = c.scratch<T>("ts_W", (size_t)N * b);
    T* X = c.scratch<T>("ts_X", (size_t)N * b);
    T* Q = c.scratch<T>("ts_Q", (size_t)N * b);
    T* sm = c.scratch<T>("ts_small", (size_t)16 * b * b);      // G, inverses, R, S, ...
    T* G = sm, *Ginv = sm + b * b, *R2 = sm + 2 * b * b, *S = sm + 3 * b * b, *S2 = sm + 4 * b * b;
    // (random data: the clocks this part sustains depend on the operands; the one product that feeds back into F is scaled to nothing
    //  so that the meaningless values stay finite over the 63 panels)
    const T one = Tr<T>::one(), zero = Tr<T>::zero(), mone = Tr<T>::make(-1e-30, 0.0), mhalf = Tr<T>::make(-0.5, 0.0);
    hipLaunchKernelGGL((fill_pseudo_random_kernel<T>), dim3((unsigned)(((size_t)N * N + 255) / 256)), dim3(256), 0, st, (size_t)N * N, F);
    hipLaunchKernelGGL((fill_pseudo_random_kernel<T>), dim3((unsigned)(((size_t)N * b + 255) / 256)), dim3(256), 0, st, (size_t)N * b, V);
    for (int k0 = 0; k0 + b < N; k0 += b) {
        const int k1 = k0 + b, m = N - k1;
        T* P = F + (size_t)k1 + (size_t)k0 * N;
        T* F22 = F + (size_t)k1 + (size_t)k1 * N;
        const int kch = std::max(256, ((m + 7) / 8 + 63) & ~63);
        if (what != 2) {
            for (int pass = 0; pass < 2; ++pass) {
                const T* src = pass == 0 ? P : Q;
                const int lds = pass == 0 ? N : m;
                gemm_splitk<T>(c, st, b, b, m, one, opA('C', src, lds), opB('N', src, lds), zero, G, b, kch);              // Gram
                hipLaunchKernelGGL((diag_block_kernel<T>), dim3(1), dim3(DGT), 0, st, b, G, b, Ginv, 1, 0, c.d_info + 3, 0);   // chol + inverse
                Operand<T> Ri = op_plain((const T*)Ginv, b, 1, 0);
                Ri.mask = M_UPPER;
                gemm<T>(c, st, m, b, b, one, opA('N', src, lds), Ri, zero, pass == 0 ? Q : V, m);                         // Q = P R^-1
            }
            gemm<T>(c, st, b, b, b, one, opA('N', (const T*)G, b), opB('N', (const T*)R2, b), zero, S, b);                 // R = R2 R1
            hipLaunchKernelGGL((diag_block_kernel<T>), dim3(1), dim3(DGT), 0, st, b, G, b, Ginv, 1, 0, c.d_info + 3, 0);       // (LU with signs, T)
            if (m > b) {
                Operand<T> Ui = op_plain((const T*)Ginv, b, 1, 0);
                Ui.mask = M_UPPER;
                gemm<T>(c, st, m - b, b, b, one, opA('N', (const T*)(V + b), m), Ui, zero, Q + b, m);                     // V_bottom = Q_bottom U^-1
            }
        }
        if (what != 1) {
            Operand<T> L = op_plain((const T*)F22, N, 0, 0);
            L.mask = M_LOWER;
            // (m x 64 outputs are 64-tile launches: split along K so that they fill the chip, like the W = C^H V products of the
            //  back-transformation -- unsplit they ran at 10 TFLOP/s and the whole trailing part at 14.8)
            gemm_splitk<T>(c, st, m, b, m, one, L, opB('N', (const T*)V, m), zero, W, m, kch);                              // W = lower(A22) V
            Operand<T> Lh = op_plain((const T*)F22, N, 1, 1);
            Lh.mask = M_LOWER;     // stored coordinates: the same lower triangle, read transposed-conjugated (its diagonal counted once is a detail of the real thing)
            gemm_splitk<T>(c, st, m, b, m, one, Lh, opB('N', (const T*)V, m), one, W, m, kch);                              // W += strict_lower(A22)^H V
            gemm<T>(c, st, m, b, b, one, opA('N', (const T*)W, m), opB('N', (const T*)S, b), zero, X, m);                   // W T
            gemm_splitk<T>(c, st, b, b, m, one, opA('C', (const T*)V, m), opB('N', (const T*)X, m), zero, S2, b, kch);      // S = V^H W
            gemm<T>(c, st, b, b, b, one, opA('C', (const T*)S, b), opB('N', (const T*)S2, b), zero, G, b);                  // T^H S
            gemm<T>(c, st, m, b, b, mhalf, opA('N', (const T*)V, m), opB('N', (const T*)G, b), one, X, m);                  // X = W - 1/2 V (T^H S)
            Operand<T> Ao, Bo;                                                                                            // A22 -= X V^H + V X^H (lower)
            Ao.p = X; Ao.ld = m; Ao.trans = 0; Ao.conj = 0; Ao.k1 = b; Ao.p2 = V; Ao.ld2 = m;
            Bo.p = V; Bo.ld = m; Bo.trans = 0; Bo.conj = 1; Bo.k1 = b; Bo.p2 = X; Bo.ld2 = m;
            Epi e; e.uplo = 2; e.herm_diag = 1;
            gemm<T>(c, st, m, m, 2 * b, mone, Ao, Bo, one, F22, N, e);
        }
    }
    EIG_HIP(hipGetLastError());
}
template void two_stage_stage1_skeleton<cplx>(Ctx&, hipStream_t, int, int);
template void two_stage_stage1_skeleton<double>(Ctx&, hipStream_t, int, int);
#endif  // EIG_TOOLS

// explicit instantiations
#define INST(T)                                                                                                          \
    template void gemm<T>(Ctx&, hipStream_t, int, int, int, T, const Operand<T>&, const Operand<T>&, T, T*, int, Epi);   \
    template void gemm_splitk<T>(Ctx&, hipStream_t, int, int, int, T, const Operand<T>&, const Operand<T>&, T, T*, int,  \
                                 int, Epi);                                                                              \
    template void her2k_un<T>(Ctx&, hipStream_t, int, int, const T*, int, const T*, int, T*, int);                       \
    template void gemm_batched<T>(Ctx&, hipStream_t, int, int, int, T, const Operand<T>&, const Operand<T>&, T, T*, int, \
                                  Epi, GemmBatch, int);                                                                  \
    template void potrf_upper<T>(Ctx&, hipStream_t, int, T*, int);                                                       \
    template void build_invU<T>(Ctx&, hipStream_t, int, const T*, int);                                                  \
    template void trsm_LUN<T>(Ctx&, hipStream_t, int, int, const T*, int, int, T*, int, T*, int, int,                    \
                              const std::function<void(int, int)>*, int, int);                            \
    template void trsm_LUC<T>(Ctx&, hipStream_t, int, int, const T*, int, int, T*, int, T*, int, int);                            \
    template void trsm_RUN<T>(Ctx&, hipStream_t, int, int, const T*, int, int, T*, int, T*, int, int);                            \
    template void build_inv256<T>(Ctx&, hipStream_t, int, const T*, int);                                                \
    template void build_inv_blocks<T>(Ctx&, hipStream_t, int, const T*, int);                                            \
    template void hegst_upper<T>(Ctx&, hipStream_t, int, T*, int, const T*, int);                                        \
    template void potrf_upper_group<T>(Ctx&, hipStream_t, int, int, T* const*, int);                                     \
    template bool pipeline_applicable<T>(const Ctx&, int);                                                               \
    template void potrf_hegst_pipelined_begin<T>(Ctx&, int, T*, int, T*, int);                                           \
    template void hegst_pipelined_finish<T>(Ctx&, int, T*, int, const T*, int);
INST(double)
INST(cplx)

}  // namespace eig
