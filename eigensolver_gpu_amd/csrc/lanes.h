// lanes.h -- wave64 cross-lane primitives for gfx950 (internal).
//
// __shfl_xor lowers to ds_bpermute_b32: every dependent step is an LDS-crossbar round trip (>100 cycles),
// and the panel kernels of the tridiagonalization sit on chains of 6-12 such steps per launch.  These
// helpers use what the hardware offers instead:
//   * DPP row operations (quad_perm, row_ror, row_half_mirror, row_mirror): a VALU move, a few cycles;
//   * v_readlane for the last two levels of a full-wave sum;
//   * v_permlane32_swap / v_permlane16_swap (new in gfx950): exchange halves / odd-even rows of two
//     registers in one instruction -- exactly one level of a transpose-reduce butterfly.
#pragma once
#include "common.h"

namespace eig {

constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_ROR8 = 0x128;        // row_ror:8  == lane ^ 8 inside a 16-lane row
constexpr int DPP_HALF_MIRROR = 0x141; // lane -> 7 - lane inside each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // lane -> 15 - lane inside each 16-lane row

// value of `v` in the lane selected by CTRL (all lanes enabled)
template <int CTRL> __device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lanes in the banks of BANK_A receive a from their CTRL partner, the other lanes receive b from theirs
template <int CTRL, int BANK_A> __device__ __forceinline__ double dpp_get2(double a, double b) {
    int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    int lo = __builtin_amdgcn_update_dpp(0, blo, CTRL, 0xf, 0xf ^ BANK_A, false);
    int hi = __builtin_amdgcn_update_dpp(0, bhi, CTRL, 0xf, 0xf ^ BANK_A, false);
    lo = __builtin_amdgcn_update_dpp(lo, alo, CTRL, 0xf, BANK_A, false);
    hi = __builtin_amdgcn_update_dpp(hi, ahi, CTRL, 0xf, BANK_A, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ cplx dpp_get(cplx v) { return cplx{dpp_get<CTRL>(v.x), dpp_get<CTRL>(v.y)}; }
template <int CTRL, int BANK_A> __device__ __forceinline__ cplx dpp_get2(cplx a, cplx b) {
    return cplx{dpp_get2<CTRL, BANK_A>(a.x, b.x), dpp_get2<CTRL, BANK_A>(a.y, b.y)};
}

// sum over the 16 lanes of each row, result in every lane of the row
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_get<DPP_XOR1>(v);
    v += dpp_get<DPP_XOR2>(v);
    v += dpp_get<DPP_HALF_MIRROR>(v);
    v += dpp_get<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ double read_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// sum over the 64 lanes, result (wave-uniform) in every lane.  Must be called with all 64 lanes active.
__device__ __forceinline__ double wave_sum(double v) {
    v = row_sum16(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
__device__ __forceinline__ cplx wave_sum(cplx v) { return cplx{wave_sum(v.x), wave_sum(v.y)}; }
__device__ __forceinline__ cplx row_sum16(cplx v) { return cplx{row_sum16(v.x), row_sum16(v.y)}; }

// a' = {lanes 0-31: a, lanes 32-63: b(lane-32)},  b' = {lanes 0-31: a(lane+32), lanes 32-63: b}
__device__ __forceinline__ void swap_halves(double& a, double& b) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// rows of 16 lanes: a' = {a.r0, b.r0, a.r2, b.r2},  b' = {a.r1, b.r1, a.r3, b.r3}
__device__ __forceinline__ void swap_rows(double& a, double& b) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap_halves(cplx& a, cplx& b) { swap_halves(a.x, b.x); swap_halves(a.y, b.y); }
__device__ __forceinline__ void swap_rows(cplx& a, cplx& b) { swap_rows(a.x, b.x); swap_rows(a.y, b.y); }

// 16 per-lane partials t[0..15] (one per column) -> the full 64-lane sum of column (lane>>2)&15 in every
// lane of the quad.  Levels: halves (permlane32_swap), rows (permlane16_swap), 8 (row_ror:8), 4 (half
// mirror), then the quad sum.  Deterministic summation order.
template <class T> __device__ __forceinline__ T transpose_reduce16(T (&t)[16], int lane) {
    T u8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        T x = t[j], y = t[j + 8];
        swap_halves(x, y);
        u8[j] = x + y;            // lanes 0-31: column j, lanes 32-63: column j+8
    }
    T u4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        T x = u8[j], y = u8[j + 4];
        swap_rows(x, y);
        u4[j] = x + y;            // + 4 * (bit 4 of lane)
    }
    const bool b3 = lane & 8, b2 = lane & 4;
    T u2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        T x = u4[j], y = u4[j + 2];
        T recv = dpp_get2<DPP_ROR8, 0x3>(x, y);       // banks 0,1 (bit 3 clear) keep x and receive x
        u2[j] = (b3 ? y : x) + recv;
    }
    T recv = dpp_get2<DPP_HALF_MIRROR, 0x5>(u2[0], u2[1]);   // banks 0,2 (bit 2 clear)
    T out = (b2 ? u2[1] : u2[0]) + recv;
    out = out + dpp_get<DPP_XOR2>(out);
    out = out + dpp_get<DPP_XOR1>(out);
    return out;
}

// The same reduction in two halves of 8 columns, so that only 8 per-lane partials (plus two reduced values of the
// first half) are live at a time.  Phase 1 takes 8 partials (local columns 0..7) through the halves / rows levels
// and leaves two values per lane: w[a] holds local column a + 2*bit4 + 4*bit5.  Phase 2 merges the two halves
// (row_ror:8 level: bit 3 selects the half), then the half-mirror level (bit 2 selects a) and the quad sum.
// Result: the full 64-lane sum of column transpose_col_of_lane(lane) in every lane of the quad.
template <class T> __device__ __forceinline__ void transpose_reduce8_phase1(T (&x)[8], T& w0, T& w1) {
    T v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        T p = x[a], q = x[a + 4];
        swap_halves(p, q);
        v[a] = p + q;             // lanes 0-31: local column a, lanes 32-63: a + 4
    }
    {
        T p = v[0], q = v[2];
        swap_rows(p, q);
        w0 = p + q;               // + 2 * (bit 4 of lane)
    }
    {
        T p = v[1], q = v[3];
        swap_rows(p, q);
        w1 = p + q;
    }
}
template <class T> __device__ __forceinline__ T transpose_reduce_phase2(T a0, T a1, T b0, T b1, int lane) {
    const bool b3 = lane & 8, b2 = lane & 4;
    T z0 = sel(b3, b0, a0) + dpp_get2<DPP_ROR8, 0x3>(a0, b0);   // banks 0,1 (bit 3 clear) keep / receive the first half
    T z1 = sel(b3, b1, a1) + dpp_get2<DPP_ROR8, 0x3>(a1, b1);
    T out = sel(b2, z1, z0) + dpp_get2<DPP_HALF_MIRROR, 0x5>(z0, z1);   // banks 0,2 (bit 2 clear)
    out = out + dpp_get<DPP_XOR2>(out);
    out = out + dpp_get<DPP_XOR1>(out);
    return out;
}
__device__ __forceinline__ int transpose_col_of_lane(int lane) {
    // local column a = bit 2, + 2 * bit 4 + 4 * bit 5 inside the half, + 8 * bit 3 for the second half
    return ((lane >> 2) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 5) & 1) + 8 * ((lane >> 3) & 1);
}

// ---- LDS-DMA (global_load_lds_dwordx4) -------------------------------------------------------------------------------------
// One wave instruction copies 64 x 16 B straight from global memory into LDS at [M0 base + 16 * lane]: no VGPR destination, so the
// number of bytes a wave keeps in flight is bounded by the LDS it owns, not by its registers (MI355X_MICROARCH.md, "LDS-DMA").
// hipcc does not count these operations (cdna_hip_programming.md 5.7): completion is waited for with explicit s_waitcnt vmcnt(N),
// and a wave that uses them issues NO other vector-memory loads, so the counts below are exact.  M0 is saved and restored around
// the instruction (it is compiler-reserved).
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(uintptr_t)p; }   // low half of the flat address
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst_wave_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_wave_uniform)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 1/x to ~1 ulp without the IEEE division sequence (x normal, nonzero): v_rcp_f64 + two Newton steps
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// 1/sqrt(x) to ~1 ulp (x normal, positive): v_rsq_f64 + two Newton steps
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// sqrt(x) to ~1 ulp for x >= 0 (x normal or zero): x * rsqrt(x) with one residual correction
__device__ __forceinline__ double fast_sqrt(double x) {
    const double y = fast_rsqrt(x > 0.0 ? x : 1.0);
    double g = x * y;
    g = fma(fma(-g, g, x), 0.5 * y, g);
    return g;
}

}  // namespace eig
