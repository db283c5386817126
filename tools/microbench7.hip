// microbench7.hip -- can the two per-column kernels of the tridiagonalization hand over INSIDE running launches instead of at
// kernel boundaries?  (round 6, experiment 12)
//
// A column of ?hetrd is row(i) -> mv(i) -> row(i+1): two dependent launches, ~2 us of boundary + ~0.8-3 us of ramp each (launch
// ramp, and for the mat-vec the first tile's trip from memory, which depends on NOTHING the row kernel writes).  If mv(i) could
// start while row(i) still runs -- its tile loads in flight, a spin on a counter in front of the first use of x -- the ramp and
// the first tile would be hidden.  This benchmark prices exactly that hand-over with kernels shaped like the real ones:
//   R(i) ("row", 256 workgroups x 256 threads): [wait: all workgroups of M(i-1) done]  read 16 rows x 64 partial sums of 16 B,
//        write 16 entries of the column x, bump counter_row;
//   M(i) ("mv", 256 workgroups x 320 threads): load its first 64 KB tile of an immutable matrix, [wait: all workgroups of R(i) done],
//        read 128 entries of x, stream T tiles, write 2 KB of partial sums per tile, bump counter_mv.
// Mode 0: ONE stream, R(0) M(0) R(1) M(1) ...: the hand-overs are kernel boundaries (plain loads / stores) -- today's form.
// Mode 1: TWO streams (R's on one, M's on the other), the hand-overs are the counters: data crosses with 16-B sc1 stores (write-
//         through, drained before the bump) and sc1 loads, the placement-independent form of MI355X_MICROARCH.md.
// Mode 2: one stream, launches flagged hipExtAnyOrderLaunch (documented as unsupported on gfx9: measured anyway).
// Every spin is bounded (a missed hand-over shows up as a timeout count, never as a hang); every value read is checked.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench7.hip -o tools/_build/microbench7
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void st_sc1(d2* p, d2 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ d2 ld_sc1(const d2* p) {
    d2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
constexpr int GR = 256, GM = 256, NT = 64, N = 4096;

// bounded wait until *ctr >= target (one lane polls, the workgroup follows through LDS)
__device__ __forceinline__ bool wait_counter(const unsigned* ctr, unsigned target, unsigned* tmo) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 0;
        // (after the first timeout every later launch gives up at once: a missed hand-over must not turn into minutes of spinning)
        const long limit = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : (1L << 21);
        for (long spins = 0; spins < limit; ++spins) {
            if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicAdd(tmo, 1u);
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

template <bool HANDOVER>
__global__ void __launch_bounds__(256) row_kernel(int i, const d2* P, d2* x, unsigned* ctr_row, const unsigned* ctr_mv, unsigned* err, unsigned* tmo) {
    const int w = blockIdx.x, tid = threadIdx.x;
    if (HANDOVER && i > 0 && !wait_counter(ctr_mv, (unsigned)i * GM, tmo)) return;
    // 16 rows x NT partial sums: thread (r = tid >> 4, q0 = tid & 15) adds NT / 16 of them; expected value of every partial: i
    const int r = w * 16 + (tid >> 4);
    double s = 0.0;
    for (int q = tid & 15; q < NT; q += 16) {
        const d2* p = P + (size_t)q * N + r;
        const d2 v = HANDOVER ? ld_sc1(p) : *p;
        s += v.x;
    }
    for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 16);
    if (i > 0 && s != (double)NT * (double)i && (tid & 15) == 0) atomicAdd(err, 1u);
    if ((tid & 15) == 0) {
        const d2 v = d2{(double)(i + 1), s};
        if (HANDOVER) st_sc1(x + r, v); else x[r] = v;
    }
    if (HANDOVER) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr_row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool HANDOVER>
__global__ void __launch_bounds__(320) mv_kernel(int i, int T, const d2* A, const d2* x, d2* P, const unsigned* ctr_row, unsigned* ctr_mv, unsigned* err,
                                                  unsigned* tmo, double* sink) {
    const int w = blockIdx.x, tid = threadIdx.x;
    // first tile: 64 KB of the immutable matrix, 256 streaming threads x 16 loads of 16 B -- issued BEFORE the wait
    d2 t[16];
    const d2* a0 = A + ((size_t)w * 4096);
    if (tid < 256)
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = a0[c * 256 + tid];
    if (HANDOVER && !wait_counter(ctr_row, (unsigned)(i + 1) * GR, tmo)) return;
    // x: 128 entries this workgroup's tiles need (here: 64 from its own block, 64 from the next)
    double xv = 0.0;
    if (tid < 128) {
        const d2* p = x + ((w * 16 + tid) & (N - 1));
        const d2 v = HANDOVER ? ld_sc1(p) : *p;
        if (v.x != (double)(i + 1)) atomicAdd(err, 1u);
        xv = v.x;
    }
    double acc = xv;
    for (int k = 0; k < T; ++k) {
        if (tid < 256) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc += t[c].x;
            if (k + 1 < T) {
                const d2* an = A + ((size_t)(w + (k + 1) * GM) * 4096);
#pragma unroll
                for (int c = 0; c < 16; ++c) t[c] = an[c * 256 + tid];
            }
        }
    }
    // partial sums for the NEXT row kernel: every (q, row) slot must hold i + 1 -- workgroup w writes slots q = w % NT ... for its share of rows
    for (int e = tid; e < NT * N / GM; e += 320) {
        const size_t idx = (size_t)w * (NT * N / GM) + e;
        const d2 v = d2{(double)(i + 1), 0.0};
        if (HANDOVER) st_sc1(P + idx, v); else P[idx] = v;
    }
    if (acc == 12345.678) sink[0] = acc;
    if (HANDOVER) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr_mv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int cols = argc > 1 ? atoi(argv[1]) : 1000;
    CK(hipSetDevice(0));
    d2 *A, *P, *x;
    unsigned* ctr;
    double* sink;
    const size_t a_elems = (size_t)GM * 9 * 4096;      // up to 9 tiles of 64 KB per workgroup
    CK(hipMalloc(&A, a_elems * sizeof(d2)));
    CK(hipMemset(A, 0, a_elems * sizeof(d2)));
    CK(hipMalloc(&P, (size_t)NT * N * sizeof(d2)));
    CK(hipMalloc(&x, N * sizeof(d2)));
    CK(hipMalloc(&ctr, 4096));
    CK(hipMalloc(&sink, 64));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int T : {1, 2, 8})
        for (int mode = 0; mode < 3; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(ctr, 0, 4096));
                CK(hipMemset(P, 0, (size_t)NT * N * sizeof(d2)));
                CK(hipDeviceSynchronize());
                unsigned *c_row = ctr, *c_mv = ctr + 64, *err = ctr + 128, *tmo = ctr + 192;
                CK(hipEventRecord(e0, s1));
                if (mode == 1) CK(hipStreamWaitEvent(s2, e0, 0));
                for (int i = 0; i < cols; ++i) {
                    if (mode == 0) {
                        hipLaunchKernelGGL(row_kernel<false>, dim3(GR), dim3(256), 0, s1, i, (const d2*)P, x, c_row, (const unsigned*)c_mv, err, tmo);
                        hipLaunchKernelGGL(mv_kernel<false>, dim3(GM), dim3(320), 0, s1, i, T, (const d2*)A, (const d2*)x, P, (const unsigned*)c_row, c_mv, err, tmo, sink);
                    } else if (mode == 1) {
                        hipLaunchKernelGGL(row_kernel<true>, dim3(GR), dim3(256), 0, s1, i, (const d2*)P, x, c_row, (const unsigned*)c_mv, err, tmo);
                        hipLaunchKernelGGL(mv_kernel<true>, dim3(GM), dim3(320), 0, s2, i, T, (const d2*)A, (const d2*)x, P, (const unsigned*)c_row, c_mv, err, tmo, sink);
                    } else {
                        hipExtLaunchKernelGGL(row_kernel<true>, dim3(GR), dim3(256), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, i, (const d2*)P, x, c_row,
                                              (const unsigned*)c_mv, err, tmo);
                        hipExtLaunchKernelGGL(mv_kernel<true>, dim3(GM), dim3(320), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, i, T, (const d2*)A, (const d2*)x, P,
                                              (const unsigned*)c_row, c_mv, err, tmo, sink);
                    }
                }
                CK(hipGetLastError());
                CK(hipEventRecord(e1, s1));
                if (mode == 1) { CK(hipEventRecord(e2, s2)); CK(hipEventSynchronize(e2)); }
                CK(hipEventSynchronize(e1));
                float ms = 0, ms2 = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (mode == 1) { CK(hipEventElapsedTime(&ms2, e0, e2)); if (ms2 > ms) ms = ms2; }
                unsigned h[256];
                CK(hipMemcpy(h, ctr, 1024, hipMemcpyDeviceToHost));
                printf("T=%d tiles  %-34s : %7.3f us per column   (errors %u, timeouts %u)\n", T,
                       mode == 0 ? "one stream, kernel boundaries" : (mode == 1 ? "two streams, in-launch hand-over" : "one stream, any-order launches"),
                       ms * 1e3 / cols, h[128], h[192]);
                fflush(stdout);
                if (h[192]) break;
            }
    return 0;
}
