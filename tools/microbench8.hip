// microbench8.hip -- the in-launch all-gather of microbench5 among P workgroups placed on ONE XCD (round 6, experiment 14).
// Workgroup b of a launch runs on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch"); a launch of 8 P workgroups whose members with
// b % 8 != 0 exit at once leaves P workgroups on XCD 0, which share ONE L2.  Variants of the data path:
//   0: sc1 stores + sc1 loads  (placement-independent: the form microbench5 priced at 2.4 us per step across XCDs)
//   1: plain stores (the L1 is write-through: they land in the XCD's L2) + sc1 loads (bypass the reader's L1, served by the shared L2)
//   2: plain stores + nt loads
// Every word read is checked (stale data = an error count); every workgroup reports its XCC_ID (placement is a hardware habit HIP
// does not promise: a product path may use it for speed, never for correctness -- variants 1 / 2 are correct only on one XCD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench8.hip -o tools/_build/microbench8
#include <hip/hip_runtime.h>
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
// four loads in flight
__device__ __forceinline__ void ld4_sc1(const d2* p0, const d2* p1, const d2* p2, const d2* p3, d2& a, d2& b, d2& c, d2& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

__device__ __forceinline__ void ld4_nt(const d2* p0, const d2* p1, const d2* p2, const d2* p3, d2& a, d2& b, d2& c, d2& d) {
    asm volatile("global_load_dwordx4 %0, %4, off nt\n\tglobal_load_dwordx4 %1, %5, off nt\n\tglobal_load_dwordx4 %2, %6, off nt\n\t"
                 "global_load_dwordx4 %3, %7, off nt\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ double val_of(int it, int idx) { return (double)(it * 4099 + idx); }

// mode 0: allgather, 1: broadcast.  Y: 2 x n entries (double buffered), flags: 2 x P words, err: error counter, tmo: timeout word
template <int VAR>
__global__ void __launch_bounds__(512) exch_kernel(int P, int n, int iters, d2* Y, unsigned* flags, unsigned* err, unsigned* tmo, int* xcc) {
    constexpr int MODE = 0;
    if (blockIdx.x & 7) return;                       // only the workgroups the dispatcher puts on XCD 0 take part
    const int w = blockIdx.x >> 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[w] = (int)(x & 0xf);
    }
    const int per = (n + P - 1) / P;
    unsigned bad = 0;
    __shared__ int fail;
    if (tid == 0) fail = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        d2* Yb = Y + (size_t)par * n;
        unsigned* fl = flags + par * P;
        // ---- publish ----
        if (MODE == 0) {
            for (int i = tid; i < per; i += 512) {
                const int idx = w * per + i;
                if (idx < n) { if (VAR == 0) st_sc1(Yb + idx, d2{val_of(it, idx), -val_of(it, idx)}); else Yb[idx] = d2{val_of(it, idx), -val_of(it, idx)}; }
            }
        } else if (w == it % P) {
            for (int idx = tid; idx < n; idx += 512) st_sc1(Yb + idx, d2{val_of(it, idx), -val_of(it, idx)});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && (MODE == 0 || w == it % P)) __hip_atomic_store(fl + (MODE == 0 ? w : 0), (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- poll (wave 0) ----
        if (wave == 0) {
            const int nfl = MODE == 0 ? P : 1;
            long spins = 0;
            for (;;) {
                bool ok = true;
                for (int q = lane; q < nfl; q += 64)
                    ok &= (int)(__hip_atomic_load(fl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)(it + 1)) >= 0;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 20)) { if (lane == 0) { fail = 1; atomicAdd(tmo, 1u); } break; }
            }
        }
        __syncthreads();
        if (fail) break;
        // ---- consume: every thread reads its share of the whole vector ----
        for (int base = tid * 4; base < n; base += 512 * 4) {
            d2 a, b, c, d;
            const int i0 = base, i1 = min(base + 1, n - 1), i2 = min(base + 2, n - 1), i3 = min(base + 3, n - 1);
            if (VAR == 2) ld4_nt(Yb + i0, Yb + i1, Yb + i2, Yb + i3, a, b, c, d); else ld4_sc1(Yb + i0, Yb + i1, Yb + i2, Yb + i3, a, b, c, d);
            bad += (a.x != val_of(it, i0)) + (b.x != val_of(it, i1)) + (c.x != val_of(it, i2)) + (d.x != val_of(it, i3));
            bad += (a.y != -val_of(it, i0));
        }
    }
    if (bad) atomicAdd(err, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    CK(hipSetDevice(0));
    d2* Y; unsigned *flags, *err; int* xcc;
    CK(hipMalloc(&Y, 2 * 4096 * sizeof(d2)));
    CK(hipMalloc(&flags, 4096));
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&xcc, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 3; ++var)
        for (int P : {8, 16, 32})
            for (int n : {256, 768, 1280}) {
                CK(hipMemset(flags, 0, 4096));
                CK(hipMemset(err, 0, 64));
                CK(hipMemset(xcc, 0xff, 4096));
                CK(hipMemset(Y, 0, 2 * 4096 * sizeof(d2)));
                CK(hipEventRecord(e0, 0));
                if (var == 0) hipLaunchKernelGGL(exch_kernel<0>, dim3(8 * P), dim3(512), 0, 0, P, n, iters, Y, flags, err, err + 1, xcc);
                else if (var == 1) hipLaunchKernelGGL(exch_kernel<1>, dim3(8 * P), dim3(512), 0, 0, P, n, iters, Y, flags, err, err + 1, xcc);
                else hipLaunchKernelGGL(exch_kernel<2>, dim3(8 * P), dim3(512), 0, 0, P, n, iters, Y, flags, err, err + 1, xcc);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned h[2];
                int hx[64];
                CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hx, xcc, P * sizeof(int), hipMemcpyDeviceToHost));
                int other = 0;
                for (int q = 0; q < P; ++q) other += hx[q] != hx[0];
                printf("%s P=%2d n=%4d : %7.3f us per step   (errors %u, timeouts %u; XCC of workgroup 0: %d, workgroups elsewhere: %d)\n",
                       var == 0 ? "sc1 stores + sc1 loads  " : (var == 1 ? "plain stores + sc1 loads" : "plain stores + nt loads "), P, n,
                       ms * 1e3 / iters, h[0], h[1], hx[0], other);
                fflush(stdout);
            }
    return 0;
}
