// microbench5.hip -- price of the in-launch exchanges a multi-workgroup, register-resident finish of the tridiagonalization
// would need: P co-resident workgroups (one per CU, 512 threads), per step
//   ALLGATHER : every workgroup publishes its slice of an n-vector (16-B sc1 write-through stores, drained, one relaxed
//               agent-scope flag store), polls the P flags (one wave, relaxed sc1 loads) and reads the whole vector with
//               16-B sc1 loads (no fences: the placement-independent {sc1 store, sc1 load} form of MI355X_MICROARCH.md);
//   BCAST     : one workgroup (round-robin owner) publishes the whole n-vector, the others poll its flag and read it.
// Every word read is checked against the value its producer must have written (stale data shows up as an error count).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench5.hip -o tools/_build/microbench5
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

__device__ __forceinline__ double val_of(int it, int idx) { return (double)(it * 4099 + idx); }

// mode 0: allgather, 1: broadcast.  Y: 2 x n entries (double buffered), flags: 2 x P words, err: error counter, tmo: timeout word
template <int MODE>
__global__ void __launch_bounds__(512) exch_kernel(int P, int n, int iters, d2* Y, unsigned* flags, unsigned* err, unsigned* tmo) {
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
                if (idx < n) st_sc1(Yb + idx, d2{val_of(it, idx), -val_of(it, idx)});
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
            ld4_sc1(Yb + i0, Yb + i1, Yb + i2, Yb + i3, a, b, c, d);
            bad += (a.x != val_of(it, i0)) + (b.x != val_of(it, i1)) + (c.x != val_of(it, i2)) + (d.x != val_of(it, i3));
            bad += (a.y != -val_of(it, i0));
        }
    }
    if (bad) atomicAdd(err, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int dev = 0;
    CK(hipSetDevice(dev));
    d2* Y; unsigned *flags, *err;
    CK(hipMalloc(&Y, 2 * 4096 * sizeof(d2)));
    CK(hipMalloc(&flags, 4096));
    CK(hipMalloc(&err, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int P : {8, 16, 32, 64, 128}) {
            for (int n : {256, 768, 1152}) {
                CK(hipMemset(flags, 0, 4096));
                CK(hipMemset(err, 0, 64));
                CK(hipMemset(Y, 0, 2 * 4096 * sizeof(d2)));
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(exch_kernel<0>, dim3(P), dim3(512), 0, 0, P, n, iters, Y, flags, err, err + 1);
                else hipLaunchKernelGGL(exch_kernel<1>, dim3(P), dim3(512), 0, 0, P, n, iters, Y, flags, err, err + 1);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned h[2];
                CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
                printf("%s P=%3d n=%4d : %7.3f us per step   (errors %u, timeouts %u)\n", mode == 0 ? "allgather" : "bcast    ", P, n,
                       ms * 1e3 / iters, h[0], h[1]);
                fflush(stdout);
            }
        }
    return 0;
}
