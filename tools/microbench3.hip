// microbench3.hip -- why does a bare MFMA loop top out below rocBLAS' zgemm?  Variants.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// NACC accumulators, NOP distinct operand pairs, data scale `dz` (0 -> all-zero operands)
template <int NACC, int NOP> __global__ void __launch_bounds__(256) mf(double* out, int iters, double dz) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a[NOP], b[NOP];
    for (int i = 0; i < NOP; ++i) { a[i] = dz * (threadIdx.x * 1e-3 + i); b[i] = dz * (1.0 + threadIdx.x * 1e-4 - i * 0.1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i % NOP], b[(i / 2) % NOP], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 4x4x4 (4 blocks) variant
template <int NACC> __global__ void __launch_bounds__(256) mf4(double* out, int iters, double dz) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = dz * threadIdx.x * 1e-3, b = dz * (1.0 + threadIdx.x * 1e-4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    double* out; CK(hipMalloc(&out, 1 << 26));
    int iters = 2000;
    for (double dz : {1.0, 0.0})
        for (int bpc : {1, 2, 4}) {
            int blocks = p.multiProcessorCount * bpc;
            float ms;
            ms = timeit([&] { mf<8, 1><<<blocks, 256>>>(out, iters, dz); });
            printf("dz=%.0f %d blk/CU  8acc 1op : %.3f ms %.1f TF\n", dz, bpc, ms, (double)blocks * 4 * iters * 8 * 2048.0 / ms * 1e-9);
            ms = timeit([&] { mf<8, 4><<<blocks, 256>>>(out, iters, dz); });
            printf("dz=%.0f %d blk/CU  8acc 4op : %.3f ms %.1f TF\n", dz, bpc, ms, (double)blocks * 4 * iters * 8 * 2048.0 / ms * 1e-9);
            ms = timeit([&] { mf<16, 4><<<blocks, 256>>>(out, iters, dz); });
            printf("dz=%.0f %d blk/CU 16acc 4op : %.3f ms %.1f TF\n", dz, bpc, ms, (double)blocks * 4 * iters * 16 * 2048.0 / ms * 1e-9);
            ms = timeit([&] { mf<4, 2><<<blocks, 256>>>(out, iters, dz); });
            printf("dz=%.0f %d blk/CU  4acc 2op : %.3f ms %.1f TF\n", dz, bpc, ms, (double)blocks * 4 * iters * 4 * 2048.0 / ms * 1e-9);
            ms = timeit([&] { mf4<8><<<blocks, 256>>>(out, iters, dz); });
            printf("dz=%.0f %d blk/CU 4x4x4 8acc: %.3f ms %.1f TF\n", dz, bpc, ms, (double)blocks * 4 * iters * 8 * 512.0 / ms * 1e-9);
        }
    // half the chip only (power headroom?): 128 blocks
    float ms = timeit([&] { mf<8, 4><<<128, 256>>>(out, iters, 1.0); });
    printf("128 blocks only (half the CUs): %.3f ms %.1f TF (x2 = %.1f)\n", ms, 128.0 * 4 * iters * 8 * 2048.0 / ms * 1e-9, 2 * 128.0 * 4 * iters * 8 * 2048.0 / ms * 1e-9);
    ms = timeit([&] { mf<8, 4><<<32, 256>>>(out, iters, 1.0); });
    printf("32 blocks only: %.3f ms -> %.1f cycles@2.4GHz per MFMA per SIMD\n", ms, ms * 1e-3 * 2.4e9 / (iters * 8.0));
    return 0;
}
