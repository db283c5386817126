// microbench2.hip -- fp64 MFMA ceiling vs occupancy, MFMA+VALU co-issue, real clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode: 0 = all waves MFMA, 1 = all waves VALU fma, 2 = even waves MFMA / odd waves VALU
template <int NACC> __global__ void __launch_bounds__(256) mix(double* out, long long* cyc, int iters, int mode) {
    int wave = threadIdx.x >> 6;
    bool do_mfma = (mode == 0) || (mode == 2 && (wave & 1) == 0);
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    double s = 0;
    if (do_mfma) {
        d4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double a[16];
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
        double b = 1.000001, c = 1e-9;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < NACC; ++r) {   // 16 fma per r: same flop count per wave as one 16x16x4 MFMA? (MFMA = 2048 flop/wave; 16 fma*64 lanes*2 = 2048)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = fma(a[i], b, c);
            }
        }
        for (int i = 0; i < 16; ++i) s += a[i];
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    double* out; CK(hipMalloc(&out, 1 << 26));
    long long* cyc; CK(hipMalloc(&cyc, 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 3000;
    const char* names[] = {"MFMA only", "VALU fma only", "MFMA(even waves)+VALU(odd waves)"};
    for (int mode = 0; mode < 3; ++mode)
        for (int bpc : {1, 2, 4, 8}) {
            int blocks = p.multiProcessorCount * bpc;
            mix<8><<<blocks, 256>>>(out, cyc, 10, mode);
            CK(hipDeviceSynchronize());
            hipEventRecord(e0); mix<8><<<blocks, 256>>>(out, cyc, iters, mode); hipEventRecord(e1); CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
            double fl = (double)blocks * 4 * iters * 8 * 2048.0;
            printf("%-34s %d waves/SIMD: %.3f ms %.1f TFLOP/s | block0: %lld shader cyc, %lld wallclk(100MHz) -> %.2f GHz, %.1f cyc per 2048-flop unit per wave\n",
                   names[mode], bpc, ms, fl / ms * 1e-9, h[0], h[1], h[0] / (h[1] * 10.0), (double)h[0] / (iters * 8.0));
        }
    return 0;
}
