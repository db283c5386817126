// microbench4: does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N) shorten a chain of dependent small kernels?
// Each kernel: 256 workgroups x 256 threads, one dependent global load -> store on data the previous launch wrote.
// Built twice (with / without the flag) by tools/r03_microbench4.sh; prints us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Blob { double* p; const double* q; int n, ld, a, b; int pad[24]; };
__global__ void __launch_bounds__(256) k_scalar(double* p, const double* q, int n, int ld, int a, int b) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) p[(size_t)t] = q[(size_t)((t + a) % n)] * 0.5 + b + ld;
}
__global__ void __launch_bounds__(256) k_struct(Blob s) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < s.n) s.p[(size_t)t] = s.q[(size_t)((t + s.a) % s.n)] * 0.5 + s.b + s.ld;
}
int main() {
    const int n = 256 * 256, reps = 4000;
    double *x, *y;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) {
                double* a = (r & 1) ? x : y; const double* b = (r & 1) ? y : x;
                if (mode == 0) hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, st, a, b, n, 1, 77, 0);
                else { Blob s{a, b, n, 1, 77, 0, {0}}; hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, st, s); }
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("%s args: %.3f us per dependent launch\n", mode == 0 ? "scalar" : "struct", ms * 1e3 / reps);
        }
    }
    return 0;
}
