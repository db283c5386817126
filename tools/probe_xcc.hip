// Which XCDs does a CU-masked stream run on?  hipExtStreamCreateWithCUMask takes a bit mask over the device's CUs; this probe launches a
// kernel on streams with a few mask layouts and reports the XCC_ID (s_getreg_b32 HW_REG_XCC_ID) of every workgroup.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_xcc.hip -o tools/_build/probe_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); return 1; } } while (0)
__global__ void who(int* xcc, int* cu) {
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    if (threadIdx.x == 0) { xcc[blockIdx.x] = (int)(x & 0xf); cu[blockIdx.x] = (int)h; }
    // keep the workgroup alive a little so that the launch spreads over everything the mask allows
    long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
}
static int run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s stream creation failed: %s\n", name, hipGetErrorName(e)); return 0; }
    const int nb = 2048;
    int *xcc, *cu;
    CK(hipMalloc(&xcc, nb * sizeof(int))); CK(hipMalloc(&cu, nb * sizeof(int)));
    hipLaunchKernelGGL(who, dim3(nb), dim3(256), 0, st, xcc, cu);
    CK(hipStreamSynchronize(st));
    std::vector<int> hx(nb), hc(nb);
    CK(hipMemcpy(hx.data(), xcc, nb * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), cu, nb * sizeof(int), hipMemcpyDeviceToHost));
    int cnt[16] = {};
    for (int i = 0; i < nb; ++i) cnt[hx[i] & 15]++;
    printf("%-28s workgroups per XCC:", name);
    for (int i = 0; i < 8; ++i) printf(" %4d", cnt[i]);
    printf("   first 16 workgroups' XCC:");
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    printf("\n");
    CK(hipFree(xcc)); CK(hipFree(cu));
    CK(hipStreamDestroy(st));
    return 0;
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = 8;   // 256 bits
    std::vector<uint32_t> all(words, 0xffffffffu);
    run("all 256 bits", all);
    { std::vector<uint32_t> m(words, 0); m[0] = 0xffffffffu; m[1] = 0xffffffffu; run("bits 0..63", m); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xffffffffu; run("bits 0..31", m); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < 256; ++b) if (b % 8 < 2) m[b / 32] |= 1u << (b % 32); run("bits with b%8 in {0,1}", m); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < 256; ++b) if (b % 8 == 0) m[b / 32] |= 1u << (b % 32); run("bits with b%8 == 0", m); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < 256; ++b) if (b % 4 == 0) m[b / 32] |= 1u << (b % 32); run("bits with b%4 == 0", m); }
    { std::vector<uint32_t> m(words, 0); for (int b = 64; b < 128; ++b) m[b / 32] |= 1u << (b % 32); run("bits 64..127", m); }
    { std::vector<uint32_t> m(words, 0); for (int b = 0; b < 256; ++b) if (b % 8 >= 6) m[b / 32] |= 1u << (b % 32); run("bits with b%8 in {6,7}", m); }
    return 0;
}
