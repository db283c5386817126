// probe444.hip -- lane layout of v_mfma_f64_4x4x4_4b_f64 incl. cbsz/abid broadcast and blgp (neg) bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int CBSZ, int ABID, int BLGP> __global__ void k(const double* a, const double* b, double* out) {
    int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, CBSZ, ABID, BLGP);
    out[l] = acc;
}
int main() {
    double *da, *db, *dout; CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dout, 512));
    std::vector<double> a(64), b(64), o(64);
    // 1) find which A lane / B lane contributes to each output lane: one-hot probing
    // A one-hot at lane la (value 1), B all = 1 + lane index pattern -> output lanes show which outputs use A[la] and with which B lanes
    printf("== plain (cbsz=0): for each output lane list (A lane, B lane) pairs contributing ==\n");
    std::vector<std::vector<std::pair<int,int>>> contrib(64);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        // to keep it cheap: use 2 runs per la: B = primes? use unique weights: b[l] = 2^l is too big -> do la x lb only for same block
        if ((la >> 4) != (lb >> 4)) continue;
        for (int i = 0; i < 64; ++i) { a[i] = 0; b[i] = 0; }
        a[la] = 1; b[lb] = 1;
        CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
        k<0, 0, 0><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; ++i) if (o[i] != 0) contrib[i].push_back({la, lb});
    }
    for (int i = 0; i < 20; ++i) { printf("out lane %2d:", i); for (auto& p : contrib[i]) printf(" (A%d,B%d)", p.first, p.second); printf("\n"); }
    // derive: A lane -> (i,k), B lane -> (k,j), out lane -> (i,j) within block 0
    // 2) broadcast test: cbsz=2, abid=r: A one-hot in block r -> which outputs light up (B all ones)
    for (int i = 0; i < 64; ++i) { a[i] = 0; b[i] = 1; }
    a[16 * 2 + 5] = 1;  // block 2, inner 5
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    k<2, 2, 0><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("== cbsz=2 abid=2, A one-hot at lane 37, B=1: nonzero output lanes:");
    for (int i = 0; i < 64; ++i) if (o[i] != 0) printf(" %d", i);
    printf("\n");
    k<2, 0, 0><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("== cbsz=2 abid=0 (A one-hot in block 2 should be ignored): nonzero output lanes:");
    for (int i = 0; i < 64; ++i) if (o[i] != 0) printf(" %d", i);
    printf("\n");
    // 3) blgp as NEG bits? a=b=1 everywhere, blgp=1,2,4
    for (int i = 0; i < 64; ++i) { a[i] = 1; b[i] = 1; }
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    k<0, 0, 0><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost)); printf("blgp=0: out[0]=%g\n", o[0]);
    k<0, 0, 1><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost)); printf("blgp=1: out[0]=%g\n", o[0]);
    k<0, 0, 2><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost)); printf("blgp=2: out[0]=%g\n", o[0]);
    k<0, 0, 4><<<1, 64>>>(da, db, dout); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost)); printf("blgp=4: out[0]=%g\n", o[0]);
    return 0;
}
