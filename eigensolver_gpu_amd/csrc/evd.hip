// evd.hip -- standard eigensolver (trd -> host stedc -> back-transform), the generalized
// drivers, and the C ABI.  Replaces zheevd_gpu / dsyevd_gpu (+ zlarft_gpu, zlarfb_gpu,
// finish_T_block_kernel; zheevd_gpu.F90:32-279, dsyevd_gpu.F90:32-276) and zhegvdx_gpu /
// dsygvdx_gpu (zhegvdx_gpu.F90:75-182, dsygvdx_gpu.F90:71-168).
#include <chrono>
#include <cstring>
#include <functional>
#include <initializer_list>

#include <climits>
#include "../../include/eigsolve_gpu.h"
#include "stedc.h"
#include "trd.h"

// smallest N*m for which the host copy of Z is issued block row by block row beside the final solve (hegvdx_core)
#ifndef EIG_ZOVERLAP_MIN
#define EIG_ZOVERLAP_MIN (2048L * 512L)
#endif
namespace eig {

// ---- small kernels ---------------------------------------------------------------------------

// Z(:, 0:m) <- Q(:, 0:m) (real host eigenvectors of T, uploaded as fp64) widened to T.
template <class T> __global__ void __launch_bounds__(256) widen_kernel(int n, int m, const double* Q, int ldq, T* Z, int ldz) {
    size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)n * m) return;
    int r = (int)(id % n), j = (int)(id / n);
    Z[(size_t)r + (size_t)j * ldz] = Tr<T>::make(Q[(size_t)r + (size_t)j * ldq], 0.0);
}

// finish_T_block_kernel (zheevd_gpu.F90:215-279): Tm holds S = V^H V (lower) on entry.
//   T(r,j) <- -tau(j) S(r,j) (r>j), T(j,j) <- tau(j), then for col = K-2..0:
//   T(r,col) <- sum_{j=col+1..r} T(j,col) T(r,j), r > col.   Upper part is zeroed.
// Batched: workgroup b finishes the T factor of reflector block b (K = min(nb2, k - b*nb2)).
template <class T> __global__ void __launch_bounds__(64) finish_T_kernel(int k, int nb2, T* Tall, int ldt, const T* tau_all) {
    __shared__ T t[64][65];  // t[col][row]
    const int tx = threadIdx.x;
    // reflector blocks of nb2 <= 512 = up to eight 64-wide parts, one workgroup per part
    const int parts = (nb2 + 63) / 64;
    const int b = blockIdx.x / parts, sub = blockIdx.x % parts;
    const int ibb = (k - b * nb2 < nb2) ? k - b * nb2 : nb2;   // reflectors in block b
    const int K = (ibb - sub * 64 < 64) ? ibb - sub * 64 : 64;
    if (K <= 0) return;
    const int i0 = b * nb2 + sub * 64;
    T* Tm = Tall + (size_t)b * ldt * ldt + (size_t)sub * 64 * (1 + ldt);
    const T* tau = tau_all + i0;
    for (int j = 0; j < K; ++j) {
        T v = Tr<T>::zero();
        if (tx < K) {
            if (tx > j) v = -(tau[j] * Tm[(size_t)tx + (size_t)j * ldt]);
            else if (tx == j) v = tau[j];
        }
        t[j][tx] = v;
    }
    __syncthreads();
    for (int col = K - 2; col >= 0; --col) {
        T cv = Tr<T>::zero();
        if (tx > col && tx < K) {
            // four independent partial sums: the LDS reads of consecutive terms overlap instead of queueing behind one
            // dependent multiply-add chain (145 -> ~60 us for the 64 blocks of N = 4096)
            T c1 = Tr<T>::zero(), c2 = Tr<T>::zero(), c3 = Tr<T>::zero();
            int j = col + 1;
            for (; j + 3 <= tx; j += 4) {
                fma_(cv, t[col][j], t[j][tx]);
                fma_(c1, t[col][j + 1], t[j + 1][tx]);
                fma_(c2, t[col][j + 2], t[j + 2][tx]);
                fma_(c3, t[col][j + 3], t[j + 3][tx]);
            }
            for (; j <= tx; ++j) fma_(cv, t[col][j], t[j][tx]);
            cv = (cv + c1) + (c2 + c3);
        }
        __syncthreads();
        if (tx > col && tx < K) t[col][tx] = cv;
        __syncthreads();
    }
    for (int j = 0; j < K; ++j)
        if (tx < K) Tm[(size_t)tx + (size_t)j * ldt] = t[j][tx];
}

// Two 64-reflector T factors of one 128-block -> the T factor of the 128 reflectors:
//   (I - V1 T1 V1^H)(I - V0 T0 V0^H) = I - [V0 V1] [[T0, 0], [T10, T1]] [V0 V1]^H,  T10 = -T1 (V1^H V0) T0.
// On entry the (1,0) block of the buffer still holds S10 = V1^H V0 (from the V^H V product), the diagonal
// blocks hold T0, T1 (lower, zero above the diagonal).  One workgroup per 128-block; thread (tr,tc) owns a
// 4x4 block of the 64x64 result; X = S10 T0 goes through LDS.
template <class T> __global__ void __launch_bounds__(256) merge_T_kernel(int k, int nb2, T* Tall, int ldt) {
    __shared__ T X[64][65];   // X[c][r]
    const int pairs = nb2 / 128;                 // 128-pairs per reflector block (nb2 = 128, 256, 512)
    const int b = blockIdx.x / pairs, pr = blockIdx.x % pairs;
    const int ibb = (k - b * nb2 < nb2) ? k - b * nb2 : nb2;
    const int K1 = min(64, ibb - pr * 128 - 64);   // rows of the pair's (1,0) block
    if (K1 <= 0) return;
    T* Tb = Tall + (size_t)b * ldt * ldt + (size_t)pr * 128 * (1 + ldt);
    const T* T0 = Tb;
    T* S10 = Tb + 64;
    const T* T1 = Tb + (size_t)64 * (1 + ldt);
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Tr<T>::zero();
    // X(r, c) = sum_p S10(r, p) T0(p, c),  T0 lower: p >= c
    for (int p = 4 * tc; p < 64; ++p) {
        T sr[4], tcv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sr[i] = (4 * tr + i < K1) ? S10[(size_t)(4 * tr + i) + (size_t)p * ldt] : Tr<T>::zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) tcv[j] = (p >= 4 * tc + j) ? T0[(size_t)p + (size_t)(4 * tc + j) * ldt] : Tr<T>::zero();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) fma_(acc[i][j], sr[i], tcv[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) X[4 * tc + j][4 * tr + i] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Tr<T>::zero();
    // T10(r, c) = - sum_p T1(r, p) X(p, c),  T1 lower: p <= r
    const int pmax = min(4 * tr + 3, K1 - 1);
    for (int p = 0; p <= pmax; ++p) {
        T t1[4], xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t1[i] = (4 * tr + i < K1 && p <= 4 * tr + i) ? T1[(size_t)(4 * tr + i) + (size_t)p * ldt] : Tr<T>::zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = X[4 * tc + j][p];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) fma_(acc[i][j], t1[i], xv[j]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * tr + i < K1) S10[(size_t)(4 * tr + i) + (size_t)(4 * tc + j) * ldt] = -acc[i][j];
}

// ---- back-transformation ------------------------------------------------------------------------
// Q = H_{N-2} ... H_0 applied to C = Z(0:N, 0:m) in blocks of nb2 reflectors, ascending
// (zheevd_gpu.F90:113-131).  All T factors are built first (they do not depend on C), then each
// block costs three MFMA launches.  V's bottom square is masked on the fly (unit upper
// triangular) instead of the reference's stash / zero / restore of A (:154-164, :203-211).
static inline int bt_norm_nb(int nb2, int N) {
    nb2 = nb2 >= 512 ? 512 : (nb2 >= 256 ? 256 : (nb2 >= 128 ? 128 : 64));
    while (nb2 > 64 && nb2 / 2 >= N) nb2 /= 2;     // small problems: no wider than needed
    return nb2;
}

// All T factors of the reflector blocks (blocks of nb2 = 64 / 128 / 256 / 512 reflectors), in a fixed number of launches:
//   1. Gram blocks S_b = V_b^H V_b (lower) for ALL blocks in one strided-batch split-K launch (+ one reduce);
//   2. the 64x64 diagonal parts by the recurrence of finish_T_block_kernel (zheevd_gpu.F90:215-279), one workgroup each;
//   3. pairwise merges 64 -> 128 (merge_T_kernel), 128 -> 256, 256 -> 512 (two batched MFMA products per level):
//        (I - V1 T1 V1^H)(I - V0 T0 V0^H) = I - [V0 V1] [[T0, 0], [T10, T1]] [V0 V1]^H,   T10 = -T1 (V1^H V0) T0,
//      where V1^H V0 is the (1,0) block of S_b already sitting in the buffer.
// Wider blocks = higher K of the rank-nb2 updates in bt_apply (the MFMA engine reaches ~50 TFLOP/s at K = 256 against ~38
// at K = 128) and half / a quarter of the launches, for (nb2 / N) extra flops in T.
template <class T>
static void bt_build_T(Ctx& c, hipStream_t st, int N, const T* A, int lda, const T* tau, int nb2) {
    const int k = N - 1;
    if (k <= 0) return;
    nb2 = bt_norm_nb(nb2, N);
    const int nblk = (k + nb2 - 1) / nb2;
    const int ldt = nb2;
    T* Tall = c.scratch<T>("bt_T", (size_t)nblk * ldt * ldt);
    kmemset(c, st, Tall, 0, sizeof(T) * (size_t)nblk * ldt * ldt);
    {
        const T* V = A + (size_t)lda;                 // block b: V + b*nb2*lda, rows 0..mi-1, mi = min((b+1) nb2, k)
        Operand<T> Va = op_plain(V, lda, 1, 1);       // (r,p) -> conj(V(p,r))
        Va.mask = M_UNITTRAP; Va.moff = 0;            // unit diagonal at row b*nb2 + column (moff = mi - ib = b*nb2)
        Operand<T> Vb = op_plain(V, lda, 1, 0);       // Bt(j,p) = V(p,j)
        Vb.mask = M_UNITTRAP; Vb.moff = 0;
        Epi e; e.uplo = 2;
        GemmBatch bt;
        bt.count = nblk; bt.sA = bt.sB = (long)nb2 * lda; bt.sC = (long)ldt * ldt;
        bt.dMoffA = bt.dMoffB = nb2; bt.dK = nb2; bt.capM = bt.capN = k; bt.dcap = nb2;
        const int ib0 = k < nb2 ? k : nb2;
        gemm_batched<T>(c, st, ib0, ib0, ib0, Tr<T>::one(), Va, Vb, Tr<T>::zero(), Tall, ldt, e, bt, 512);
    }
    const int parts = nb2 / 64;
    klaunch(c, st, (finish_T_kernel<T>), dim3(nblk * parts), dim3(64), k, nb2, Tall, ldt, tau);
    if (nb2 >= 128) klaunch(c, st, (merge_T_kernel<T>), dim3(nblk * (nb2 / 128)), dim3(256), k, nb2, Tall, ldt);
    EIG_HIP(hipGetLastError());
    for (int s_ = 128; 2 * s_ <= nb2; s_ *= 2) {      // merge pairs of s_-blocks; everything beyond the last reflector is zero
        T* X = c.scratch<T>("bt_X", (size_t)nblk * s_ * s_);
        for (int mg = 0; mg < nb2 / (2 * s_); ++mg) {
            T* base = Tall + (size_t)mg * 2 * s_ * (1 + ldt);
            GemmBatch bt;
            bt.count = nblk; bt.sA = bt.sB = (long)ldt * ldt; bt.sC = (long)s_ * s_;
            Operand<T> S10 = op_plain((const T*)(base + s_), ldt, 0, 0);
            Operand<T> T0 = op_plain((const T*)base, ldt, 1, 0);              // Bt(j,p) = T0(p,j), T0 lower
            T0.mask = M_LOWER;
            gemm_batched<T>(c, st, s_, s_, s_, Tr<T>::one(), S10, T0, Tr<T>::zero(), X, s_, Epi(), bt);          // X = S10 T0
            GemmBatch b2;
            b2.count = nblk; b2.sA = (long)ldt * ldt; b2.sB = (long)s_ * s_; b2.sC = (long)ldt * ldt;
            Operand<T> T1 = op_plain((const T*)(base + (size_t)s_ * (1 + ldt)), ldt, 0, 0);
            T1.mask = M_LOWER;
            gemm_batched<T>(c, st, s_, s_, s_, Tr<T>::make(-1.0, 0.0), T1, opB('N', (const T*)X, s_), Tr<T>::zero(), base + s_, ldt,
                            Epi(), b2);                                                                          // T10 = -T1 X
        }
    }
}

template <class T>
static void bt_apply(Ctx& c, hipStream_t st, int N, int m, const T* A, int lda, T* Z, int ldz, int nb2) {
    const int k = N - 1;
    if (k <= 0 || m <= 0) return;
    nb2 = bt_norm_nb(nb2, N);
    const int nblk = (k + nb2 - 1) / nb2;
    const int ldt = nb2;
    T* Tall = c.scratch<T>("bt_T", (size_t)nblk * ldt * ldt);
    T* Wk = c.scratch<T>("bt_Wk", (size_t)m * nb2);
    T* Wk2 = c.scratch<T>("bt_Wk2", (size_t)m * nb2);
    for (int b = 0; b < nblk; ++b) {
        int i = b * nb2, ib = (k - i < nb2) ? k - i : nb2, mi = i + ib;
        const T* V = A + (size_t)(i + 1) * lda;
        const T* Tb = Tall + (size_t)b * ldt * ldt;
        // Wk = C^H V                                  (:193)
        Operand<T> Ca = op_plain((const T*)Z, ldz, 1, 1);
        Operand<T> Vb = op_plain(V, lda, 1, 0);
        Vb.mask = M_UNITTRAP; Vb.moff = mi - ib;
        int tiles = ((m + 63) / 64) * ((ib + 63) / 64);     // 64x64 output tiles of Wk (m x ib)
        int want = (2 * c.n_cu + tiles - 1) / tiles;        // splits that fill the resident workgroups once
        if (want < 1) want = 1;
        int kchunk = (mi + want - 1) / want;
        if (kchunk < 128) kchunk = 128;
        gemm_splitk<T>(c, st, m, ib, mi, Tr<T>::one(), Ca, Vb, Tr<T>::zero(), Wk, m, kchunk);
        // Wk2 = Wk T^H                                (:197-198)
        Operand<T> Tt = op_plain(Tb, ldt, 0, 1);
        Tt.mask = M_LOWER;
        gemm<T>(c, st, m, ib, ib, Tr<T>::one(), opA('N', (const T*)Wk, m), Tt, Tr<T>::zero(), Wk2, m);
        // C -= V Wk2^H                                (:201)
        Operand<T> Vn = op_plain(V, lda, 0, 0);
        Vn.mask = M_UNITTRAP; Vn.moff = mi - ib;
        gemm<T>(c, st, mi, m, ib, Tr<T>::make(-1.0, 0.0), Vn, op_plain((const T*)Wk2, m, 0, 1), Tr<T>::one(), Z, ldz);
    }
}

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// EIGSOLVE_TRACE_MARKS=1: an empty kernel with grid = 1 + 2*phase (+1 at the end of the phase) is launched at every phase
// boundary, so that a rocprofv3 --kernel-trace of a solve can be segmented exactly (tools/trace_phases.py) -- kernel names
// alone do not say which phase a gemm belongs to.
__global__ void phase_marker_kernel(int) {}
static void phase_mark(const Ctx& c, hipStream_t st, int id) {
    if (c.trace_marks) hipLaunchKernelGGL(phase_marker_kernel, dim3(1 + id), dim3(64), 0, st, id);
}

struct PhaseTimer {
    Ctx& c;
    hipStream_t st;
    explicit PhaseTimer(Ctx& cc) : c(cc), st(cc.s1) {}
    void begin(int ph) { phase_mark(c, st, 2 * ph); (void)hipEventRecord(c.ev[2 * ph], st); }
    void end(int ph) { (void)hipEventRecord(c.ev[2 * ph + 1], st); phase_mark(c, st, 2 * ph + 1); }
    void collect(int ph) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c.ev[2 * ph], c.ev[2 * ph + 1]) == hipSuccess) c.phase_ms[ph] += ms;
    }
};

// ---- heevd: trd -> host dstedc -> upload wanted vectors -> back-transform -----------------------
// e_d/tau_d/W_d: device workspace pieces.  e_h[N], Q_h (N x N, ldq), swork/lswork, iwork: host.
template <class T>
static int heevd_core(Ctx& c, int il, int iu, int N, T* A, int lda, T* Z, int ldz, double* w_d, double* e_d, T* tau_d,
                      T* W_d, double* w_h, double* e_h, double* Q_h, int ldq, double* swork, long lswork, int* iwork,
                      int liwork) {
    hipStream_t st = c.s1;
    PhaseTimer pt(c);
    const int m = iu - il + 1;
    // The real reference path copies the eigenvectors from column 1 whatever il is (dsyevd_gpu.F90:108; the complex path
    // honours il, zheevd_gpu.F90:110).  Default: honour il in both; option "real_il_reference" = 1 reproduces the quirk.
    if (!Tr<T>::cx && c.real_il_reference) { iu = iu - il + 1; il = 1; }
    const T* Vsrc = A;      // where the reflectors live for the back-transformation
    int ldv = lda;
    const T* tau_bt = tau_d;
    {
    PhaseRange trd_range(Tr<T>::cx ? "zhetrd" : "dsytrd");   // zheevd_gpu.F90:80
    pt.begin(PH_TRD);
    if (c.use_graph && N > 64) {
        // hipGraph path: fixed-address internal working set, launch sequence captured once per (type, N).
        const size_t NN = (size_t)N * N;
        T* Aw = c.scratch<T>(Tr<T>::cx ? "g_Az" : "g_Ad", NN);
        T* Ww = c.scratch<T>(Tr<T>::cx ? "g_Wz" : "g_Wd", (size_t)N * 64);
        T* tauw = c.scratch<T>(Tr<T>::cx ? "g_tauz" : "g_taud", (size_t)N + 8);
        double* dw = c.scratch<double>(Tr<T>::cx ? "g_dz" : "g_dd", (size_t)N + 8);
        double* ew = c.scratch<double>(Tr<T>::cx ? "g_ez" : "g_ed", (size_t)N + 8);
        const void* cur[16] = {};
        (void)hemv_scratch_touch<T>(c, N, cur + 5);   // all scratch used inside the captured region exists before capture
        cur[0] = Aw; cur[1] = Ww; cur[2] = tauw; cur[3] = dw; cur[4] = ew;
        char key[64];
        snprintf(key, sizeof key, "trd_%c_%d_%d_%d_%d_%d", Tr<T>::cx ? 'z' : 'd', N, c.trd_nb, c.hemv_blocks, c.trd_finish, c.mv_dma);   // everything the captured launch sequence bakes in
        Ctx::GraphEntry& ge = c.graphs[key];
        bool valid = ge.exec != nullptr;
        for (int q = 0; q < 16 && valid; ++q) valid = (ge.ptrs[q] == cur[q]);
        if (!valid) {
            if (ge.exec) { (void)hipGraphExecDestroy(ge.exec); ge.exec = nullptr; }
            if (ge.graph) { (void)hipGraphDestroy(ge.graph); ge.graph = nullptr; }
            c.sync(st);
            EIG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            try {
                hetrd_upper<T>(c, st, N, Aw, N, dw, ew, tauw, Ww, c.trd_nb);
            } catch (...) {
                // leave the stream usable: end the capture, drop the partial graph, then report the failure
                hipGraph_t partial = nullptr;
                (void)hipStreamEndCapture(st, &partial);
                if (partial) (void)hipGraphDestroy(partial);
                c.graphs.erase(key);
                throw;
            }
            EIG_HIP(hipStreamEndCapture(st, &ge.graph));
            if (hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGraphDestroy(ge.graph);
                c.graphs.erase(key);
                throw HipFail{hipErrorUnknown};
            }
            for (int q = 0; q < 16; ++q) ge.ptrs[q] = cur[q];
        }
        EIG_HIP(hipMemcpy2DAsync(Aw, sizeof(T) * N, A, sizeof(T) * lda, sizeof(T) * N, N, hipMemcpyDeviceToDevice, st));
        EIG_HIP(hipGraphLaunch(ge.exec, st));
        EIG_HIP(hipMemcpyAsync(w_d, dw, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
        if (N > 1) EIG_HIP(hipMemcpyAsync(e_d, ew, sizeof(double) * (N - 1), hipMemcpyDeviceToDevice, st));
        EIG_HIP(hipMemcpyAsync(tau_d, tauw, sizeof(T) * (N - 1), hipMemcpyDeviceToDevice, st));
        Vsrc = Aw; ldv = N; tau_bt = tauw;
    } else {
        // (A padded working copy for the one column stride that streams slower from HBM -- exactly 128 KB, complex lda = 8192 =
        //  BASELINE's C4: the mat-vec launches alone gain 5 % there, profiles/r04_experiments.txt 18 -- was built in round 5 and
        //  LOSES in the whole tridiagonalization: 384 -> 396 ms at C4, twice, profiles/r05_experiments.txt 3.  Removed.)
        hetrd_upper<T>(c, st, N, A, lda, w_d, e_d, tau_d, W_d, c.trd_nb);
    }
    pt.end(PH_TRD);
    }
    // larft T factors depend only on the reflectors: build them on the second stream while the tridiagonal
    // eigenproblem is being solved (zheevd_gpu.F90:125 does the same per block with stream1/stream2)
    // Only for a solve that has the device to itself, and only ENQUEUED once the tridiagonalization has finished: a queue that sits
    // blocked on an event while the per-column kernels run costs every one of their dispatches ~1 us (measured: trd + 4.7 ms).
    const bool ovT = (c.overlap & 2) != 0 && !c.in_batch && streams_in_use(c.dev) <= c.own_streams();
    auto build_T_beside = [&]() {
        hipStream_t stT = c.second_stream();
        bt_build_T<T>(c, stT, N, Vsrc, ldv, tau_bt, c.bt_nb);
        EIG_HIP(hipEventRecord(c.evB, stT));
    };
    {
    PhaseRange stedc_range(Tr<T>::cx ? "zstedc" : "dstedc");   // zheevd_gpu.F90:100
    if (c.tridiag_device) {
        // device-side divide & conquer: no N x N host round trip at all
        c.sync(st);
        pt.collect(PH_TRD);
        if (ovT) build_T_beside();
        double t0 = now_ms();
        double* Qd = nullptr;
        int ldq_d = 0;
        phase_mark(c, st, 2 * PH_STEDC);
        if (stedc_device(c, st, N, w_d, e_d, w_d, &Qd, &ldq_d, il, iu) != 0) {
            printf(" eigsolve error: device tridiagonal eigensolver failed!\n");
            return -1;
        }
        size_t tot = (size_t)N * m;
        hipLaunchKernelGGL((widen_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, m,
                           (const double*)(Qd + (size_t)(il - 1) * ldq_d), ldq_d, Z, ldz);
        EIG_HIP(hipMemcpyAsync(w_h, w_d, sizeof(double) * N, hipMemcpyDeviceToHost, st));
        phase_mark(c, st, 2 * PH_STEDC + 1);
        c.sync(st);
        c.phase_ms[PH_STEDC] += now_ms() - t0;
    } else {
    // d, e -> host (zheevd_gpu.F90:85-86)
    EIG_HIP(hipMemcpyAsync(w_h, w_d, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    if (N > 1) EIG_HIP(hipMemcpyAsync(e_h, e_d, sizeof(double) * (N - 1), hipMemcpyDeviceToHost, st));
    c.sync(st);
    pt.collect(PH_TRD);
    if (ovT) build_T_beside();
    double t0 = now_ms();
    stedc_fn f = get_dstedc();
    if (!f) {
        printf(" eigsolve error: no host LAPACK dstedc available (set EIGSOLVE_LAPACK_LIB or call eigsolve_set_lapack)\n");
        return -1;
    }
    int info = 0, n = N, ldqi = ldq;
    int lw = lswork > 2147483647L ? 2147483647 : (int)lswork;
    f("I", &n, w_h, e_h, Q_h, &ldqi, swork, &lw, iwork, &liwork, &info, 1);  // :101
    if (info != 0) {
        printf(" eigsolve error: dstedc failed! (info=%d)\n", info);
        return -1;
    }
    // wanted vectors (real) and all eigenvalues back to the device (:110-111)
    double* Qd = c.scratch<double>("evd_Q", (size_t)N * m);
    EIG_HIP(hipMemcpy2DAsync(Qd, sizeof(double) * N, Q_h + (size_t)(il - 1) * ldq, sizeof(double) * ldq, sizeof(double) * N,
                             m, hipMemcpyHostToDevice, st));
    EIG_HIP(hipMemcpyAsync(w_d, w_h, sizeof(double) * N, hipMemcpyHostToDevice, st));
    size_t tot = (size_t)N * m;
    hipLaunchKernelGGL((widen_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, m, (const double*)Qd, N, Z,
                       ldz);
    c.sync(st);
    c.phase_ms[PH_STEDC] += now_ms() - t0;
    }
    }
    PhaseRange bt_range(Tr<T>::cx ? "zunmtr" : "dormtr");   // zheevd_gpu.F90:115
    pt.begin(PH_BT);
    if (ovT) EIG_HIP(hipStreamWaitEvent(st, c.evB, 0));
    else bt_build_T<T>(c, st, N, Vsrc, ldv, tau_bt, c.bt_nb);
    bt_apply<T>(c, st, N, m, Vsrc, ldv, Z, ldz, c.bt_nb);
    pt.end(PH_BT);
    return 0;
}

// hipHostMalloc'ed / hipHostRegister'ed memory?  (a device-to-host copy into anything else is staged and blocks the caller)
static bool host_ptr_is_pinned(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static void clear_phases(Ctx& c) {
    for (double& v : c.phase_ms) v = 0.0;
}

// ---- generalized driver -----------------------------------------------------------------------------
template <class T>
static int hegvdx_core(Ctx& c, int N, T* A, int lda, T* B, int ldb, T* Z, int ldz, int il, int iu, double* w_d, double* e_d,
                       T* tau_d, T* W_d, double* w_h, double* e_h, double* Q_h, int ldq, double* swork, long lswork,
                       int* iwork, int liwork, T* Z_h, int ldz_h, int skip_host_copy, const char* name) {
    hipStream_t st = c.s1;
    PhaseTimer pt(c);
    clear_phases(c);
    double t_all = now_ms();
    const int m = iu - il + 1;
    // potrf || the U11-only part of hegst (blas3.hip: potrf_hegst_pipelined_begin) when this solve has the device to itself;
    // phase times: "potrf" = until the factor is complete, "gst" = what is left of hegst after that.
    const bool pipe = (c.overlap & 1) && !c.in_batch && pipeline_applicable<T>(c, N) && streams_in_use(c.dev) <= c.own_streams();
    // Cholesky of B (zhegvdx_gpu.F90:135-142)
    {
        PhaseRange r(Tr<T>::cx ? "cusolverdnZpotrf" : "cusolverdnDpotrf");   // the reference's range name, :134
        pt.begin(PH_POTRF);
        if (pipe) potrf_hegst_pipelined_begin<T>(c, N, A, lda, B, ldb);
        else potrf_upper<T>(c, st, N, B, ldb);
        pt.end(PH_POTRF);
    }
    EIG_HIP(hipMemcpyAsync(c.h_info, c.d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    c.sync(st);
    pt.collect(PH_POTRF);
    if (c.h_info[0] != 0) {
        if (pipe) c.sync(c.second_stream());   // (the part of hegst already queued works on A)
        if (c.h_info[0] < 0) printf(" %s error: potrf failed! (the block-row kernel could not synchronise its workgroups; use option potrf = 0)\n", name);
        else printf(" %s error: potrf failed! (B is not positive definite, pivot %d)\n", name, c.h_info[0]);
        return -1;
    }
    // The reference saves strict-lower(A) in Z here and restores it later (:144-152) because
    // its gst/td2 overwrite parts of it; this implementation never writes below the diagonal.
    {
        PhaseRange r(Tr<T>::cx ? "zhegst_gpu" : "dsygst_gpu");   // :155
        pt.begin(PH_GST);   // (potrf_upper has merged the inverse diagonal blocks already)
        if (pipe) hegst_pipelined_finish<T>(c, N, A, lda, (const T*)B, ldb);
        else hegst_upper<T>(c, st, N, A, lda, B, ldb);  // :156-158
        pt.end(PH_GST);
    }
    int info;
    // The eigenvectors of the standard problem are formed in library scratch (N x m) and the final solve writes Z = U^-1 Zs
    // out of place: no staging copies inside the solve (blas3.hip, trsm_LUN), and the caller's Z is written exactly once.
    // That copy costs sizeof(T) N m of device memory per context (C4: 1 GiB).  Beyond "zs_cap_mb" (default 4 GiB), or when the
    // device cannot provide it, the vectors are formed in the CALLER's Z instead (as the reference does) and the solve runs in column
    // chunks of mc vectors through a smaller block: Z(:, chunk) -> Zs, Zs -> Z(:, chunk).
    const char* zs_slot = Tr<T>::cx ? "evd_Zsz" : "evd_Zsd";
    int mc = m;
    T* Zs = nullptr;
    {
        const size_t cap_elems = (size_t)c.zs_cap_mb * 1048576 / sizeof(T);
        if ((size_t)N * m <= cap_elems) Zs = reinterpret_cast<T*>(c.try_scratch_bytes(zs_slot, sizeof(T) * (size_t)N * m));
        if (!Zs) {
            mc = (int)std::min<size_t>((size_t)m, std::max<size_t>(64, cap_elems / (size_t)N / 64 * 64));
            if (mc >= m) mc = std::max(64, ((m / 2 + 63) / 64) * 64);
            while (!(Zs = reinterpret_cast<T*>(c.try_scratch_bytes(zs_slot, sizeof(T) * (size_t)N * mc)))) {
                if (mc <= 64) throw HipFail{hipErrorOutOfMemory};
                mc = std::max(64, ((mc / 2 + 63) / 64) * 64);
            }
        }
    }
    const bool chunked = mc < m;
    {
        PhaseRange r(Tr<T>::cx ? "zheevd_gpu" : "dsyevd_gpu");   // :161
        info = heevd_core<T>(c, il, iu, N, A, lda, chunked ? Z : Zs, chunked ? ldz : N, w_d, e_d, tau_d, W_d, w_h, e_h, Q_h, ldq, swork,
                             lswork, iwork, liwork);  // :163
    }
    if (info != 0) return -1;
    // The substitution finishes the row blocks of Z bottom-up, and the host copy of a finished block (a quarter of the rows) runs on
    // the second stream beside the rest of the solve (full-spectrum C4: Z is 1 GB, 19 ms over PCIe after a 39 ms solve -> 4.7 ms
    // exposed; C3: 1.19 -> 0.32 ms; below N*m = 2^20 the events cost what the copy gains).  Only for a solve that has the device
    // to itself and only into PINNED host memory: a copy into pageable memory is staged by the runtime and blocks the calling thread
    // until the block's event has fired -- the rest of the solve would not even be queued meanwhile.  The solve itself is unchanged,
    // results are bit-identical.
    // (Splitting the COLUMNS instead changes the split-K decisions of the products, and at m = 1024 half-width solves lose what the
    //  copy gains -- round 3.)
    const bool zoverlap = !skip_host_copy && !chunked && (long)N * m >= (long)EIG_ZOVERLAP_MIN && (c.overlap & 2) && !c.in_batch &&
                          streams_in_use(c.dev) <= c.own_streams() && host_ptr_is_pinned(Z_h);
    bool copy_failed = false;
    {
        PhaseRange r(Tr<T>::cx ? "cublasZtrsm" : "cublasDtrsm");   // :167
        pt.begin(PH_TRSM);
        if (chunked) {
            for (int j0 = 0; j0 < m; j0 += mc) {
                const int wj = std::min(mc, m - j0);
                T* Zj = Z + (size_t)j0 * ldz;
                EIG_HIP(hipMemcpy2DAsync(Zs, sizeof(T) * N, Zj, sizeof(T) * ldz, sizeof(T) * N, wj, hipMemcpyDeviceToDevice, st));
                trsm_LUN<T>(c, st, N, wj, B, ldb, 0, Zs, N, Zj, ldz, c.trsm_base);
            }
        } else if (!zoverlap) {
            trsm_LUN<T>(c, st, N, m, B, ldb, 0, Zs, N, Z, ldz, c.trsm_base);  // :169  Z = U^-1 Zs
        } else {
            hipStream_t sc = c.second_stream();
            int nev = 0;
            const std::function<void(int, int)> rows_done = [&](int r0, int nr) {
                hipEvent_t& ev = c.evStage[nev++ & 15];
                if (!ev) EIG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                EIG_HIP(hipEventRecord(ev, st));
                EIG_HIP(hipStreamWaitEvent(sc, ev, 0));
                if (hipMemcpy2DAsync(Z_h + r0, sizeof(T) * ldz_h, Z + r0, sizeof(T) * ldz, sizeof(T) * nr, m, hipMemcpyDeviceToHost, sc) !=
                    hipSuccess)
                    copy_failed = true;
            };
            trsm_LUN<T>(c, st, N, m, B, ldb, 0, Zs, N, Z, ldz, c.trsm_base, &rows_done, 2);
            EIG_HIP(hipEventRecord(c.evB, sc));
        }
        pt.end(PH_TRSM);
    }
    pt.begin(PH_D2H);   // (overlapped form: what is left of the copies after the solve)
    if (zoverlap) {
        EIG_HIP(hipStreamWaitEvent(st, c.evB, 0));
        if (copy_failed) {
            c.sync(st);
            printf(" %s error: hipMemcpy2D failed!\n", name);
            return -1;
        }
    } else if (!skip_host_copy) {
        hipError_t e = hipMemcpy2DAsync(Z_h, sizeof(T) * ldz_h, Z, sizeof(T) * ldz, sizeof(T) * N, m, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            printf(" %s error: hipMemcpy2D failed!\n", name);
            return -1;
        }
    }
    pt.end(PH_D2H);
    c.sync(st);
    pt.collect(PH_GST); pt.collect(PH_BT); pt.collect(PH_TRSM); pt.collect(PH_D2H);
    c.phase_ms[PH_TOTAL] = now_ms() - t_all;
    return 0;
}

// ---- batch of same-order problems: lockstep tridiagonalization ----------------------------------------------
// QE k-point style batches (BASELINE.json configs[4]) are many problems of ONE order.  The per-column kernels of the
// tridiagonalization -- 2/3 of a solve -- are latency-bound for most of the reduction, so nprob problems share every
// per-column launch (hetrd_upper_batch); the BLAS-3 phases and the tridiagonal eigensolver run problem after problem on the
// same stream with the context's scratch.  Per-problem results are bit-identical to the single-problem driver's.
// Device tridiagonal solver only (the host dstedc path solves the problems one by one through hegvdx_core).
template <class T>
static int hegvdx_batch_core(Ctx& c, int nprob, int N, T* const* A, int lda, T* const* B, int ldb, T* const* Z, int ldz, int il, int iu,
                             double* const* w_d, double* const* e_d, T* const* tau_d, T* const* W_d, double* const* w_h,
                             T* const* Z_h, int ldz_h, int skip_host_copy, int* infos, const char* name) {
    hipStream_t st = c.s1;
    clear_phases(c);
    const double t_all = now_ms();
    const int m = iu - il + 1;
    if (!Tr<T>::cx && c.real_il_reference) { iu = iu - il + 1; il = 1; }   // as heevd_core (dsyevd_gpu.F90:108)
    int* h_inf = reinterpret_cast<int*>(c.host_scratch_bytes("batch_info", sizeof(int) * (size_t)nprob));
    // The BLAS-3 phases of a group: every problem's launches are RECORDED (blas3.h: GroupRecorder; problem j of the group works in
    // scratch slots "<name>#g<j>"), then replayed position by position -- a product on the MFMA engine as one launch for the whole
    // group, anything else once per problem.  A recording during which a slot was (re)allocated holds stale pointers: repeated.
    struct RecGuard { Ctx& c; ~RecGuard() { c.rec = nullptr; c.grp_q = 0; } } rec_guard{c};
    auto grouped = [&](int nq, const std::function<void(int)>& body) {      // body(j): the launches of problem j of the group
        GroupRecorder recs[4];
        for (int attempt = 0; attempt < 3; ++attempt) {
            const unsigned long gen = c.slot_gen;
            for (int j = 0; j < nq; ++j) {
                recs[j].seq.clear();
                c.rec = &recs[j]; c.grp_q = j;
                body(j);
            }
            c.rec = nullptr; c.grp_q = 0;
            if (c.slot_gen == gen) break;
        }
        replay_group(st, recs, nq);
    };
    const bool zipA = (c.batch_zip & 1) != 0, zipC = (c.batch_zip & 2) != 0;
    {
        PhaseRange r("batch: potrf + hegst");
        if (c.potrf_mode != 0 && nprob > 1) {
            // the block-row chains of the group in lockstep (64 x 42 us of latency per factorization, shared), then its inverse
            // diagonal blocks and its reduction to standard form
            for (int q0 = 0; q0 < nprob; q0 += 4) {
                const int nq = std::min(4, nprob - q0);
                potrf_upper_group<T>(c, st, N, nq, B + q0, ldb);
                for (int q = q0; q < q0 + nq; ++q)
                    EIG_HIP(hipMemcpyAsync(&h_inf[q], c.d_info + 4 + (q - q0), sizeof(int), hipMemcpyDeviceToHost, st));
                auto body = [&](int j) {
                    const int q = q0 + j;
                    build_invU<T>(c, st, N, (const T*)B[q], ldb);
                    build_inv_blocks<T>(c, st, N, (const T*)B[q], ldb);
                    hegst_upper<T>(c, st, N, A[q], lda, B[q], ldb);   // (meaningless if B[q] was not positive definite: checked below)
                };
                if (zipA) grouped(nq, body);
                else for (int j = 0; j < nq; ++j) body(j);
            }
        } else {
            for (int q = 0; q < nprob; ++q) {
                potrf_upper<T>(c, st, N, B[q], ldb);
                EIG_HIP(hipMemcpyAsync(&h_inf[q], c.d_info, sizeof(int), hipMemcpyDeviceToHost, st));
                hegst_upper<T>(c, st, N, A[q], lda, B[q], ldb);     // (meaningless if B[q] was not positive definite: checked below)
            }
        }
    }
    {
        PhaseRange r(Tr<T>::cx ? "batch: zhetrd lockstep" : "batch: dsytrd lockstep");
        hetrd_upper_batch<T>(c, st, N, nprob, A, lda, w_d, e_d, tau_d, W_d, c.trd_nb);
    }
    c.sync(st);
    int bad = 0;
    for (int q = 0; q < nprob; ++q) {
        infos[q] = 0;
        if (h_inf[q] != 0) {
            printf(" %s error: potrf failed! (B is not positive definite, pivot %d; batch problem %d)\n", name, h_inf[q], q);
            infos[q] = -1;
            bad = 1;
        }
    }
    const size_t tot = (size_t)N * m;
    // everything behind the tridiagonal eigensolver for problem q, its eigenvectors of T in Qd
    auto tail = [&](int q, const double* Qd, int ldq_d) {
        T* Zs = c.scratch<T>(Tr<T>::cx ? "evd_Zsz" : "evd_Zsd", (size_t)N * m);   // as in hegvdx_core
        klaunch(c, st, (widen_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), N, m, (const double*)(Qd + (size_t)(il - 1) * ldq_d),
                ldq_d, Zs, N);
        {
            PhaseRange r(Tr<T>::cx ? "zunmtr" : "dormtr");
            bt_build_T<T>(c, st, N, A[q], lda, tau_d[q], c.bt_nb);
            bt_apply<T>(c, st, N, m, A[q], lda, Zs, N, c.bt_nb);
        }
        // the inverse diagonal blocks in the context's scratch are those of the LAST factorization: rebuild problem q's
        build_invU<T>(c, st, N, (const T*)B[q], ldb);
        build_inv_blocks<T>(c, st, N, (const T*)B[q], ldb);
        trsm_LUN<T>(c, st, N, m, B[q], ldb, 0, Zs, N, Z[q], ldz, c.trsm_base);
    };
    auto host_copy = [&](int q) {
        if (skip_host_copy) return;
        hipError_t e = hipMemcpy2DAsync(Z_h[q], sizeof(T) * ldz_h, Z[q], sizeof(T) * ldz, sizeof(T) * N, m, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            printf(" %s error: hipMemcpy2D failed!\n", name);
            infos[q] = -1;
            bad = 1;
        }
    };
    auto tridiag = [&](int q, double** Qd, int* ldq_d) -> bool {
        PhaseRange r(Tr<T>::cx ? "zstedc" : "dstedc");
        if (stedc_device(c, st, N, w_d[q], e_d[q], w_d[q], Qd, ldq_d, il, iu) != 0) {
            printf(" eigsolve error: device tridiagonal eigensolver failed! (batch problem %d)\n", q);
            infos[q] = -1;
            bad = 1;
            return false;
        }
        EIG_HIP(hipMemcpyAsync(w_h[q], w_d[q], sizeof(double) * N, hipMemcpyDeviceToHost, st));
        return true;
    };
    if (zipC && nprob > 1) {
        // the tridiagonal eigensolvers of a group one after the other (host-side deflation scans in between), each into slots of its
        // own; then the back-transformations and final solves of the group as one zipped sequence
        for (int q0 = 0; q0 < nprob; q0 += 4) {
            const int nq = std::min(4, nprob - q0);
            int good[4], ng = 0;
            double* Qd[4];
            int ldq_d[4];
            for (int j = 0; j < nq; ++j) {
                const int q = q0 + j;
                if (infos[q] != 0) continue;
                c.grp_q = j;
                const bool ok = tridiag(q, &Qd[ng], &ldq_d[ng]);
                c.grp_q = 0;
                if (ok) good[ng++] = q;
            }
            if (ng == 0) continue;
            // (grouped() numbers the group's scratch slots 0 .. ng-1: the eigenvectors of T stay where their solver put them)
            grouped(ng, [&](int j) { tail(good[j], Qd[j], ldq_d[j]); });
            for (int j = 0; j < ng; ++j) host_copy(good[j]);
        }
    } else {
        for (int q = 0; q < nprob; ++q) {
            if (infos[q] != 0) continue;
            double* Qd = nullptr;
            int ldq_d = 0;
            if (!tridiag(q, &Qd, &ldq_d)) continue;
            tail(q, Qd, ldq_d);
            host_copy(q);
        }
    }
    c.sync(st);
    c.phase_ms[PH_TOTAL] = now_ms() - t_all;
    return bad ? -1 : 0;
}

template <class F> static int guarded(int* info, F&& f);

// ---- batch of problems on the library's worker threads ----------------------------------------------------------------
// One call, one caller thread, `c0.batch_workers` problems in flight: every problem is an ordinary single-problem solve
// (hegvdx_core) on a worker thread's own context and stream, so per-problem results are bit-identical to the one-problem
// driver's and the latency-bound phases of one solve (the Cholesky chain, the per-column kernels of the tridiagonalization,
// the divide & conquer tree) fill under the kernels of the others -- what a caller otherwise needs a host thread per
// problem for (bench.py --inflight).  The reference solves one problem per call (zhegvdx_gpu.F90:75).
template <class T>
static int hegvdx_batch_workers(Ctx& c0, int nprob, int N, T* const* A, int lda, T* const* B, int ldb, T* const* Z, int ldz, int il,
                                int iu, double* const* w_d, double* const* e_d, T* const* tau_d, T* const* W_d, double* const* w_h,
                                T* const* Z_h, int ldz_h, int skip_host_copy, int* infos, const char* name) {
    clear_phases(c0);
    const double t_all = now_ms();
    const int nworkers = c0.batch_workers < 0 ? auto_batch_workers() : c0.batch_workers;
    // problems per launch chain that share the per-column launches of the tridiagonalization (lockstep groups, hegvdx_batch_core)
    // Automatic: while a matrix is small (<= 96 MiB: complex N <= 2508, real N <= 3547) a solve is a chain of latency-bound
    // launches and sharing them pays (C5, N = 2048: 83.8 -> 95-98 problems/s; complex N = 1024: 246 -> 361); above that the
    // mat-vec streams and groups only take concurrency away (complex N = 3072: 34.6 -> 33.9).  profiles/r03_experiments.txt 15.
    int fuse = c0.batch_fuse;
    if (fuse < 1) {
        const double mat_bytes = (double)sizeof(T) * N * (double)N;
        fuse = (mat_bytes <= 96.0 * 1048576.0 && c0.tridiag_device) ? std::max(1, std::min(4, nprob / std::max(1, nworkers))) : 1;
    }
    if (fuse > 4) fuse = 4;   // (MAXB of the lockstep tridiagonalization)
    if (!c0.tridiag_device) fuse = 1;   // (the lockstep core needs the device tridiagonal solver)
    const int ngroups = (nprob + fuse - 1) / fuse;
    const int several = ngroups > 1 && nworkers > 1;
    batch_run(c0.dev, nworkers, ngroups, [&](int gi) {
        const int q0 = gi * fuse, nq = std::min(fuse, nprob - q0);
        for (int q = q0; q < q0 + nq; ++q) infos[q] = -2;   // (sentinel: a group cut short by an exception fails as a whole)
        guarded(nullptr, [&]() -> int {
            Ctx& c = ctx();
            copy_options(c, c0);
            struct InBatch { Ctx& c; explicit InBatch(Ctx& c_, int v) : c(c_) { c.in_batch = v; } ~InBatch() { c.in_batch = 0; } } ib(c, several);
            if (nq == 1)
                return infos[q0] = hegvdx_core<T>(c, N, A[q0], lda, B[q0], ldb, Z[q0], ldz, il, iu, w_d[q0], e_d[q0], tau_d[q0], W_d[q0],
                                                  w_h[q0], nullptr, nullptr, N, nullptr, 0, nullptr, 0, Z_h ? Z_h[q0] : nullptr, ldz_h,
                                                  skip_host_copy, name);
            return hegvdx_batch_core<T>(c, nq, N, A + q0, lda, B + q0, ldb, Z + q0, ldz, il, iu, w_d + q0, e_d + q0, tau_d + q0, W_d + q0,
                                        w_h + q0, Z_h ? Z_h + q0 : nullptr, ldz_h, skip_host_copy, infos + q0, name);
        });
    });
    int bad = 0;
    for (int q = 0; q < nprob; ++q) {
        if (infos[q] == -2) infos[q] = -1;
        bad |= (infos[q] != 0);
    }
    // (the caller's thread is one of the workers: its context holds the phase times of the last problem it solved itself --
    //  a batch call reports the total only)
    clear_phases(c0);
    c0.phase_ms[PH_TOTAL] = now_ms() - t_all;
    return bad ? -1 : 0;
}

// info[] of a batch call: pre-filled with a sentinel, so that whatever path rejects or aborts the call (argument checks, a HIP
// failure before the first synchronisation), every entry that was not explicitly reported reads -1 afterwards
// (include/eigsolve_gpu.h: "a rejected batch call returns -1 and every info[q] = -1").
constexpr int kInfoUnset = INT_MIN;
static void batch_info_begin(int nprob, int* info) {
    if (info && nprob >= 1 && nprob <= 64)
        for (int q = 0; q < nprob; ++q) info[q] = kInfoUnset;
}
static int batch_info_end(int rc, int nprob, int* info) {
    if (info && nprob >= 1 && nprob <= 64)
        for (int q = 0; q < nprob; ++q)
            if (info[q] == kInfoUnset) info[q] = rc == 0 ? 0 : -1;
    return rc;
}

// the batch ABI hands over arrays of pointers: none of them, and none of their entries, may be null
static bool batch_ptrs_ok(int nprob, std::initializer_list<const void* const*> arrays) {
    for (const void* const* a : arrays) {
        if (!a) return false;
        for (int q = 0; q < nprob; ++q)
            if (!a[q]) return false;
    }
    return true;
}

template <class F> static int guarded(int* info, F&& f) {
    int r;
    try {
        StreamLease lease(ctx());   // the context's compute stream for this call (nested calls share it)
        r = f();
    } catch (const HipFail&) {
        r = -1;
    } catch (...) {
        r = -1;
    }
    if (info) *info = r;
    return r;
}

template <class T> static int bench_loop(Ctx& c, int reps, double* ms_avg, const std::function<void()>& body) {
    if (reps < 1) reps = 1;
    body();  // warm-up
    c.sync(c.s1);
    EIG_HIP(hipEventRecord(c.ev[0], c.s1));
    for (int r = 0; r < reps; ++r) body();
    EIG_HIP(hipEventRecord(c.ev[1], c.s1));
    c.sync(c.s1);
    float ms = 0.f;
    EIG_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    if (ms_avg) *ms_avg = (double)ms / reps;
    return 0;
}
}  // namespace eig

using namespace eig;

// liwork_h: the reference announces 3+5N but only rejects liwork_h < N (zhegvdx_gpu.F90:123, dsygvdx_gpu.F90:109).
// With the device tridiagonal solver iwork_h is not read at all, so the reference's actual acceptance (>= N) is kept;
// the host dstedc path really needs 3+5N and says so.
static bool liwork_bad(const Ctx& c, long liwork_h, long n) { return liwork_h < (c.tridiag_device ? n : 3 + 5 * n); }

// (all functions below were declared extern "C" in eigsolve_gpu.h and keep C linkage)

int eigsolve_zhegvdx(int N, void* A_d, int lda, void* B_d, int ldb, void* Z_d, int ldz, int il, int iu, double* w_d,
                     void* work_d, int lwork, double* rwork_d, int lrwork, void* work_h, int lwork_h, double* rwork_h,
                     int lrwork_h, int* iwork_h, int liwork_h, void* Z_h, int ldz_h, double* w_h, int* info,
                     int skip_host_copy) {
    (void)work_h;
    return guarded(info, [&]() -> int {
        const long n = N;
        // workspace checks, same conditions and wording as zhegvdx_gpu.F90:107-127
        if (lwork < 2 * 64 * 64 + 65 * n) { printf(" zhegvdx_gpu error: lwork must be at least 2*64*64 + 65*N\n"); return -1; }
        if (lrwork < n) { printf(" zhegvdx_gpu error: lrwork must be at least N\n"); return -1; }
        if (lwork_h < n) { printf(" zhegvdx_gpu error: lwork_h must be at least N\n"); return -1; }
        if (lrwork_h < 1 + 5 * n + 2 * n * n) { printf(" zhegvdx_gpu error: lrwork_h must be at least 1 + 5*N + 2*N*N\n"); return -1; }
        if (N <= 0 || il < 1 || iu > N || iu < il) { printf(" zhegvdx_gpu error: invalid N/il/iu\n"); return -1; }
        Ctx& c = ctx();
        if (liwork_bad(c, liwork_h, n)) { printf(" zhegvdx_gpu error: liwork_h must be at least 3 + 5*N\n"); return -1; }
        // carve-up as zheevd_gpu.F90:68-75: tau = work(1:N), e = rwork(1:N), rest of work = panel W
        cplx* work = (cplx*)work_d;
        cplx* tau = work;
        cplx* W = work + n;  // 8192 + 64 N elements >= N * 64
        double* e_d = rwork_d;
        double* e_h = rwork_h;            // rwork_h(1:N)
        double* Q_h = rwork_h + n;        // N x N real eigenvectors of T
        double* swork = Q_h + n * n;      // 1 + 4N + N^2 for dstedc('I')
        long lswork = (long)lrwork_h - n - n * n;
        return hegvdx_core<cplx>(c, N, (cplx*)A_d, lda, (cplx*)B_d, ldb, (cplx*)Z_d, ldz, il, iu, w_d, e_d, tau, W, w_h, e_h,
                                 Q_h, N, swork, lswork, iwork_h, liwork_h, (cplx*)Z_h, ldz_h, skip_host_copy, "zhegvdx_gpu");
    });
}

int eigsolve_dsygvdx(int N, double* A_d, int lda, double* B_d, int ldb, double* Z_d, int ldz, int il, int iu, double* w_d,
                     double* work_d, int lwork, double* work_h, int lwork_h, int* iwork_h, int liwork_h, double* Z_h,
                     int ldz_h, double* w_h, int* info, int skip_host_copy) {
    return guarded(info, [&]() -> int {
        const long n = N;
        if (lwork < 2 * 64 * 64 + 66 * n) { printf(" dsygvdx_gpu error: lwork must be at least 2*64*64 + 66*N\n"); return -1; }
        if (lwork_h < 1 + 6 * n + 2 * n * n) { printf(" dsygvdx_gpu error: lwork_h must be at least 1 + 6*N + 2*N*N\n"); return -1; }
        if (N <= 0 || il < 1 || iu > N || iu < il) { printf(" dsygvdx_gpu error: invalid N/il/iu\n"); return -1; }
        Ctx& c = ctx();
        if (liwork_bad(c, liwork_h, n)) { printf(" dsygvdx_gpu error: liwork_h must be at least 3 + 5*N\n"); return -1; }
        // dsyevd_gpu.F90:68-74: e = work(1:N), tau = work(N+1:2N), W = work(2N+1:)
        double* e_d = work_d;
        double* tau = work_d + n;
        double* W = work_d + 2 * n;
        double* e_h = work_h;                  // work_h(1:N)
        double* Q_h = work_h + 2 * n;          // N x N, ld N (the reference puts it in Z_h; Z_h then receives the result)
        double* swork = Q_h + n * n;
        long lswork = (long)lwork_h - 2 * n - n * n;
        return hegvdx_core<double>(c, N, A_d, lda, B_d, ldb, Z_d, ldz, il, iu, w_d, e_d, tau, W, w_h, e_h, Q_h, N, swork, lswork,
                                   iwork_h, liwork_h, Z_h, ldz_h, skip_host_copy, "dsygvdx_gpu");
    });
}

int eigsolve_zhegvdx_batch(int nprob, int N, void* const* A_d, int lda, void* const* B_d, int ldb, void* const* Z_d, int ldz, int il,
                           int iu, double* const* w_d, void* const* work_d, int lwork, double* const* rwork_d, int lrwork,
                           void* const* Z_h, int ldz_h, double* const* w_h, int* info, int skip_host_copy) {
    batch_info_begin(nprob, info);
    int rc = guarded(nullptr, [&]() -> int {
        const long n = N;
        if (nprob < 1 || nprob > 64 || !info) { printf(" zhegvdx_gpu batch error: nprob must be in 1..64 and info an array of nprob ints\n"); return -1; }
        if (lwork < 2 * 64 * 64 + 65 * n) { printf(" zhegvdx_gpu error: lwork must be at least 2*64*64 + 65*N\n"); return -1; }
        if (lrwork < n) { printf(" zhegvdx_gpu error: lrwork must be at least N\n"); return -1; }
        if (N <= 0 || il < 1 || iu > N || iu < il) { printf(" zhegvdx_gpu error: invalid N/il/iu\n"); return -1; }
        if (!batch_ptrs_ok(nprob, {(const void* const*)A_d, (const void* const*)B_d, (const void* const*)Z_d, (const void* const*)w_d,
                                   (const void* const*)work_d, (const void* const*)rwork_d, (const void* const*)w_h}) ||
            (!skip_host_copy && !batch_ptrs_ok(nprob, {(const void* const*)Z_h}))) {
            printf(" zhegvdx_gpu batch error: null pointer in the argument arrays\n");
            return -1;
        }
        Ctx& c = ctx();
        if (!c.tridiag_device) { printf(" zhegvdx_gpu batch error: the batch driver needs the device tridiagonal solver (tridiag = 1)\n"); return -1; }
        std::vector<cplx*> A(nprob), B(nprob), Z(nprob), tau(nprob), W(nprob), Zh(nprob);
        std::vector<double*> e(nprob);
        for (int q = 0; q < nprob; ++q) {
            A[q] = (cplx*)A_d[q]; B[q] = (cplx*)B_d[q]; Z[q] = (cplx*)Z_d[q]; Zh[q] = Z_h ? (cplx*)Z_h[q] : nullptr;
            tau[q] = (cplx*)work_d[q]; W[q] = (cplx*)work_d[q] + n; e[q] = rwork_d[q];       // carve-up of zheevd_gpu.F90:68-75
        }
        if (c.batch_workers != 0)
            return hegvdx_batch_workers<cplx>(c, nprob, N, A.data(), lda, B.data(), ldb, Z.data(), ldz, il, iu, w_d, e.data(), tau.data(),
                                              W.data(), w_h, Z_h ? Zh.data() : nullptr, ldz_h, skip_host_copy, info, "zhegvdx_gpu");
        return hegvdx_batch_core<cplx>(c, nprob, N, A.data(), lda, B.data(), ldb, Z.data(), ldz, il, iu, w_d, e.data(), tau.data(),
                                       W.data(), w_h, Zh.data(), ldz_h, skip_host_copy, info, "zhegvdx_gpu");
    });
    return batch_info_end(rc, nprob, info);
}

int eigsolve_dsygvdx_batch(int nprob, int N, double* const* A_d, int lda, double* const* B_d, int ldb, double* const* Z_d, int ldz, int il,
                           int iu, double* const* w_d, double* const* work_d, int lwork, double* const* Z_h, int ldz_h,
                           double* const* w_h, int* info, int skip_host_copy) {
    batch_info_begin(nprob, info);
    const int rc = guarded(nullptr, [&]() -> int {
        const long n = N;
        if (nprob < 1 || nprob > 64 || !info) { printf(" dsygvdx_gpu batch error: nprob must be in 1..64 and info an array of nprob ints\n"); return -1; }
        if (lwork < 2 * 64 * 64 + 66 * n) { printf(" dsygvdx_gpu error: lwork must be at least 2*64*64 + 66*N\n"); return -1; }
        if (N <= 0 || il < 1 || iu > N || iu < il) { printf(" dsygvdx_gpu error: invalid N/il/iu\n"); return -1; }
        if (!batch_ptrs_ok(nprob, {(const void* const*)A_d, (const void* const*)B_d, (const void* const*)Z_d, (const void* const*)w_d,
                                   (const void* const*)work_d, (const void* const*)w_h}) ||
            (!skip_host_copy && !batch_ptrs_ok(nprob, {(const void* const*)Z_h}))) {
            printf(" dsygvdx_gpu batch error: null pointer in the argument arrays\n");
            return -1;
        }
        Ctx& c = ctx();
        if (!c.tridiag_device) { printf(" dsygvdx_gpu batch error: the batch driver needs the device tridiagonal solver (tridiag = 1)\n"); return -1; }
        std::vector<double*> e(nprob), tau(nprob), W(nprob);
        for (int q = 0; q < nprob; ++q) { e[q] = work_d[q]; tau[q] = work_d[q] + n; W[q] = work_d[q] + 2 * n; }   // dsyevd_gpu.F90:68-74
        if (c.batch_workers != 0)
            return hegvdx_batch_workers<double>(c, nprob, N, A_d, lda, B_d, ldb, Z_d, ldz, il, iu, w_d, e.data(), tau.data(), W.data(), w_h,
                                                Z_h, ldz_h, skip_host_copy, info, "dsygvdx_gpu");
        return hegvdx_batch_core<double>(c, nprob, N, A_d, lda, B_d, ldb, Z_d, ldz, il, iu, w_d, e.data(), tau.data(), W.data(), w_h, Z_h,
                                         ldz_h, skip_host_copy, info, "dsygvdx_gpu");
    });
    return batch_info_end(rc, nprob, info);
}

int eigsolve_zheevd(int il, int iu, int N, void* A_d, int lda, void* Z_d, int ldz, double* w_d, void* work_d, int lwork,
                    double* rwork_d, int lrwork, void* work_h, int lwork_h, double* rwork_h, int lrwork_h, int* iwork_h,
                    int liwork_h, void* Z_h, int ldz_h, double* w_h, int* info) {
    (void)work_h; (void)lwork_h; (void)Z_h; (void)ldz_h;
    return guarded(info, [&]() -> int {
        const long n = N;
        if (N <= 0 || il < 1 || iu > N || iu < il) { printf(" zheevd_gpu error: invalid N/il/iu\n"); return -1; }
        Ctx& c = ctx();
        if (lwork < 2 * 64 * 64 + 65 * n || lrwork < n || lrwork_h < 1 + 5 * n + 2 * n * n || liwork_bad(c, liwork_h, n)) {
            printf(" zheevd_gpu error: workspace too small\n");
            return -1;
        }
        clear_phases(c);
        cplx* work = (cplx*)work_d;
        int r = heevd_core<cplx>(c, il, iu, N, (cplx*)A_d, lda, (cplx*)Z_d, ldz, w_d, rwork_d, work, work + n, w_h, rwork_h,
                                 rwork_h + n, N, rwork_h + n + n * n, (long)lrwork_h - n - n * n, iwork_h, liwork_h);
        c.sync(c.s1);
        return r;
    });
}

int eigsolve_dsyevd(int il, int iu, int N, double* A_d, int lda, double* Z_d, int ldz, double* w_d, double* work_d, int lwork,
                    double* work_h, int lwork_h, int* iwork_h, int liwork_h, double* Z_h, int ldz_h, double* w_h, int* info) {
    (void)Z_h; (void)ldz_h;
    return guarded(info, [&]() -> int {
        const long n = N;
        if (N <= 0 || il < 1 || iu > N || iu < il) { printf(" dsyevd_gpu error: invalid N/il/iu\n"); return -1; }
        Ctx& c = ctx();
        if (lwork < 2 * 64 * 64 + 66 * n || lwork_h < 1 + 6 * n + 2 * n * n || liwork_bad(c, liwork_h, n)) {
            printf(" dsyevd_gpu error: workspace too small\n");
            return -1;
        }
        clear_phases(c);
        int r = heevd_core<double>(c, il, iu, N, A_d, lda, Z_d, ldz, w_d, work_d, work_d + n, work_d + 2 * n, w_h, work_h,
                                   work_h + 2 * n, N, work_h + 2 * n + n * n, (long)lwork_h - 2 * n - n * n, iwork_h, liwork_h);
        c.sync(c.s1);
        return r;
    });
}

int eigsolve_zhegst(int N, void* A_d, int lda, const void* B_d, int ldb, int nb) {
    (void)nb;
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        build_invU<cplx>(c, c.s1, N, (const cplx*)B_d, ldb);
        build_inv_blocks<cplx>(c, c.s1, N, (const cplx*)B_d, ldb);
        hegst_upper<cplx>(c, c.s1, N, (cplx*)A_d, lda, (const cplx*)B_d, ldb);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_dsygst(int N, double* A_d, int lda, const double* B_d, int ldb, int nb) {
    (void)nb;
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        build_invU<double>(c, c.s1, N, B_d, ldb);
        build_inv_blocks<double>(c, c.s1, N, B_d, ldb);
        hegst_upper<double>(c, c.s1, N, A_d, lda, B_d, ldb);
        c.sync(c.s1);
        return 0;
    });
}

template <class T> static int hetrd_entry(int N, T* A, int lda, double* d, double* e, T* tau, T* work, int lwork, int nb) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        if (nb <= 0) nb = c.trd_nb;
        if (nb > 64) nb = 64;
        T* W = work;
        if (!W || (long)lwork < (long)N * nb) W = c.scratch<T>("trd_W", (size_t)N * nb);
        hetrd_upper<T>(c, c.s1, N, A, lda, d, e, tau, W, nb);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_zhetrd(int N, void* A_d, int lda, double* d_d, double* e_d, void* tau_d, void* work_d, int lwork, int nb) {
    return hetrd_entry<cplx>(N, (cplx*)A_d, lda, d_d, e_d, (cplx*)tau_d, (cplx*)work_d, lwork, nb);
}
int eigsolve_dsytrd(int N, double* A_d, int lda, double* d_d, double* e_d, double* tau_d, double* work_d, int lwork, int nb) {
    return hetrd_entry<double>(N, A_d, lda, d_d, e_d, tau_d, work_d, lwork, nb);
}

template <class T> static int potrf_entry(int N, T* B, int ldb, int* info_h) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        potrf_upper<T>(c, c.s1, N, B, ldb);
        EIG_HIP(hipMemcpyAsync(c.h_info, c.d_info, sizeof(int), hipMemcpyDeviceToHost, c.s1));
        c.sync(c.s1);
        if (info_h) *info_h = c.h_info[0];
        return 0;
    });
}
int eigsolve_zpotrf(int N, void* B_d, int ldb, int* info_h) { return potrf_entry<cplx>(N, (cplx*)B_d, ldb, info_h); }
int eigsolve_dpotrf(int N, double* B_d, int ldb, int* info_h) { return potrf_entry<double>(N, B_d, ldb, info_h); }

template <class T> static int hemv_entry(int n, const T* A, int lda, const T* x, T* y) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        hemv_upper<T>(c, c.s1, n, A, lda, x, y, true);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_zhemv(int n, const void* A_d, int lda, const void* x_d, void* y_d) {
    return hemv_entry<cplx>(n, (const cplx*)A_d, lda, (const cplx*)x_d, (cplx*)y_d);
}
int eigsolve_dsymv(int n, const double* A_d, int lda, const double* x_d, double* y_d) {
    return hemv_entry<double>(n, A_d, lda, x_d, y_d);
}
template <class T> static int hemv_bench_entry(int n, const T* A, int lda, const T* x, T* y, int reps, double* ms) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        return bench_loop<T>(c, reps, ms, [&]() { hemv_upper<T>(c, c.s1, n, A, lda, x, y, false); });
    });
}
int eigsolve_zhemv_bench(int n, const void* A_d, int lda, const void* x_d, void* y_d, int reps, double* ms_avg) {
    return hemv_bench_entry<cplx>(n, (const cplx*)A_d, lda, (const cplx*)x_d, (cplx*)y_d, reps, ms_avg);
}
int eigsolve_dsymv_bench(int n, const double* A_d, int lda, const double* x_d, double* y_d, int reps, double* ms_avg) {
    return hemv_bench_entry<double>(n, A_d, lda, x_d, y_d, reps, ms_avg);
}

template <class T> static T scal_from(const double* p) { return Tr<T>::make(p[0], Tr<T>::cx ? p[1] : 0.0); }

template <class T>
static int gemm_entry(char ta, char tb, int M, int N, int K, const double* alpha, const T* A, int lda, const T* B, int ldb,
                      const double* beta, T* C, int ldc) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        gemm<T>(c, c.s1, M, N, K, scal_from<T>(alpha), opA(ta, A, lda), opB(tb, B, ldb), scal_from<T>(beta), C, ldc);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_zgemm(char ta, char tb, int M, int N, int K, const double* alpha, const void* A_d, int lda, const void* B_d, int ldb,
                   const double* beta, void* C_d, int ldc) {
    return gemm_entry<cplx>(ta, tb, M, N, K, alpha, (const cplx*)A_d, lda, (const cplx*)B_d, ldb, beta, (cplx*)C_d, ldc);
}
int eigsolve_dgemm(char ta, char tb, int M, int N, int K, const double* alpha, const double* A_d, int lda, const double* B_d,
                   int ldb, const double* beta, double* C_d, int ldc) {
    return gemm_entry<double>(ta, tb, M, N, K, alpha, A_d, lda, B_d, ldb, beta, C_d, ldc);
}
template <class T>
static int gemm_bench_entry(char ta, char tb, int M, int N, int K, const T* A, int lda, const T* B, int ldb, T* C, int ldc,
                            int reps, double* ms) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        return bench_loop<T>(c, reps, ms, [&]() {
            gemm<T>(c, c.s1, M, N, K, Tr<T>::one(), opA(ta, A, lda), opB(tb, B, ldb), Tr<T>::zero(), C, ldc);
        });
    });
}
int eigsolve_zgemm_bench(char ta, char tb, int M, int N, int K, const void* A_d, int lda, const void* B_d, int ldb, void* C_d,
                         int ldc, int reps, double* ms_avg) {
    return gemm_bench_entry<cplx>(ta, tb, M, N, K, (const cplx*)A_d, lda, (const cplx*)B_d, ldb, (cplx*)C_d, ldc, reps, ms_avg);
}
int eigsolve_dgemm_bench(char ta, char tb, int M, int N, int K, const double* A_d, int lda, const double* B_d, int ldb,
                         double* C_d, int ldc, int reps, double* ms_avg) {
    return gemm_bench_entry<double>(ta, tb, M, N, K, A_d, lda, B_d, ldb, C_d, ldc, reps, ms_avg);
}

#ifdef EIG_TOOLS   // experiment hooks: only in the tools-side build (make -C eigensolver_gpu_amd/csrc tools; tools/eigsolve_tools.h)
// experiment hook (tools/gemm_shapes.py): like ?gemm_bench with beta = 1 and/or operand masks (enum Mask, stored coordinates)
template <class T>
static int gemm_probe_entry(char ta, char tb, int M, int N, int K, const T* A, int lda, const T* B, int ldb, T* C, int ldc,
                            int reps, int beta_one, int maskA, int moffA, int maskB, int moffB, double* ms) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        Operand<T> a = opA(ta, A, lda), b = opB(tb, B, ldb);
        a.mask = maskA; a.moff = moffA; b.mask = maskB; b.moff = moffB;
        return bench_loop<T>(c, reps, ms, [&]() {
            gemm<T>(c, c.s1, M, N, K, Tr<T>::one(), a, b, beta_one ? Tr<T>::one() : Tr<T>::zero(), C, ldc);
        });
    });
}
extern "C" int eigsolve_zgemm_probe(char ta, char tb, int M, int N, int K, const void* A_d, int lda, const void* B_d, int ldb,
                                    void* C_d, int ldc, int reps, int beta_one, int maskA, int moffA, int maskB, int moffB,
                                    double* ms_avg) {
    return gemm_probe_entry<cplx>(ta, tb, M, N, K, (const cplx*)A_d, lda, (const cplx*)B_d, ldb, (cplx*)C_d, ldc, reps, beta_one,
                                  maskA, moffA, maskB, moffB, ms_avg);
}
extern "C" int eigsolve_dgemm_probe(char ta, char tb, int M, int N, int K, const double* A_d, int lda, const double* B_d, int ldb,
                                    double* C_d, int ldc, int reps, int beta_one, int maskA, int moffA, int maskB, int moffB,
                                    double* ms_avg) {
    return gemm_probe_entry<double>(ta, tb, M, N, K, A_d, lda, B_d, ldb, C_d, ldc, reps, beta_one, maskA, moffA, maskB, moffB, ms_avg);
}

// timing hook (tools/two_stage_model.py): the launch skeleton of stage 1 of a two-stage reduction, ms per pass
extern "C" int eigsolve_debug_two_stage_model(int N, int cplx_, int what, int reps, double* ms_avg) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        if (cplx_) return bench_loop<cplx>(c, reps, ms_avg, [&]() { two_stage_stage1_skeleton<cplx>(c, c.s1, N, what); });
        return bench_loop<double>(c, reps, ms_avg, [&]() { two_stage_stage1_skeleton<double>(c, c.s1, N, what); });
    });
}
#endif  // EIG_TOOLS

template <class T> static int her2k_entry(int n, int k, const T* V, int ldv, const T* W, int ldw, T* C, int ldc, int reps, double* ms) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        if (reps > 0) return bench_loop<T>(c, reps, ms, [&]() { her2k_un<T>(c, c.s1, n, k, V, ldv, W, ldw, C, ldc); });
        her2k_un<T>(c, c.s1, n, k, V, ldv, W, ldw, C, ldc);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_zher2k(int n, int k, const void* V_d, int ldv, const void* W_d, int ldw, void* C_d, int ldc) {
    return her2k_entry<cplx>(n, k, (const cplx*)V_d, ldv, (const cplx*)W_d, ldw, (cplx*)C_d, ldc, 0, nullptr);
}
int eigsolve_dsyr2k(int n, int k, const double* V_d, int ldv, const double* W_d, int ldw, double* C_d, int ldc) {
    return her2k_entry<double>(n, k, V_d, ldv, W_d, ldw, C_d, ldc, 0, nullptr);
}
int eigsolve_zher2k_bench(int n, int k, const void* V_d, int ldv, const void* W_d, int ldw, void* C_d, int ldc, int reps,
                          double* ms_avg) {
    return her2k_entry<cplx>(n, k, (const cplx*)V_d, ldv, (const cplx*)W_d, ldw, (cplx*)C_d, ldc, reps < 1 ? 1 : reps, ms_avg);
}
int eigsolve_dsyr2k_bench(int n, int k, const double* V_d, int ldv, const double* W_d, int ldw, double* C_d, int ldc, int reps,
                          double* ms_avg) {
    return her2k_entry<double>(n, k, V_d, ldv, W_d, ldw, C_d, ldc, reps < 1 ? 1 : reps, ms_avg);
}

template <class T> static int trsm_entry(int N, int m, const T* U, int ldu, T* Z, int ldz) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        build_invU<T>(c, c.s1, N, U, ldu);
        build_inv_blocks<T>(c, c.s1, N, U, ldu);
        T* Xs = c.scratch<T>(Tr<T>::cx ? "evd_Zsz" : "evd_Zsd", (size_t)N * m);     // the solve is out of place: X = copy of Z
        EIG_HIP(hipMemcpy2DAsync(Xs, sizeof(T) * N, Z, sizeof(T) * ldz, sizeof(T) * N, m, hipMemcpyDeviceToDevice, c.s1));
        trsm_LUN<T>(c, c.s1, N, m, U, ldu, 0, Xs, N, Z, ldz, c.trsm_base);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_ztrsm_lun(int N, int m, const void* U_d, int ldu, void* Z_d, int ldz) {
    return trsm_entry<cplx>(N, m, (const cplx*)U_d, ldu, (cplx*)Z_d, ldz);
}
int eigsolve_dtrsm_lun(int N, int m, const double* U_d, int ldu, double* Z_d, int ldz) {
    return trsm_entry<double>(N, m, U_d, ldu, Z_d, ldz);
}

template <class T> static int mv_sweep_entry(int N, T* A, int lda, int nb, int reps, double* ms_total, long* nlaunch, double* bytes) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        if (nb <= 0) nb = c.trd_nb;
        T* W = c.scratch<T>("trd_W", (size_t)N * 64);
        double* e = c.scratch<double>("sweep_e", (size_t)N + 8);
        T* tau = c.scratch<T>("sweep_tau", (size_t)N + 8);
        EIG_HIP(hipMemsetAsync(W, 0, sizeof(T) * (size_t)N * 64, c.s1));
        long nl = 0; double by = 0;
        hetrd_mv_sweep<T>(c, c.s1, N, A, lda, W, nb, e, tau, &nl, &by);  // warm-up
        c.sync(c.s1);
        if (reps < 1) reps = 1;
        EIG_HIP(hipEventRecord(c.ev[0], c.s1));
        for (int r = 0; r < reps; ++r) hetrd_mv_sweep<T>(c, c.s1, N, A, lda, W, nb, e, tau, &nl, &by);
        EIG_HIP(hipEventRecord(c.ev[1], c.s1));
        c.sync(c.s1);
        float ms = 0.f;
        EIG_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        if (ms_total) *ms_total = (double)ms / reps;
        if (nlaunch) *nlaunch = nl;
        if (bytes) *bytes = by;
        return 0;
    });
}
int eigsolve_zhetrd_mv_sweep(int N, void* A_d, int lda, int nb, int reps, double* ms_total, long* nlaunch, double* algo_bytes) {
    return mv_sweep_entry<cplx>(N, (cplx*)A_d, lda, nb, reps, ms_total, nlaunch, algo_bytes);
}
int eigsolve_dsytrd_mv_sweep(int N, double* A_d, int lda, int nb, int reps, double* ms_total, long* nlaunch, double* algo_bytes) {
    return mv_sweep_entry<double>(N, A_d, lda, nb, reps, ms_total, nlaunch, algo_bytes);
}

template <class T> static int her2k_sweep_entry(int N, T* A, int lda, T* W, int nb, int reps, double* ms_total, long* nlaunch, double* flops) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        if (nb <= 0) nb = c.trd_nb;
        long nl = 0; double fl = 0;
        hetrd_her2k_sweep<T>(c, c.s1, N, A, lda, W, nb, &nl, &fl);  // warm-up
        c.sync(c.s1);
        if (reps < 1) reps = 1;
        EIG_HIP(hipEventRecord(c.ev[0], c.s1));
        for (int r = 0; r < reps; ++r) hetrd_her2k_sweep<T>(c, c.s1, N, A, lda, W, nb, &nl, &fl);
        EIG_HIP(hipEventRecord(c.ev[1], c.s1));
        c.sync(c.s1);
        float ms = 0.f;
        EIG_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        if (ms_total) *ms_total = (double)ms / reps;
        if (nlaunch) *nlaunch = nl;
        if (flops) *flops = fl;
        return 0;
    });
}
int eigsolve_zhetrd_her2k_sweep(int N, void* A_d, int lda, void* W_d, int nb, int reps, double* ms_total, long* nlaunch, double* flops) {
    return her2k_sweep_entry<cplx>(N, (cplx*)A_d, lda, (cplx*)W_d, nb, reps, ms_total, nlaunch, flops);
}
int eigsolve_dsytrd_her2k_sweep(int N, double* A_d, int lda, double* W_d, int nb, int reps, double* ms_total, long* nlaunch, double* flops) {
    return her2k_sweep_entry<double>(N, A_d, lda, W_d, nb, reps, ms_total, nlaunch, flops);
}

// ---- stage-level entry points of the back-transformation (zlarft_gpu / zlarfb_gpu, zheevd_gpu.F90:136-213) ----
template <class T> static int larft_entry(int N, const T* A, int lda, const T* tau, int nb2, T* T_out, int ldt_out) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        const int k = N - 1;
        if (k <= 0) return 0;
        bt_build_T<T>(c, c.s1, N, A, lda, tau, nb2);
        const int nb = bt_norm_nb(nb2, N);
        const int nblk = (k + nb - 1) / nb, ldt = nb;
        if (ldt_out < ldt) return -1;
        const T* Tall = c.scratch<T>("bt_T", 0);
        for (int b = 0; b < nblk; ++b)
            EIG_HIP(hipMemcpy2DAsync(T_out + (size_t)b * ldt_out * ldt_out, sizeof(T) * ldt_out, Tall + (size_t)b * ldt * ldt,
                                     sizeof(T) * ldt, sizeof(T) * ldt, ldt, hipMemcpyDeviceToDevice, c.s1));
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_zlarft(int N, const void* A_d, int lda, const void* tau_d, int nb, void* T_d, int ldt) {
    return larft_entry<cplx>(N, (const cplx*)A_d, lda, (const cplx*)tau_d, nb, (cplx*)T_d, ldt);
}
int eigsolve_dlarft(int N, const double* A_d, int lda, const double* tau_d, int nb, double* T_d, int ldt) {
    return larft_entry<double>(N, A_d, lda, tau_d, nb, T_d, ldt);
}
template <class T> static int unmtr_entry(int N, int m, const T* A, int lda, const T* tau, T* Z, int ldz, int nb2) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        bt_build_T<T>(c, c.s1, N, A, lda, tau, nb2);
        bt_apply<T>(c, c.s1, N, m, A, lda, Z, ldz, nb2);
        c.sync(c.s1);
        return 0;
    });
}
int eigsolve_zunmtr(int N, int m, const void* A_d, int lda, const void* tau_d, void* Z_d, int ldz, int nb) {
    return unmtr_entry<cplx>(N, m, (const cplx*)A_d, lda, (const cplx*)tau_d, (cplx*)Z_d, ldz, nb);
}
int eigsolve_dormtr(int N, int m, const double* A_d, int lda, const double* tau_d, double* Z_d, int ldz, int nb) {
    return unmtr_entry<double>(N, m, A_d, lda, tau_d, Z_d, ldz, nb);
}

// Device divide & conquer on (d,e): w_d[N] ascending, Q_d (N x N, ld ldq) eigenvectors.  Test / bench entry point.
int eigsolve_dstedc_device(int N, const double* d_d, const double* e_d, double* w_d, double* Q_d, int ldq, double* ms) {
    return guarded(nullptr, [&]() -> int {
        Ctx& c = ctx();
        double t0 = now_ms();
        double* Qs = nullptr;
        int lds_ = 0;
        int r = stedc_device(c, c.s1, N, d_d, e_d, w_d, &Qs, &lds_);
        if (r == 0 && Q_d)
            EIG_HIP(hipMemcpy2DAsync(Q_d, sizeof(double) * ldq, Qs, sizeof(double) * lds_, sizeof(double) * N, N, hipMemcpyDeviceToDevice, c.s1));
        c.sync(c.s1);
        if (ms) *ms = now_ms() - t0;
        return r;
    });
}
