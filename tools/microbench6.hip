// microbench6.hip -- which TILE SHAPE streams an HBM-resident column-major matrix fastest?  (the mat-vec sweep
// of the tridiagonalization reaches 6.2 TB/s on the Infinity-Cache-resident C3 triangle but 4.9-5.2 TB/s at N = 8192.)
// One workgroup per CU (256 threads = 4 waves), tiles dealt round-robin, 16-byte elements, 16 loads of 16 B in flight per lane --
// the load structure of panel_mv_kernel -- with the tile cut as
//     64 x 64   : wave w = columns 16w..16w+15, lane = row            (64 column segments of 1 KB per tile, today's shape)
//    128 x 32   : wave w = rows 64(w&1).., columns 16(w>>1)..           (32 segments of 2 KB)
//    256 x 16   : wave w = rows 64w.., all 16 columns                   (16 segments of 4 KB)
//    512 x 8    : wave w = rows 128w + {lane, lane+64}, 8 columns       (8 segments of 8 KB)
// over the upper triangle (tile-level) of an n x n matrix with leading dimension lda.  Output: GB/s per shape.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench6.hip -o tools/_build/microbench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// tile (ti, tj) of TR x TC elements, ti*TR <= (tj+1)*TC - 1  (tiles that touch the upper triangle)
template <int TR, int TC>
__global__ void __launch_bounds__(256) stream_kernel(const d2* A, long lda, int n, const int2* tiles, int ntiles, d2* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d2 acc = {0.0, 0.0};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int2 tt = tiles[t];
        if (tt.x < 0) continue;       // (hole of a run-ordered list, see reorder())
        const long r0 = (long)tt.x * TR, c0 = (long)tt.y * TC;
        d2 v[16];
        // wave layout inside the tile
        constexpr int WR = TR >= 256 ? 4 : (TR >= 128 ? 2 : 1);      // waves along the rows
        constexpr int WC = 4 / WR;                                   // waves along the columns
        constexpr int RPW = TR / WR;                                 // rows per wave (64 or 128)
        constexpr int CPW = TC / WC;                                 // columns per wave
        constexpr int RI = RPW / 64;                                 // row chunks of 64 per wave
        static_assert(RI * CPW == 16, "16 loads per lane");
        const long rb = r0 + (wave % WR) * RPW, cb = c0 + (wave / WR) * CPW;
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int q = 0; q < RI; ++q) {
                long r = rb + lane + 64 * q, c = cb + j;
                r = r < n ? r : n - 1; c = c < n ? c : n - 1;
                v[j * RI + q] = A[r + c * lda];
            }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j];
    }
    if (acc.x == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = acc;     // (keeps the loads alive)
}

// The same stream with panel_mv_kernel's per-tile rhythm (64 x 64 tiles): every wave spends `work` dependent multiply-adds per tile
// on the data (the products + the transpose-reduce of the real kernel: ~1500 cycles) and the workgroup meets at a barrier per tile.
//   MODE 1: the next tile's loads are issued AFTER the work on the current one (the round-4 kernel): nothing in flight meanwhile.
//   MODE 2: two tiles in flight (double-buffered registers, loop unrolled by two): the loads of tile k+2 go out right after tile k
//           has been worked on, tile k+1 is in flight throughout.
template <int MODE>
__global__ void __launch_bounds__(320) rhythm_kernel(const d2* A, long lda, int n, const int2* tiles, int ntiles, d2* out, int work) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double sh[5][64];
    d2 acc = {0.0, 0.0};
    d2 v[16];
    int t = blockIdx.x;
    // tile tt = J(J+1)/2 + I of the upper triangle, decoded arithmetically once per tile; returns the lane's base pointer
    auto base_of = [&](int tt) -> const d2* {
        tt = tt < ntiles ? tt : ntiles - 1;
        int J = (int)((__builtin_sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
        while ((J + 1) * (J + 2) / 2 <= tt) ++J;
        while (J * (J + 1) / 2 > tt) --J;
        const int I = tt - J * (J + 1) / 2;
        long r = (long)I * 64 + lane, c = (long)J * 64 + (wave & 3) * 16;
        r = r < n ? r : n - 1; c = c + 15 < n ? c : n - 16;
        return A + r + c * lda;
    };
    const int G = gridDim.x;
    auto work_on = [&](const d2 (&w)[16]) {
        d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 8; ++j) s0 += w[j];
#pragma unroll
        for (int j = 8; j < 16; ++j) s1 += w[j];
        double x = s0.x + s1.y, y = s0.y + s1.x;
        for (int k = 0; k < work; ++k) { x = fma(x, 1.0000001, y); y = fma(y, 0.9999999, x); }     // dependent chain: ~22 cycles per iteration
        acc.x += x; acc.y += y;
        sh[wave][lane] = x;
    };
    auto finish = [&]() {
        __syncthreads();
        if (wave == 4) acc.x += sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane];
    };
    if (MODE == 1) {
        const d2* bp = base_of(t);
        if (wave < 4 && t < ntiles) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = bp[j * lda];
        }
        for (; t < ntiles; t += G) {
            if (wave < 4) {
                const bool more = t + G < ntiles;
                const d2* np = base_of(t + G);
                work_on(v);
                if (more) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = np[j * lda];
                }
            }
            finish();
        }
    } else {
        // two tiles in flight: buffer u holds tile t + G while tile t (buffer v) is worked on; the loop is unrolled by two so that
        // no register copies (and the waits they imply) separate the iterations
        d2 u[16];
        if (wave < 4) {
            const d2* b0 = base_of(t);
            const d2* b1 = base_of(t + G);
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = b0[j * lda];
#pragma unroll
            for (int j = 0; j < 16; ++j) u[j] = b1[j * lda];
        }
        for (; t < ntiles; t += 2 * G) {
            if (wave < 4) {
                work_on(v);
                const d2* np = base_of(t + 2 * G);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = np[j * lda];
            }
            finish();
            if (t + G < ntiles) {
                if (wave < 4) {
                    work_on(u);
                    const d2* np = base_of(t + 3 * G);
#pragma unroll
                    for (int j = 0; j < 16; ++j) u[j] = np[j * lda];
                }
                finish();
            }
        }
    }
    if (acc.x == 12345.678) out[blockIdx.x * 320 + threadIdx.x] = acc;
}

template <int MODE> static double run_rhythm(const d2* A, long lda, int n, d2* out, int reps, int grid, int work) {
    std::vector<int2> tl;
    const int nt = (n + 63) / 64;
    for (int j = 0; j < nt; ++j)
        for (int i = 0; i <= j; ++i) tl.push_back(make_int2(i, j));
    int2* d_t;
    CK(hipMalloc(&d_t, tl.size() * sizeof(int2)));
    CK(hipMemcpy(d_t, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rhythm_kernel<MODE><<<grid, 320>>>(A, lda, n, d_t, (int)tl.size(), out, work);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) rhythm_kernel<MODE><<<grid, 320>>>(A, lda, n, d_t, (int)tl.size(), out, work);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(d_t));
    return (double)tl.size() * 64 * 64 * 16.0 * reps / (ms * 1e-3) * 1e-9;
}

// Tile ORDER experiment (round 5, section 10 of r05_experiments.txt): the kernels deal list entry t to workgroup t mod G.
//   order 0: the list as built (column-major over the triangle): at any moment the G workgroups read G consecutive tiles -- a compact
//            window of one or two tile columns;
//   order 1: workgroup hb reads a RUN of L consecutive tiles of the column-major order (entry k G + hb = tile hb L + k): every
//            workgroup streams down its own tile column, the G workgroups are spread over the whole matrix;
//   order 2: the same with the tiles first put in strips of 4 tile columns, row by row (the order of trd.hip's SlotMap).
static int g_order = 0;
static std::vector<int2> reorder(const std::vector<int2>& cm, int G) {
    if (g_order == 0) return cm;
    std::vector<int2> src = cm;
    if (g_order == 2) {
        int nt = 0;
        for (auto& t : cm) nt = t.y + 1 > nt ? t.y + 1 : nt;
        src.clear();
        const int w = 4, ws0 = nt - w * ((nt - 1) / w);
        for (int J0 = 0, wd = ws0; J0 < nt; J0 += wd, wd = w)
            for (int I = 0; I < J0 + wd; ++I)
                for (int J = (I > J0 ? I : J0); J < J0 + wd; ++J) src.push_back(make_int2(I, J));
    }
    const int M = (int)src.size(), L = (M + G - 1) / G;
    std::vector<int2> out((size_t)L * G, make_int2(-1, -1));
    for (int hb = 0; hb < G; ++hb)
        for (int k = 0; k < L && hb * L + k < M; ++k) out[(size_t)k * G + hb] = src[hb * L + k];
    return out;
}

template <int TR, int TC> static double run(const d2* A, long lda, int n, d2* out, int reps, int grid) {
    std::vector<int2> tl;
    const int ntr = (n + TR - 1) / TR, ntc = (n + TC - 1) / TC;
    for (int j = 0; j < ntc; ++j)
        for (int i = 0; i < ntr; ++i)
            if ((long)i * TR <= (long)(j + 1) * TC - 1) tl.push_back(make_int2(i, j));
    const size_t nreal = tl.size();
    if (TR == 64 && TC == 64) tl = reorder(tl, grid);
    if (TR == 256 && TC == 16 && g_order == 3) {
        // balanced ROW runs: the tiles in row-major order (row block i, then j), workgroup hb takes [hb M / G, (hb + 1) M / G)
        std::vector<int2> rm;
        for (int i = 0; i < ntr; ++i)
            for (int j = 0; j < ntc; ++j)
                if ((long)i * TR <= (long)(j + 1) * TC - 1) rm.push_back(make_int2(i, j));
        const long M = (long)rm.size();
        const int L = (int)((M + grid - 1) / grid);
        std::vector<int2> out((size_t)L * grid, make_int2(-1, -1));
        for (int hb = 0; hb < grid; ++hb) {
            const long a = hb * M / grid, b = (hb + 1) * M / grid;
            for (long k = a; k < b; ++k) out[(size_t)(k - a) * grid + hb] = rm[k];
        }
        tl = out;
    }
    int2* d_t;
    CK(hipMalloc(&d_t, tl.size() * sizeof(int2)));
    CK(hipMemcpy(d_t, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    stream_kernel<TR, TC><<<grid, 256>>>(A, lda, n, d_t, (int)tl.size(), out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) stream_kernel<TR, TC><<<grid, 256>>>(A, lda, n, d_t, (int)tl.size(), out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(d_t));
    const double bytes = (double)nreal * TR * TC * 16.0;
    return bytes * reps / (ms * 1e-3) * 1e-9;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    const long pad = argc > 2 ? atol(argv[2]) : 0;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const int grid = argc > 4 ? atoi(argv[4]) : 256;
    const long lda = n + pad;
    const int orders = argc > 5 ? atoi(argv[5]) : 0;     // 1: also time the 64 x 64 stream in the run orders
    d2 *A, *out;
    CK(hipMalloc(&A, (size_t)lda * n * sizeof(d2)));
    CK(hipMemset(A, 0, (size_t)lda * n * sizeof(d2)));
    CK(hipMalloc(&out, 1 << 22));
    printf("n %d lda %ld (column stride %ld KB) grid %d: ", n, lda, lda * 16 / 1024, grid);
    printf(" 64x64 %.0f", run<64, 64>(A, lda, n, out, reps, grid));
    printf("  128x32 %.0f", run<128, 32>(A, lda, n, out, reps, grid));
    printf("  256x16 %.0f", run<256, 16>(A, lda, n, out, reps, grid));
    printf("  512x8 %.0f GB/s\n", run<512, 8>(A, lda, n, out, reps, grid));
    if (orders) {
        printf("   64x64 stream by tile order: ");
        for (int rep = 0; rep < 2; ++rep)
            for (g_order = 0; g_order < 3; ++g_order)
                printf(" %s %.0f", g_order == 0 ? "column-major round-robin" : (g_order == 1 ? "column runs" : "strip-4 runs"), run<64, 64>(A, lda, n, out, reps, grid));
        printf(" GB/s\n   256x16 stream by tile order: ");
        for (int rep = 0; rep < 2; ++rep) {
            g_order = 0; printf(" column-major round-robin %.0f", run<256, 16>(A, lda, n, out, reps, grid));
            g_order = 3; printf(" balanced row runs %.0f", run<256, 16>(A, lda, n, out, reps, grid));
        }
        g_order = 0;
        printf(" GB/s\n");
    }
    for (int work : {0, 30, 60, 90})
        printf("   rhythm (64x64, barrier per tile, %4d dependent fma pairs of work per tile):  issue-after-work %.0f   two tiles in flight %.0f GB/s\n",
               work, run_rhythm<1>(A, lda, n, out, reps, grid, work), run_rhythm<2>(A, lda, n, out, reps, grid, work));
    return 0;
}
