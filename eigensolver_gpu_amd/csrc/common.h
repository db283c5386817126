// common.h -- shared types for the gfx950 eigensolver library (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace eig {

// ---- complex(8): interleaved (re,im), same layout as Fortran complex(8) -------------------
struct alignas(16) cplx {
    double x, y;
};

__host__ __device__ inline cplx mkc(double r, double i) { return cplx{r, i}; }
__host__ __device__ inline cplx operator+(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
__host__ __device__ inline cplx operator-(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
__host__ __device__ inline cplx operator-(cplx a) { return cplx{-a.x, -a.y}; }
__host__ __device__ inline cplx operator*(cplx a, cplx b) {
    return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__host__ __device__ inline cplx operator*(double s, cplx a) { return cplx{s * a.x, s * a.y}; }
__host__ __device__ inline cplx operator*(cplx a, double s) { return cplx{s * a.x, s * a.y}; }
__host__ __device__ inline cplx& operator+=(cplx& a, cplx b) { a.x += b.x; a.y += b.y; return a; }
__host__ __device__ inline cplx& operator-=(cplx& a, cplx b) { a.x -= b.x; a.y -= b.y; return a; }

// ---- scalar traits: one code path for real(8) and complex(8) -----------------------------
__host__ __device__ inline double conj_(double a) { return a; }
__host__ __device__ inline cplx conj_(cplx a) { return cplx{a.x, -a.y}; }
__host__ __device__ inline double real_(double a) { return a; }
__host__ __device__ inline double real_(cplx a) { return a.x; }
__host__ __device__ inline double imag_(double) { return 0.0; }
__host__ __device__ inline double imag_(cplx a) { return a.y; }
__host__ __device__ inline double abs2_(double a) { return a * a; }
__host__ __device__ inline double abs2_(cplx a) { return a.x * a.x + a.y * a.y; }
// a += b*c
__host__ __device__ inline void fma_(double& a, double b, double c) { a = fma(b, c, a); }
__host__ __device__ inline void fma_(cplx& a, cplx b, cplx c) {
    a.x = fma(b.x, c.x, a.x); a.x = fma(-b.y, c.y, a.x);
    a.y = fma(b.x, c.y, a.y); a.y = fma(b.y, c.x, a.y);
}
// a -= b*c
__host__ __device__ inline void fms_(double& a, double b, double c) { a = fma(-b, c, a); }
__host__ __device__ inline void fms_(cplx& a, cplx b, cplx c) {
    a.x = fma(-b.x, c.x, a.x); a.x = fma(b.y, c.y, a.x);
    a.y = fma(-b.x, c.y, a.y); a.y = fma(-b.y, c.x, a.y);
}
// a -= conj(b)*c
__host__ __device__ inline void fmsc_(double& a, double b, double c) { a = fma(-b, c, a); }
__host__ __device__ inline void fmsc_(cplx& a, cplx b, cplx c) {
    a.x = fma(-b.x, c.x, a.x); a.x = fma(-b.y, c.y, a.x);
    a.y = fma(-b.x, c.y, a.y); a.y = fma(b.y, c.x, a.y);
}
// a += conj(b)*c
__host__ __device__ inline void fmac_(double& a, double b, double c) { a = fma(b, c, a); }
__host__ __device__ inline void fmac_(cplx& a, cplx b, cplx c) {
    a.x = fma(b.x, c.x, a.x); a.x = fma(b.y, c.y, a.x);
    a.y = fma(b.x, c.y, a.y); a.y = fma(-b.y, c.x, a.y);
}

// c ? a : b as scalar selects.  A ternary on the struct type becomes control flow and LLVM then sinks a
// preceding load into the taken branch -- a conditional load with its own wait; this keeps loads unconditional.
__host__ __device__ inline double sel(bool c, double a, double b) { return c ? a : b; }
__host__ __device__ inline cplx sel(bool c, cplx a, cplx b) { return cplx{c ? a.x : b.x, c ? a.y : b.y}; }

template <class T> struct Tr;
template <> struct Tr<double> {
    static constexpr bool cx = false;
    __host__ __device__ static double make(double r, double) { return r; }
    __host__ __device__ static double zero() { return 0.0; }
    __host__ __device__ static double one() { return 1.0; }
    __host__ __device__ static double realpart(double a) { return a; }  // value with imag dropped
};
template <> struct Tr<cplx> {
    static constexpr bool cx = true;
    __host__ __device__ static cplx make(double r, double i) { return cplx{r, i}; }
    __host__ __device__ static cplx zero() { return cplx{0.0, 0.0}; }
    __host__ __device__ static cplx one() { return cplx{1.0, 0.0}; }
    __host__ __device__ static cplx realpart(cplx a) { return cplx{a.x, 0.0}; }
};

// ---- error handling ----------------------------------------------------------------------
#define EIG_HIP(call)                                                                                 \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            fprintf(stderr, "eigsolve: HIP error %s at %s:%d: %s\n", hipGetErrorName(e__), __FILE__,  \
                    __LINE__, #call);                                                                 \
            throw eig::HipFail{e__};                                                                  \
        }                                                                                             \
    } while (0)

struct HipFail {
    hipError_t err;
};

// ---- per-(thread, device) context (replaces module eigsolve_vars) ------------------------
using stedc_fn = void (*)(const char*, const int*, double*, double*, double*, const int*, double*, const int*,
                          int*, const int*, int*, size_t);

enum Phase { PH_POTRF = 0, PH_GST, PH_TRD, PH_STEDC, PH_BT, PH_TRSM, PH_D2H, PH_TOTAL, PH_COUNT };

// Default for the tridiagonal step: 1 = device divide & conquer (~100x faster than
// the host dstedc at N=4096); EIGSOLVE_TRIDIAG=host / eigsolve_set_option("tridiag", 0) restores the
// reference behaviour (host LAPACK).
constexpr int kTridiagDefault = 1;
// Reflectors per block in the back-transformation: 64 (the reference's larfb width, zheevd_gpu.F90:113-131), or 128 / 256 / 512 =
// 64-blocks whose T factors are merged pairwise (bt_build_T in evd.hip): K of the rank-k updates grows with the block.
constexpr int kBtNbDefault = 256;
inline int norm_bt_nb(int v) { return v <= 0 ? kBtNbDefault : (v >= 512 ? 512 : (v >= 256 ? 256 : (v >= 128 ? 128 : 64))); }
constexpr int kOverlapDefault = 3;   // (bit 2, the look-ahead factorization, is built and bit-identical but gains nothing: profiles/r06_experiments.txt 7)
constexpr int kPotrfDefault = 2;
// Largest library-side copy of the standard problem's eigenvectors (N x m, MiB) the generalized drivers allocate; beyond it (or when
// that much device memory is not available) the vectors are formed in the caller's Z and the final triangular solve runs in column
// chunks through a smaller block (hegvdx_core in evd.hip)
constexpr int kZsCapMbDefault = 4096;
// Panel width of the tridiagonalization: 32, the reference's own (zheevd_gpu.F90:63).  Rounds 1-3 used 64 (fewer, deeper rank-2nb
// updates); with the round-4 mat-vec grid the narrower panel wins everywhere -- the per-column row kernel carries half the pending
// columns: C3 trd 67.6 -> 66.6 ms, batch 17.66 -> 17.90 problems/s, C5 105.3 -> 109.9, C2 188.5 -> 194.0 (profiles/r04_experiments.txt 7).
constexpr int kTrdNbDefault = 32;
// Reduction to standard form: 0 symmetric recursion to 64x64 blocks, 1 two full triangular solves, 2 hybrid (symmetric
// algorithm while the diagonal blocks are larger than gst_thr, two solves below); see hegst_upper in blas3.hip
constexpr int kGstModeDefault = 2;
constexpr int kGstThrDefault = 1024;
// Order of the inverted diagonal blocks the triangular solves outside potrf stop at: 64 (as produced by the
// factorization) or 256 (merged after it, build_inv256 in blas3.hip)
constexpr int kTrsmBaseDefault = 256;
inline int norm_trsm_base(int v) { return v >= 1024 ? 1024 : (v >= 512 ? 512 : (v >= 256 ? 256 : 64)); }
// MFMA engine, complex 64 x 64 tiles: 0 = K-slabs staged through registers (gemm_fast_kernel), 1 = by LDS-DMA (gemm_dma_kernel), 2 = LDS-DMA,
// persistent form, 3 = LDS-DMA where a work item has at least kGemmDmaMinK of K (measured: +8-9 % at K >= 1024, +4-6 % at 128-256, -0..3 %
// at K = 64, profiles/r06_experiments.txt section 2); results are bit-identical in every form
constexpr int kGemmDmaDefault = 3;
constexpr int kGemmDmaMinK = 96;
// MFMA engine, complex products with fewer than one 64 x 64 tile per CU: 0 = 32 x 32 tiles on four-wave workgroups (gemm_fast_kernel), 1 =
// 32 x 32 tiles on whole-CU workgroups of 16 waves with K split inside the workgroup where the launch has at most one tile per CU
// (gemm_wide_kernel<.., 4>), 2 = that, and 8-wave workgroups (two per CU) up to two tiles per CU (dispatch_gemm_now in blas3.hip; the
// choice is a function of the product's shape only).  Off: 0-10 % faster on the bare shapes (256 x 1024 x 256: 18.9 -> 17.1 us), within
// noise in the C3 solve (gst + back-transformation + trsm 12.8 -> 12.6 ms), 1-2 % SLOWER in batches (C3 18.0 -> 17.8, C5 108.8 -> 106.4
// problems/s: a 1024-thread workgroup with 128 KB of LDS needs an empty CU, which the other launch chains rarely leave), and the K-split
// changes the summation order (profiles/r06_experiments.txt section 8)
constexpr int kGemmWideDefault = 0;
// MFMA engine, complex 64 x 64 tiles: work items with at most this much of K run on the lean LDS-DMA form (K-slabs of 8, C fetched in the
// epilogue, 116 VGPRs and 32 KB of LDS: four workgroups per CU, gemm_dma_kernel<., 8, 3>) when the launch has at least four tiles per CU;
// 0 = never.  Bit-identical to the other forms.  Measured (profiles/r06_experiments.txt section 9): her2k n = 4096 k = 32 / 64 +5 / +6 %,
// n = 3000 +7 %, the factorization's rank-128 update of order 4032 +7 %; in the C3 solve -0.3 ms, C4 -2.5 ms, batch rates unchanged.
constexpr int kGemmLeanDefault = 128;
constexpr int kMvDmaDefault = 0;   // (set from the measurements of round 6, profiles/r06_experiments.txt)
// Upper bound of the "hemv_blocks" knob (workgroups of the panel mat-vec kernel; sizes the per-workgroup partial array)
constexpr int kHemvBlocksMax = 8192;

struct GroupRecorder;   // blas3.h

struct Ctx {
    int dev = -1;
    hipStream_t s1 = nullptr;  // compute stream: LEASED from the library's stream pool for the duration of an API call
                               // (StreamLease, context.cpp); wait with sync(), never hipStreamSynchronize
    int lease_depth = 0;
    hipStream_t s2 = nullptr;  // private overlap stream, created on first use (second_stream(); "overlap" options only)
    hipEvent_t evSync = nullptr;
    hipEvent_t ev[2 * PH_COUNT] = {};
    hipEvent_t evA = nullptr, evB = nullptr;
    hipStream_t s3 = nullptr;  // third stream: the trailing updates of the look-ahead factorization ("overlap" bit 2), leased like s2
    hipEvent_t evLA[2] = {};   // look-ahead factorization: block rows of a pair done (chain -> updates), update done (updates -> chain)
    hipEvent_t evStage[16] = {};   // block-row stages of the factorization (potrf || hegst pipeline), created on first use
    std::map<std::string, std::pair<void*, size_t>> slots;  // named grow-only device scratch
    // Lockstep groups of a batch call (hegvdx_batch_core in evd.hip): while `rec` is set the BLAS-3 drivers append their launches to it
    // instead of queueing them (blas3.h: GroupRecorder), and while grp_q > 0 every scratch slot name gets the suffix "#g<grp_q>" --
    // each problem of a group works in slots of its own, so the recorded sequences of the group can be replayed position by
    // position (one launch carrying all problems where the position is a product on the MFMA engine).  slot_gen counts slot
    // (re)allocations: a recording during which it moved holds stale pointers and is repeated.
    GroupRecorder* rec = nullptr;
    int grp_q = 0;
    unsigned long slot_gen = 0;
    std::map<std::string, std::pair<void*, size_t>> hslots; // named grow-only pinned host scratch
    int* d_info = nullptr;   // device int (replaces devInfo_d)
    int* h_info = nullptr;   // pinned host mirror
    double phase_ms[PH_COUNT] = {};
    int n_cu = 256;
    // tunables
    int trd_nb = kTrdNbDefault;
    int bt_nb = kBtNbDefault;
    int hemv_blocks = 0;  // 0 = auto
    int gemm_dma = kGemmDmaDefault;   // staging path of the complex 64 x 64 tiles (see kGemmDmaDefault)
    int gemm_lean = kGemmLeanDefault; // largest K of a work item served by the lean LDS-DMA form (see kGemmLeanDefault)
    int gemm_wide = kGemmWideDefault; // whole-CU workgroups for the complex 32 x 32 tiles (see kGemmWideDefault)
    int mv_dma = kMvDmaDefault;   // smallest trailing order for which the panel mat-vec streams its tiles through the LDS-DMA ring
                             // (panel_mv_kernel<T, NB, true> in trd.hip); 0 = never (the register-staged form everywhere)
    int use_graph = 0;       // replay the tridiagonalization launch sequence as a hipGraph (EIGSOLVE_GRAPH=1 / option "graph");
                             // measured neutral on MI355X/ROCm 7.2 (dispatch latency is device-side), so off by default
    int in_batch = 0;        // set while this context solves one problem of a batch call with several problems in flight
    int overlap = kOverlapDefault;   // bit 0: hegst on a second stream beside the factorization, released stage by stage
                             // (potrf_hegst_pipelined_begin); bit 1: the larft T factors on the second stream while the tridiagonal
                             // eigenproblem is solved (zheevd_gpu.F90:125 overlaps the same work).  Both only for a solve that has the
                             // device to itself -- a best-effort test (streams_in_use() sampled once per phase, not a lock: two caller
                             // threads that start together may both decide they are alone; results are identical either way)
    struct GraphEntry {
        hipGraphExec_t exec = nullptr;
        hipGraph_t graph = nullptr;
        const void* ptrs[16] = {};
    };
    std::map<std::string, GraphEntry> graphs;
    int trsm_base = kTrsmBaseDefault;
    int potrf_mode = kPotrfDefault;   // 2: right-looking block rows in pairs (chol_row2_kernel: elimination blocked by 16 on MFMA; rank-128
                             // trailing updates), 1: the round-2 form (chol_row_kernel + a rank-64 update per block row), 0: recursive (potrf_rec)
    int gst_mode = kGstModeDefault;
    int gst_thr = kGstThrDefault;
    int real_il_reference = 0;  // 1: real path copies eigenvectors 1..m whatever il is, like dsyevd_gpu.F90:108
    int tridiag_device = kTridiagDefault;  // 0: host LAPACK dstedc (reference behaviour), 1: device divide & conquer
    int trd_finish = -1;     // order at which the tridiagonalization hands the rest of the matrix to the one-workgroup LDS kernel
                             // (hetd2_wide_kernel in trd.hip): -1 = the largest order its LDS holds, 32 = the reference's cut-over
                             // (zhetrd_gpu.F90:84-87)
    int tile_map = 1;        // 1: XCD-aware super-tile map of the MFMA engine's workgroups (tile_of in blas3.hip), 0: plain grids
    int batch_zip = 3;       // lockstep groups: the BLAS-3 phases of a group as zipped launch sequences (one launch per product position for the
                             // whole group; hegvdx_batch_core in evd.hip), 0 = problem after problem (round 3-5 form)
    int batch_fuse = -1;     // problems per launch chain of a batch call that share the per-column launches of the tridiagonalization
                             // (lockstep groups of <= 4): -1 = automatic (groups while a matrix is <= 96 MiB), 1 = none
    int batch_workers = -1;  // problems in flight inside one eigsolve_?hegvdx_batch call (internal worker threads, one context +
                             // stream each); 0 = the lockstep form on the caller's own context (hegvdx_batch_core in evd.hip);
                             // -1 = automatic, see auto_batch_workers()
    int trace_marks = 0;     // EIGSOLVE_TRACE_MARKS=1: marker kernels at the phase boundaries (profiling aid, see evd.hip)
    int zs_cap_mb = kZsCapMbDefault;

    template <class T> T* scratch(const char* name, size_t count) {
        return reinterpret_cast<T*>(scratch_bytes(name, count * sizeof(T)));
    }
    void* scratch_bytes(const char* name, size_t bytes);
    void* try_scratch_bytes(const char* name, size_t bytes);   // nullptr instead of an exception when the device is out of memory
    void* host_scratch_bytes(const char* name, size_t bytes);
    void release();
    // Wait until everything THIS context has enqueued on `st` so far is done.  On a shared stream a
    // hipStreamSynchronize would also wait for whatever the other contexts enqueue meanwhile.
    void sync(hipStream_t st);
    void sync() { sync(s1); }
    hipStream_t second_stream();
    hipStream_t third_stream();
    int own_streams() const { return 1 + (s2 ? 1 : 0) + (s3 ? 1 : 0); }   // streams this context holds from the pool
    void drop_graphs();   // forget captured launch sequences (an option they bake in has changed)
};

Ctx& ctx();  // lazily created for the current device + calling thread

// Every public entry point holds one of these while it runs: the context's compute stream comes from a process-wide pool and
// goes back when the call returns (after a sync), so the library never owns more streams than it has calls in flight.
struct StreamLease {
    Ctx& c;
    explicit StreamLease(Ctx& ctx_);
    ~StreamLease();
    StreamLease(const StreamLease&) = delete;
    StreamLease& operator=(const StreamLease&) = delete;
};
void copy_options(Ctx& dst, const Ctx& src);   // all tunables of src (eigsolve_set_option / environment) into dst
bool apply_option(Ctx& c, const std::string& name, int value);   // eigsolve_set_option / EIGSOLVE_<NAME>; false: unknown name
int streams_in_use(int dev);   // API calls in flight on the device (leased compute streams)
int auto_batch_workers();   // 4 when the process has asked for >= 5 hardware queues (GPU_MAX_HW_QUEUES), else 3

// The library's own worker threads (context.cpp): runs fn(0) ... fn(ntasks-1) on `nworkers` of them, device `dev` current,
// each worker taking the next task as it finishes one; returns when all are done.  A worker keeps its per-thread context
// (stream, scratch) from call to call.
void batch_run(int dev, int nworkers, int ntasks, const std::function<void(int)>& fn);
void batch_workers_finalize(int dev);          // every worker releases its context for `dev`
void stream_pool_finalize(int dev);            // idle pool streams of `dev` are destroyed

// host LAPACK plumbing (context.cpp)
stedc_fn get_dstedc();
int load_lapack(const char* path);
void set_host_threads(int n);

// roctx (context.cpp)
void range_push(const char* name);
void range_pop();
void phase_range_push(const char* name);   // no device sync (inside the drivers)
void phase_range_pop();
struct PhaseRange {
    explicit PhaseRange(const char* n) { phase_range_push(n); }
    ~PhaseRange() { phase_range_pop(); }
};

}  // namespace eig
