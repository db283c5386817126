// blas3.h -- fp64 MFMA tile engine: one kernel family behind every BLAS-3 call site of the
// reference: gemm (NN/CN/NC), her2k/syr2k, herk/syrk, trmm, trsm (via
// inverted diagonal blocks + gemm), potrf, hegst.  Internal header.
#pragma once
#include <climits>
#include <memory>

#include "common.h"

namespace eig {

// Masks are expressed in the coordinates (sr, sc) of the STORED matrix element.
enum Mask {
    M_NONE = 0,
    M_UPPER = 1,     // keep sr <= sc
    M_SUPPER = 2,    // keep sr <  sc
    M_LOWER = 3,     // keep sr >= sc
    M_UNITTRAP = 4,  // d = sr - sc - moff : d == 0 -> 1, d > 0 -> 0   (larfb's V block)
};

// An operand is a logical (nidx x K) view X(idx,k):
//   trans == 0 : X(idx,k) = p[idx + k*ld]   (idx contiguous in memory)
//   trans == 1 : X(idx,k) = p[k + idx*ld]   (k contiguous in memory)
// The A operand of C = A*B uses idx = row of C; the B operand is given through its
// transpose view Bt(j,k) = B(k,j), idx = column of C.  k >= k1 switches to (p2, ld2) at
// k - k1 (K-concatenation, used by her2k so C is touched once).
template <class T> struct Operand {
    const T* p = nullptr;
    int ld = 0;
    int trans = 0;
    int conj = 0;
    int mask = M_NONE;
    int moff = 0;
    const T* p2 = nullptr;
    int ld2 = 0;
    int k1 = INT_MAX;
};

template <class T> Operand<T> op_plain(const T* p, int ld, int trans, int conj) {
    Operand<T> o;
    o.p = p; o.ld = ld; o.trans = trans; o.conj = conj;
    return o;
}
// BLAS-style helpers.  A operand (M x K) from op in {'N','T','C'} of a column-major matrix.
template <class T> Operand<T> opA(char t, const T* p, int ld) {
    return op_plain(p, ld, (t == 'N' || t == 'n') ? 0 : 1, (t == 'C' || t == 'c') ? 1 : 0);
}
// B operand (K x N) -> transpose view (N x K).
template <class T> Operand<T> opB(char t, const T* p, int ld) {
    return op_plain(p, ld, (t == 'N' || t == 'n') ? 1 : 0, (t == 'C' || t == 'c') ? 1 : 0);
}

struct Epi {
    int uplo = 0;       // 0 full, 1 write only i<=j, 2 write only i>=j
    int herm_diag = 0;  // force Im C(i,i) = 0
    void* aux = nullptr;  // != nullptr: alpha * A * B (the product before beta * C is added) is ALSO stored here (type T, ld ldaux)
    int ldaux = 0;
};

// Strided batch descriptor: entry zb = 0..count-1 uses A.p + zb*sA, Bt.p + zb*sB, C + zb*sC, mask offsets + zb*dMoff*,
// K + zb*dK, and M, N, K clipped to cap* - zb*dcap (INT_MAX = no clipping).  `splits` is filled in by gemm_batched.
struct GemmBatch {
    int count = 0;
    int splits = 1;
    long sA = 0, sB = 0, sC = 0;
    int dMoffA = 0, dMoffB = 0, dK = 0;
    int capM = INT_MAX, capN = INT_MAX, capK = INT_MAX, dcap = 0;
};

// ---- deferred launches: the lockstep groups of a batch call (evd.hip, hegvdx_batch_core) --------------------------------------
// The problems of a group run the same sequence of launches with different pointers.  While c.rec is set, every launch the
// BLAS-3 drivers would queue is appended to the recorder instead: products on the MFMA engine as their argument block (kind 1),
// everything else (small kernels, memsets) as a closure (kind 0).  replay_group() then walks the sequences of the group position
// by position: a product position becomes ONE launch that carries all problems (pointer table in the kernel arguments,
// blockIdx.z = problem x K-split: same tiles, same K order per problem as the problem's own launch -> bit-identical results),
// any other position one launch per problem.  Sequences that do not line up are replayed one problem after the other.
struct LaunchRec {
    int kind = 0;
    std::function<void(hipStream_t)> run;                                          // this launch alone
    std::function<bool(hipStream_t, const LaunchRec* const*, int)> run_group;       // kind 1: with n - 1 peers as one launch (false: not compatible)
    std::shared_ptr<const void> gemm_args;                                          // kind 1: the engine's argument block
    int gemm_type = 0;                                                              // 1 real, 2 complex
};
struct GroupRecorder {
    std::vector<LaunchRec> seq;
};
void replay_group(hipStream_t st, GroupRecorder* recs, int n);

// kernel launch / memset that honours the recorder
template <class... KArgs, class... Args>
inline void klaunch(Ctx& c, hipStream_t st, void (*kern)(KArgs...), dim3 grid, dim3 block, Args... args) {
    if (c.rec) {
        LaunchRec r;
        r.run = [=](hipStream_t s) { hipLaunchKernelGGL(kern, grid, block, 0, s, static_cast<KArgs>(args)...); };
        c.rec->seq.push_back(std::move(r));
    } else {
        hipLaunchKernelGGL(kern, grid, block, 0, st, static_cast<KArgs>(args)...);
    }
}
inline void kmemset(Ctx& c, hipStream_t st, void* p, int value, size_t bytes) {
    if (c.rec) {
        LaunchRec r;
        r.run = [=](hipStream_t s) { EIG_HIP(hipMemsetAsync(p, value, bytes, s)); };
        c.rec->seq.push_back(std::move(r));
    } else {
        EIG_HIP(hipMemsetAsync(p, value, bytes, st));
    }
}

// C(MxN) = alpha * A * B + beta * C.
template <class T>
void gemm(Ctx& c, hipStream_t st, int M, int N, int K, T alpha, const Operand<T>& A, const Operand<T>& Bt, T beta,
          T* C, int ldc, Epi epi = Epi());

// Split-K variant for skinny outputs (larft's V^H V, larfb's C^H V): partial products are
// written to scratch and summed in a fixed order (deterministic), then alpha/beta applied.
template <class T>
void gemm_splitk(Ctx& c, hipStream_t st, int M, int N, int K, T alpha, const Operand<T>& A, const Operand<T>& Bt,
                 T beta, T* C, int ldc, int kchunk, Epi epi = Epi());

// bt.count products of one shape in one launch; kchunk > 0 additionally splits K (partials summed in a fixed order).
template <class T>
void gemm_batched(Ctx& c, hipStream_t st, int M, int N, int K, T alpha, const Operand<T>& A, const Operand<T>& Bt, T beta,
                  T* C, int ldc, Epi epi, GemmBatch bt, int kchunk = 0);

// C(upper) -= V W^H + W V^H  (trans='N'), V,W n x k.
template <class T> void her2k_un(Ctx& c, hipStream_t st, int n, int k, const T* V, int ldv, const T* W, int ldw, T* C, int ldc);

// Blocked upper Cholesky of B (N x N, ld ldb); also leaves the inverses of the 64x64
// diagonal blocks of U in scratch slot "invU" (used by every trsm).  Device info flag in
// c.d_info[0] (0 = ok, else 1-based index of first bad pivot).
template <class T> void potrf_upper(Ctx& c, hipStream_t st, int N, T* B, int ldb);
// Rebuilds slot "invU" from an existing factor.
template <class T> void build_invU(Ctx& c, hipStream_t st, int N, const T* U, int ldu);
// block-row factorizations of nprob <= 4 matrices in lockstep (one block-row launch per block row for all of them); no inverse
// blocks are built; info of problem q in c.d_info[4 + q]
template <class T> void potrf_upper_group(Ctx& c, hipStream_t st, int N, int nprob, T* const* B, int ldb);

// Triangular solves with the Cholesky factor (block offsets are multiples of 64 from U(0,0)), OUT OF PLACE: the result goes to
// Y, X is used as workspace and destroyed (its blocks receive the updates of the substitution).  No staging copies: the
// base-case products read X and write Y, the updates read Y and modify X.  X and Y must not overlap.
// (rows_done, optional: called on the host right after the launches that make rows [r0, r0 + nr) of Y final have been queued on st,
//  for the row blocks of recursion depth `hook_depth` -- lets the caller start consuming finished row blocks, e.g. the host copy)
template <class T> void trsm_LUN(Ctx& c, hipStream_t st, int n, int m, const T* U, int ldu, int k0, T* X, int ldx, T* Y, int ldy, int base = 64,
                                 const std::function<void(int, int)>* rows_done = nullptr, int hook_depth = 0, int row0 = 0);  // Y = U^-1 X
template <class T> void trsm_LUC(Ctx& c, hipStream_t st, int n, int m, const T* U, int ldu, int k0, T* X, int ldx, T* Y, int ldy, int base = 64);  // Y = U^-H X
template <class T> void trsm_RUN(Ctx& c, hipStream_t st, int n, int m, const T* U, int ldu, int k0, T* X, int ldx, T* Y, int ldy, int base = 64);  // Y(mxn) = X U^-1
// merged 256x256 inverse diagonal blocks for base = 256 (needs the 64-block inverses of potrf_upper / build_invU)
template <class T> void build_inv256(Ctx& c, hipStream_t st, int N, const T* U, int ldu);
// all merged inverse blocks the "trsm_base" option asks for (256, 512, 1024), from the 64-block inverses
template <class T> void build_inv_blocks(Ctx& c, hipStream_t st, int N, const T* U, int ldu);

// A <- U^-H A U^-1 (upper triangle only is read/written).
template <class T> void hegst_upper(Ctx& c, hipStream_t st, int N, T* A, int lda, const T* U, int ldu);

// potrf(B) || hegst(A, U) pipeline (option "overlap" bit 0; see blas3.hip): *_begin enqueues the whole factorization on
// c.s1 and the steps of hegst's top level that only need the leading half of the factor on the second stream; the caller
// synchronises c.s1 for potrf's info, then calls *_finish (join + the two steps that need U22).  Bit-identical to
// potrf_upper + hegst_upper.  On a non-positive-definite B the upper triangle of A is unspecified (the caller must
// synchronise the second stream before returning).
template <class T> bool pipeline_applicable(const Ctx& c, int N);
template <class T> void potrf_hegst_pipelined_begin(Ctx& c, int N, T* A, int lda, T* B, int ldb);
template <class T> void hegst_pipelined_finish(Ctx& c, int N, T* A, int lda, const T* U, int ldu);

// launch skeleton of stage 1 of a two-stage reduction (timing only; see blas3.hip)
#ifdef EIG_TOOLS
template <class T> void two_stage_stage1_skeleton(Ctx& c, hipStream_t st, int N, int what);
#endif

inline constexpr int kDiagBlk = 64;  // order of the inverted diagonal blocks

}  // namespace eig
