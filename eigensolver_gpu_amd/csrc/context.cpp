// context.cpp -- per-(thread,device) runtime state, host LAPACK binding, roctx ranges.
// Replaces module eigsolve_vars (eigsolve_vars.F90:25-61) and nvtx_inters
// (lib_eigsolve/toolbox.F90:25-99) of the reference.
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "../../include/eigsolve_gpu.h"
#include "common.h"

namespace eig {

namespace {
// A context belongs to the thread that uses it.  When that thread ends, its contexts go back to a process-wide pool
// (no HIP call is made from a thread-local destructor: the runtime -- or an attached profiler, rocprofv3 aborts -- may
// already be tearing that thread down) and the next thread that needs a context for the same device takes one from the
// pool, streams, events and cached scratch included.  Threads that come and go therefore neither leak device memory nor
// pay for hipMalloc / stream creation again.
std::mutex g_pool_mu;
std::vector<Ctx*> g_pool;
struct CtxHolder {
    std::map<int, Ctx*> m;
    ~CtxHolder() {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto& kv : m) g_pool.push_back(kv.second);
        m.clear();
    }
};
thread_local CtxHolder t_ctx;

// Stream pool.  Every hipStream occupies one of the process' few hardware queues (GPU_MAX_HW_QUEUES, default 4, one of them
// taken by the null stream) and what a batch of solves achieves depends on how its launch chains are spread over them:
// measured at C3 with three solves in flight, 16.0 problems/s when every chain has a hardware queue of its own, 12.4-12.8
// when two of them share one (profiles/r03_experiments.txt; which queue a new stream gets once all are taken is the
// runtime's choice -- rocprofv3 showed the 4th library stream landing on an already busy queue or on the idle null
// stream's, depending on creation order).  The library therefore keeps its stream count at the number of API calls that
// are actually in flight: a context has no stream of its own; every public entry point leases one from this pool for the
// duration of the call (StreamLease) and gives it back, synchronised, on return.  A single-threaded caller uses one
// stream for ever, a batch with three problems in flight three, however many threads have ever called the library.
struct PooledStream {
    hipStream_t s;
    bool busy;
};
std::mutex g_stream_mu;
std::map<int, std::vector<PooledStream>> g_streams;   // per device
hipStream_t lease_stream(int dev, hipStream_t prefer) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto& v = g_streams[dev];
    for (auto& p : v)
        if (!p.busy && p.s == prefer) { p.busy = true; return p.s; }
    for (auto& p : v)
        if (!p.busy) { p.busy = true; return p.s; }
    hipStream_t s = nullptr;
    EIG_HIP(hipStreamCreate(&s));   // blocking streams, like the reference's cudaStreamCreate (eigsolve_vars.F90:50-52)
    v.push_back(PooledStream{s, true});
    return s;
}
void return_stream(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& p : g_streams[dev])
        if (p.s == s) p.busy = false;
}
}  // namespace
int streams_in_use(int dev) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    int n = 0;
    for (auto& p : g_streams[dev]) n += p.busy ? 1 : 0;
    return n;
}
namespace {

std::mutex g_lapack_mu;
void* g_lapack_handle = nullptr;
stedc_fn g_dstedc = nullptr;
void (*g_set_threads)(int) = nullptr;

using roctx_push_t = int (*)(const char*);
using roctx_pop_t = int (*)();
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
bool g_roctx_tried = false;

void try_roctx() {
    if (g_roctx_tried) return;
    g_roctx_tried = true;
    const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                           "libroctx64.so.4"};
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_LAZY | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
        g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) return;
    }
    g_roctx_push = nullptr;
    g_roctx_pop = nullptr;
}
}  // namespace

// may_fail: "not enough device memory" is an answer (nullptr; the slot is then empty, nothing is printed), not an exception
static void* scratch_grow(Ctx& c, const char* name0, size_t bytes, bool may_fail) {
    char nbuf[64];
    const char* name = name0;
    if (c.grp_q > 0) { snprintf(nbuf, sizeof nbuf, "%s#g%d", name0, c.grp_q); name = nbuf; }   // (a group's problems: slots of their own)
    auto& s = c.slots[name];
    if (s.second < bytes) {
        ++c.slot_gen;
        if (s.first) {
            if (c.lease_depth > 0) c.sync(c.s1);     // (outside a call nothing of this context is in flight: calls end synchronised)
            if (c.s2) EIG_HIP(hipStreamSynchronize(c.s2));
            if (c.s3) EIG_HIP(hipStreamSynchronize(c.s3));
            void* old = s.first;
            s.first = nullptr; s.second = 0;   // (a failing hipMalloc below must not leave a dangling pointer in the slot)
            EIG_HIP(hipFree(old));
        }
        const size_t cap = bytes + bytes / 8 + 256;
        const hipError_t e = hipMalloc(&s.first, cap);
        if (e != hipSuccess) {
            s.first = nullptr; s.second = 0;
            if (may_fail && (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation)) {
                (void)hipGetLastError();       // (handled: clear the sticky error, print nothing)
                return nullptr;
            }
            EIG_HIP(e);
        }
        s.second = cap;
    }
    return s.first;
}
void* Ctx::scratch_bytes(const char* name, size_t bytes) { return scratch_grow(*this, name, bytes, false); }
void* Ctx::try_scratch_bytes(const char* name, size_t bytes) { return scratch_grow(*this, name, bytes, true); }

void* Ctx::host_scratch_bytes(const char* name, size_t bytes) {
    auto& s = hslots[name];
    if (s.second < bytes) {
        if (s.first) EIG_HIP(hipHostFree(s.first));
        size_t cap = bytes + bytes / 8 + 256;
        EIG_HIP(hipHostMalloc(&s.first, cap, hipHostMallocDefault));
        s.second = cap;
    }
    return s.first;
}

StreamLease::StreamLease(Ctx& ctx_) : c(ctx_) {
    if (c.lease_depth == 0) c.s1 = lease_stream(c.dev, c.s1);   // may throw (hipStreamCreate): the depth is only counted on success
    ++c.lease_depth;
}
StreamLease::~StreamLease() {
    if (--c.lease_depth == 0 && c.s1) {
        // nothing of this call may still be running on the stream when another context takes it (an exception may have
        // cut the call short of its final sync)
        if (c.evSync && hipEventRecord(c.evSync, c.s1) == hipSuccess) (void)hipEventSynchronize(c.evSync);
        if (c.s2) {
            (void)hipStreamSynchronize(c.s2);
            return_stream(c.dev, c.s2);
            c.s2 = nullptr;
        }
        if (c.s3) {
            (void)hipStreamSynchronize(c.s3);
            return_stream(c.dev, c.s3);
            c.s3 = nullptr;
        }
        return_stream(c.dev, c.s1);
    }
}

void Ctx::sync(hipStream_t st) {
    if (st == s2 && s2) { EIG_HIP(hipStreamSynchronize(s2)); return; }
    EIG_HIP(hipEventRecord(evSync, st));
    EIG_HIP(hipEventSynchronize(evSync));
}

// The second stream of a call is LEASED from the same pool as its first one (and handed back when the call returns): the pool's
// streams are created once, in order, and map to distinct hardware queues / pipes (profiles/r03_experiments.txt 5, 10, 14).  A
// private per-context stream -- created whenever a solve first overlaps two chains -- stays behind idle, takes a hardware queue
// out of that order, and a later batch finds two of its launch chains on one pipe: 16.4 -> 11 problems/s at C3 (measured).
hipStream_t Ctx::second_stream() {
    if (!s2) s2 = lease_stream(dev, nullptr);
    return s2;
}
hipStream_t Ctx::third_stream() {
    if (!s3) s3 = lease_stream(dev, nullptr);
    return s3;
}

void Ctx::drop_graphs() {
    if (graphs.empty()) return;
    for (auto& kv : graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    graphs.clear();
}

void Ctx::release() {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;   // runtime already gone (process teardown): nothing to free
    if (dev >= 0 && cur != dev) (void)hipSetDevice(dev);
    if (s2) (void)hipStreamSynchronize(s2);
    if (s3) (void)hipStreamSynchronize(s3);
    drop_graphs();
    for (auto& kv : slots)
        if (kv.second.first) (void)hipFree(kv.second.first);
    slots.clear();
    for (auto& kv : hslots)
        if (kv.second.first) (void)hipHostFree(kv.second.first);
    hslots.clear();
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    if (evA) (void)hipEventDestroy(evA);
    for (auto& e : evStage) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (evB) (void)hipEventDestroy(evB);
    for (auto& e : evLA) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (d_info) (void)hipFree(d_info);
    if (h_info) (void)hipHostFree(h_info);
    if (evSync) (void)hipEventDestroy(evSync);
    evSync = nullptr;
    // s1 belongs to the stream pool: never destroyed
    if (s2) return_stream(dev, s2);   // (pool streams live until eigsolve_finalize)
    if (s3) return_stream(dev, s3);
    for (auto& e : ev) e = nullptr;
    evA = evB = nullptr; d_info = nullptr; h_info = nullptr; s1 = s2 = s3 = nullptr;
    if (dev >= 0 && cur != dev) (void)hipSetDevice(cur);
}

// How many launch chains of a batch run side by side: every chain wants a hardware queue of its own next to the null stream's, and
// the part serves FOUR queues concurrently -- measured (profiles/r03_experiments.txt, section 10): 3 / 4 chains with 8 queues allowed
// 16.0 / 16.3 problems/s at C3 and 67.7 / 82.8 at C5, but 5, 6, 8 chains 55 / 52 / 46 (C5) however many queues are allowed, and 4
// chains on the default 4 queues (one shared with another chain) 55.9.  So: 4 when the process allows >= 5 queues
// (GPU_MAX_HW_QUEUES, read by the HIP runtime at start-up; ROCm's default is 4), otherwise 3.
int auto_batch_workers() {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    return (e && atoi(e) >= 5) ? 4 : 3;
}

void copy_options(Ctx& c, const Ctx& d) {
    if (c.trd_nb != d.trd_nb || c.hemv_blocks != d.hemv_blocks || c.mv_dma != d.mv_dma) c.drop_graphs();   // baked into captured launch sequences
    c.trd_nb = d.trd_nb; c.bt_nb = d.bt_nb; c.hemv_blocks = d.hemv_blocks; c.use_graph = d.use_graph; c.overlap = d.overlap;
    c.trsm_base = d.trsm_base; c.potrf_mode = d.potrf_mode; c.gst_mode = d.gst_mode; c.gst_thr = d.gst_thr;
    c.tridiag_device = d.tridiag_device; c.real_il_reference = d.real_il_reference; c.tile_map = d.tile_map;
    c.batch_workers = d.batch_workers; c.trace_marks = d.trace_marks; c.batch_fuse = d.batch_fuse; c.batch_zip = d.batch_zip; c.trd_finish = d.trd_finish;
    c.zs_cap_mb = d.zs_cap_mb; c.mv_dma = d.mv_dma; c.gemm_dma = d.gemm_dma; c.gemm_wide = d.gemm_wide; c.gemm_lean = d.gemm_lean;
}

// ---- the library's worker threads ------------------------------------------------------------------------------------
// eigsolve_?hegvdx_batch keeps several problems in flight on ONE caller thread by handing them to these threads (QE's k-point
// loop is a single-threaded Fortran caller; SURVEY.md 8(b) "Threading").  The threads are created on first use, never
// joined (they idle on a condition variable; the process exit ends them) and own ordinary per-thread contexts.
namespace {
struct WorkerPool {
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::deque<std::function<void()>> q;
    int nthreads = 0;
    // eigsolve_finalize(dev): every worker thread releases its context for `dev`.  Requests are numbered (epoch); a thread
    // handles the requests newer than the last one it has seen whenever it is between two tasks, and acknowledges.
    unsigned rel_epoch = 0;
    std::vector<std::pair<unsigned, int>> rel;   // (epoch, device)
    int rel_acks = 0;
    void ensure(int n) {   // mu held
        while (nthreads < n) {
            const unsigned born = rel_epoch;
            std::thread([this, born] {
                unsigned seen = born;
                for (;;) {
                    std::function<void()> f;
                    std::vector<int> todo;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !q.empty() || rel_epoch != seen; });
                        if (rel_epoch != seen) {
                            for (auto& r : rel)
                                if ((int)(r.first - seen) > 0) todo.push_back(r.second);
                            seen = rel_epoch;
                        } else {
                            f = std::move(q.front());
                            q.pop_front();
                        }
                    }
                    if (!f) {
                        for (int dev : todo) {
                            auto it = t_ctx.m.find(dev);
                            if (it == t_ctx.m.end()) continue;
                            (void)hipSetDevice(dev);
                            it->second->release();
                            delete it->second;
                            t_ctx.m.erase(it);
                        }
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            ++rel_acks;
                        }
                        cv_done.notify_all();
                        continue;
                    }
                    f();
                }
            }).detach();
            ++nthreads;
        }
    }
};
WorkerPool& workers() {
    static WorkerPool* p = new WorkerPool();   // leaked on purpose: no destructor may run while a worker still waits on it
    return *p;
}
std::mutex g_finalize_mu;   // one eigsolve_finalize at a time (the acknowledgement count belongs to one request)
}  // namespace

void batch_run(int dev, int nworkers, int ntasks, const std::function<void(int)>& fn) {
    if (ntasks <= 0) return;
    if (nworkers > ntasks) nworkers = ntasks;
    if (nworkers < 1) nworkers = 1;
    struct Call {
        std::atomic<int> next{0};
        std::mutex mu;
        std::condition_variable cv;
        int running = 0;
    } call;
    auto loop = [&call, &fn, ntasks] {
        for (int t; (t = call.next.fetch_add(1)) < ntasks;) {
            try { fn(t); } catch (...) { }   // fn reports through its own info slot
        }
    };
    // The CALLER is one of the workers (it would only wait otherwise): a batch with w problems in flight then uses w streams
    // in all -- the caller's context and w - 1 pool threads' -- and w = 3 fits the 4 hardware queues ROCm gives a process by
    // default next to the null stream.  (With the caller idle and 3 pool threads, i.e. 4 library streams, two of the three
    // active launch chains ended up sharing a hardware queue: 12.8 instead of 16.0 problems/s at C3.)
    const int helpers = nworkers - 1;
    call.running = helpers;
    if (helpers > 0) {
        WorkerPool& wp = workers();
        {
            std::lock_guard<std::mutex> lk(wp.mu);
            wp.ensure(helpers);
            for (int w = 0; w < helpers; ++w)
                wp.q.emplace_back([&call, &loop, dev] {
                    (void)hipSetDevice(dev);
                    loop();
                    std::lock_guard<std::mutex> lk2(call.mu);
                    if (--call.running == 0) call.cv.notify_all();
                });
        }
        wp.cv.notify_all();
    }
    loop();
    std::unique_lock<std::mutex> lk(call.mu);
    call.cv.wait(lk, [&] { return call.running == 0; });
}

// Every worker thread of the library releases its context for `dev`; returns when all of them have (a worker that is in the
// middle of another caller's batch task does so when that task ends).  Callers are serialised; no thread spins.
void batch_workers_finalize(int dev) {
    std::lock_guard<std::mutex> fin(g_finalize_mu);
    WorkerPool& wp = workers();
    std::unique_lock<std::mutex> lk(wp.mu);
    const int n = wp.nthreads;
    if (n == 0) return;
    ++wp.rel_epoch;
    if (wp.rel.size() > 64) wp.rel.erase(wp.rel.begin(), wp.rel.begin() + 32);   // (every live thread has long seen those)
    wp.rel.emplace_back(wp.rel_epoch, dev);
    wp.rel_acks = 0;
    wp.cv.notify_all();
    wp.cv_done.wait(lk, [&] { return wp.rel_acks >= n; });
}

// idle streams of the pool for `dev` (nothing of the library runs on them: calls end synchronised)
void stream_pool_finalize(int dev) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto& v = g_streams[dev];
    for (size_t i = 0; i < v.size();) {
        if (!v[i].busy) {
            (void)hipStreamDestroy(v[i].s);
            v.erase(v.begin() + i);
        } else {
            ++i;
        }
    }
}

// ---- tunables ---------------------------------------------------------------------------------------------------------
// One table: eigsolve_set_option(name, value) and the environment variable EIGSOLVE_<NAME> (read when a context is created;
// same value semantics, plus the words "host" / "device" for TRIDIAG and "rec" for POTRF) go through apply_option.
static const char* const kOptionNames[] = {"trd_nb", "bt_nb", "hemv_blocks", "real_il_reference", "graph", "overlap", "trsm_base",
                                           "potrf", "gst", "gst_thr", "batch_workers", "batch_fuse", "batch_zip", "tridiag", "tile_map",
                                           "trd_finish", "trace_marks", "zs_cap_mb", "mv_dma", "gemm_dma", "gemm_wide", "gemm_lean"};
bool apply_option(Ctx& c, const std::string& s, int value) {
    if (s == "trd_nb") { c.trd_nb = (value <= 0 || value > 64) ? kTrdNbDefault : value; c.drop_graphs(); }
    else if (s == "bt_nb") c.bt_nb = norm_bt_nb(value);
    else if (s == "hemv_blocks") { c.hemv_blocks = value < 0 ? 0 : (value > kHemvBlocksMax ? kHemvBlocksMax : value); c.drop_graphs(); }
    else if (s == "real_il_reference") c.real_il_reference = value > 0;
    else if (s == "graph") c.use_graph = value > 0;
    else if (s == "overlap") c.overlap = value < 0 ? kOverlapDefault : (value & 7);
    else if (s == "trsm_base") c.trsm_base = value <= 0 ? kTrsmBaseDefault : norm_trsm_base(value);
    else if (s == "potrf") c.potrf_mode = value == 0 ? 0 : (value == 1 ? 1 : kPotrfDefault);
    else if (s == "gst") c.gst_mode = (value < 0 || value > 3) ? kGstModeDefault : value;
    else if (s == "gst_thr") c.gst_thr = value <= 0 ? kGstThrDefault : (value < 256 ? 256 : value);
    else if (s == "batch_workers") c.batch_workers = (value < 0 || value > 16) ? -1 : value;
    else if (s == "batch_fuse") c.batch_fuse = (value < 1 || value > 4) ? -1 : value;
    else if (s == "batch_zip") c.batch_zip = (value < 0 || value > 3) ? 3 : value;
    else if (s == "tridiag") c.tridiag_device = value < 0 ? kTridiagDefault : (value > 0 ? 1 : 0);
    else if (s == "tile_map") c.tile_map = value != 0;
    else if (s == "trd_finish") { c.trd_finish = value < 0 ? -1 : value; c.drop_graphs(); }
    else if (s == "trace_marks") c.trace_marks = value > 0;
    else if (s == "zs_cap_mb") c.zs_cap_mb = value <= 0 ? kZsCapMbDefault : value;
    else if (s == "gemm_dma") c.gemm_dma = (value < 0 || value > 3) ? kGemmDmaDefault : value;
    else if (s == "gemm_lean") c.gemm_lean = value < 0 ? kGemmLeanDefault : value;
    else if (s == "gemm_wide") c.gemm_wide = (value < 0 || value > 2) ? kGemmWideDefault : value;
    else if (s == "mv_dma") { c.mv_dma = value < 0 ? kMvDmaDefault : value; c.drop_graphs(); }
    else return false;
    return true;
}

static void init_options(Ctx& c) {
    Ctx d;   // compile-time defaults
    copy_options(c, d);
    c.trace_marks = 0;
    for (const char* name : kOptionNames) {
        std::string env = "EIGSOLVE_";
        for (const char* p = name; *p; ++p) env += (char)toupper((unsigned char)*p);
        const char* e = getenv(env.c_str());
        if (!e || !*e) continue;
        // Values are integers.  Words are accepted only where the header documents them: TRIDIAG=host|device, POTRF=rec.
        // Anything else that is not a number is ignored with one line on stderr (it used to go through atoi and silently select
        // setting 0, e.g. EIGSOLVE_OVERLAP=default switched the overlap off).
        const std::string nm = name, val = e;
        int v = 0;
        char* end = nullptr;
        const long lv = strtol(e, &end, 10);
        while (end && (*end == ' ' || *end == '\t')) ++end;
        if (end != e && end && *end == 0) v = (int)lv;
        else if (nm == "tridiag" && (val == "host" || val == "HOST")) v = 0;
        else if (nm == "tridiag" && (val == "device" || val == "DEVICE")) v = 1;
        else if (nm == "potrf" && (val == "rec" || val == "REC")) v = 0;
        else {
            fprintf(stderr, "eigsolve: ignoring %s=%s (not an integer)\n", env.c_str(), e);
            continue;
        }
        (void)apply_option(c, name, v);
    }
}

Ctx& ctx() {
    int dev = 0;
    EIG_HIP(hipGetDevice(&dev));
    auto it = t_ctx.m.find(dev);
    if (it != t_ctx.m.end()) return *it->second;
    {   // a context a finished thread left behind for this device?
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = g_pool.size(); i-- > 0;)      // most recently returned first: a stream of short-lived threads keeps
            if (g_pool[i]->dev == dev) {              // re-using ONE warm context instead of cycling through (and growing) all
                Ctx* c = g_pool[i];
                g_pool.erase(g_pool.begin() + i);
                init_options(*c);          // tunables are per owner: back to the defaults / environment
                t_ctx.m[dev] = c;
                return *c;
            }
    }
    Ctx* c = new Ctx();
    c->dev = dev;
    for (auto& e : c->ev) EIG_HIP(hipEventCreate(&e));
    EIG_HIP(hipEventCreateWithFlags(&c->evSync, hipEventDisableTiming));
    EIG_HIP(hipEventCreateWithFlags(&c->evA, hipEventDisableTiming));
    EIG_HIP(hipEventCreateWithFlags(&c->evB, hipEventDisableTiming));
    EIG_HIP(hipMalloc((void**)&c->d_info, 64));
    EIG_HIP(hipHostMalloc((void**)&c->h_info, 64, hipHostMallocDefault));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    init_options(*c);
    t_ctx.m[dev] = c;
    return *c;
}

int load_lapack(const char* path) {
    std::lock_guard<std::mutex> lk(g_lapack_mu);
    void* h = nullptr;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            fprintf(stderr, "eigsolve: cannot dlopen LAPACK library '%s': %s\n", path, dlerror());
            return -1;
        }
    } else {
        h = dlopen(nullptr, RTLD_NOW);
    }
    const char* names[] = {"dstedc_", "scipy_dstedc_", "dstedc", "dstedc_64_", "DSTEDC"};
    stedc_fn f = nullptr;
    for (const char* n : names) {
        f = (stedc_fn)dlsym(h, n);
        if (f) break;
    }
    if (!f) return -1;
    g_dstedc = f;
    g_lapack_handle = h;
    const char* tn[] = {"openblas_set_num_threads", "scipy_openblas_set_num_threads", "openblas_set_num_threads64_",
                        "MKL_Set_Num_Threads"};
    g_set_threads = nullptr;
    for (const char* n : tn) {
        g_set_threads = (void (*)(int))dlsym(h, n);
        if (g_set_threads) break;
    }
    return 0;
}

stedc_fn get_dstedc() {
    if (g_dstedc) return g_dstedc;
    const char* env = getenv("EIGSOLVE_LAPACK_LIB");
    if (env && load_lapack(env) == 0) return g_dstedc;
    if (load_lapack(nullptr) == 0) return g_dstedc;
    const char* guesses[] = {"liblapack.so.3", "liblapack.so", "libopenblas.so.0", "libopenblas.so", "libmkl_rt.so"};
    for (const char* g : guesses) {
        void* h = dlopen(g, RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            dlclose(h);
            if (load_lapack(g) == 0) return g_dstedc;
        }
    }
    return nullptr;
}

void set_host_threads(int n) {
    (void)get_dstedc();
    if (g_set_threads && n > 0) g_set_threads(n);
}

void range_push(const char* name) {
    try_roctx();
    if (g_roctx_push) {
        (void)hipDeviceSynchronize();
        g_roctx_push(name);
    }
}
void range_pop() {
    try_roctx();
    if (g_roctx_pop) {
        (void)hipDeviceSynchronize();
        g_roctx_pop();
    }
}
// Phase ranges inside the drivers (the reference wraps potrf/gst/evd/trsm and trd/stedc/unmtr in NVTX ranges,
// zhegvdx_gpu.F90:134-170, zheevd_gpu.F90:80-132).  No device synchronisation here: they mark the host-side issue
// window of a phase (several solves may be in flight on one GPU), the per-phase device times are in
// eigsolve_get_phase_times.
void phase_range_push(const char* name) {
    try_roctx();
    if (g_roctx_push) g_roctx_push(name);
}
void phase_range_pop() {
    if (g_roctx_pop) g_roctx_pop();
}

}  // namespace eig

extern "C" {

int eigsolve_init(void) {
    try {
        (void)eig::ctx();
        return 0;
    } catch (...) {
        return -1;
    }
}

int eigsolve_finalize(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    eig::batch_workers_finalize(dev);
    {   // contexts finished threads left behind for this device
        std::lock_guard<std::mutex> lk(eig::g_pool_mu);
        for (size_t i = 0; i < eig::g_pool.size();) {
            if (eig::g_pool[i]->dev == dev) {
                eig::g_pool[i]->release();
                delete eig::g_pool[i];
                eig::g_pool.erase(eig::g_pool.begin() + i);
            } else {
                ++i;
            }
        }
    }
    auto it = eig::t_ctx.m.find(dev);
    if (it != eig::t_ctx.m.end()) {
        it->second->release();
        delete it->second;
        eig::t_ctx.m.erase(it);
    }
    eig::stream_pool_finalize(dev);
    return 0;
}

int eigsolve_set_lapack(const char* path) {
    if (path && *path) return eig::load_lapack(path);
    return eig::get_dstedc() ? 0 : -1;
}

int eigsolve_set_host_threads(int n) {
    eig::set_host_threads(n);
    return 0;
}

int eigsolve_set_option(const char* name, int value) {
    try {
        return eig::apply_option(eig::ctx(), std::string(name ? name : ""), value) ? 0 : -1;
    } catch (...) {
        return -1;
    }
}

void eigsolve_range_push(const char* name, int) { eig::range_push(name ? name : "range"); }
void eigsolve_range_pop(void) { eig::range_pop(); }

int eigsolve_get_phase_times(double* ms, int n) {
    try {
        eig::Ctx& c = eig::ctx();
        int k = n < eig::PH_COUNT ? n : (int)eig::PH_COUNT;
        for (int i = 0; i < k; ++i) ms[i] = c.phase_ms[i];
        return k;
    } catch (...) {
        return 0;
    }
}

const char* eigsolve_version(void) { return "eigensolver_gpu_amd 0.1 (gfx950, fp64 MFMA, wave64)"; }
}
