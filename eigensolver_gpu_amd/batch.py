"""Batch-of-independent-problems sharding (QE k-point style, BASELINE.json configs[4]).

The reference has no multi-GPU code (single GPU, process-global state, eigsolve_vars.F90:29-35);
a single solve is a chain of dependent Householder steps and does not shard.  A batch shards
embarrassingly: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), static
block-cyclic assignment, NO collective on the data path.  RCCL is used only for the optional
gather of results and for the benchmark's timing reduction.

Inside one process several problems are kept in flight on the GPU either by the library itself (eigsolve_?hegvdx_batch:
one call from one host thread, the library's worker threads keep `batch_workers` problems in flight -- the default of
bench.py) or by a small pool of PERSISTENT host threads issuing one-problem calls (InflightPool): the library keeps one
context (events, cached scratch) per (host thread, device), so such threads should live as long as the batch does."""
import queue
import threading


def shard_problems(n_problems, rank, world):
    """Static block-cyclic partition p -> rank (p mod world)."""
    return [p for p in range(n_problems) if p % world == rank]


def host_threads_per_rank(cores, world):
    """Host LAPACK / BLAS threads one rank may use when `world` ranks share a node's cores (the host-side
    tridiagonal step of 8 ranks must not oversubscribe the CPU).  Capped at 64: OpenBLAS stops scaling long before."""
    return max(1, min(64, int(cores) // max(int(world), 1)))


MIN_CPUS_PER_RANK = 6   # below this share per rank, pinning would stack a rank's launch-chain threads on top of each other


def rank_cpu_slice(cpus, local_rank, local_world):
    """The CPUs local rank `local_rank` of `local_world` ranks on one node may use: a contiguous, equal share of the sorted
    list `cpus` (the process' current affinity mask), disjoint from every other rank's.  A rank drives its GPU from a handful
    of host threads (the caller, the library's batch workers, the divide & conquer's deflation scans, OpenBLAS for the host
    tridiagonal option): eight unpinned ranks would migrate over all 256 cores of the node and share caches at random."""
    cpus = sorted(int(c) for c in cpus)
    local_world = max(1, int(local_world))
    local_rank = int(local_rank) % local_world
    per = len(cpus) // local_world
    if per < MIN_CPUS_PER_RANK:      # a share too small for a rank's own threads (four launch chains + the caller): nobody is restricted
        return cpus
    return cpus[local_rank * per:(local_rank + 1) * per]


def pin_rank_to_cpu_slice(local_rank, local_world):
    """Restricts EVERY thread this process has at the moment of the call (os.sched_setaffinity(0, ...) alone changes only the
    calling thread: OpenBLAS / OpenMP pools that torch or numpy started at import would keep the node-wide mask) -- and, by
    inheritance, every thread started afterwards (the library's batch workers) -- to this rank's rank_cpu_slice.  Call before
    the library creates its worker threads.  Returns the CPU list now in force for the calling thread; a platform without
    sched_setaffinity, or a refusal by the OS, leaves the mask alone and returns it."""
    import os
    if not hasattr(os, "sched_getaffinity"):
        return list(range(os.cpu_count() or 1))
    cur = sorted(os.sched_getaffinity(0))
    want = rank_cpu_slice(cur, local_rank, local_world)
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return cur
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = []
    for tid in tids:
        try:
            os.sched_setaffinity(tid, want)      # (on Linux the pid argument is a thread id)
        except OSError:
            pass                                 # a thread that ended in between
    return sorted(os.sched_getaffinity(0))


class InflightPool:
    """`nthr` persistent worker threads.  `init(t)` runs once in worker t (device selection, solver options: the
    library's options are per context, i.e. per thread); `map(fn, items)` hands items[t::nthr] to worker t, which
    calls fn(item, t) for each, and returns the results in item order.  The first worker exception is re-raised."""

    def __init__(self, nthr, init=None):
        self.nthr = max(1, int(nthr))
        self._in = [queue.Queue() for _ in range(self.nthr)]
        self._out = queue.Queue()
        self._threads = []
        # Workers are started one at a time, each finishing its init before the next starts (deterministic context order).
        # (Round 2 needed this to steer which hardware queue a worker's streams landed on; since round 3 the library leases
        #  one stream per call from a process-wide pool and the order no longer matters.)
        for t in range(self.nthr):
            ready = threading.Event()
            th = threading.Thread(target=self._run, args=(t, init, ready), daemon=True)
            th.start()
            ready.wait()
            self._threads.append(th)

    def _run(self, t, init, ready):
        err = None
        try:
            if init is not None:
                init(t)
        except BaseException as ex:  # noqa: BLE001 -- reported to the caller of map()
            err = ex
        finally:
            ready.set()
        while True:
            job = self._in[t].get()
            if job is None:
                return
            fn, chunk = job
            res = []
            try:
                if err is not None:
                    raise err
                for idx, item in chunk:
                    res.append((idx, fn(item, t)))
                self._out.put((t, res, None))
            except BaseException as ex:  # noqa: BLE001
                self._out.put((t, res, ex))

    def map(self, fn, items):
        items = list(items)
        chunks = [[(i, items[i]) for i in range(t, len(items), self.nthr)] for t in range(self.nthr)]
        for t in range(self.nthr):
            self._in[t].put((fn, chunks[t]))
        out = [None] * len(items)
        first = None
        for _ in range(self.nthr):
            _, res, ex = self._out.get()
            for idx, r in res:
                out[idx] = r
            if ex is not None and first is None:
                first = ex
        if first is not None:
            raise first
        return out

    def close(self):
        for q in self._in:
            q.put(None)
        for th in self._threads:
            th.join(timeout=60)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def run_sharded_batch(n_problems, rank, world, solve, pool, fuse=1):
    """Solves this rank's share of a batch of independent problems: p -> rank (p mod world), the rank's problems
    spread over the pool's in-flight workers.  fuse = 1: `solve(p, t)` returns the eigenvalues of problem p (a 1-D tensor)
    computed by worker t.  fuse = F > 1: the rank's problems are handed out in groups of F (same order, one
    `eigsolve_?hegvdx_batch` call: tridiagonalizations in lockstep) and `solve(group, t)` returns one tensor per problem
    of the group.  No communication.  Returns {problem id: eigenvalues}."""
    mine = shard_problems(n_problems, rank, world)
    if fuse <= 1:
        vals = pool.map(solve, mine)
        return dict(zip(mine, vals))
    groups = [mine[i:i + fuse] for i in range(0, len(mine), fuse)]
    outs = pool.map(solve, groups)
    return {p: v for g, vs in zip(groups, outs) for p, v in zip(g, vs)}


def gather_eigenvalues(local, n_problems, m):
    """Optional result gather: local = {problem_id: tensor[m]} -> [n_problems, m] tensor on rank 0
    (other ranks get None).  One all_gather of a padded block per rank: sizes are tiny
    (64 x 512 x 8 B = 262 KB in C5), xGMI is nowhere near a limit."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    per = (n_problems + world - 1) // world
    any_t = next(iter(local.values())) if local else torch.zeros(m, dtype=torch.float64)
    buf = torch.zeros((per, m + 1), dtype=torch.float64, device=any_t.device)
    buf[:, 0] = -1.0
    for k, (p, wv) in enumerate(sorted(local.items())):
        buf[k, 0] = float(p)
        buf[k, 1:] = wv[:m].to(torch.float64)
    if dist.is_initialized():       # (also with ONE rank: the collective then runs over the process group's backend all the same)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    if rank != 0:
        return None
    out = torch.zeros((n_problems, m), dtype=torch.float64, device=any_t.device)
    for part in parts:
        ids = part[:, 0].to(torch.int64)
        keep = ids >= 0
        out[ids[keep]] = part[keep, 1:]
    return out


def gather_eigenvectors(local, n_problems, n, m, dtype=None, device=None):
    """Optional gather of the eigenvector blocks ("w ... and optionally Z(1:N,1:m)"), off by default in
    bench.py (--gather-z): local = {problem_id: Z} with Z the solver's output block in its device layout -- a tensor whose first
    m rows are the m eigenvectors (column-major N x m = row-major m x N) -- -> [n_problems, m, n] on rank 0 (None elsewhere).
    One gather (dst = rank 0) of this rank's padded [per, m, n] block plus one of the problem ids: only rank 0 allocates the
    world-sized receive list.  This is the only place the batch path moves bulk data between GPUs (C5: 64 x 512 x 2048 complex =
    1 GiB in all, 128 MiB per rank over xGMI).
    dtype / device: what an EMPTY rank (n_problems < world) sends -- every rank must contribute a block of the same shape, dtype
    and device kind.  When omitted they are agreed on with one all_gather_object (first rank that holds a block decides; a rank
    that holds none uses its current CUDA device for a CUDA peer, the CPU otherwise)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    per = (n_problems + world - 1) // world
    any_t = next(iter(local.values())) if local else None
    # (the decision to run the agreement collective depends on the CALLER's arguments only: every rank takes the same branch)
    agree = dist.is_initialized() and (dtype is None or device is None)
    dev_kind = torch.device(device).type if device is not None else None
    if any_t is not None:
        dtype, dev_kind, device = any_t.dtype, any_t.device.type, any_t.device
    if agree:
        mine = (str(any_t.dtype).replace("torch.", ""), any_t.device.type) if any_t is not None else None
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        first = next((x for x in seen if x is not None), None)
        if first is not None and any_t is None:
            dtype = dtype or getattr(torch, first[0])
            dev_kind = dev_kind or first[1]
    dtype = dtype or torch.float64
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dev_kind == "cuda" else torch.device("cpu")
    buf = torch.zeros((per, m, n), dtype=dtype, device=device)
    ids = torch.full((per,), -1, dtype=torch.int64, device=device)
    for k, (p, Z) in enumerate(sorted(local.items())):
        ids[k] = p
        buf[k] = Z[:m, :n]
    if dist.is_initialized():
        # (complex blocks travel as (re, im) pairs: every backend moves real tensors)
        sbuf = torch.view_as_real(buf) if buf.is_complex() else buf
        idl = [torch.empty_like(ids) for _ in range(world)] if rank == 0 else None
        rparts = [torch.empty_like(sbuf) for _ in range(world)] if rank == 0 else None
        dist.gather(ids, idl, dst=0)
        dist.gather(sbuf, rparts, dst=0)
        if rank != 0:
            return None
        parts = [torch.view_as_complex(x) if buf.is_complex() else x for x in rparts]
    else:
        parts, idl = [buf], [ids]
    out = torch.zeros((n_problems, m, n), dtype=dtype, device=device)
    for part, pid in zip(parts, idl):
        keep = pid >= 0
        out[pid[keep]] = part[keep]
    return out
