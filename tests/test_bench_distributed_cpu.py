"""world_size-2 and world_size-8 gloo tests of the batch path bench.py uses for the multi-GPU legs (no GPU here): the SAME functions --
batch.shard_problems / InflightPool / run_sharded_batch / gather_eigenvalues -- driven with a CPU stub solver.
Problems are partitioned p -> rank (p mod G) with no data-path collective; the only collectives are the timing
all_gather and the result gather."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys, threading
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from eigensolver_gpu_amd.batch import (InflightPool, shard_problems, run_sharded_batch, gather_eigenvalues,
                                           gather_eigenvectors, rank_cpu_slice, pin_rank_to_cpu_slice)
    # every rank keeps to its own share of the node's CPUs (bench.py does this before the library starts its workers)
    before = sorted(os.sched_getaffinity(0))
    lr, lw = int(os.environ["LOCAL_RANK"]), int(os.environ["LOCAL_WORLD_SIZE"])
    mask = pin_rank_to_cpu_slice(lr, lw)
    assert mask == sorted(os.sched_getaffinity(0)) and set(mask) <= set(before)
    dist.init_process_group(backend="gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    masks = [None] * w
    dist.all_gather_object(masks, (before, mask))
    from eigensolver_gpu_amd.batch import MIN_CPUS_PER_RANK
    if len(before) // w >= MIN_CPUS_PER_RANK:   # enough CPUs: equal, disjoint, contiguous shares of the common mask
        assert all(b == before for b, _ in masks)
        assert all(len(mk) == len(before) // w for _, mk in masks)
        allc = sum((mk for _, mk in masks), [])
        assert len(allc) == len(set(allc)), "rank CPU masks overlap"
        assert mask == rank_cpu_slice(before, lr, lw)
    else:                                     # a share too small for a rank's own threads: nobody is restricted
        assert mask == before
    assert rank_cpu_slice(range(16), 1, 2) == list(range(8, 16)) and rank_cpu_slice(range(16), 1, 8) == list(range(16))
    assert rank_cpu_slice(range(256), 3, 8) == list(range(96, 128)) and rank_cpu_slice([5, 1, 3], 1, 8) == [1, 3, 5]
    NP, M = (64, 4) if w == 8 else (13, 4)          # world 8: BASELINE configs[4], 64 problems -> 8 per rank
    mine = shard_problems(NP, r, w)
    if w == 8:
        assert len(mine) == 8 and mine == list(range(r, 64, 8))
        from eigensolver_gpu_amd.batch import host_threads_per_rank
        assert host_threads_per_rank(256, 8) == 32 and host_threads_per_rank(8, 8) == 1 and host_threads_per_rank(4, 8) == 1
    allp = [None] * w
    dist.all_gather_object(allp, mine)
    assert sorted(sum(allp, [])) == list(range(NP))
    assert all(p %% w == r for p in mine)

    def matrix(p):            # problem p: a seeded symmetric matrix whose eigenvalues the stub solver returns
        g = torch.Generator().manual_seed(1000 + 17 * p)
        a = torch.rand((8, 8), generator=g, dtype=torch.float64)
        return a + a.T + 8 * torch.eye(8, dtype=torch.float64)

    seen = {}
    def init(t):
        seen[t] = threading.get_ident()
    def solve(p, t):          # CPU stand-in for api.hegvdx on worker t's context
        assert threading.get_ident() == seen[t]      # persistent threads: the worker that was initialised
        return torch.linalg.eigvalsh(matrix(p))[:M]

    with InflightPool(2, init=init) as pool:
        for step in range(3):                         # same threads across steps (contexts live with them)
            local = run_sharded_batch(NP, r, w, solve, pool)
        assert sorted(local) == mine
        ids = {threading.get_ident()}
        assert len(seen) == 2 and not (set(seen.values()) & ids)
        try:                                          # worker exceptions surface in the caller
            pool.map(lambda p, t: 1 // 0, [1, 2, 3])
            raise SystemExit("exception was swallowed")
        except ZeroDivisionError:
            pass
    got = gather_eigenvalues(local, NP, M)
    if r == 0:
        assert got.shape == (NP, M)
        for p in range(NP):
            assert torch.equal(got[p], torch.linalg.eigvalsh(matrix(p))[:M]), p
    else:
        assert got is None
    # optional gather of the eigenvector blocks (SURVEY.md 8(e)): problem p's block = m rows of length n, complex
    NZ, MZ = 6, 3
    def zblock(p):
        g = torch.Generator().manual_seed(77 + p)
        return torch.complex(torch.rand((MZ + 2, NZ), generator=g, dtype=torch.float64), torch.rand((MZ + 2, NZ), generator=g, dtype=torch.float64))
    gz = gather_eigenvectors({p: zblock(p) for p in mine}, NP, NZ, MZ)
    if r == 0:
        assert gz.shape == (NP, MZ, NZ) and gz.dtype == torch.complex128
        for p in range(NP):
            assert torch.equal(gz[p], zblock(p)[:MZ]), p
    else:
        assert gz is None
    # a rank WITHOUT a block (n_problems < world) still contributes one of the right shape, dtype and device kind
    g1 = gather_eigenvectors({0: zblock(0)} if r == 0 else {}, 1, NZ, MZ)
    g2 = gather_eigenvectors({0: zblock(0)} if r == w - 1 else {}, 1, NZ, MZ, dtype=torch.complex128, device="cpu")
    if r == 0:
        assert g1.shape == (1, MZ, NZ) and g1.dtype == torch.complex128 and torch.equal(g1[0], zblock(0)[:MZ])
        assert g2.shape == (1, MZ, NZ) and torch.equal(g2[0], zblock(0)[:MZ])
    else:
        assert g1 is None and g2 is None
    t = torch.tensor([1.0 + r], dtype=torch.float64)
    allt = [torch.empty_like(t) for _ in range(w)]
    dist.all_gather(allt, t)
    assert max(float(x) for x in allt) == float(w)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", r, "ok")
""") % ROOT


def test_two_rank_gloo(tmp_path):
    f = tmp_path / "w2.py"
    f.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(f)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_eight_rank_gloo(tmp_path):
    """The 8-way split a driver with an 8-GPU node will run (no such node here): 64 problems -> 8 per rank, p -> rank p mod 8,
    gathered [64, m] on rank 0, max-over-ranks timing reduction -- the same batch.py functions, 8 gloo ranks on the CPU."""
    f = tmp_path / "w8.py"
    f.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", "29519", str(f)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 8


def test_bench_uses_the_batch_module():
    """bench.py's multi-GPU legs go through eigensolver_gpu_amd/batch.py (the functions the gloo test drives)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ("InflightPool", "run_sharded_batch", "gather_eigenvalues", "shard_problems", "host_threads_per_rank",
                 "pin_rank_to_cpu_slice", "gather_eigenvectors"):
        assert name in src


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py --gpus N must mean N ranks: with WORLD_SIZE set by a launcher to something else the script refuses to run
    (before it touches a GPU) instead of labelling a 1-rank measurement as an N-GPU one; launched plainly with --gpus N > 1 it
    re-executes itself under torch.distributed.run (GPU test: test_bench_gpus_flag_starts_its_own_ranks)."""
    envv = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=envv)
    assert out.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=1" in (out.stdout + out.stderr)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "torch.distributed.run" in src and "args.gpus" in src
