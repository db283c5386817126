"""world_size-2 gloo test of the batch sharding logic used by bench.py / the multi-GPU path
(no GPU here): problems are partitioned p -> rank (p mod G) with no data-path collective, the
only collectives are the timing MAX-reduce and the optional result gather."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from eigensolver_gpu_amd.batch import shard_problems, gather_eigenvalues
    dist.init_process_group(backend="gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    mine = shard_problems(13, r, w)
    allp = [None] * w
    dist.all_gather_object(allp, mine)
    flat = sorted(sum(allp, []))
    assert flat == list(range(13)), flat
    assert all(p %% w == r for p in mine)
    # result gather: each rank contributes (problem id, eigenvalues)
    local = {p: torch.arange(4, dtype=torch.float64) + p for p in mine}
    got = gather_eigenvalues(local, 13, 4)
    if r == 0:
        assert got.shape == (13, 4)
        for p in range(13):
            assert float(got[p, 0]) == p
    t = torch.tensor([1.0 + r], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(w)
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
