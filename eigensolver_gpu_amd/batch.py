"""Batch-of-independent-problems sharding (QE k-point style, BASELINE.json configs[4]).

The reference has no multi-GPU code (single GPU, process-global state, eigsolve_vars.F90:29-35);
a single solve is a chain of dependent Householder steps and does not shard.  A batch shards
embarrassingly: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), static
block-cyclic assignment, NO collective on the data path.  RCCL is used only for the optional
gather of results and for the benchmark's timing reduction."""


def shard_problems(n_problems, rank, world):
    """Static block-cyclic partition p -> rank (p mod world)."""
    return [p for p in range(n_problems) if p % world == rank]


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
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    if rank != 0:
        return None
    out = torch.zeros((n_problems, m), dtype=torch.float64, device=any_t.device)
    for part in parts:
        for row in part:
            p = int(row[0].item())
            if p >= 0:
                out[p] = row[1:]
    return out
