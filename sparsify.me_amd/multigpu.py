"""Multi-GPU roll-up for the layer sweep (new capability: the reference is single-GPU, device 0 only --
examples/spmma.cu:27-28).  The path shards with NO data-path collective: every (layer, batch) unit is an
independent prune -> compress -> matmul, so each rank (one process per GPU) runs its own units and the
only communication is one tiny all-reduce of {dense-equivalent flops done (SUM), elapsed seconds (MAX)}
over RCCL ("nccl" backend on ROCm) -- or gloo in the CPU tests.  Aggregate GF/s = sum(flops) / max(t)."""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_units(num_layers, batch, world, rank, mode="weak"):
    """Units this rank runs, as (layer, batch_begin, batch_end).
    weak  : every rank runs every layer on its own full batch (per-GPU work fixed as N grows)
    strong: the batch dimension of every layer is split across ranks (total work fixed)"""
    if mode == "weak":
        return [(l, 0, batch) for l in range(num_layers)]
    if mode != "strong":
        raise ValueError(mode)
    per, extra = divmod(batch, world)
    lo = rank * per + min(rank, extra)
    hi = lo + per + (1 if rank < extra else 0)
    return [(l, lo, hi) for l in range(num_layers) if hi > lo]


def rollup(flops_done, seconds, device=None):
    """(total flops over ranks, max seconds over ranks).  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(flops_done), float(seconds)
    f = torch.tensor([float(flops_done)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item()), float(t.item())
