"""Multi-GPU sharding plan and roll-up for the layer sweep (new capability: the reference is single-GPU, device 0
only -- examples/spmma.cu:27-28).  The path shards with NO data-path collective: every (layer, batch index) unit is
an independent prune -> compress -> matmul, so each rank (one process per GPU) runs its own units
and the only communication is one tiny all-reduce of {dense-equivalent flops done (SUM), elapsed seconds (MAX)} over
RCCL ("nccl" backend on ROCm) -- or gloo in the CPU tests.  Aggregate GF/s = sum(flops) / max(t).

A plan is a list of units (layer, batch_begin, batch_end): `layer` indexes the concatenated shape tables, the batch
range is in GLOBAL batch indices.  Operand data are generated per (layer, global batch index) -- see unit_seed() --
so a sharded run multiplies exactly the matrices the unsharded run does, whatever the world size."""
import os

MODES = ("weak", "strong", "lpt", "hybrid")

# hybrid: a layer is split by batch index only while a rank's share still fills the chip with resident row tiles: 256 CUs x
# the three 128-row workgroups a CU holds of the kernels that serve the tall layers.  Below that a tile's time is set by
# its K stages, not by its rows (128 tiles of a 784 x 256 x 2304 layer take the time of 256), so a batch
# slice of such a layer costs every rank most of the whole layer's time.  Whole layers balance well on their own: the
# ResNet tables repeat each shape 1-6 times (bytes within 1.2 % at 8 ranks).
HYBRID_FILL_ROWS = 768 * 128


def layer_cost(m, n, k, b):
    """Modelled time of one layer (arbitrary unit): the elements it streams over the rate its kernel family reaches alone
    (bench.py `families`: direct 4.9, wide 3.0, A-stationary 3.1, span 2.9 TB/s of algorithmic bytes, round 3)."""
    by = b * (m * k + m * n) + k * n
    if k % 64:
        rate = 2.9
    elif n <= 128 or (n <= 256 and k <= 64):
        rate = 4.9
    elif n > 256 and k <= 512:
        rate = 3.1
    else:
        rate = 3.0
    return by / rate


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def _split(batch, world, rank):
    per, extra = divmod(batch, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def shard_units(num_layers, batch, world, rank, mode="weak"):
    """Units this rank runs when every layer has the same batch, as (layer, batch_begin, batch_end).
    weak  : every rank runs every layer on its own full batch (per-GPU work fixed as N grows); rank r owns global
            batch indices [r*batch, (r+1)*batch) of a virtual batch of world*batch
    strong: the batch dimension of every layer is split across ranks (total work fixed)"""
    return plan_units([(0, 0, 0, batch)] * num_layers, world, rank, mode)


def plan_units(shapes, world, rank, mode="weak"):
    """shapes: [(m, n, k, b)] of the concatenated tables.  Returns this rank's [(layer, batch_begin, batch_end)].
    weak  : all layers, global batch indices [rank*b, (rank+1)*b)
    strong: all layers, [g*b/G, (g+1)*b/G) of each layer's batch (SURVEY.md 8(e) primary partitioning; B replicated)
    lpt   : whole layers, greedy longest-processing-time assignment by 2*m*n*k*b (SURVEY.md 8(e) alternative; the
            config-4 sweep over several tables), deterministic: ties by layer index, equal loads to the lower rank
    hybrid: strong scaling (total work fixed) with the granularity chosen per layer: a layer whose per-rank batch share
            still has >= HYBRID_FILL_ROWS rows is split by batch index as in `strong`; a smaller one stays whole and goes
            to the least-loaded rank (LPT by the layer's modelled time, layer_cost(), on top of the split layers' equal shares).
            What `bench.py --gpus N` uses by default on one table"""
    if mode not in MODES:
        raise ValueError(mode)
    if world < 1 or not (0 <= rank < world):
        raise ValueError((world, rank))
    if mode == "weak":
        return [(l, rank * b, (rank + 1) * b) for l, (_, _, _, b) in enumerate(shapes)]
    if mode == "strong":
        out = []
        for l, (_, _, _, b) in enumerate(shapes):
            lo, hi = _split(b, world, rank)
            if hi > lo:
                out.append((l, lo, hi))
        return out
    if mode == "hybrid":
        by = layer_cost  # modelled time: the layers are HBM-bound, at a rate that depends on the kernel family
        out, whole, load = [], [], [0.0] * world
        for l, (m, n, k, b) in enumerate(shapes):
            if world > 1 and (b < world or m * (b // world) < HYBRID_FILL_ROWS):
                whole.append(l)
                continue
            for r in range(world):
                lo, hi = _split(b, world, r)
                load[r] += by(m, n, k, hi - lo)
                if r == rank and hi > lo:
                    out.append((l, lo, hi))
        for l in sorted(whole, key=lambda l: (-by(*shapes[l]), l)):
            g = min(range(world), key=lambda r: (load[r], r))
            load[g] += by(*shapes[l])
            if g == rank:
                out.append((l, 0, shapes[l][3]))
        return sorted(out)
    cost = [2.0 * m * n * k * b for (m, n, k, b) in shapes]
    order = sorted(range(len(shapes)), key=lambda l: (-cost[l], l))
    load = [0.0] * world
    mine = []
    for l in order:
        g = min(range(world), key=lambda r: (load[r], r))
        load[g] += cost[l]
        if g == rank:
            mine.append(l)
    return [(l, 0, shapes[l][3]) for l in sorted(mine)]


def unit_flops(shapes, units):
    """Dense-equivalent flops (2*m*n*k per batch index) of a list of units."""
    return sum(2.0 * shapes[l][0] * shapes[l][1] * shapes[l][2] * (hi - lo) for l, lo, hi in units)


def unit_seed(base, layer, batch_index):
    """Seed of the operand of (layer, global batch index): independent of world size, rank and sharding mode."""
    return (int(base) + 0x9E3779B1 * (layer + 1) + 0x85EBCA77 * (batch_index + 1)) & 0xFFFFFFFFFFFF


def rollup(flops_done, seconds, device=None, force_collective=False):
    """(total flops over ranks, max seconds over ranks).  No-op without an initialised process group, and -- unless
    `force_collective` -- with a group of one rank (force_collective: run the two all-reduces anyway, so that a one-GPU
    box can put the RCCL path of an N-GPU run through the backend: tests/test_bench_multirank.py)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return float(flops_done), float(seconds)
    f = torch.tensor([float(flops_done)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item()), float(t.item())
