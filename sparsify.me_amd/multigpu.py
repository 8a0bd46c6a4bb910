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

# hybrid (round 4 form): a layer is split by batch index when its SHAPE GROUP -- the same-shape layers a rank launches as one
# grouped grid -- still fills the chip on a rank's batch share: count x m x (b / world) >= HYBRID_FILL_ROWS rows (512 resident
# 128-row workgroups: the direct kernels hold two to three per CU).  On the ResNet-50 table that is every 12544-row layer at
# up to 8 ranks.  The other layers stay WHOLE (a batch slice of a few-tile layer costs a rank most of the whole layer's time:
# a tile's time is set by its K stages) and are placed longest-first by measured time, in chunks of the same
# shape so that a rank still launches them grouped.
HYBRID_FILL_ROWS = 512 * 128

# Per-shape costs.  Since round 5 bench.py MEASURES them in its setup on the box it runs on (measure_costs(): one grouped launch per
# unique shape, outside the timed region; rank 0's figures are broadcast so that every rank plans with the same numbers) and installs
# them with set_measured_costs().  The table below is only the FALLBACK for planning without a GPU (the CPU tests, `--emulate-world`
# children that are handed no cost file): one MI355X's round-4 figures, us per instance at b = 32 inside a grouped launch of the
# shape (profiles/sweep_r04_f16_resnet50.txt, column `fused`); shapes in neither fall back to bytes over their kernel family's rate.
_measured_us = {}   # (m, n, k) -> {b it was measured at: us per instance}


def set_measured_costs(costs):
    """costs: {(m, n, k): (us_per_instance, b)} or {(m, n, k): (us, b, us2, b2, ...)} as measure_costs() returns them (or None / {} to
    forget them).  Round 6: a shape may carry its time at SEVERAL batch sizes -- the full b and a rank's share b / world -- because a
    batch slice does not cost b_share / b of the whole layer (few-tile shapes: a tile's time is its K stages)."""
    _measured_us.clear()
    for key, v in (costs or {}).items():
        v = list(v)
        _measured_us[tuple(int(x) for x in key)] = {int(v[i + 1]): float(v[i]) for i in range(0, len(v) - 1, 2)}


def measure_costs(sm, torch, shapes, dtype, reps=3, world=1):
    """Time ONE grouped launch of every unique (m, n, k, b) of `shapes` with its instance count (capped at 8) on the current device:
    the fused 2:4 path where it takes the shape, compress + spmma elsewhere.  The instances share one operand set (timing only: the
    launch geometry is what costs; C is overwritten by every instance).  Returns {(m, n, k): (us per instance, b)} -- and, with
    world > 1 (an int or several), for the shapes the hybrid plan would split by batch index also their time at a rank's share:
    (us, b, us_share, b // world, ...)."""
    import collections
    cnt = collections.Counter(shapes)
    out = {}
    work = []
    worlds = sorted(set(w for w in ([world] if isinstance(world, int) else list(world)) if w > 1))
    for (m, n, k, b), c in cnt.items():
        work.append((m, n, k, b, c))
        for w in worlds:   # the batch share of a rank, for the shapes the hybrid plan splits at that world size
            if b >= w and c * m * (b // w) >= HYBRID_FILL_ROWS and (m, n, k, b // w, c) not in work:
                work.append((m, n, k, b // w, c))
    for (m, n, k, b, c) in work:
        c = min(8, c)
        A = torch.empty(b * m * k, dtype=dtype, device="cuda"); sm.fill_uniform(A, 17, 0.0, 1.0)
        B = torch.empty(k * n, dtype=dtype, device="cuda"); sm.fill_uniform(B, 18, 0.0, 1.0)
        C = torch.empty(b * m * n, dtype=dtype, device="cuda")
        fused = ((n % 8 == 0 and (k % 64 == 0 or (n <= 128 and (b * m * k * A.element_size()) % 16 == 0))) or (n < 8 and k <= 64)) and dtype != torch.float32
        if fused:
            try:
                sm.spmma_fused_grouped([A] * c, [B] * c, [C] * c, m, n, k, batch=b)
                call = lambda: sm.spmma_fused_grouped([A] * c, [B] * c, [C] * c, m, n, k, batch=b)
            except sm.SparsifymeError:
                fused = False
        if not fused:
            blob = torch.empty(sm.compress24_size(m, k, A.element_size(), b), dtype=torch.uint8, device="cuda")

            def call():
                for _ in range(c):
                    sm.compress24(A, m, k, k, b, m * k, blob)
                    sm.spmma(blob, B, C, m, n, k, b, 0)
        t = min(sm.graph_time_ms(call, iters=2, replays=3) for _ in range(reps)) * 1e3 / c
        out[(m, n, k)] = tuple(out.get((m, n, k), ())) + (t, b)
        del A, B, C
    return out


MEASURED_US_B32 = {
    (12544, 64, 147): 55.7, (12544, 64, 64): 21.3, (12544, 64, 576): 91.4, (12544, 256, 64): 53.6, (12544, 64, 256): 48.2,
    (12544, 128, 256): 64.6, (3136, 128, 1152): 48.3, (3136, 512, 128): 31.8, (3136, 128, 512): 27.1, (3136, 256, 512): 43.4,
    (784, 256, 2304): 31.1, (784, 1024, 256): 19.2, (784, 256, 1024): 15.2, (784, 512, 1024): 33.9, (196, 512, 4608): 39.1,
    (196, 2048, 512): 21.1, (196, 512, 2048): 16.5,
}


def layer_cost(m, n, k, b):
    """Modelled time of one layer in us: the per-instance time of its shape measured on this box (set_measured_costs), else the
    fallback table's, scaled by the batch share; else the elements it streams over the rate its kernel family reaches alone
    (TB/s of algorithmic bytes)."""
    if (m, n, k) in _measured_us:
        tab = _measured_us[(m, n, k)]
        if b in tab:
            return tab[b]
        b0 = min(tab, key=lambda x: (abs(x - b), x))   # the nearest measured batch size, scaled
        return tab[b0] * b / float(b0)
    t = MEASURED_US_B32.get((m, n, k))
    if t is not None:
        return t * b / 32.0
    by = 2.0 * (b * (m * k + m * n) + k * n)
    if k % 64:
        rate = 2.9
    elif n <= 128 or (n <= 256 and k <= 64):
        rate = 4.9
    elif n > 256 and k <= 512:
        rate = 3.1
    else:
        rate = 3.0
    return by / rate * 1e-6


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


def plan_units(shapes, world, rank, mode="weak", rank_bias=None):
    """shapes: [(m, n, k, b)] of the concatenated tables.  Returns this rank's [(layer, batch_begin, batch_end)].
    weak  : all layers, global batch indices [rank*b, (rank+1)*b)
    strong: all layers, [g*b/G, (g+1)*b/G) of each layer's batch (SURVEY.md 8(e) primary partitioning; B replicated)
    lpt   : whole layers, greedy longest-processing-time assignment by 2*m*n*k*b (SURVEY.md 8(e) alternative; the
            config-4 sweep over several tables), deterministic: ties by layer index, equal loads to the lower rank
    hybrid: strong scaling (total work fixed) with the granularity chosen per SHAPE GROUP: the layers of a shape whose grouped
            launch still fills the chip on a rank's batch share (count x m x b / world >= HYBRID_FILL_ROWS rows) are split by
            batch index as in `strong`; the others stay whole and are placed in same-shape chunks, longest measured time first
            (layer_cost()), on top of the split layers' equal shares.  What `bench.py --gpus N` uses by default on one table.
            rank_bias (round 6; hybrid only): per-rank microseconds added to a rank's load before anything is placed -- the difference
            between what a rank's step MEASURED and what the cost model said (a step of ~10 launches over 8 streams is not the sum of
            its launches alone); bench.py measures it in setup, all-gathers it and re-plans (closed-loop balancing; the same list on
            every rank, so every rank computes the same plan)"""
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
        groups = {}
        for l, sh in enumerate(shapes):
            groups.setdefault(sh, []).append(l)
        out, load, whole_groups = [], ([float(x) for x in rank_bias] if rank_bias else [0.0] * world), []
        if len(load) != world:
            raise ValueError("rank_bias needs one entry per rank")
        for (m, n, k, b), ls in groups.items():
            if world == 1 or (b >= world and len(ls) * m * (b // world) >= HYBRID_FILL_ROWS):
                for l in ls:  # split by batch index as in `strong`
                    for r in range(world):
                        lo, hi = _split(b, world, r)
                        load[r] += layer_cost(m, n, k, hi - lo)
                        if r == rank and hi > lo:
                            out.append((l, lo, hi))
            else:
                whole_groups.append(((m, n, k, b), ls))
        # whole layers: chunks of one shape (a rank launches a chunk as one grouped grid), no chunk above ~ 0.6 of a rank's
        # fair share of the whole-layer time, placed longest-first on the least-loaded rank
        total = sum(layer_cost(*sh) * len(ls) for sh, ls in whole_groups)
        cap = 0.6 * total / world if world > 1 else float("inf")
        chunks = []
        for sh, ls in whole_groups:
            c1 = layer_cost(*sh)
            per = max(1, int(cap // c1)) if c1 > 0 else len(ls)
            nch = -(-len(ls) // per)
            base, extra = divmod(len(ls), nch)
            i = 0
            for j in range(nch):  # nch chunks of nearly equal size
                sz = base + (1 if j < extra else 0)
                chunks.append((c1 * sz, ls[i:i + sz]))
                i += sz
        owner = {}
        for cost, ls in sorted(chunks, key=lambda c: (-c[0], c[1][0])):
            g = min(range(world), key=lambda r: (load[r], r))
            load[g] += cost
            for l in ls:
                owner[l] = g
        # (round 6) refinement of the longest-first placement, deterministic: while it lowers the maximum, move ONE whole layer from the
        # most loaded rank to the least loaded one, or swap one of its layers for a cheaper one of any other rank.  The chunks above keep
        # same-shape layers together where that costs nothing; this step trades a little of that for balance: with another box's costs
        # (+-20 % per shape) the plain placement left up to 12.5 % between the most loaded rank and the mean at 8 ranks, this keeps it
        # under 8 % (tests/test_multigpu_gloo.py::test_hybrid_plan_balance_survives_perturbed_costs).
        lc = {l: layer_cost(*shapes[l]) for l in owner}
        for _ in range(4 * len(owner) + 8):
            hi_r = max(range(world), key=lambda r: (load[r], -r))
            best = None   # (new maximum of the ranks touched, kind, a, b, partner rank)
            mine_hi = sorted(l for l in owner if owner[l] == hi_r)
            for lo_r in range(world):
                if lo_r == hi_r:
                    continue
                for a in mine_hi:
                    new_max = max(load[hi_r] - lc[a], load[lo_r] + lc[a])
                    if new_max < load[hi_r] - 1e-9 and (best is None or new_max < best[0] - 1e-12):
                        best = (new_max, "move", a, None, lo_r)
                    for b_ in sorted(l for l in owner if owner[l] == lo_r):
                        if lc[b_] >= lc[a]:
                            continue
                        d = lc[a] - lc[b_]
                        new_max = max(load[hi_r] - d, load[lo_r] + d)
                        if new_max < load[hi_r] - 1e-9 and (best is None or new_max < best[0] - 1e-12):
                            best = (new_max, "swap", a, b_, lo_r)
            if best is None:
                break
            _, kind, a, b_, lo_r = best
            owner[a] = lo_r
            load[hi_r] -= lc[a]
            load[lo_r] += lc[a]
            if kind == "swap":
                owner[b_] = hi_r
                load[lo_r] -= lc[b_]
                load[hi_r] += lc[b_]
        out += [(l, 0, shapes[l][3]) for l in owner if owner[l] == rank]
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


def rebalance_bias(measured_us, modelled_us, old_bias=None):
    """The per-rank bias of the next planning round: what each rank's step measured minus what the cost model says its units cost
    (without any bias), shifted so that the smallest entry is 0 (only differences between ranks matter to the planner)."""
    raw = [float(t) - float(m) for t, m in zip(measured_us, modelled_us)]
    lo = min(raw)
    return [x - lo for x in raw]


def plan_loads(shapes, world, mode="hybrid", rank_bias=None):
    """Modelled time (us, layer_cost()) of every rank's units under `mode`: what the plan balances.  max / mean of it is the plan's
    predicted imbalance -- bench.py records it for N = 2 / 4 / 8 from the costs it measured on the box at N = 1, and
    tests/test_multigpu_gloo.py holds it under 8 % for cost tables perturbed by +-20 % (another box's numbers)."""
    loads = []
    for r in range(world):
        loads.append(sum(layer_cost(shapes[l][0], shapes[l][1], shapes[l][2], hi - lo) for l, lo, hi in plan_units(shapes, world, r, mode, rank_bias)))
    return loads
