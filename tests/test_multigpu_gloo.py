"""CPU test of the N>1 path: world_size-2 gloo run of the sharding plan and the throughput roll-up that
bench.py uses across GPUs (SUM of flops, MAX of time; no data-path collective)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    mg = ge.load_package_module("multigpu")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    weak = mg.shard_units(49, 32, world, rank, "weak")
    strong = mg.shard_units(49, 32, world, rank, "strong")
    flops = 1000.0 * (rank + 1)          # pretend work
    seconds = 2.0 if rank == 0 else 5.0  # the slow rank sets the time
    tot, tmax = mg.rollup(flops, seconds)
    out[rank] = (len(weak), weak[0], strong[0], tot, tmax)
    dist.barrier()
    dist.destroy_process_group()


def test_rollup_and_sharding_world2():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        nweak, w0, s0, tot, tmax = res[rank]
        assert nweak == 49 and w0 == (0, 0, 32)
        assert s0 == (0, 16 * rank, 16 * rank + 16)
        assert tot == 3000.0 and tmax == 5.0  # every rank sees sum(flops) and max(time)


def test_strong_sharding_covers_the_batch_exactly_once():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    mg = ge.load_package_module("multigpu")
    for world in (1, 2, 3, 4, 8):
        for batch in (1, 5, 32):
            seen = []
            for rank in range(world):
                for (_, lo, hi) in mg.shard_units(1, batch, world, rank, "strong"):
                    seen += list(range(lo, hi))
            assert sorted(seen) == list(range(batch))
    # rollup without a process group is the identity
    assert mg.rollup(7.0, 3.0) == (7.0, 3.0)
