"""`bench.py --gpus N` without a launcher, and `--emulate-world N`: neither imports torch in this process."""
import json
import os
import sys

from .common import BENCH_PY, read_shapes, table_path


def emulate_world(args):
    """`--emulate-world N`: one child process per emulated rank, each alone on the one GPU; this process never imports torch.
    A strong (pure batch split) plan with b % N == 0 gives every rank the same shapes: rank 0 is measured and the others
    are stated to be identical."""
    import subprocess
    N = args.emulate_world
    if N < 1:
        raise SystemExit("bench: --emulate-world must be >= 1")
    ntab = len((args.tables or args.table or "x").split(","))
    mode = args.scaling or ("weak" if N == 1 else ("lpt" if ntab > 1 else "hybrid"))
    argv = []
    skip = 0
    for a in sys.argv[1:]:  # the child's command line: ours without --emulate-world / --scaling / --gpus
        if skip:
            skip -= 1
            continue
        if a in ("--emulate-world", "--scaling", "--gpus", "--rebalance"):
            skip = 1
            continue
        if a.startswith(("--emulate-world=", "--scaling=", "--gpus=", "--rebalance=")):
            continue
        argv.append(a)

    def child(extra):
        cmd = [sys.executable, BENCH_PY] + argv + ["--no-extras", "--no-cpu-baseline"] + extra
        res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or len(lines) != 1:
            raise SystemExit(f"bench --emulate-world: child {extra} failed (rc {res.returncode})")
        return json.loads(lines[0])

    import tempfile
    costs_file = os.path.join(tempfile.mkdtemp(prefix="sm_emu_"), "costs.json")
    argv += ["--costs-file", costs_file]   # the N = 1 child measures the per-shape costs, the rank children plan with them
    base = child(["--scaling", "weak"])  # N = 1: the whole table on the one GPU
    shapes = [s_ for t in (args.tables or args.table or ("resnet18" if args.dtype == "f32" else "resnet50")).split(",") for s_ in read_shapes(table_path(t))]
    identical = mode == "weak" or (mode == "strong" and all(b % N == 0 for _, _, _, b in shapes))
    ranks = [0] if identical else list(range(N))
    def run_ranks(bias_file=None):
        extra = ["--rank-bias-file", bias_file] if bias_file else []
        per_ = {r: child(["--scaling", mode, "--emulate-world", str(N), "--emu-rank", str(r)] + extra) for r in ranks}
        return ([per_[r if not identical else 0]["ms_per_step"] for r in range(N)],
                [per_[r if not identical else 0]["emulated"]["dense_equiv_gflop_per_step"] for r in range(N)])
    ms, gf = run_ranks()
    # Closed-loop balancing (round 6; hybrid only, --rebalance R rounds): what a real N-GPU run does in its setup -- every rank measures its
    # own step, the times are all-gathered, the difference to the cost model becomes a per-rank bias and every rank re-plans with it
    # (multigpu.rebalance_bias / plan_units(rank_bias=...)) -- emulated here by running the rank children again with the bias in a file.
    rounds = []
    if mode == "hybrid" and not identical and getattr(args, "rebalance", 0) > 0 and N > 1:
        import __graft_entry__ as ge
        mg = ge.load_package_module("multigpu")
        costs = {tuple(int(x) for x in k_.split("x")): tuple(v) for k_, v in json.load(open(costs_file)).items()}
        mg.set_measured_costs(costs)
        best = (max(ms), ms, gf, None)
        bias = None
        rounds.append({"round": 0, "max_ms": max(ms), "spread": (max(ms) - min(ms)) / (sum(ms) / len(ms))})
        for it in range(args.rebalance):
            modelled = mg.plan_loads(shapes, N, "hybrid", rank_bias=bias)
            bias = mg.rebalance_bias([t * 1e3 for t in ms], modelled)
            bf = os.path.join(os.path.dirname(costs_file), "bias_%d.json" % it)
            json.dump(bias, open(bf, "w"))
            ms, gf = run_ranks(bf)
            rounds.append({"round": it + 1, "max_ms": max(ms), "spread": (max(ms) - min(ms)) / (sum(ms) / len(ms)), "bias_us": [round(x, 1) for x in bias]})
            if max(ms) < best[0]:
                best = (max(ms), ms, gf, bias)
        _, ms, gf, _ = best   # a real run keeps the best plan it measured (one more re-plan when the last round was not the best)
    tmax = max(ms)
    total = sum(gf)
    out = {"metric": base["metric"], "value": total / (tmax * 1e-3), "unit": "GF/s", "n_gpus": N,
           "label": "predicted, single-GPU emulation: every rank's units ran ALONE on one MI355X, each in a fresh process; no RCCL, no "
                    "contention between ranks, one box's clock -- not a measured N-GPU run",
           "predicted": True, "partition_mode": mode, "scaling": "weak" if mode == "weak" else "strong",
           "per_rank_ms": ms, "per_rank_gflop": gf, "max_ms": tmax, "min_ms": min(ms),
           "spread": (max(ms) - min(ms)) / (sum(ms) / len(ms)),
           "ranks_measured": ranks, "ranks_identical_by_construction": identical, "rebalance_rounds": rounds,
           "n1_ms": base["ms_per_step"], "n1_value": base["value"],
           "predicted_speedup_vs_n1": (total / (tmax * 1e-3)) / base["value"],
           "steps": args.steps, "warmup": args.warmup, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": base["config"]["workload"], "library": base["config"]["library"]}}
    print(json.dumps(out))
    return 0


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run as a CHILD
    process, relay what rank 0 prints and return the exit code; non-zero when the job fails or does not report N ranks."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH_PY] + sys.argv[1:]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # stderr goes straight through
    sys.stdout.write(res.stdout)
    sys.stdout.flush()
    if res.returncode != 0:
        sys.stderr.write(f"bench: the {args.gpus}-rank job exited with {res.returncode}\n")
        return res.returncode
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    try:
        ok = len(lines) == 1 and json.loads(lines[0])["n_gpus"] == args.gpus
    except Exception:
        ok = False
    if not ok:
        sys.stderr.write(f"bench: expected one JSON line with n_gpus == {args.gpus}\n")
        return 3
    return 0
