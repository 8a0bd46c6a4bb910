#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE.json's headline workload.

Metric: effective GF/s (2:4 spmma vs dense gemm) on the ResNet-50 layer shapes, fp16, b = 32.
A "step" is one pass of the hot path over this rank's units of the table(s): for each layer the 2:4 prune + compress
of the per-batch activation operand A and the sparse x dense matmul -- as ONE fused kernel (sm_spmma_fused_f16) on the
layers where that wins (--path auto) and as sm_compress24_f16 + sm_spmma_f16 on the rest; both give the same C bit for
bit.  (This is compress(STRIP)+spmma on the UNPRUNED A, bit-identical to the reference's spmma() call sequence only for
an A that is already 2:4; the API-faithful sequence -- TILE prune in place, check, compress, multiply, spmma.hxx:82-113
-- is timed beside it as stages.api_spmma_ms.)  value = dense-equivalent flops (2*m*n*k per batch index, summed over
the units of every rank) / time.  Inputs are generated on the device, per (layer, global batch index), and are
resident in HBM before the timed region starts.  A step's launches, spread over 4 HIP streams, are captured once into a
hipGraph and replayed (the launches are 7-100 us each; without the graph the Python/ctypes call cost would be on the
clock).

Multi-GPU (--scaling): one process per GPU, no data-path collective, one tiny all-reduce (RCCL) of sum(flops) and
max(time).  weak (default): every rank runs every layer on its own b batch indices (rank r = global indices
[r*b, (r+1)*b)).  strong: every layer's batch is split [g*b/G, (g+1)*b/G), B replicated.  lpt: whole
layers, longest-processing-time assignment -- the config-4 sweep: --tables resnet50,resnet101,resnet152 --scaling lpt.

Besides the contract line's fields the JSON carries `stages` (matmul only, compress only, the API-faithful sequence and
the dense GEMMs that are the metric's denominator), `roofline` for the dominant kernel of the timed step and
`cpu_baseline` (the oracle's arithmetic timed on the host cores; rank 0, N = 1 only).
"""
import argparse
import csv
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
GUIDE_COPY_GBS = 6290.0    # the same guide, line 36: a float4 device copy measured at 6.29 TB/s (79 % of the specification)
F32_MATRIX_PEAK_TFS = 157.3  # same guide: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD


def read_shapes(path):
    with open(path, newline="") as f:
        rows = list(csv.reader(f))[1:]
    return [tuple(int(x) for x in r[:4]) for r in rows if r]


def table_path(name):
    if os.path.exists(name):
        return name
    p = os.path.join(ROOT, "datasets", name if name.endswith(".csv") else name + ".csv")
    if not os.path.exists(p):
        raise SystemExit(f"bench: no shape table {name!r}")
    return p


def file_tag(path, measured_on=None, loaded=None):
    """provenance of a replayed (not measured-in-this-run) profile file: relative path + content hash, the library it was
    measured on (tools/pmc_*.py record it) and whether that is the library THIS process loaded: `stale` = it is not (or the
    file does not say), and the caller then drops the replayed numbers instead of reporting another build's counters"""
    with open(path, "rb") as fh:
        return {"file": os.path.relpath(path, ROOT), "sha256_12": hashlib.sha256(fh.read()).hexdigest()[:12],
                "measured_in_this_run": False, "measured_on_library_sha256_16": measured_on, "loaded_library_sha256_16": loaded,
                "stale": (measured_on is None) or (measured_on != loaded),
                "how": "rocprofv3 --pmc passes of an earlier run of the same step (tools/pmc_traffic.py, tools/pmc_mfma.py); "
                       "the committed file is replayed here, the counters are not collected by bench.py itself"}


def library_tag(sm):
    """Which shared library this process loaded (SPARSIFYME_LIB can redirect it): path, version string, content hash."""
    with open(sm.LIB_PATH, "rb") as fh:
        h = hashlib.sha256(fh.read()).hexdigest()[:16]
    return {"lib_path": os.path.relpath(sm.LIB_PATH, ROOT) if sm.LIB_PATH.startswith(ROOT) else sm.LIB_PATH,
            "sm_version": sm.version(), "sha256_16": h, "redirected_by_env": bool(os.environ.get("SPARSIFYME_LIB"))}


def fused_variant(n, k, m=None, b=None, count=1, cus=256):
    """Which kernel sm_spmma_fused_f16[_grouped] dispatches a layer to (csrc/spmma_f16_fused.hip: spmma_fused16).  With m, b and
    the instance count of the launch given, the round-4 rule for the 256-row big form is applied too (it depends on how many
    tiles the launch has); without them the (n, k)-only families of rounds 1-3 are returned."""
    if n < 8 and k <= 64:
        return "thin"
    if k % 64 != 0:
        return "span"
    if n <= 128 or (n <= 256 and k <= 64):
        return "direct"
    astat = n > 256 and k <= 512
    if m is not None:
        rows = m * b                       # the batches of a shared-B launch are one tall matrix
        eff = lambda t: t / (-(-t // cus) * cus)
        t_big = -(-rows // 256) * -(-n // 256) * count
        t_wide = -(-rows // 128) * -(-n // 256) * count
        big = eff(t_big) >= eff(t_wide)
        if astat:
            panels, ns, tn = -(-rows // 128) * count, 1, -(-n // 128)
            while panels * ns * 4 < 3 * cus and -(-tn // (2 * ns)) >= 2:
                ns *= 2
            big = eff(t_big) > eff(panels * ns) + 0.1
        if big:
            return "big"
    return "astat" if astat else "wide"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--plan-costs", choices=["measure", "table"], default="measure",
                    help="per-shape costs of the hybrid plan: measured in setup on this box (one grouped launch per unique shape, outside "
                         "the timed region; rank 0's figures are broadcast) or the library's fallback table (sparsify.me_amd/multigpu.py)")
    ap.add_argument("--costs-file", default=None,
                    help="--emulate-world plumbing: the N = 1 child writes the costs it measured here, the rank children read them (so that "
                         "every emulated rank plans with the same numbers, as the broadcast does in a real N-GPU run)")
    ap.add_argument("--config4-stage", choices=["auto", "on", "off"], default="auto",
                    help="also time the config-4 sweep (resnet50 + 101 + 152, 300 layers, lpt) in the same run and report it as "
                         "stages.config4_sweep; auto = when N > 1 (north_star's >= 7x at 8 GPUs is stated on that sweep)")
    ap.add_argument("--streamk", choices=["on", "off"], default="off",
                    help="on: give the grouped fused launches a workspace (sm_spmma_fused_*_grouped_ws): the library then runs the stream-K form "
                         "on the shapes its rule names (round 5).  Default off: the form shortens a launch that runs ALONE (196 x 512 x 4608 x 3: "
                         "123 -> 106 us) but inside the 8-stream step, where other kernels fill a few-tile launch's idle CUs, its extra partial-sum "
                         "traffic costs more than its balance returns (1.600 / 1.608 ms against 1.591 / 1.588, profiles/bench_streamk_ab_r05f.txt)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="setup, before the W warm-up steps: untimed replays of the step for this long (clock ramp after an idle "
                         "GPU, first-use state of a fresh graph); 0 = none.  Stated in config.launch")
    ap.add_argument("--tables", default=None,
                    help="comma-separated shape tables (names under datasets/ or paths), concatenated into one work list; "
                         "default resnet50 (fp16 / bf16) or resnet18 (f32: BASELINE config 2); "
                         "config 4 = resnet50,resnet101,resnet152")
    ap.add_argument("--table", default=None, help="one shape table (alias of --tables)")
    ap.add_argument("--scaling", choices=["weak", "strong", "lpt", "hybrid"], default=None,
                    help="multi-GPU partitioning of the (layer, batch) units: weak = every rank the whole table on its own "
                         "batch; strong = batch split b/G per layer; lpt = whole layers by longest-processing-time; hybrid = "
                         "batch split where a rank's share still fills the chip, whole layers (LPT) elsewhere.  Default: total "
                         "work fixed as N grows -- hybrid on one table, lpt on several (--tables); N = 1: all the same")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-stage / denominator passes")
    ap.add_argument("--eager", action="store_true", help="launch from Python instead of replaying a hipGraph")
    ap.add_argument("--path", choices=["auto", "staged"], default="auto",
                    help="auto: fused prune+compress+matmul kernel on the layers where it wins (n <= --fused-max-n), the "
                         "staged compress24 + spmma pair elsewhere; staged: the pair on every layer")
    ap.add_argument("--fused-max-n", type=int, default=512,
                    help="auto path: widest n served by sm_spmma_fused_f16 whatever k (one workgroup spans up to 256 columns: up to "
                         "there A is loaded and selected once, at 512 twice).  512 (default since round 3): inside the step, which is "
                         "bound by the bytes it moves, fusing the n = 512 long-K layers too is 3-4 % faster than compress + spmma for "
                         "them (no blob written and re-read) although each such launch alone is slower; 256 = round 2's choice")
    ap.add_argument("--fused-max-k-wide", type=int, default=512,
                    help="auto path: wider layers (n > --fused-max-n) are still fused when k <= this (the A-stationary "
                         "kernel keeps the 2:4 image of a row panel in LDS across its column tiles); 0 = never")
    ap.add_argument("--batch-split", type=int, default=1,
                    help="run every layer's batch as this many independent problems of b / S entries (more, smaller work items "
                         "for the streams to interleave; same kernels, same results)")
    ap.add_argument("--dtype", choices=["f16", "bf16", "f32"], default="f16",
                    help="element type (BASELINE's metric is quoted on f16; bf16 runs the same kernels with the bfloat16 matrix "
                         "instructions; f32 is BASELINE config 2: sm_compress24_f32 + sm_spmma_f32 against the fp32 dense GEMMs, "
                         "default table resnet18)")
    ap.add_argument("--streams", type=int, default=8,
                    help="HIP streams the independent layers of a step are spread over (fork/join inside the step)")
    ap.add_argument("--graphs", choices=["single", "per-stream"], default="single",
                    help="hipGraph form of a step: one graph holding every chain (default) or one linear graph per stream")
    ap.add_argument("--sched", choices=["rr", "split", "lpt"], default="rr",
                    help="layer -> stream assignment: round-robin, or chip-filling layers (>= 784 row tiles) on the first half "
                         "of the streams and the under-filling ones on the second half")
    ap.add_argument("--group", choices=["on", "off"], default="on",
                    help="on (default): the fused layers of one (m, n, k, b) shape run as ONE grouped launch per 8 instances "
                         "(sm_spmma_fused_*_grouped: same kernels, same C bit for bit; the instances share the chip instead of each "
                         "paying its own last partial round of workgroups); off: one launch per layer")
    ap.add_argument("--no-span", action="store_true", help="auto path: k % 64 != 0 layers on sm_compress24 + sm_spmma instead of the span-form fused kernel")
    ap.add_argument("--cost", choices=["bytes", "model"], default="bytes", help="what the longest-first spreading of work items balances")
    ap.add_argument("--item-order", choices=["big-first", "small-first"], default="big-first", help="order of a stream's items")
    ap.add_argument("--big-streams", type=int, default=2, help="--sched split: streams reserved for the chip-filling items")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="predict an N-GPU run on ONE GPU: the data path has no collective, so rank r's time is measurable alone. "
                         "For r = 0..N-1 a fresh child process (started before this process touches a GPU; it never does) runs rank "
                         "r's units of the --scaling plan; the parent reports max_r t_r, sum(flops) / max t and the spread, labelled "
                         "'predicted, single-GPU emulation' (no RCCL, no contention between ranks, one box's clock)")
    ap.add_argument("--emu-rank", type=int, default=None, help=argparse.SUPPRESS)  # child of --emulate-world: whose units to run
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank rehearsal on a ONE-GPU box: gloo backend, every rank on cuda:0 (control flow only; "
                         "the ranks share the device, so the numbers mean nothing)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if args.emulate_world and args.emu_rank is None:
        raise SystemExit(emulate_world(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks.  It has not imported torch and
        # never touches a GPU; the ranks are child processes (no exec), rank 0's JSON line and the exit code are relayed.
        raise SystemExit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # a launcher that realised another rank count than asked for must not pass as an N-GPU run
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}`, which starts its own ranks)")
    if world > 1 and not args.rehearse_gloo and torch.cuda.device_count() < world:
        raise SystemExit(f"bench: {world} ranks asked for, {torch.cuda.device_count()} GPU(s) visible "
                         "(--rehearse-gloo rehearses the control flow on one GPU)")
    plan_world, plan_rank = (args.emulate_world, args.emu_rank) if args.emu_rank is not None else (world, rank)
    if args.scaling is None:
        ntab = len((args.tables or args.table or "x").split(","))
        args.scaling = "weak" if plan_world == 1 else ("lpt" if ntab > 1 else "hybrid")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_gloo:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    sm = ge.load_package()
    mg = ge.load_package_module("multigpu")
    sm.device_check()  # raises when the HIP library or a gfx950 device is missing: no fallback

    f32 = args.dtype == "f32"
    tables = (args.tables or args.table or ("resnet18" if f32 else "resnet50")).split(",")
    tables = [table_path(t) for t in tables]
    shapes = [s_ for t in tables for s_ in read_shapes(t)]
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    # Costs the hybrid plan balances with (round 5): measured HERE, in setup, on the box that runs -- not a table from another box.
    # Every rank must plan with the same numbers: rank 0 measures and broadcasts (setup, not the data path); emulated ranks are
    # separate processes on one GPU and read what the N = 1 child measured (--costs-file).
    plan_costs = {"source": "fallback table (sparsify.me_amd/multigpu.py)"}
    if args.scaling == "hybrid" and plan_world > 1 and args.plan_costs == "measure" and not f32:
        costs = None
        if args.costs_file and os.path.exists(args.costs_file):
            costs = {tuple(int(x) for x in k_.split("x")): tuple(v) for k_, v in json.load(open(args.costs_file)).items()}
            plan_costs = {"source": "measured by the N = 1 child of this emulation (--costs-file)"}
        elif world > 1:
            box = [mg.measure_costs(sm, torch, shapes, tdt) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            costs = box[0]
            plan_costs = {"source": "measured in setup by rank 0 on this node (one grouped launch per unique shape), broadcast to all ranks"}
        if costs:
            mg.set_measured_costs(costs)
            plan_costs["us_per_instance"] = {"x".join(map(str, k_)): round(v[0], 2) for k_, v in costs.items()}
    elif args.costs_file and plan_world == 1 and not f32:   # the N = 1 child of an emulation: measure for the rank children
        costs = mg.measure_costs(sm, torch, shapes, tdt)
        json.dump({"x".join(map(str, k_)): list(v) for k_, v in costs.items()}, open(args.costs_file, "w"))
    units = mg.plan_units(shapes, plan_world, plan_rank, args.scaling)
    es = 4 if f32 else 2

    def build_layers(shapes_, units_):
        """device operands of the units (layer, batch_begin, batch_end): A, B, blob, C per unit, seeded per (layer, global batch index)"""
        layers = []
        for (li, lo, hi) in units_:
            m, n, k, _ = shapes_[li]
            b = hi - lo
            A = torch.empty(b * m * k, dtype=tdt, device=dev)
            B = torch.empty(k * n, dtype=tdt, device=dev)
            for bi in range(lo, hi):  # operand of (layer, global batch index): the same matrix whatever the sharding
                sm.fill_uniform(A[(bi - lo) * m * k:(bi - lo + 1) * m * k], mg.unit_seed(0x5EED0000, li, bi), 0.0, 1.0)
            sm.fill_uniform(B, mg.unit_seed(0xB0000000, li, -1), 0.0, 1.0)
            blob = torch.empty(sm.compress24_size(m, k, es, b), dtype=torch.uint8, device=dev)
            C = torch.empty(b * m * n, dtype=tdt, device=dev)
            S = args.batch_split if args.batch_split > 1 and b % args.batch_split == 0 else 1
            if S == 1:
                layers.append(dict(li=li, m=m, n=n, k=k, b=b, A=A, B=B, blob=blob, C=C))
            else:
                # --batch-split S: the layer's batch as S independent problems of b / S batch entries (views of the same operands; own
                # blobs): more, smaller work items for the streams to interleave -- same kernels, same C
                bs = b // S
                for s_ in range(S):
                    layers.append(dict(li=li, m=m, n=n, k=k, b=bs, A=A[s_ * bs * m * k:(s_ + 1) * bs * m * k], B=B,
                                       blob=torch.empty(sm.compress24_size(m, k, es, bs), dtype=torch.uint8, device=dev),
                                       C=C[s_ * bs * m * n:(s_ + 1) * bs * m * n]))
        return layers
    layers = build_layers(shapes, units)
    flops = mg.unit_flops(shapes, units)

    # The layers of a step are independent problems (the reference's sweep runs them as separate
    # processes, examples/profiling.py:6-17), so a step forks them over a few HIP streams and joins:
    # one layer's ramp-up and tail overlap another layer's streaming phase.  compress -> spmma of one
    # layer stay ordered on one stream.
    side = [torch.cuda.Stream() for _ in range(max(0, args.streams - 1))]
    nstreams = len(side) + 1
    chains = [[] for _ in range(nstreams)]  # stream w runs chains[w] in order
    cnt = [0, 0]
    for i, L in enumerate(layers):
        w = i % nstreams
        if args.sched == "split" and nstreams >= 2:
            big = 0 if L["m"] * L["b"] >= 784 * 128 else 1
            half = [nstreams // 2, nstreams - nstreams // 2]
            w = (0 if big == 0 else half[0]) + cnt[big] % half[big]
            cnt[big] += 1
        chains[w].append(L)
    if args.sched == "lpt":  # longest chain first by the layers' bytes (A + B + C: what the kernels stream), largest layers first
        def cost(L):
            return L["b"] * (L["m"] * L["k"] + L["m"] * L["n"]) + L["k"] * L["n"]
        chains = [[] for _ in range(nstreams)]
        load = [0] * nstreams
        for L in sorted(layers, key=cost, reverse=True):
            w = load.index(min(load))
            chains[w].append(L)
            load[w] += cost(L)

    class Forked:
        """A step whose layers are spread over the streams: fork, one chain of layers per stream, join."""

        def __init__(self, per_layer, only=None):
            self.per_layer = per_layer
            self.only = only  # optional predicate: the layers this pass runs

        def fork_join(self, run_chain):
            main_s = torch.cuda.current_stream()
            for s_ in side:
                s_.wait_stream(main_s)
            run_chain(0)
            for w, s_ in enumerate(side, start=1):
                with torch.cuda.stream(s_):
                    run_chain(w)
            for s_ in side:
                main_s.wait_stream(s_)

        def chain(self, w):
            for L in chains[w]:
                if self.only is None or self.only(L):
                    self.per_layer(L)

        def __call__(self):  # launched kernel by kernel
            self.fork_join(self.chain)

    def layer_staged(L):
        sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])
        sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)

    def use_fused(L):
        if f32:  # sm_spmma_fused_f32: the STRIP rule in the registers of the dense fp32 MFMA kernel (no blob, no compress pass)
            return args.path == "auto" and L["k"] % 32 == 0 and L["n"] % 4 == 0
        if L["n"] < 8 and L["k"] <= 64:  # the thin form (round 5): depthwise layers as im2col products, on the vector ALUs
            return args.path == "auto" and (L["b"] * L["m"] * L["k"] * 2) % 16 == 0
        if L["k"] % 64 != 0:  # the span form of sm_spmma_fused_*: ragged k (the stem layer, k = 147), n <= 128, span + B within the LDS
            return (args.path == "auto" and not args.no_span and L["n"] % 8 == 0 and L["n"] <= 128 and (L["b"] * L["m"] * L["k"] * 2) % 16 == 0 and
                    128 * L["k"] * 2 + 1152 + (L["k"] + 63) // 64 * 64 * (64 if L["n"] <= 64 else 128) * 2 <= 160 * 1024)
        return args.path == "auto" and (L["n"] <= args.fused_max_n or L["k"] <= args.fused_max_k_wide)

    # (f-1) the fused kernel computes the same C bit for bit straight from the dense A (the 2:4 selection
    # and compaction happen in registers / LDS; no blob goes to HBM)
    def layer_path(L):
        if use_fused(L):
            sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])
        else:
            layer_staged(L)

    # Work items of the timed step: with --group on the fused layers of one shape are one item (a grouped launch per 8
    # instances), every other layer is its own item; items are spread over the streams longest-first by their bytes.
    def layer_bytes(L):
        return L["b"] * (L["m"] * L["k"] + L["m"] * L["n"]) + L["k"] * L["n"]

    def fused_groups(Ls):
        """[(shape key, [layers])] of the fused layers in Ls, in table order of first appearance"""
        g = {}
        for L in Ls:
            g.setdefault((L["m"], L["n"], L["k"], L["b"]), []).append(L)
        return list(g.items())

    sk_ws = {}   # shape -> the stream-K workspace of that shape's grouped launches (one per work item: items run concurrently)

    def sk_takes(L0, cnt):
        """does the library run the stream-K form on a grouped launch of `cnt` instances of L0's shape? (its own rule, asked through
        sm_spmma_fused_streamk_plan -- nothing is mirrored here)"""
        if f32 or args.streamk != "on" or L0["k"] % 64 != 0 or L0["n"] <= 128:
            return False
        return sm.spmma_fused_streamk_plan(L0["m"] * L0["b"], L0["n"], L0["k"], cnt)[0]

    def run_group(Ls):
        L0 = Ls[0]
        key = (L0["m"], L0["n"], L0["k"], L0["b"])
        if key not in sk_ws:   # decided once per shape, at the first (untimed) call
            chunks = {min(8, len(Ls) - i) for i in range(0, len(Ls), 8)}
            sk_ws[key] = sm.spmma_fused_workspace() if any(sk_takes(L0, c) for c in chunks) else None
        sm.spmma_fused_grouped([L["A"] for L in Ls], [L["B"] for L in Ls], [L["C"] for L in Ls], L0["m"], L0["n"], L0["k"], batch=L0["b"],
                               workspace=sk_ws[key])

    grouped = args.group == "on" and not f32

    def item_cost(it):
        """what a work item costs a stream: its bytes (--cost bytes) or those bytes over the rate its kernel family reaches
        alone (--cost model: direct 4.9, span 2.9, wide 3.0, A-stationary 3.1 TB/s, profiles/bench_r03k.json)"""
        by = sum(layer_bytes(L) for L in it[1])
        if args.cost == "bytes" or f32:
            return by
        L0 = it[1][0]
        rate = {"direct": 4.9, "span": 2.9, "wide": 3.0, "astat": 3.1, "big": 3.2, "thin": 3.0}[fused_variant(L0["n"], L0["k"])] if use_fused(L0) else 3.0
        return by / rate

    def spread(items):
        """items [(kind, [layers])] -> per-stream chains, longest-first onto the least-loaded stream."""
        items = sorted(items, key=lambda it: -item_cost(it))
        ch, load = [[] for _ in range(nstreams)], [0] * nstreams
        nbig = min(max(1, args.big_streams), nstreams - 1) if (args.sched == "split" and nstreams >= 2) else 0
        for it in items:
            # --sched split: the chip-filling items (>= 784 row tiles per instance: they stream at the HBM rate whatever runs
            # beside them) go to the first `--big-streams` streams, the few-tile items (bound by per-tile latency, they
            # leave CUs idle) to the others, so that a few-tile kernel always has a streaming kernel beside it
            if nbig:
                big = it[1][0]["m"] * it[1][0]["b"] >= 784 * 128
                cand = range(0, nbig) if big else range(nbig, nstreams)
            else:
                cand = range(nstreams)
            w = min(cand, key=lambda c: load[c])
            ch[w].append(it)
            load[w] += item_cost(it)
        if args.item_order == "small-first":  # a stream ends on its chip-filling items: the step's tail is not left to few-tile kernels
            ch = [list(reversed(c)) for c in ch]
        return ch

    class ForkedItems(Forked):
        """A step of work items (a group of same-shape layers launched as one grid, or a single layer) spread over the streams."""

        def __init__(self, ch, run_group_fn, run_single_fn):
            Forked.__init__(self, run_single_fn)
            self.ch, self.run_group_fn = ch, run_group_fn

        def chain(self, w):
            for kind, Ls in self.ch[w]:
                if kind == "group":
                    self.run_group_fn(Ls)
                else:
                    self.per_layer(Ls[0])

    if grouped:
        items = [("group", Ls) for _, Ls in fused_groups([L for L in layers if use_fused(L)])]
        items += [("single", [L]) for L in layers if not use_fused(L)]
        step_full = ForkedItems(spread(items), run_group, layer_path)
        n_launch_groups = sum((len(Ls) + 7) // 8 for kind, Ls in items if kind == "group")
    else:
        step_full = Forked(layer_path)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_runner(fn):
        """fn replayed from a hipGraph (one graph = one call of fn), or fn itself with --eager.  --graphs per-stream
        turns a Forked step into one linear graph per stream, replayed into its own stream between the fork and the
        join (measured the same as the single graph, profiles/ktrace_r01l.txt)."""
        fn()  # first call outside capture: lazy module loads, function attributes
        torch.cuda.synchronize()
        if args.eager:
            return fn
        try:
            if isinstance(fn, Forked) and args.graphs == "per-stream":
                graphs = []
                for w in range(nstreams):
                    gw = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gw, stream=torch.cuda.Stream()):
                        fn.chain(w)
                    graphs.append(gw)
                return lambda: fn.fork_join(lambda w: graphs[w].replay())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.Stream()):
                fn()
            return g.replay
        except Exception as e:  # capture unsupported: fall back to eager launches, and say so
            sys.stderr.write(f"bench: hipGraph capture failed ({e}); launching eagerly\n")
            return fn

    def timed(run, steps, warmup, collective=True):
        """Wall time of `steps` calls.  collective=False: rank-local timing (the per-stage / per-family passes only rank
        0 runs: a distributed barrier there would pair with the other ranks' final barrier and hang the job)."""
        sync = barrier if collective else torch.cuda.synchronize
        for _ in range(warmup):
            run()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        sync()
        return wall

    def event_seconds(run, reps, warmup=2):
        """Device time per call by a HIP event pair on the stream the work is launched on (torch's current stream: the
        graph replays and the eager launches both go there)."""
        for _ in range(warmup):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    run_full = make_runner(step_full)
    if args.settle_ms > 0:  # part of the setup, like the buffer fills and the graph capture: not one of the K timed steps
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                run_full()
            torch.cuda.synchronize()
    wall = timed(run_full, args.steps, args.warmup)
    tot_flops, wall_max = mg.rollup(flops * args.steps, wall, None if args.rehearse_gloo else dev)
    ms_per_step = wall_max / args.steps * 1e3
    value = tot_flops / wall_max / 1e9

    # BASELINE config 4 in the same run (every rank takes part: the timed loop is bracketed by collectives): the three ResNet tables
    # as 300 whole layers placed by LPT, launched like the step (grouped, over the streams).  north_star's ">= 7x aggregate at 8 GPUs"
    # is stated on this sweep; with it in every N > 1 line the first real SCALE run reports it beside the one-table headline.
    config4 = None
    want_c4 = args.config4_stage == "on" or (args.config4_stage == "auto" and world > 1)
    if want_c4 and grouped and not f32 and args.emu_rank is None:
        shapes4 = [s_ for t in ("resnet50", "resnet101", "resnet152") for s_ in read_shapes(table_path(t))]
        units4 = mg.plan_units(shapes4, world, rank, "lpt")
        layers4 = build_layers(shapes4, units4)
        items4 = [("group", Ls) for _, Ls in fused_groups([L for L in layers4 if use_fused(L)])] + [("single", [L]) for L in layers4 if not use_fused(L)]
        run4 = make_runner(ForkedItems(spread(items4), run_group, layer_path))
        steps4 = max(3, args.steps // 4)
        wall4 = timed(run4, steps4, 2)
        f4, w4 = mg.rollup(mg.unit_flops(shapes4, units4) * steps4, wall4, None if args.rehearse_gloo else dev)
        config4 = {"workload": "datasets/resnet50.csv + resnet101.csv + resnet152.csv: %d layer instances at b = %d, whole layers by LPT over %d rank(s)" % (len(shapes4), shapes4[0][3], world),
                   "partition_mode": "lpt", "layers_this_rank": len(layers4), "steps": steps4, "ms_per_step": w4 / steps4 * 1e3, "value": f4 / w4 / 1e9, "unit": "GF/s",
                   "n_gpus": world, "dense_equiv_gflop_per_step": f4 / steps4 / 1e9,
                   "how": "same process, same ranks, after the headline loop; max over ranks of the wall time, one all-reduce of {flops, seconds}"}
        del layers4, items4, run4

    names = ",".join(os.path.basename(t) for t in tables)
    nfused = sum(use_fused(L) for L in layers)
    sfx = args.dtype
    if f32:
        path_desc = ("auto: sm_spmma_fused_f32 on %d layers (k %% 32 == 0), sm_compress24_f32 + sm_spmma_f32 on %d" % (nfused, len(layers) - nfused)
                     if args.path == "auto" else "staged: sm_compress24_f32 + sm_spmma_f32 on every layer")
    elif args.path == "auto":
        path_desc = ("auto: sm_spmma_fused_%s on %d layers (n <= %d or k <= %d), sm_compress24_%s + sm_spmma_%s on %d"
                     % (sfx, nfused, args.fused_max_n, args.fused_max_k_wide, sfx, sfx, len(layers) - nfused))
    else:
        path_desc = "staged: sm_compress24_%s + sm_spmma_%s on every layer" % (sfx, sfx)
    split = {"weak": f"every rank runs all {len(shapes)} layers on its own batch (rank r = global batch indices [r*b, (r+1)*b))",
             "strong": f"batch split: rank g runs batch indices [g*b/{world}, (g+1)*b/{world}) of every layer, B replicated",
             "lpt": f"whole layers by longest-processing-time over {len(shapes)} layer instances",
             "hybrid": f"batch split [g*b/{world}, (g+1)*b/{world}) of the layers whose per-rank share keeps >= {mg.HYBRID_FILL_ROWS} rows, "
                       f"the other layers whole, longest modelled time first (this rank: {len(layers)} units)"}[args.scaling]
    if grouped:
        path_desc += ("; the fused layers run as %d grouped launches (sm_spmma_fused_%s_grouped: one grid per <= 8 same-shape instances, "
                      "same kernels, same C)" % (n_launch_groups, sfx))
    out = {
        "metric": "effective GF/s (2:4 spmma vs dense gemm) on ResNet-50 layer shapes",
        "value": value, "unit": "GF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.scaling == "weak" else "strong",
        "partition_mode": args.scaling,  # the real mode name (`scaling` keeps the contract's two values: hybrid / lpt fix the total work = strong)
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "datasets/%s: %d conv layers as im2col GEMMs (m,n,k) at b=%d, %s; step = per layer "
                               "2:4 prune(STRIP)+compress+matmul of the unpruned A"
                               % (names, len(shapes), shapes[0][3], {"f16": "fp16", "bf16": "bfloat16", "f32": "fp32"}[args.dtype]),
                   "path": path_desc,
                   "layers": len(shapes), "layers_this_rank": len(layers), "batch": shapes[0][3],
                   "dense_equiv_gflop_per_step": tot_flops / args.steps / 1e9,
                   "launch": ("eager" if args.eager else ("hipGraph replay, one linear graph per stream" if args.graphs == "per-stream" else "hipGraph replay of one step"))
                             + (f"; setup runs {args.settle_ms:.0f} ms of untimed replays before the W warm-up steps" if args.settle_ms > 0 else ""),
                   "streams": args.streams, "sched": args.sched, "partition": args.scaling,
                   "library": library_tag(sm),
                   "parallelism": f"{args.scaling} x{world}: {split}; no data-path collective, one all-reduce of "
                                  "{sum flops, max seconds}"},
    }

    # What was timed, checked (not timed): C of one layer per kernel family, as the last timed step left it, against
    # sm_compress24 + sm_spmma on the same operands, bit for bit (the fused kernels' contract; tests/test_gpu_parity.py holds
    # the full matrix of cases).  A mismatch fails the run: a fast step with different results is not a measurement.
    if not f32 and args.path == "auto":
        checked, ok = [], True
        seen = set()
        for L in layers:
            if not use_fused(L):
                continue
            cnt_ = min(8, sum(1 for X in layers if (X["m"], X["n"], X["k"], X["b"]) == (L["m"], L["n"], L["k"], L["b"]))) if args.group == "on" else 1
            fam = fused_variant(L["n"], L["k"], L["m"], L["b"], cnt_)
            if grouped and sk_ws.get((L["m"], L["n"], L["k"], L["b"])) is not None and sk_takes(L, cnt_):
                fam = "sk"
            if fam in seen:
                continue
            seen.add(fam)
            Cref = torch.empty_like(L["C"])
            sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])
            sm.spmma(L["blob"], L["B"], Cref, L["m"], L["n"], L["k"], L["b"], 0)
            if fam == "thin":
                # the thin form (vector ALUs): the staged pair's products in another order of fp32 additions -- within one rounding of it
                af, bf_ = Cref.float(), L["C"].float()
                rel = float(((af - bf_).abs() / torch.maximum(af.abs(), bf_.abs()).clamp_min(2.0 ** -14)).max().item())
                same = rel <= 2.0 ** -9
                checked.append({"family": fam, "m": L["m"], "n": L["n"], "k": L["k"], "b": L["b"], "max_relative_difference_vs_compress_plus_spmma": rel,
                                "within_one_fp16_rounding": same})
            elif fam == "sk":
                # the stream-K form: the tiles its plan cuts are sums of fp32 partials in a fixed order -- equal to compress + spmma to
                # one fp16 rounding of the result (+ the few fp32 re-associations); every row panel the plan leaves whole: bit for bit
                _, plan = sm.spmma_fused_streamk_plan(L["m"] * L["b"], L["n"], L["k"], cnt_)
                M_, nkt_ = L["m"] * L["b"], L["k"] // 64
                tm_ = (M_ + 255) // 256
                inst = [X for X in layers if (X["m"], X["n"], X["k"], X["b"]) == (L["m"], L["n"], L["k"], L["b"])].index(L) % 8
                whole = [t - inst * tm_ for t in sm.streamk_whole_panels(plan, cnt_ * tm_, nkt_) if inst * tm_ <= t < (inst + 1) * tm_]
                a_, b_ = Cref.view(M_, L["n"]), L["C"].view(M_, L["n"])
                same_whole = all(bool(torch.equal(a_[t * 256:(t + 1) * 256].view(torch.int16), b_[t * 256:(t + 1) * 256].view(torch.int16))) for t in whole)
                af, bf_ = a_.float(), b_.float()
                rel = float(((af - bf_).abs() / torch.maximum(af.abs(), bf_.abs()).clamp_min(2.0 ** -14)).max().item())
                same = same_whole and rel <= 2.0 ** -9   # two neighbouring fp16 values of the larger magnitude differ by at most 2^-10 of it
                checked.append({"family": fam, "m": L["m"], "n": L["n"], "k": L["k"], "b": L["b"], "whole_row_panels": len(whole),
                                "whole_panels_bit_identical_to_compress_plus_spmma": same_whole,
                                "cut_tiles_max_relative_difference": rel, "within_one_fp16_rounding": rel <= 2.0 ** -9})
            else:
                same = bool(torch.equal(Cref.view(torch.int16), L["C"].view(torch.int16)))
                checked.append({"family": fam, "m": L["m"], "n": L["n"], "k": L["k"], "b": L["b"], "bit_identical_to_compress_plus_spmma": same})
            ok = ok and same
            del Cref
        out["verified"] = ok
        out["verified_layers"] = checked
        if not ok:
            sys.stderr.write("bench: the timed step's C differs from compress + spmma: " + json.dumps(checked) + "\n")
            raise SystemExit(4)
    out["config"]["plan_costs"] = plan_costs
    if config4 is not None:
        out.setdefault("stages", {})["config4_sweep"] = config4
    if args.emu_rank is not None:
        out["emulated"] = {"world": plan_world, "rank": plan_rank, "units": len(units), "dense_equiv_gflop_per_step": flops / 1e9}
    if rank == 0 and not args.no_extras:
        extras(args, sm, torch, dev, layers, flops, wall / args.steps, Forked, make_runner, timed, event_seconds, use_fused, out,
               (fused_groups, run_group, spread, ForkedItems, sk_takes) if grouped else None, step_full)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ge, shapes)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


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
        if a in ("--emulate-world", "--scaling", "--gpus"):
            skip = 1
            continue
        if a.startswith(("--emulate-world=", "--scaling=", "--gpus=")):
            continue
        argv.append(a)

    def child(extra):
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-extras", "--no-cpu-baseline"] + extra
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
    per = {r: child(["--scaling", mode, "--emulate-world", str(N), "--emu-rank", str(r)]) for r in ranks}
    ms = [per[r if not identical else 0]["ms_per_step"] for r in range(N)]
    gf = [per[r if not identical else 0]["emulated"]["dense_equiv_gflop_per_step"] for r in range(N)]
    tmax = max(ms)
    total = sum(gf)
    out = {"metric": base["metric"], "value": total / (tmax * 1e-3), "unit": "GF/s", "n_gpus": N,
           "label": "predicted, single-GPU emulation: every rank's units ran ALONE on one MI355X, each in a fresh process; no RCCL, no "
                    "contention between ranks, one box's clock -- not a measured N-GPU run",
           "predicted": True, "partition_mode": mode, "scaling": "weak" if mode == "weak" else "strong",
           "per_rank_ms": ms, "per_rank_gflop": gf, "max_ms": tmax, "min_ms": min(ms),
           "spread": (max(ms) - min(ms)) / (sum(ms) / len(ms)),
           "ranks_measured": ranks, "ranks_identical_by_construction": identical,
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
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
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


def extras(args, sm, torch, dev, layers, flops, t_full, Forked, make_runner, timed, event_seconds, use_fused, out, grouping=None, step_full=None):
    """Rank 0 only: per-stage times, the dense denominators, the API-faithful sequence and the roofline of the
    dominant kernel family."""
    f32 = args.dtype == "f32"
    s = 4 if f32 else 2
    R = max(5, args.steps)

    def sec_per_call(fn):
        return timed(make_runner(fn), R, 2, collective=False) / R

    gfs = lambda t: flops / t / 1e9
    spmma_only = Forked(lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0))
    compress_only = Forked(lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]))
    dense_rowmajor = Forked(lambda L: sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]))

    # the reference's dense path: column-major pointer-array batched GEMM, B shared (examples/gemm.cu:60,86)
    for L in layers:
        m, n, k, b = L["m"], L["n"], L["k"], L["b"]
        L["Ap"] = torch.tensor([L["A"].data_ptr() + s * i * m * k for i in range(b)], dtype=torch.int64, device=dev)
        L["Bp"] = torch.tensor([L["B"].data_ptr()] * b, dtype=torch.int64, device=dev)
        L["Cp"] = torch.tensor([L["C"].data_ptr() + s * i * m * n for i in range(b)], dtype=torch.int64, device=dev)
    has_batched = args.dtype in ("f16", "f32")  # cublas{H,S}gemmBatched's role; no bf16 form in the reference
    dense_batched = Forked(lambda L: sm.gemm_batched(L["Ap"], L["Bp"], L["Cp"], L["m"], L["n"], L["k"], L["b"], args.dtype))

    t_mul, t_cmp = sec_per_call(spmma_only), sec_per_call(compress_only)
    t_drm = sec_per_call(dense_rowmajor)
    # The dense comparator given the treatment the timed step gets (--group on): the instances of one shape as ONE grid.  The
    # row-major product of the (b*m) x k stacked operand IS the column-major pointer-array entry with the operands swapped
    # (C^T = B^T A^T: the same kernel, sm_gemm_batched_* maps it back), so a group is one call with `count` pointer triples.
    t_drm_grouped = None
    if grouping and has_batched:
        fused_groups, _, spread, ForkedItems = grouping[:4]
        ditems = []
        dense_ws = {}   # the dense twin gets the stream-K workspace too (sm_gemm_batched_f16_ws): the library's rule decides, as for the 2:4 launches
        for _, Ls in fused_groups(layers):
            if len(Ls) == 1:
                ditems.append(("single", Ls))
                continue
            for L in Ls[:1]:
                L["gAp"] = torch.tensor([x["A"].data_ptr() for x in Ls], dtype=torch.int64, device=dev)
                L["gBp"] = torch.tensor([x["B"].data_ptr() for x in Ls], dtype=torch.int64, device=dev)
                L["gCp"] = torch.tensor([x["C"].data_ptr() for x in Ls], dtype=torch.int64, device=dev)
            ditems.append(("group", Ls))

        def dense_group(Ls):
            L0 = Ls[0]
            key = (L0["m"], L0["n"], L0["k"])
            if key not in dense_ws:
                dense_ws[key] = sm.spmma_fused_workspace() if (args.streamk == "on" and args.dtype == "f16" and 128 < L0["n"] <= 256 and L0["k"] >= 2048) else None
            sm.gemm_batched(L0["gBp"], L0["gAp"], L0["gCp"], L0["n"], L0["m"] * L0["b"], L0["k"], len(Ls), args.dtype, workspace=dense_ws[key])
        t_drm_grouped = sec_per_call(ForkedItems(spread(ditems), dense_group,
                                                 lambda L: sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])))
    # the 2:4 matmul on prepared blobs given the same treatment (round 4: sm_spmma_*_grouped, one grid per <= 8 same-shape blobs)
    t_mul_grouped = None
    if grouping and not f32 and hasattr(sm, "spmma_grouped"):
        fused_groups, _, spread, ForkedItems = grouping[:4]
        mitems = [("group" if len(Ls) > 1 else "single", Ls) for _, Ls in fused_groups(layers)]

        def mul_group(Ls):
            L0 = Ls[0]
            sm.spmma_grouped([x["blob"] for x in Ls], [x["B"] for x in Ls], [x["C"] for x in Ls], L0["m"], L0["n"], L0["k"], batch=L0["b"])
        t_mul_grouped = sec_per_call(ForkedItems(spread(mitems), mul_group,
                                                 lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
    t_dcm = sec_per_call(dense_batched) if has_batched else None
    t_staged = t_full if args.path == "staged" else sec_per_call(Forked(lambda L: (
        sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
        sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0))))
    prev_stages = out.get("stages", {})
    out["stages"] = {
        "spmma_mul_gfs": gfs(t_mul), "spmma_mul_ms": t_mul * 1e3, "compress_ms": t_cmp * 1e3,
        "dense_gemm_rowmajor_gfs": gfs(t_drm), "dense_gemm_rowmajor_ms": t_drm * 1e3,
        "dense_gemm_batched_colmajor_gfs": gfs(t_dcm) if t_dcm else None, "dense_gemm_batched_colmajor_ms": t_dcm * 1e3 if t_dcm else None,
        "speedup_mul_vs_dense_rowmajor": t_drm / t_mul, "speedup_mul_vs_dense_batched": t_dcm / t_mul if t_dcm else None,
        "speedup_full_vs_dense_rowmajor": t_drm / t_full, "speedup_full_vs_dense_batched": t_dcm / t_full if t_dcm else None,
        "dense_gemm_rowmajor_grouped_ms": t_drm_grouped * 1e3 if t_drm_grouped else None,
        "dense_gemm_rowmajor_grouped_gfs": gfs(t_drm_grouped) if t_drm_grouped else None,
        "speedup_full_vs_dense_rowmajor_grouped": t_drm_grouped / t_full if t_drm_grouped else None,
        "speedup_mul_vs_dense_rowmajor_grouped": t_drm_grouped / t_mul if t_drm_grouped else None,
        "spmma_mul_grouped_ms": t_mul_grouped * 1e3 if t_mul_grouped else None,
        "spmma_mul_grouped_gfs": gfs(t_mul_grouped) if t_mul_grouped else None,
        "speedup_mul_grouped_vs_dense_rowmajor_grouped": t_drm_grouped / t_mul_grouped if t_drm_grouped and t_mul_grouped else None,
        "speedup_mul_grouped_vs_dense_batched": t_dcm / t_mul_grouped if t_dcm and t_mul_grouped else None,
        "full_path_staged_gfs": gfs(t_staged), "full_path_staged_ms": t_staged * 1e3,
        "timed_path": args.path, "timed_path_ms": t_full * 1e3,
        # what 2:4 can buy on these shapes when both products are HBM-bound (fp16: they are): the ratio of
        # the algorithmic bytes, dense (A + B + C) over sparse (9/16 A + B + C)
        "hbm_bound_speedup_ceiling": sum(L["b"] * s * (L["m"] * L["k"] + L["m"] * L["n"]) + s * L["k"] * L["n"] for L in layers)
        / sum(L["b"] * (L["m"] * L["k"] * (s / 2 + 1.0 / 8) + s * L["m"] * L["n"]) + s * L["k"] * L["n"] for L in layers),
    }
    out["stages"].update(prev_stages)

    # What the per-step join costs (NOT the headline: `value` keeps one fork / join per step).  The steps are independent
    # batches; here four of them are captured as one graph in which every stream runs its chain four times back to back --
    # a layer's consecutive executions stay ordered on their stream, nothing waits for another stream between steps -- so one
    # step's ramp-up and tail (<= 2 kernels active for ~ 25 % of a replayed step, profiles/ktrace_r03f.txt) overlap its neighbours.
    if step_full is not None and not args.eager:
        class Pipelined(object):
            def __call__(self):
                step_full.fork_join(lambda w: [step_full.chain(w) for _ in range(4)])
        t_pipe = timed(make_runner(Pipelined()), max(2, R // 2), 2, collective=False) / max(2, R // 2) / 4.0
        out["stages"]["pipelined_4_steps_ms_per_step"] = t_pipe * 1e3
        out["stages"]["pipelined_4_steps_gfs"] = gfs(t_pipe)
        out["stages"]["pipelined_note"] = ("four steps per graph replay, no cross-stream join between them (per-stream order kept); "
                                           "reported beside the headline, which joins every step")

    # The API-faithful sequence of sparsifyme::spmma() (reference spmma.hxx:82-113, include/sparsify.me/spmma.hxx):
    # TILE prune -> prune check -> compress -> multiply.  The prune reads the step's dense A and writes the pruned
    # operand to a second buffer (the bytes of the in-place prune, without turning the bench's operand into an already
    # pruned one for the next step).
    if hasattr(sm, "api_spmma_step"):  # (fp32 too since round 3: sm_prune24_compress24_f32)
        valid = torch.zeros(1, dtype=torch.int32, device=dev)
        for L in layers:
            L["Aapi"] = torch.empty_like(L["A"])
        t_api = sec_per_call(Forked(lambda L: sm.api_spmma_step(L["A"], L["Aapi"], L["B"], L["C"], L["blob"], valid, L["m"], L["n"], L["k"], L["b"])))
        out["stages"]["api_spmma_ms"] = t_api * 1e3
        out["stages"]["api_spmma_gfs"] = gfs(t_api)
        out["stages"]["api_spmma_sequence"] = sm.API_SPMMA_SEQUENCE
        if not f32 and hasattr(sm, "api_spmma_step_fused"):
            # round 4: the same sequence as ONE kernel (sm_prune24_spmma_*: TILE prune written to the second buffer, flag, multiply,
            # no blob) on the layers it takes (n <= 128, k % 64 == 0, m % 4 == 0); the two-launch pair on the others
            t_api1 = sec_per_call(Forked(lambda L: sm.api_spmma_step_fused(L["A"], L["Aapi"], L["B"], L["C"], L["blob"], valid, L["m"], L["n"], L["k"], L["b"])))
            n_one = sum(1 for L in layers if L["n"] <= 128 and L["n"] % 8 == 0 and L["k"] % 64 == 0 and L["m"] % 4 == 0)
            out["stages"]["api_spmma_one_kernel_ms"] = t_api1 * 1e3
            out["stages"]["api_spmma_one_kernel_gfs"] = gfs(t_api1)
            out["stages"]["api_spmma_one_kernel_layers"] = n_one
        for L in layers:
            del L["Aapi"]

    if f32 and hasattr(sm, "spmma_fused_f32_split"):
        # round 4: the fp32 2:4 product on the SPARSE matrix instruction through exact bfloat16 splits of both operands
        # (sm_spmma_fused_f32_split; planes = 3: |error| <= 2^-21 sum|a||b|, planes = 2: 2^-13) where it applies (k % 64 == 0,
        # n % 8 == 0), the exact fused kernel elsewhere.  Reported BESIDE the headline, which stays the exact fp32 form.
        split = {"kernel": "spmma_f32_split_kernel (v_smfmac_f32_16x16x64_bf16 on three / two truncated bfloat16 pieces per fp32 value, fp32 "
                           "accumulation; mask = the exact path's) + split_planes_kernel (B's pieces, once per call, into a workspace)"}
        for L in layers:
            L["ws"] = torch.empty(max(16, sm.spmma_fused_f32_split_workspace(L["n"], L["k"], planes=3)), dtype=torch.uint8, device=dev)
        for planes in (3, 2):
            def layer_split(L, planes=planes):
                if sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False) != 0:
                    (sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]) if use_fused(L) else
                     (sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                      sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
            t_sp = sec_per_call(Forked(layer_split))
            key = "planes%d" % planes
            split[key + "_ms"] = t_sp * 1e3
            split[key + "_gfs"] = gfs(t_sp)
            split[key + "_speedup_vs_dense_rowmajor"] = t_drm / t_sp
            split[key + "_speedup_vs_exact_fused"] = t_full / t_sp
            split[key + "_hbm_frac"] = sum(L["b"] * 4 * (L["m"] * L["k"] + L["m"] * L["n"]) + 4 * L["k"] * L["n"] for L in layers) / t_sp / (HBM_PEAK_GBS * 1e9)
        # B's planes kept across calls (round 5: sm_spmma_fused_f32_split_prepare once, untimed -- B is the layer's weights in the reference's
        # use -- then sm_spmma_fused_f32_split_prepared per step): the same C bit for bit, without the per-call pass over B
        if hasattr(sm, "spmma_fused_f32_split_prepared"):
            for planes in (3, 2):
                for L in layers:
                    L["prep"] = sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False) == 0
                    if L["prep"]:
                        sm.spmma_fused_f32_split_prepare(L["B"], L["n"], L["k"], L["ws"], planes=planes)
                def layer_prepared(L, planes=planes):
                    if L["prep"]:
                        sm.spmma_fused_f32_split_prepared(L["A"], L["ws"], L["C"], L["m"], L["n"], L["k"], batch=L["b"], planes=planes)
                    else:
                        (sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]) if use_fused(L) else
                         (sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                          sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
                t_pp = sec_per_call(Forked(layer_prepared))
                split["planes%d_prepared_ms" % planes] = t_pp * 1e3
                split["planes%d_prepared_speedup_vs_dense_rowmajor" % planes] = t_drm / t_pp
        # the dense product by the same pieces (sm_gemm_rowmajor_f32_split): what the 2:4 split form should be held against
        for planes in (3, 2):
            def layer_dense_split(L, planes=planes):
                if sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False, dense=True) != 0:
                    sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])
            t_ds = sec_per_call(Forked(layer_dense_split))
            split["dense_planes%d_ms" % planes] = t_ds * 1e3
            split["planes%d_speedup_vs_dense_split" % planes] = t_ds / (split["planes%d_ms" % planes] * 1e-3)
        # the API-faithful sequence of spmma<float> with spmma_options().f32_planes: TILE prune in place (here: into a second buffer, as
        # stages.api_spmma_ms does) + check in one pass, no blob, then the split multiply straight from the pruned dense operand
        for L in layers:
            L["Aapi"] = torch.empty_like(L["A"])
        vflag = torch.zeros(1, dtype=torch.int32, device=dev)
        for planes in (3, 2):
            def layer_api_split(L, planes=planes):
                sm.prune24_compress24(L["A"], L["Aapi"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], None, vflag, sm.PRUNE_TILE)
                if sm.spmma_fused_f32_split(L["Aapi"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False) != 0:
                    sm.compress24(L["Aapi"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])
                    sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)
            split["api_spmma_planes%d_ms" % planes] = sec_per_call(Forked(layer_api_split)) * 1e3
        for L in layers:
            del L["Aapi"]
        split["layers_on_split_form"] = sum(1 for L in layers if sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"],
                                                                                      planes=3, check=False) == 0)
        # error of the split forms against the exact kernel on the first layer they take (max |diff| / max sum|a||b| bound proxy)
        L = next((L for L in layers if L["k"] % 64 == 0 and L["n"] % 8 == 0), None)
        if L is not None:
            Ce = torch.empty_like(L["C"])
            sm.spmma_fused(L["A"], L["B"], Ce, L["m"], L["n"], L["k"], batch=L["b"])
            for planes in (3, 2):
                sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes)
                torch.cuda.synchronize()
                d = (L["C"].double() - Ce.double()).abs().max().item()
                split["planes%d_max_abs_diff_vs_exact" % planes] = d
                split["planes%d_max_rel_diff_vs_exact" % planes] = d / max(Ce.double().abs().max().item(), 1e-30)
            split["diff_layer"] = [L["m"], L["n"], L["k"], L["b"]]
            del Ce
        for L in layers:
            del L["ws"]
        out["stages"]["f32_split"] = split

    if not f32 and os.path.basename(args.tables.split(",")[0] if args.tables else (args.table or "resnet50")).startswith("resnet50"):
        out["stages"]["conv_path"] = conv_path_stage(sm, torch, dev, args.dtype)
        if grouping and len(layers) == 49:
            out["stages"]["conv_step"] = conv_step_stage(args, sm, torch, dev, layers, grouping, make_runner, event_seconds, out["stages"])
        out["stages"]["config5_coo_spmm"] = config5_stage(sm, torch, dev)
        out["stages"]["bell_spmm"] = bell_stage(sm, torch, dev)
    if not args.no_cpu_baseline:
        out["stages"]["config1_cpu"] = config1_cpu(ge_mod())

    if not f32:
        # matrix-pipe view of the 2:4 matmul (north_star: "MFMA utilisation for the matmul against chip peak"):
        # dense-equivalent rate of the matmul-only pass against 2 x the dense fp16 peak (v_smfmac does a 16x16x64
        # product in the cycles of a dense 16x16x32), plus the PMC MfmaUtil per kernel when a profile is present
        mfma = {"achieved_TFs": gfs(t_mul) / 1e3, "peak_TFs": 2.0 * 2500.0, "frac": gfs(t_mul) / 1e3 / 5000.0,
                "peak": "2 x 2.5 PF/s dense fp16 (MI355X_MICROARCH.md); the v_smfmac issue rate measured on this chip "
                        "is 3.4-3.8 PF/s dense-equivalent (profiles/mfma_rate_r01.txt)",
                "pmc_mfma_util_percent": None, "pmc_source": None}
        mpath = os.path.join(ROOT, "profiles", "mfma_util_latest.json")  # tools/pmc_mfma.py, from a rocprofv3 --pmc pass
        if os.path.exists(mpath):
            try:
                mtab = json.load(open(mpath))
                mfma["pmc_source"] = file_tag(mpath, (mtab.get("_library") or {}).get("sha256_16"), out["config"]["library"]["sha256_16"])
                if not mfma["pmc_source"]["stale"]:   # counters of another build are not this build's utilisation
                    mfma["pmc_mfma_util_percent"] = {k: round(v["mfma_util_percent"], 2) for k, v in mtab.items() if not k.startswith("_")}
            except Exception:
                pass
        out["stages"]["matmul_mfma"] = mfma

    # roofline of the dominant kernel family of the timed step: algorithmic bytes / device
    # time (HIP events on the launch stream) of a single-stream pass that launches only that family on its layers
    A_sp = lambda L: L["b"] * (L["m"] * L["k"] * s / 2 + L["m"] * L["k"] / 8 + L["m"] * L["n"] * s) + s * L["k"] * L["n"]
    A_fu = lambda L: L["b"] * s * (L["m"] * L["k"] + L["m"] * L["n"]) + s * L["k"] * L["n"]
    fam = {}
    if f32:
        staged = [L for L in layers if not use_fused(L)]
        fam["spmma_f32"] = dict(names=["spmma_f32_dma_kernel", "spmma_f32_kernel"], layers=staged,
                                call=lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0), bytes=A_sp)
        fam["compress"] = dict(names=["compress_kernel"], layers=staged,
                               call=lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                               bytes=lambda L: L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8))
        fam["spmma_f32_fused"] = dict(names=["gemm_f32_dma_kernel"], layers=[L for L in layers if use_fused(L)],
                                      call=lambda L: sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]), bytes=A_fu)
    else:
        staged = [L for L in layers if not use_fused(L)]
        fam["spmma_f16"] = dict(names=["spmma_f16_dma_kernel", "spmma_f16_pc_kernel", "spmma_f16_kernel", "spmma_f16_splitk_kernel"],
                                layers=staged, call=lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0), bytes=A_sp)
        fam["compress"] = dict(names=["compress_flat_kernel", "compress_rowspan_f16_kernel", "compress_kernel"], layers=staged,
                               call=lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                               bytes=lambda L: L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8))
        shape_count = {}
        for L in layers:
            if use_fused(L):
                key = (L["m"], L["n"], L["k"], L["b"])
                shape_count[key] = shape_count.get(key, 0) + 1

        def variant_of(L):  # the kernel the timed step's (grouped) launch of this layer's shape runs
            cnt = min(8, shape_count[(L["m"], L["n"], L["k"], L["b"])]) if grouping else 1
            if grouping and grouping[4](L, cnt):   # the library's own rule (sm_spmma_fused_streamk_plan): the stream-K form
                return "sk"
            return fused_variant(L["n"], L["k"], L["m"], L["b"], cnt)
        for var in ("direct", "big", "wide", "astat", "span", "sk", "thin"):
            fam["spmma_f16_fused_" + var] = dict(names=["spmma_f16_thin_kernel"] if var == "thin" else ["spmma_f16_fused_%s_kernel" % var] + (["spmma_f16_fused_widep_kernel"] if var == "wide" else []),
                                                 layers=[L for L in layers if use_fused(L) and variant_of(L) == var],
                                                 call=lambda L: sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]),
                                                 bytes=A_fu)
    traffic_tab, tsrc = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")  # tools/pmc_traffic.py, from rocprofv3 --pmc passes
    if os.path.exists(tpath):
        try:
            traffic_tab = json.load(open(tpath))
            tsrc = file_tag(tpath, (traffic_tab.get("_library") or {}).get("sha256_16"), out["config"]["library"]["sha256_16"])
            if tsrc["stale"]:   # measured on another build of the library: not replayed (roofline.traffic stays null, the tag says why)
                traffic_tab = {}
        except Exception:
            traffic_tab = {}
    rows = {}
    for name, f in fam.items():
        if not f["layers"]:
            continue

        nlaunch = len(f["layers"])
        if grouping and name.startswith("spmma_f16_fused"):  # the family as the timed step launches it: one grid per <= 8 instances of a shape
            gl = grouping[0](f["layers"])
            nlaunch = sum((len(Ls) + 7) // 8 for _, Ls in gl)

            def serial(gl=gl):
                for _, Ls in gl:
                    grouping[1](Ls)
        else:
            def serial(f=f):
                for L in f["layers"]:
                    f["call"](L)
        t = event_seconds(make_runner(serial), R)
        by = sum(f["bytes"](L) for L in f["layers"])
        fl = sum(2.0 * L["m"] * L["n"] * L["k"] * L["b"] for L in f["layers"])
        tb = [(traffic_tab[n]["hbm_bytes_per_launch"], traffic_tab[n]["launches_profiled"]) for n in f["names"] if n in traffic_tab]
        traffic = sum(b_ * c_ for b_, c_ in tb) / sum(c_ for _, c_ in tb) if tb else None
        rows[name] = dict(seconds=t, launches=nlaunch, layers=len(f["layers"]), bytes=by, GBs=by / t / 1e9, TFs=fl / t / 1e12, traffic=traffic)
    # the fused variants are one family for the "dominant kernel" choice (they are one entry point), reported each
    groups = {}
    for n_, r_ in rows.items():
        groups.setdefault("spmma_f16_fused" if n_.startswith("spmma_f16_fused") else n_, []).append(r_)
    gsum = {g: dict(seconds=sum(r["seconds"] for r in rs), launches=sum(r["launches"] for r in rs), bytes=sum(r["bytes"] for r in rs),
                    TFs=None, traffic=(sum(r["traffic"] * r["launches"] for r in rs) / sum(r["launches"] for r in rs)
                                       if all(r["traffic"] is not None for r in rs) else None)) for g, rs in groups.items()}
    dom = max(gsum, key=lambda g: gsum[g]["seconds"])
    d = gsum[dom]
    fams_out = {n_: {"ms_per_step": r_["seconds"] * 1e3, "launches": r_["launches"], "layers": r_["layers"], "GBs": r_["GBs"], "frac_of_hbm_peak": r_["GBs"] / HBM_PEAK_GBS,
                     "hbm_traffic_per_launch": r_["traffic"]} for n_, r_ in rows.items()}
    if f32:
        domf = max((n_ for n_ in rows if n_.startswith("spmma_f32")), key=lambda n_: rows[n_]["seconds"])
        r_ = rows[domf]
        out["roofline"] = {"bound": "mfma", "achieved": r_["TFs"], "peak": F32_MATRIX_PEAK_TFS, "unit": "TFLOP/s",
                           "frac": r_["TFs"] / F32_MATRIX_PEAK_TFS, "traffic": r_["traffic"], "traffic_source": tsrc if (r_["traffic"] is not None or (tsrc and tsrc["stale"])) else None,
                           "kernel": domf,
                           "launches_per_step": r_["launches"], "avg_launch_us": r_["seconds"] / r_["launches"] * 1e6,
                           "algorithmic_flops_per_launch": r_["TFs"] * 1e12 * r_["seconds"] / r_["launches"],
                           "note": "the fp32 2:4 kernel expands to dense fp32 MFMA (no fp32 sparse matrix instruction exists): executed = dense-equivalent flops",
                           "measured": "single stream, HIP events on the launch stream, hipGraph replay", "families": fams_out}
    else:
        GBs = d["bytes"] / d["seconds"] / 1e9
        # yardstick measured in THIS process: sm_copy_bytes (16-byte non-temporal loads + stores) moving the timed step's own
        # algorithmic byte count (half read, half written), same event timing as the families above
        step_bytes = sum((A_fu(L) if use_fused(L) else A_sp(L) + L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8)) for L in layers)
        half = int(step_bytes / 2) // 4096 * 4096
        ysrc = torch.empty(half, dtype=torch.uint8, device=dev)
        ydst = torch.empty(half, dtype=torch.uint8, device=dev)
        sm.fill_uniform(ysrc.view(torch.float16), 0xC0B1, 0.0, 1.0)
        t_copy = event_seconds(make_runner(lambda: sm.copy_bytes(ysrc, ydst)), R)
        copy_GBs = 2.0 * half / t_copy / 1e9
        del ysrc, ydst
        out["roofline"] = {"bound": "hbm", "achieved": GBs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": GBs / HBM_PEAK_GBS, "traffic": d["traffic"], "traffic_source": tsrc if (d["traffic"] is not None or (tsrc and tsrc["stale"])) else None,
                           "kernel": dom, "launches_per_step": d["launches"], "avg_launch_us": d["seconds"] / d["launches"] * 1e6,
                           "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                           "measured": "single stream, one kernel family at a time, HIP events on the launch stream, hipGraph replay",
                           "yardstick": {"device_copy_GBs": copy_GBs, "frac_of_device_copy": GBs / copy_GBs,
                                         "guide_float4_copy_GBs": GUIDE_COPY_GBS, "frac_of_guide_copy": GBs / GUIDE_COPY_GBS,
                                         "step_frac_of_guide_copy": step_bytes / t_full / 1e9 / GUIDE_COPY_GBS,
                                         "copy_bytes": 2 * half, "copy_ms": t_copy * 1e3,
                                         "step_algorithmic_bytes": step_bytes, "step_GBs": step_bytes / t_full / 1e9,
                                         "step_frac_of_device_copy": step_bytes / t_full / 1e9 / copy_GBs,
                                         "source": "measured in this run: sm_copy_bytes (16-byte streaming copy kernel of libsparsifyme.so) over the "
                                                   "timed step's own algorithmic byte count, half read + half written; guide_float4_copy_GBs = the float4 copy "
                                                   "/opt/skills/guides/MI355X_MICROARCH.md:36 measures (6.29 TB/s), the stricter of the two yardsticks; context "
                                                   "only -- `frac` is against the 8 TB/s specification"},
                           "families": fams_out}


def conv_path_stage(sm, torch, dev, dtype):
    """The 3 x 3 convolution layers of the ResNet-50 table through the implicit-GEMM kernel (sm_conv_spmma_fused_*: NCHW
    activations in, C out, neither the 9 x larger A nor its blob in HBM), stride 1 / padding 1 so that m = H * W, b = 32.
    Own roofline: bytes = activations + B + C; bound = max(bytes / HBM peak, dense-equivalent flops / 2 x 2.5 PF).  Not part
    of `value`: the headline step is defined on the reference's (m, n, k) operands."""
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    N = 32
    rows, tot_ms, tot_fl, tot_by, tot_roof, tot_routed = [], 0.0, 0.0, 0.0, 0.0, 0.0
    for Cin, HW, n, cnt in [(64, 112, 64, 3), (128, 56, 128, 4), (256, 28, 256, 6), (512, 14, 512, 3)]:
        L, K = HW * HW, Cin * 9
        X = torch.empty(N * Cin * L, dtype=tdt, device=dev)
        sm.fill_uniform(X, 7 + Cin, -1.0, 1.0)
        B = torch.empty(K * n, dtype=tdt, device=dev)
        sm.fill_uniform(B, 9 + n, -1.0, 1.0)
        C = torch.empty(N * L * n, dtype=tdt, device=dev)
        ms = sm.graph_time_ms(lambda: sm.conv_spmma_fused(X, B, C, N, Cin, HW, HW, 3, 3, 1, 1, 1, n), iters=10)
        fl, by = 2.0 * N * L * n * K, 2.0 * (N * Cin * L + K * n + N * L * n)
        roof = max(by / (HBM_PEAK_GBS * 1e9), fl / 5.0e15)
        row = {"m": L, "n": n, "k": K, "count": cnt, "ms": ms, "GBs": by / ms / 1e6, "eff_TFs": fl / ms / 1e9, "frac": roof * 1e3 / ms}
        # round 4: sm_conv_spmma_* picks the faster route per layer (small-spatial long-K layers: im2col-to-blob + staged matmul)
        ms_r = ms
        if hasattr(sm, "conv_spmma"):
            need = sm.conv_spmma_workspace(N, Cin, HW, HW, 3, 3, 1, 1, 1)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            ms_r = sm.graph_time_ms(lambda: sm.conv_spmma(X, B, C, N, Cin, HW, HW, 3, 3, 1, 1, 1, n, workspace=ws), iters=10)
            row["ms_routed"] = ms_r
            row["route"] = "im2col_compress24 + spmma" if need else "implicit GEMM"
            del ws
        rows.append(row)
        tot_ms += ms * cnt; tot_fl += fl * cnt; tot_by += by * cnt; tot_roof += roof * cnt; tot_routed += ms_r * cnt
        del X, C
    return {"kernel": "conv_spmma_fused_kernel", "layers": rows, "table_weighted_ms": tot_ms, "table_weighted_routed_ms": tot_routed,
            "eff_TFs": tot_fl / tot_ms / 1e9,
            "roofline": {"bound": "hbm", "achieved": tot_by / tot_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": tot_by / tot_ms / 1e6 / HBM_PEAK_GBS, "frac_of_per_layer_roofline": tot_roof * 1e3 / tot_ms},
            "note": "bytes = activations + B + C (no A); DESIGN.md 4.4: bound by the selection / gather instruction stream, not by HBM"}


def conv_step_stage(args, sm, torch, dev, layers, grouping, make_runner, event_seconds, stages):
    """The whole ResNet-50 table FROM ACTIVATIONS as one replayed step: the 33 1 x 1 layers are the fused
    kernel on their NHWC activations (which ARE the (m x k) operand: the timed step's own launches), the 16 3 x 3 layers and the
    7 x 7 stem go through sm_conv_spmma_* from NCHW activations -- implicit GEMM, or im2col-to-blob + matmul where the routing
    rule / the geometry says so -- so the kh x kw times larger A of those 17 layers is never materialised.  Same fork / join over
    the streams as the headline step, one hipGraph replay.  Verified AFTER the loop: every convolution layer's C against
    sm_im2col_compress24 + sm_spmma of the same activations, bit for bit.  Reported beside the dense GEMM on the materialised A;
    not `value` (the headline is defined on the reference's (m, n, k) operands, datasets/get_shapes.py:30-40,66-73)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_shapes", os.path.join(ROOT, "datasets", "gen_shapes.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fused_groups, run_group, spread, ForkedItems = grouping[:4]
    tdt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    N, size, geo = 32, 224, []
    for (cin, cout, ksz, stride, pad) in gen.resnet_convs("resnet50"):   # the reference's walk: sizes chained conv to conv
        osz = gen.conv_out(size, ksz, stride, pad)
        geo.append((cin, cout, ksz, stride, pad, size, osz))
        size = osz
    by_li = sorted(layers, key=lambda L: L["li"])
    convs, ones = [], []
    for L, (cin, cout, ksz, stride, pad, hin, hout) in zip(by_li, geo):
        assert (L["m"], L["n"], L["k"]) == (hout * hout, cout, cin * ksz * ksz) and L["b"] == N, "table row / architecture mismatch"
        if ksz == 1:
            ones.append(L)
            continue
        X = torch.empty(N * cin * hin * hin, dtype=tdt, device=dev)
        sm.fill_uniform(X, 0xC0 + L["li"], 0.0, 1.0)
        need = sm.conv_spmma_workspace(N, cin, hin, hin, ksz, ksz, stride, pad, 1)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        convs.append(dict(L=L, X=X, ws=ws, need=need, g=(N, cin, hin, hin, ksz, ksz, stride, pad, 1), C=torch.empty_like(L["C"])))

    def run_conv(cv):
        sm.conv_spmma(cv["X"], cv["L"]["B"], cv["C"], *cv["g"], cv["L"]["n"], workspace=cv["ws"] if cv["need"] else None)
    items = [("group", Ls) for _, Ls in fused_groups(ones)] + [("single", [dict(cv["L"], conv=cv)]) for cv in convs]
    step = ForkedItems(spread(items), run_group, lambda L: run_conv(L["conv"]))
    t = event_seconds(make_runner(step), max(5, args.steps))
    # verification, after the timed loop: every convolution layer against the pair from the same activations
    ok, checked = True, 0
    for cv in convs:
        L = cv["L"]
        blob = torch.empty(sm.compress24_size(L["m"], L["k"], 2, N), dtype=torch.uint8, device=dev)
        sm.im2col(cv["X"], *cv["g"], blob, compress=True)
        Cref = torch.empty_like(cv["C"])
        sm.spmma(blob, L["B"], Cref, L["m"], L["n"], L["k"], N, 0)
        ok = ok and bool(torch.equal(Cref.view(torch.int16), cv["C"].view(torch.int16)))
        checked += 1
        del blob, Cref
    act_bytes = sum(cv["X"].numel() * 2 for cv in convs)
    a_bytes = sum(cv["L"]["A"].numel() * 2 for cv in convs)
    routes = {}
    for cv in convs:
        r = "im2col-to-blob + matmul" if cv["need"] else "implicit GEMM"
        routes[r] = routes.get(r, 0) + 1
    dense = stages.get("dense_gemm_rowmajor_grouped_ms") or stages.get("dense_gemm_rowmajor_ms")
    res = {"conv_step_ms": t * 1e3, "layers": 49, "conv_layers_from_activations": len(convs), "pointwise_layers_fused_on_nhwc": len(ones), "routes": routes,
           "verified_bit_identical_to_im2col_compress24_plus_spmma": ok, "verified_layers": checked,
           "activation_bytes_instead_of_A_bytes": [act_bytes, a_bytes],
           "dense_gemm_on_materialised_A_ms": dense, "speedup_vs_dense_gemm_on_materialised_A": (dense / (t * 1e3)) if dense else None,
           "headline_step_ms": stages.get("timed_path_ms"),
           "note": "not `value`: the headline stays on the materialised-A configuration the reference's tables define"}
    if not ok:
        sys.stderr.write("bench: conv_step differs from im2col_compress24 + spmma\n")
        raise SystemExit(4)
    return res


def config5_stage(sm, torch, dev):
    """BASELINE config 5: 90 %-sparse COO (one A, density 0.1, values U(-1,1)) x dense, fp32, on four ResNet-50 shapes at
    b = 32 through sm_spmm_coo_f32_ws; HBM GB/s of the algorithmic bytes (B read once + C written once + A) vs the peak."""
    import ctypes
    L_ = sm.lib()
    rows = []
    g = torch.Generator(device=dev).manual_seed(5)
    seen = []
    for sh in read_shapes(table_path("resnet50")):   # every unique shape of the table (four of 17 were timed)
        if sh not in seen:
            seen.append(sh)
    for (m, n, k, b) in seen:
        dense = torch.rand(m, k, generator=g, device=dev) < 0.1
        idx = dense.nonzero()            # row-major scan: sorted by row, then column
        r, c = idx[:, 0].to(torch.int32).contiguous(), idx[:, 1].to(torch.int32).contiguous()
        nnz = int(r.numel())
        v = (torch.rand(nnz, generator=g, device=dev) * 2 - 1).float()
        B = torch.empty(b * k * n, dtype=torch.float32, device=dev)
        sm.fill_uniform(B, 55 + n, -1.0, 1.0)
        C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
        nb = ctypes.c_size_t(0)
        L_.sm_spmm_coo_workspace_size(m, ctypes.byref(nb))
        ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
        nb2 = ctypes.c_size_t(0)
        L_.sm_spmm_coo_packed_workspace_size(m, nnz, ctypes.byref(nb2))
        ws2 = torch.zeros(nb2.value, dtype=torch.uint8, device=dev)

        def call_rowptr():
            rc = L_.sm_spmm_coo_f32_ws(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                       ws.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())

        def call_packed():
            rc = L_.sm_spmm_coo_f32_packed(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                            ws2.data_ptr(), nb2.value, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())
        nb3 = ctypes.c_size_t(0)
        L_.sm_spmm_coo_fast_workspace_size(m, k, n, b, ctypes.byref(nb3))
        ws3 = torch.zeros(nb3.value, dtype=torch.uint8, device=dev)

        def call_fast():
            rc = L_.sm_spmm_coo_f32_fast(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                          ws3.data_ptr(), nb3.value, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())
        ms_rowptr = sm.graph_time_ms(call_rowptr, iters=5)
        ms = sm.graph_time_ms(call_packed, iters=5)
        by = nnz * 8.0 + (m + 1) * 4.0 + 4.0 * b * (k * n + m * n)
        row = {"m": m, "n": n, "k": k, "b": b, "nnz": nnz, "ms_exact": ms, "GBs_exact": by / ms / 1e6, "frac_exact": by / ms / 1e6 / HBM_PEAK_GBS,
               "TFs_exact": 2.0 * nnz * n * b / ms / 1e9, "ms_exact_rowptr_form": ms_rowptr}
        form = L_.sm_spmm_coo_fast_form(m, k, nnz, n, b, 0.0) if hasattr(L_, "sm_spmm_coo_fast_form") else (1 if k % 64 == 0 else 0)
        if form:
            ms_fast = sm.graph_time_ms(call_fast, iters=5)
            flag = ctypes.c_int(-1)
            L_.sm_spmm_coo_fast_flag(ws3.data_ptr(), ctypes.byref(flag), None)
            row.update({"ms": ms_fast, "GBs": by / ms_fast / 1e6, "frac": by / ms_fast / 1e6 / HBM_PEAK_GBS, "range_flag": flag.value,
                        "form": ("sparse matrix instruction" if form == 2 else "dense-MFMA") + " (opt-in: strided_coo_options().fast)"})
        else:  # the stem layer's k = 147: the dense-MFMA form takes whole 64-deep stages only; strided_coo runs the exact form
            row.update({"ms": ms, "GBs": by / ms / 1e6, "frac": by / ms / 1e6 / HBM_PEAK_GBS, "form": "exact (packed)", "range_flag": None})
        rows.append(row)
        del B, C, ws, ws2, ws3
    return {"kernel": "ms / GBs / frac = the OPT-IN fast form of sparsifyme::batched::strided_coo (strided_coo_options().fast; the default is the exact fp32 form, ms_exact): sm_spmm_coo_f32_fast -- round 5, `form` = sparse matrix instruction: "
                      "spmm_coo_smfmac_kernel (a 2:4 image of A, hi + lo fp16 planes, + its few third / fourth non-zeros per strip as fp32 entries; B converted in the loader; v_smfmac_f32_16x16x64_f16) after scan / scatter / image kernels, "
                      "whole call timed; `form` = dense-MFMA: the round-4 pipeline (dense operand and A scaled by powers of two computed on the "
                      "device, rounded to fp16 / split hi + lo, fp16 MFMA with fp32 accumulation, inverse scales on the fp32 sums; result within 2^-11 of "
                      "sum|a||b| at any magnitude; a range flag + untouched C when an operand does not convert -> exact fallback; whole call incl. its scan / "
                      "conversion / scatter passes) = ms / GBs / frac; the exact forms beside it: ms_exact = sm_spmm_coo_f32_packed (re-ordering of A + product), "
                      "ms_exact_rowptr_form = sm_spmm_coo_f32_ws", "shapes": rows,
            "unit": "GB/s of algorithmic bytes (SURVEY.md 8(d): nnz*(s+4) + (m+1)*4 + b*s*(k*n + m*n))", "peak": HBM_PEAK_GBS}


def ge_mod():
    import __graft_entry__ as ge
    return ge


def bell_stage(sm, torch, dev):
    """batched::spmm on Blocked-ELL operands (the reference's only recorded sparse number, examples/compare.csv column
    `spmm`; call spmm.hxx:94-111) as examples/spmm.cu builds them: 2 x 2 blocks, half of the block columns present, one A per
    batch index, B shared, fp32, b = 32, all batches in one submission (sm_spmm_bell_batched_f32).  Bytes = stored values +
    block indices + B + C; the kernel pair expands the blocks and runs the dense fp32 MFMA product, so the binding roofline is
    max(bytes / HBM peak, dense flops / fp32 matrix peak).  Host pointer tables: not graph-capturable, wall clock over 5 calls."""
    import ctypes
    L_ = sm.lib()
    rows = []
    g = torch.Generator(device=dev).manual_seed(11)
    for (m, n, k, b) in [(784, 256, 2304, 32), (12544, 64, 576, 32), (196, 512, 4608, 32), (3136, 128, 1152, 32)]:
        bs, ell_cols = 2, k // 2
        bcols = ell_cols // bs
        vals, idxs = [], []
        for _ in range(b):
            ci = torch.rand(m // bs, k // bs, generator=g, device=dev).argsort(dim=1)[:, :bcols].sort(dim=1).values
            idxs.append(ci.to(torch.int64).contiguous().view(-1))
            v = torch.empty(m * ell_cols, dtype=torch.float32, device=dev)
            sm.fill_uniform(v, 77 + len(vals), -0.5, 0.5)
            vals.append(v)
        B = torch.empty(k * n, dtype=torch.float32, device=dev)
        sm.fill_uniform(B, 78, -0.5, 0.5)
        Cs = [torch.empty(m * n, dtype=torch.float32, device=dev) for _ in range(b)]
        nb = ctypes.c_size_t(0)
        L_.sm_spmm_bell_batched_workspace_size(m, k, b, ctypes.byref(nb))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        PA = ctypes.c_void_p * b
        pv, pi, pc = PA(*[v.data_ptr() for v in vals]), PA(*[i.data_ptr() for i in idxs]), PA(*[c.data_ptr() for c in Cs])

        def call():
            rc = L_.sm_spmm_bell_batched_f32(pv, pi, m, k, bs, ell_cols, B.data_ptr(), pc, n, b, 1.0, 0.0, ws.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # device time by a HIP event pair on the launch stream (round 3 timed this stage by wall clock)
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        by = b * (m * ell_cols * 4.0 + (m // bs) * bcols * 8.0 + m * n * 4.0) + k * n * 4.0
        fl = 2.0 * m * n * k * b  # what the expanded fp32 MFMA product executes (stored-value flops are half of it)
        roof_ms = max(by / (HBM_PEAK_GBS * 1e9), fl / (F32_MATRIX_PEAK_TFS * 1e12)) * 1e3
        rows.append({"m": m, "n": n, "k": k, "b": b, "block": bs, "ell_cols": ell_cols, "ms": ms, "bytes": by, "GBs": by / ms / 1e6,
                     "executed_TFs": fl / ms / 1e9, "stored_value_TFs": fl / 2 / ms / 1e9,
                     "bound": "mfma" if fl / (F32_MATRIX_PEAK_TFS * 1e12) > by / (HBM_PEAK_GBS * 1e9) else "hbm", "frac": roof_ms / ms})
        del vals, idxs, Cs, ws
    return {"kernel": "bell_expand_rows_kernel + gemm_f32_dma_kernel (sm_spmm_bell_batched_f32, one submission for all batches)",
            "shapes": rows, "frac": "roofline time / measured time, roofline = max(algorithmic bytes / 8 TB/s, executed dense flops / 157.3 TF/s)",
            "timing": "HIP event pair on the launch stream around 5 calls after one warm-up (host pointer tables: the entry point is not graph-capturable)"}


def config1_cpu(ge):
    """BASELINE config 1 (examples/sparsify.cu:43-47 path, no GPU): one 512 x 512 x 512 fp32 layer on the host, timed in
    full -- the positional sparsify, the magnitude prune to 2:4 (STRIP), compress, the dense GEMM and the 2:4 product, all the
    oracle's arithmetic (`port`).  Seeded U(0,1) operands; best of 5 after one warm-up each."""
    import numpy as np
    orc = ge.load_oracle()
    m = n = k = 512
    rng = np.random.default_rng(0x5EED)
    A = rng.uniform(0, 1, m * k).astype(np.float32)
    B = rng.uniform(0, 1, k * n).astype(np.float32)
    C = np.zeros(m * n, dtype=np.float32)

    def best(fn, reps=5):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3

    w, mask = A.copy(), np.ones(m * k, dtype=np.uint64)
    t_pos = best(lambda: orc.sparsify_positional(w, mask, m, k, 0.5))
    Au = A.view(np.uint32)
    t_prune = best(lambda: orc.prune24(Au, m, k, k, orc.STRIP))
    P = orc.prune24(Au, m, k, k, orc.STRIP)
    t_cmp = best(lambda: orc.compress24(P, m, k, k))
    t_gemm = best(lambda: orc.cpu_gemm_f32(A, B, C, m, n, k))
    t_sp = best(lambda: orc.cpu_spmma_f32(A, B, C, m, n, k))
    fl = 2.0 * m * n * k
    return {"m": m, "n": n, "k": k, "b": 1, "dtype": "f32", "kind": "port", "cores": orc.num_threads(),
            "threads": "dense GEMM and 2:4 product: OpenMP over rows on `cores` threads; sparsify / prune / compress: 1 thread",
            "sparsify_positional_ms": t_pos, "prune24_strip_ms": t_prune, "compress24_ms": t_cmp,
            "dense_gemm_ms": t_gemm, "dense_gemm_gfs": fl / t_gemm / 1e6,
            "spmma_2to4_ms": t_sp, "spmma_2to4_eff_gfs": fl / t_sp / 1e6,
            "prune_compress_dense_gemm_ms": t_prune + t_cmp + t_gemm,
            "note": "untuned restatement (naive loops, no cache blocking): a reported baseline, not a target"}


def cpu_baseline(ge, shapes):
    """The oracle's arithmetic ('port': fp32 accumulate, OpenMP over rows) on the host cores, on a
    bounded sample: one batch (b = 1) of every unique (m,n,k) of the table, repeated; both the dense
    product and the 2:4 path (STRIP selection fused with the two kept MACs per strip)."""
    import numpy as np
    orc = ge.load_oracle()
    uniq = sorted(set((m, n, k) for m, n, k, _ in shapes))
    rng = np.random.default_rng(0x5EED)
    reps = 64
    fl = t_dense = t_sparse = 0.0
    for (m, n, k) in uniq:
        r = m
        A = rng.uniform(0, 1, r * k).astype(np.float32)
        B = rng.uniform(0, 1, k * n).astype(np.float32)
        C = np.zeros(r * n, dtype=np.float32)
        orc.cpu_gemm_f32(A, B, C, r, n, k)  # warm
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.cpu_gemm_f32(A, B, C, r, n, k)
        t_dense += time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.cpu_spmma_f32(A, B, C, r, n, k)
        t_sparse += time.perf_counter() - t0
        fl += 2.0 * r * n * k * reps
    return {"value": fl / t_sparse / 1e9, "unit": "GF/s", "cores": orc.num_threads(), "kind": "port",
            "dense_value": fl / t_dense / 1e9,
            "sample": f"UNTUNED port (naive row-parallel loops, no cache blocking): oracle sm_cpu_spmma_f32 (2:4 path) / sm_cpu_gemm_f32 (dense_value), fp32, one batch (b=1) of each of "
                      f"the {len(uniq)} unique shapes of the table x {reps} repetitions ({fl / 1e9:.1f} dense-equivalent GFLOP, "
                      f"{t_dense + t_sparse:.1f} s of CPU work); effective GF/s = dense-equivalent flops / time"}


if __name__ == "__main__":
    main()
