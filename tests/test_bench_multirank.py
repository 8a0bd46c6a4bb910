"""bench.py under torch.distributed.run with two ranks (the driver's N > 1 launch), rehearsed on ONE GPU: gloo
backend, both ranks on cuda:0 (--rehearse-gloo).  Guards the control flow -- every rank must reach the same
collectives (rank 0's per-stage passes once called a distributed barrier the other ranks never matched)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_complete_and_roll_up():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rehearse-gloo"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["config"]["partition"] == "hybrid"
    assert d["value"] > 0 and "roofline" in d and "stages" in d and "cpu_baseline" not in d


@pytest.mark.gpu
def test_bench_prints_one_small_contract_line_and_writes_the_detail_file(tmp_path):
    """VERDICT round 5, item 1, on the GPU: `python bench.py` prints exactly ONE stdout line, < 4 KB, that parses and carries the
    contract's fields with `roofline` and `cpu_baseline`; the stages / families / per-shape tables are in the detail file it names."""
    detail = tmp_path / "detail.json"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--settle-ms", "50", "--detail", str(detail)],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0].encode()) < 4096
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f16" and d["verified"] is True and d["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "spmma_f16_fused_direct_kernel" and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    full = json.load(open(detail))
    assert "stages" in full and "families" in full["roofline"] and "verified_layers" in full and full["stages"]["api_spmma_no_blob_layers"] == 49


@pytest.mark.gpu
def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it: the process starts two ranks itself (child processes),
    relays rank 0's line and reports n_gpus == 2 (it used to run one rank silently)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-gloo",
           "--no-extras", "--settle-ms", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    # total work fixed: the two ranks together ran every (layer, batch index) of the table exactly once
    assert abs(d["config"]["dense_equiv_gflop_per_step"] - 931.6) < 1.0


@pytest.mark.gpu
def test_rollup_runs_through_rccl_at_world_size_one():
    """The only collective of an N-GPU run -- multigpu.rollup's two all-reduces -- executed by the `nccl` (= RCCL) backend on
    the one GPU of the box: world size 1, device-bound process group exactly as bench.py initialises it (bench.py: init_process_group
    ("nccl", device_id=...)).  In a child process, so that the group's state never meets the test process's."""
    code = (
        "import os, sys, json, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "mg = ge.load_package_module('multigpu')\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "dev = torch.device('cuda', 0)\n"
        "tot, tmax = mg.rollup(931.6e9, 1.5e-3, dev, force_collective=True)\n"
        "x = torch.arange(8, dtype=torch.float64, device=dev); dist.all_reduce(x); torch.cuda.synchronize()\n"
        "print(json.dumps({'backend': dist.get_backend(), 'tot': tot, 'tmax': tmax, 'x': x.tolist()}))\n"
        "dist.barrier(); dist.destroy_process_group()\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["tot"] == 931.6e9 and d["tmax"] == 1.5e-3 and d["x"] == [float(i) for i in range(8)]


def test_bench_refuses_a_rank_count_it_was_not_asked_for():
    """A launcher that realised another world size than --gpus must not pass as an N-GPU run (exits before any GPU call)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and not out.stdout.strip()


def test_bench_self_launch_fails_loudly_without_gpus():
    """No GPU in the CPU container: `bench.py --gpus 2` must start its ranks, see them fail and exit non-zero with no JSON."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
