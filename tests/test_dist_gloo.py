"""The N > 1 launch path of bench.py (one process per GPU, batch sharding, barrier fences, max-over-ranks timing),
exercised with world_size 2 on the gloo backend.  CPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_selftest(nproc, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "5", "--warmup", "1", "--batch", "8", "--selftest"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=180, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_and_max_over_ranks():
    r2 = run_selftest(2, 29611)
    assert r2["n_gpus"] == 2 and r2["steps"] == 5 and r2["scaling"] == "weak"
    # rank 1 sleeps 20 ms per step, rank 0 only 10 ms: the reported time is the slowest rank's
    assert r2["ms_per_step"] >= 19.0
    # whole-job throughput counts every rank's frames: 2 ranks x 8 frames x 5 steps / time
    assert abs(r2["value"] - 2 * 8 * 5 / (r2["ms_per_step"] * 5 / 1e3)) / r2["value"] < 1e-6
    # ranks hold different shards (seed 1234 + rank): the all-reduced checksum differs from 2 x rank 0's
    import torch
    s0 = float(torch.rand(8, 64, generator=torch.Generator().manual_seed(1234)).sum())
    s1 = float(torch.rand(8, 64, generator=torch.Generator().manual_seed(1235)).sum())
    assert abs(r2["checksum"] - (s0 + s1)) < 1e-3 and abs(s0 - s1) > 1e-6


def _plain_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_plain_invocation_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as the driver runs it — no torch.distributed environment — launches its two ranks by itself
    and rank 0 prints the one JSON line (the reference picks DDP by itself when it sees >= 2 devices, pl_helpers.py:365-374)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8",
                          "--selftest"], capture_output=True, text=True, timeout=180, cwd=ROOT, env=dict(_plain_env(), OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["ms_per_step"] >= 19.0
    assert abs(line["value"] - 2 * 8 * 4 / (line["ms_per_step"] * 4 / 1e3)) / line["value"] < 1e-6


def test_mismatched_world_size_is_refused():
    """An environment that says WORLD_SIZE = 1 while --gpus 2 is asked for is a launch error, not something to paper over."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(_plain_env(), WORLD_SIZE="1", RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_one_and_two_rank_lines_carry_the_same_keys():
    """A scaling curve is built from the per-N lines: N = 1 (no process group at all) and N = 2 must be the same record."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--selftest"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(_plain_env(), OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    r1 = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    r2 = run_selftest(2, 29613)
    assert set(r1) == set(r2) and r1["n_gpus"] == 1 and r2["n_gpus"] == 2
    assert r1["unit"] == r2["unit"] and r1["scaling"] == r2["scaling"] == "weak"


def test_device_count_is_read_without_a_hip_context(tmp_path, monkeypatch):
    """self_launch sizes the job from the visibility variables / the KFD topology, never from torch.cuda in the parent."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    assert bench.visible_gpu_count() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    import inspect
    assert "torch.cuda" not in inspect.getsource(bench.self_launch) and "torch.cuda" not in inspect.getsource(bench.visible_gpu_count)


def test_scale_script_builds_the_curve_from_per_n_lines(tmp_path):
    """tools/scale.sh --selftest: runs the launcher for N = 1, 2 and writes ONE json with per-N frames/s and efficiency vs N = 1."""
    out_path = tmp_path / "scale.json"
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale.sh"), "--gpus", "1,2", "--out", str(out_path), "--",
                          "--selftest", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=dict(_plain_env(), OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    rec = json.loads(out_path.read_text())
    assert [p["n_gpus"] for p in rec["points"]] == [1, 2]
    assert rec["points"][0]["efficiency"] == 1.0
    # rank r sleeps 10 (r + 1) ms per step: two ranks deliver 2 x 8 frames in 20 ms against 8 in 10 ms -> efficiency ~0.5
    assert 0.35 < rec["points"][1]["efficiency"] < 0.65
