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


def test_ranks_are_pinned_to_disjoint_cores_and_report_their_own_step_times():
    """Round-4 verdict item 7: with N ranks on one host every rank binds itself to its own whole cores (bench.affinity_plan) and the
    line says who ran where and how far apart the ranks' own step times are."""
    r2 = run_selftest(2, 29617)
    per = r2["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    masks = [set(p["cpus"]) for p in per]
    assert all(masks) and not (masks[0] & masks[1]), masks
    assert masks[0] | masks[1] <= set(os.sched_getaffinity(0))
    # rank 1 sleeps 10 ms longer per step than rank 0: the spread the line reports
    assert 8.0 <= r2["rank_spread_ms"] <= 14.0 and per[1]["own_ms_per_step"] > per[0]["own_ms_per_step"]


def test_affinity_plan_keeps_whole_cores_together_and_follows_the_gpus_numa_nodes():
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    # 2 sockets x 8 cores x 2 threads: cpu c and c + 16 share a core; node 0 = cores 0-7, node 1 = cores 8-15
    core_of = lambda c: c % 16  # noqa: E731
    node_cpus = {0: list(range(0, 8)) + list(range(16, 24)), 1: list(range(8, 16)) + list(range(24, 32))}
    avail = list(range(32))
    # GPUs 0-3 on node 0, 4-7 on node 1
    plan = bench.affinity_plan(avail, 8, [0, 0, 0, 0, 1, 1, 1, 1], node_cpus, core_of)
    assert len(plan) == 8 and all(len(p) == 4 for p in plan)
    flat = [c for p in plan for c in p]
    assert len(flat) == len(set(flat)) == 32                                  # disjoint, nothing left over
    for r, p in enumerate(plan):
        assert {core_of(c) for c in p} == {c for c in p if c < 16}              # both threads of every core it owns
        assert set(p) <= set(node_cpus[0 if r < 4 else 1])                      # on its GPU's node
    # unknown topology: even split in CPU order, still whole cores, still disjoint
    even = bench.affinity_plan(avail, 4, None, None, core_of)
    assert [sorted({core_of(c) for c in p}) for p in even] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    # a node without enough cores for its ranks: everybody falls back to the even split (never an empty or shared set)
    starved = bench.affinity_plan(list(range(0, 4)) + list(range(8, 16)), 4, [0, 0, 0, 1], {0: [0, 1], 1: list(range(8, 16))}, lambda c: c)
    assert all(starved) and len({c for p in starved for c in p}) == sum(len(p) for p in starved)
    # fewer cores than ranks: nothing disjoint to hand out, every rank keeps the whole set
    assert bench.affinity_plan([0, 1], 4, None, None, lambda c: c) == [[0, 1]] * 4
