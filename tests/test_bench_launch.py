"""bench.py's launcher on the CPU: `python bench.py --gpus 2` with no WORLD_SIZE must itself start two ranks (gloo, stub
workload: DG_BENCH_STUB=1 replaces the GPU step by a trivial CPU one) and print ONE line with n_gpus == 2; a world size
that differs from --gpus is refused instead of being reported."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DG_BENCH_STUB="1", **kw)
    return env


def test_bench_gpus2_spawns_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world_size_seen"] == 2 and line["config"]["backend"] == "gloo"
    assert line["scaling"] == "weak" and line["config"]["global_batch_images"] == 512 and line["config"]["images_per_gpu"] == 256
    assert line["steps"] == 3 and line["warmup"] == 1 and line["value"] > 0 and line["data"] == "stub"
    assert "world_size=2" in p.stderr


def test_bench_strong_scaling_splits_one_batch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--scaling", "strong"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["global_batch_images"] == 256 and line["config"]["images_per_gpu"] == 128


def test_bench_refuses_world_size_mismatch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in p.stderr
