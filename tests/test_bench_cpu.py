"""CPU: bench.py's launcher logic -- ``--gpus N`` without a torchrun environment must spawn N ranks or fail loudly; it must never
print an ``n_gpus: 1`` line for an N > 1 request (VERDICT r3 missing #1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_without_devices_is_an_error_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VOICEMAP_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] is None and "only" in out["error"] and "n_gpus" not in out


def test_bench_world_size_mismatch_is_an_error_line():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "WORLD_SIZE=1" in out["error"]
