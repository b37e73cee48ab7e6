"""bench.py's rank launcher (CPU tier): `--gpus N` must start N ranks itself, and must never print an n_gpus line
for fewer ranks than asked.  The rendezvous path is rehearsed with gloo ranks (`--dry-run-cpu`: no model, no GPU)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_launcher_spawns_two_gloo_ranks():
    r = _run(["--gpus", "2", "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # rank 0 only
    line = lines[0]
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert line["allreduce_sum"] == line["expected_sum"] == 3.0   # both ranks took part in the collective


def test_more_ranks_than_devices_fails_loudly():
    """No GPU in the CPU tier: `--gpus 2` must exit non-zero and print no JSON line (never a silent n_gpus: 1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
    assert "visible HIP devices" in r.stderr


def test_gpus_flag_must_match_world_size():
    r = _run(["--gpus", "2", "--dry-run-cpu"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "WORLD_SIZE=4" in r.stderr


def test_workload_label_follows_the_arguments():
    sys.path.insert(0, str(ROOT))
    import bench
    a = bench.parse_args(["--config", "c5"])
    assert (a.batch, a.seq_len, a.z_dim) == (512, 200, 16)
    assert "configs[4]" in bench.workload_name(a, (16, 16, 2))
    b = bench.parse_args(["--batch", "64"])
    assert "configs[1]" not in bench.workload_name(b, (4, 4, 2)) and "batch=64" in bench.workload_name(b, (4, 4, 2))
    assert "configs[1]" in bench.workload_name(bench.parse_args([]), (4, 4, 2))
