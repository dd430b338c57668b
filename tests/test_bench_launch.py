"""CPU: bench.py's N > 1 path without GPUs.  `python bench.py --gpus 2 --dry-run-cpu ...` from a plain shell must start its own two
ranks (torch.distributed.run on 127.0.0.1), shard the channels, time with a barrier on both sides and the maximum over the ranks, and
print ONE JSON line from rank 0 with n_gpus = 2 - with gloo between the ranks and the CPU oracle as each rank's compute (the mode is
labelled a self-test in the line it prints: it is never a measurement).  Asking for more GPUs than the node has is one line on stderr
and exit code 2, not a traceback."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=ROOT)


def test_gpus_2_launches_its_own_ranks(built):
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--channels", "3", "--samples", "24000", "--dry-run-cpu")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run_cpu"] is True and line["scaling"] == "weak"
    assert line["config"]["channels_total"] == 6 and line["config"]["channels_per_gpu"] == 3
    assert line["value"] > 0 and line["work"]["symbols"] > 6 * 2300 and line["work"]["syncs"] >= 6
    # the same work in one process: the two ranks together saw exactly the channels a single rank of six would
    q = _run("--gpus", "1", "--steps", "1", "--warmup", "0", "--channels", "6", "--samples", "24000", "--dry-run-cpu")
    assert q.returncode == 0, q.stderr[-2000:]
    one = json.loads([ln for ln in q.stdout.splitlines() if ln.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["work"] == line["work"]


def test_gpus_8_the_drivers_scaling_shape(built):
    """the shape the driver's scaling run uses (--gpus 8): eight ranks over gloo, one channel each, one line with the whole job's work"""
    p = _run("--gpus", "8", "--steps", "1", "--warmup", "0", "--channels", "1", "--samples", "24000", "--dry-run-cpu", timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["dry_run_cpu"] is True and line["scaling"] == "weak"
    assert line["config"]["channels_total"] == 8 and line["config"]["channels_per_gpu"] == 1
    assert line["value"] > 0 and line["work"]["symbols"] > 8 * 2300 and line["work"]["syncs"] >= 8


def test_more_gpus_than_the_node_has_is_a_clear_error(built):
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    p = _run("--gpus", str(have + 3), "--steps", "1", "--warmup", "0")
    assert p.returncode == 2
    assert "GPU(s) visible" in p.stderr and "Traceback" not in p.stderr and not p.stdout.strip()
