"""bench.py started without a launcher: `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts its own ranks as
fresh child processes.  Runs on the CPU: the children here are small scripts, or bench.py itself failing for want of
a GPU, which is the exit-status path."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys
keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
with open(os.path.join(sys.argv[1], "rank%s.json" % os.environ["RANK"]), "w") as f:
    json.dump({k: os.environ.get(k) for k in keys}, f)
print("line from rank", os.environ["RANK"])
"""


def test_children_see_their_rank_environment(tmp_path):
    import bench
    out = io.BytesIO()
    rc = bench.launch_ranks(3, [sys.executable, "-c", CHILD, str(tmp_path)], out=out)
    assert rc == 0
    assert out.getvalue().decode().strip() == "line from rank 0"       # only rank 0's standard output is relayed
    seen = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    ports = {s["MASTER_PORT"] for s in seen}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r, s in enumerate(seen):
        assert s["RANK"] == str(r) and s["LOCAL_RANK"] == str(r) and s["WORLD_SIZE"] == "3"
        assert s["MASTER_ADDR"] == "127.0.0.1"
        assert s["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_a_failing_rank_fails_the_launch_and_ends_the_others():
    import bench
    child = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)\n"
    t0 = time.perf_counter()
    rc = bench.launch_ranks(3, [sys.executable, "-c", child], out=io.BytesIO())
    assert rc == 7
    assert time.perf_counter() - t0 < 20            # ranks 0 and 2 were ended, not waited for


def test_bench_without_launcher_starts_ranks_and_propagates_their_failure():
    """No GPU here: every rank of `bench.py --gpus 2` stops with "no GPU visible", and so must the launch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stderr.count("no GPU visible") == 2, p.stderr[-2000:]
    assert p.stdout.strip() == ""


def test_bench_under_a_launcher_does_not_launch_again():
    """WORLD_SIZE set (as torch.distributed.run does) but different from --gpus: refused, no children."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
