"""bench.py with two ranks, end to end, on the one GPU of the box: `python bench.py --gpus 2` starts its ranks itself;
PT_BENCH_REHEARSE=1 puts both on GPU 0 and sends the collectives through gloo (RCCL refuses two ranks on one device).
Everything but the RCCL transport runs as it would on two GPUs: the launcher, the interleaved partition, the gather and
reordering, two frames in flight, the single-frame timing, the reductions, and the bitwise comparison of the gathered
image with the frame one rank renders alone."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_bench_with_several_ranks_on_one_gpu(world):
    env = dict(os.environ, PT_BENCH_REHEARSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 4 and d["scaling"] == "strong"
    assert d["bitwise_equal_to_single_gpu"] is True
    assert d["pipelined"] is True and d["frames_in_flight"] == 2 and d["single_frame_ms"] > 0.0
    assert d["value"] > 0.0 and d["roofline"]["traffic"] is None
    assert "REHEARSAL" in d["config"]["partition"]


def test_bench_under_torch_distributed_run():
    """The way the driver starts it: one rank per process under torch.distributed.run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PT_BENCH_REHEARSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["bitwise_equal_to_single_gpu"] is True
