"""World-size-2 (and 3, ragged) check of the row-tile partition + in-place
all-gather used for N > 1 GPUs, on CPU with the gloo backend.  The tile
renderer here is the oracle (the GPU kernels cannot run in this container); the
partition, padding and gather code is the same code bench.py runs under RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, spp, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_binding as ob
    from pathtracer_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = ob.Scene(*ob.default_scene())

    def render_rows(view, begin, end):
        cfg = ob.make_config(width, height, spp, row_begin=begin, row_end=end)
        view[begin:end] = torch.from_numpy(sc.render(cfg, threads=2)[begin:end])

    img = multigpu.render_tiled(render_rows, height, width, rank, world)
    q.put((rank, img.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 32), (3, 20)])
def test_row_tiles_gather_to_the_single_process_image(oracle, world, height):
    width, spp = 48, 2
    ref = oracle.Scene(*oracle.default_scene()).render(oracle.make_config(width, height, spp))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, width, spp, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert np.array_equal(got[r], ref), "rank %d holds a different image" % r


def test_tile_arithmetic():
    from pathtracer_amd import multigpu
    for h in (1, 7, 135, 1080, 2160):
        for w in (1, 2, 3, 4, 8):
            tiles = [multigpu.row_tile(h, r, w) for r in range(w)]
            covered = [j for b, e in tiles for j in range(b, e)]
            assert covered == list(range(h))
            assert all(e - b <= multigpu.tile_rows(h, w) for b, e in tiles)
    assert multigpu.row_tile(1080, 3, 8) == (405, 540)
