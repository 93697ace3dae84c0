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

    def render_rows(local, row_begin, row_step):
        rows = np.arange(row_begin, height, row_step, dtype=np.int32)
        full = sc.render(ob.make_config(width, height, spp), threads=2, rows=rows)
        local[:len(rows)] = torch.from_numpy(full[rows])

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
            rows = [list(multigpu.rank_rows(h, r, w)) for r in range(w)]
            assert sorted(j for rr in rows for j in rr) == list(range(h))
            assert all(len(rr) <= multigpu.tile_rows(h, w) for rr in rows)
    assert list(multigpu.rank_rows(1080, 3, 8))[:3] == [3, 11, 19]


def test_single_rank_frame_is_the_local_tile():
    from pathtracer_amd import multigpu
    f = multigpu.TiledFrame(5, 4, 1, "cpu")
    f.local[:] = 1.0
    assert f.gather() is f.image and f.image.shape == (5, 4, 3) and bool((f.image == 1.0).all())


def _id_worker(rank, world, port, q):
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pathtracer_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def make_id():                       # stands in for pt_comm_unique_id (ncclGetUniqueId): 128 opaque bytes
        calls.append(rank)
        return bytes((7 * k + 3) % 256 for k in range(128))
    ident = multigpu.exchange_comm_id(rank, make_id)
    q.put((rank, ident, list(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_id_made_by_rank_0_reaches_every_rank():
    """`bench.py --gather abi` and any one-process-per-GPU host: rank 0 makes the id of the library's own RCCL
    communicator and the process group that is already up hands it to the others; only rank 0 makes one."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = bytes((7 * k + 3) % 256 for k in range(128))
    for rank, ident, calls in got:
        assert ident == want, rank
        assert calls == ([0] if rank == 0 else []), (rank, calls)
