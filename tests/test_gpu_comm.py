"""GPU tests of the multi-GPU frame assembly of the C ABI (pt_comm_*, pt_gather_framebuffer, pt_assemble_rows).

A box with one GPU can hold a communicator of world size 1 only (RCCL refuses two ranks on one device), so the
all-gather itself runs here with one rank; the row reordering, which is where the world size matters, is checked
for several world sizes against buffers laid out as the all-gather leaves them, and the N-rank partition is checked
by rendering every rank's rows on this GPU into that layout.  Everything is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gathered_layout(torch, image, world, rows):
    """What all_gather leaves: rank r's tile (image rows r, r+world, ...; padded to `rows`) at gathered[r]."""
    h, w, _ = image.shape
    g = torch.full((world, rows, w, 3), -7.0, dtype=torch.float32, device=image.device)
    for r in range(world):
        mine = image[r::world]
        g[r, :mine.shape[0]] = mine
    return g


@pytest.mark.parametrize("height,width,world", [(8, 64, 1), (9, 64, 2), (10, 33, 3), (1080, 1920, 8), (7, 5, 8), (270, 480, 4)])
def test_assemble_rows_restores_image_order(torch, pt, height, width, world):
    rows = pt.binding.lib.pt_tile_rows(height, world)
    assert rows == -(-height // world)
    g = torch.Generator(device="cuda").manual_seed(height * 131 + width)
    image = torch.rand((height, width, 3), dtype=torch.float32, device="cuda", generator=g)
    gathered = _gathered_layout(torch, image, world, rows)
    full = torch.zeros_like(image)
    pt.assemble_rows(gathered, full, height, width, world)
    torch.cuda.synchronize()
    assert torch.equal(full, image)
    # unaligned destination: the scalar path
    if width * 3 % 4 == 0:
        spare = torch.zeros(height * width * 3 + 1, dtype=torch.float32, device="cuda")
        off = spare[1:].view(height, width, 3)
        pt.assemble_rows(gathered, off, height, width, world)
        torch.cuda.synchronize()
        assert torch.equal(off, image)


def test_world_of_one_gather_is_the_render(torch, pt):
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    w, h, spp = 160, 90, 8
    ref = sc.render(pt.make_config(w, h, spp))
    comm = pt.Comm(pt.comm_unique_id(), 0, 1)
    assert pt.binding.lib.pt_comm_rank(comm.h) == 0 and pt.binding.lib.pt_comm_world(comm.h) == 1
    tile = sc.render(pt.make_config(w, h, spp, row_begin=0, row_step=1, fb_compact=1))
    gathered = torch.zeros_like(tile)
    full = torch.zeros_like(tile)
    comm.gather(tile, gathered, full, h, w)
    torch.cuda.synchronize()
    assert torch.equal(full, ref)
    comm.close()


def test_rank_tiles_assemble_to_the_single_gpu_frame(torch, pt):
    """Every rank's pt_render call (row_begin = r, row_step = world, compact) on this GPU, placed where the all-gather
    puts it, then pt_assemble_rows: the single-GPU frame bit for bit."""
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    w, h, spp = 200, 101, 4
    ref = sc.render(pt.make_config(w, h, spp))
    for world in (2, 3, 8):
        rows = pt.binding.lib.pt_tile_rows(h, world)
        gathered = torch.zeros((world, rows, w, 3), dtype=torch.float32, device="cuda")
        for r in range(world):
            sc.render(pt.make_config(w, h, spp, row_begin=r, row_step=world, fb_compact=1), out=gathered[r])
        full = torch.zeros_like(ref)
        pt.assemble_rows(gathered, full, h, w, world)
        torch.cuda.synchronize()
        assert torch.equal(full, ref), world


def test_comm_init_all_with_one_device(torch, pt):
    import ctypes as C
    lib = pt.binding.lib
    devs = (C.c_int * 1)(0)
    out = (C.c_void_p * 1)()
    rc = lib.pt_comm_init_all(1, devs, out)
    assert rc == 0, lib.pt_last_error()
    assert lib.pt_comm_world(out[0]) == 1 and lib.pt_comm_rank(out[0]) == 0
    x = torch.arange(6 * 4 * 3, dtype=torch.float32, device="cuda").view(6, 4, 3)
    g = torch.zeros_like(x); f = torch.zeros_like(x)
    rc = lib.pt_gather_framebuffer(out[0], C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(f.data_ptr()), 6, 4,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.pt_last_error()
    torch.cuda.synchronize()
    assert torch.equal(f, x)
    assert lib.pt_comm_free(out[0]) == 0


def test_bad_arguments(pt):
    import ctypes as C
    lib = pt.binding.lib
    assert lib.pt_tile_rows(0, 4) == 0 and lib.pt_tile_rows(5, 0) == 0
    assert lib.pt_assemble_rows(None, None, 4, 4, 2, None) != 0
    assert b"pt_assemble_rows" in lib.pt_last_error()
    h = C.c_void_p(0)
    assert lib.pt_comm_init_rank(b"\0" * 128, 2, 2, 0, C.byref(h)) != 0     # rank outside the world
    assert lib.pt_gather_framebuffer(None, None, None, None, 4, 4, None) != 0
    assert lib.pt_comm_free(None) == 0
