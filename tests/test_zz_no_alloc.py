"""pt_gather_framebuffer_noalloc — the gather that allocates nothing (its staging area is a static device array inside
the library) — against pt_gather_framebuffer, through the C ABI, bit for bit: two calls in a row on one communicator,
the second with a smaller frame, so that what the first call left in the shared staging area must not show."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _gather_both(torch, lib, comm, tile, height, width):
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = lib.pt_tile_rows(height, 1)
    assert tile.numel() == rows * width * 3
    gathered = torch.full_like(tile, -3.0)
    want = torch.full((height, width, 3), -1.0, dtype=torch.float32, device="cuda")
    got = torch.full((height, width, 3), -2.0, dtype=torch.float32, device="cuda")
    rc = lib.pt_gather_framebuffer(comm, C.c_void_p(tile.data_ptr()), C.c_void_p(gathered.data_ptr()), C.c_void_p(want.data_ptr()),
                                   height, width, stream)
    assert rc == 0, lib.pt_last_error()
    rc = lib.pt_gather_framebuffer_noalloc(comm, C.c_void_p(tile.data_ptr()), C.c_void_p(got.data_ptr()), height, width, stream)
    assert rc == 0, lib.pt_last_error()
    torch.cuda.synchronize()
    return want, got


def test_noalloc_gather_equals_gather_twice_second_smaller(torch, pt):
    lib = pt.binding.lib
    lib.pt_gather_framebuffer_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    prims, n, mats, nm = pt.default_scene()
    sc = pt.Scene(prims, n, mats, nm)
    comm = pt.Comm(pt.comm_unique_id(), 0, 1)
    try:
        for (w, h, spp) in ((160, 90, 8), (64, 40, 4)):
            tile = sc.render(pt.make_config(w, h, spp, row_begin=0, row_step=1, fb_compact=1))
            ref = sc.render(pt.make_config(w, h, spp))
            want, got = _gather_both(torch, lib, comm.h, tile, h, w)
            assert torch.equal(want, ref)
            assert torch.equal(got, want), (w, h)
        # a frame beyond the static staging area is refused, with the limit in the message, and nothing is written
        big_h, big_w = 2160, 3840
        dummy = torch.zeros(4, dtype=torch.float32, device="cuda")
        rc = lib.pt_gather_framebuffer_noalloc(comm.h, C.c_void_p(dummy.data_ptr()), C.c_void_p(dummy.data_ptr()), big_h, big_w, None)
        assert rc != 0
        msg = lib.pt_last_error()
        assert b"pt_gather_framebuffer_noalloc" in msg and b"exceed" in msg, msg
        torch.cuda.synchronize()
        assert float(dummy.abs().sum()) == 0.0
        assert lib.pt_gather_framebuffer_noalloc(None, None, None, 4, 4, None) != 0
    finally:
        comm.close()


def test_noalloc_gather_with_comm_init_all(torch, pt):
    lib = pt.binding.lib
    lib.pt_gather_framebuffer_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    devs = (C.c_int * 1)(0)
    out = (C.c_void_p * 1)()
    assert lib.pt_comm_init_all(1, devs, out) == 0, lib.pt_last_error()
    try:
        g = torch.Generator(device="cuda").manual_seed(5)
        for (h, w) in ((1080, 1920), (33, 7)):      # the largest frame the staging area is sized for, then a tiny one
            x = torch.rand((h, w, 3), dtype=torch.float32, device="cuda", generator=g)
            want, got = _gather_both(torch, lib, out[0], x, h, w)
            assert torch.equal(want, x) and torch.equal(got, x)
    finally:
        assert lib.pt_comm_free(out[0]) == 0
