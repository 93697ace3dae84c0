"""pt_render allocates nothing and only launches on the caller's stream, so it can be recorded into a HIP graph and
replayed: three frames (different seeds, math modes and row sets) captured once, replayed twice, equal to the same
frames rendered eagerly."""
import pytest

pytestmark = pytest.mark.gpu


def test_render_inside_a_hip_graph(torch, pt):
    sc = pt.Scene(*pt.default_scene())
    cfgs = [pt.make_config(320, 180, 8, seed=1),
            pt.make_config(320, 180, 8, seed=2, math_mode=pt.PT_MATH_PARITY),
            pt.make_config(320, 180, 8, seed=3, row_begin=1, row_step=2, fb_compact=1)]
    shapes = [(180, 320, 3), (180, 320, 3), (90, 320, 3)]
    eager = [sc.render(c, out=torch.zeros(s, device="cuda")) for c, s in zip(cfgs, shapes)]
    bufs = [torch.full(s, -1.0, device="cuda") for s in shapes]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c, b in zip(cfgs, bufs):
            sc.render(c, out=b)
    for _ in range(2):
        for b in bufs:
            b.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        for b, e in zip(bufs, eager):
            assert torch.equal(b, e)
