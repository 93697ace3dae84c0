"""Row-tile data parallelism over the GPUs of one node.

Every (pixel, sample) is independent and the RNG is keyed on the GLOBAL pixel
index, so the image does not depend on how rows are split.  Rank r of `world`
renders one contiguous row tile into its slice of a full-size framebuffer and a
single in-place all-gather (RCCL over xGMI under the "nccl" backend) gives every
rank the whole image.  The tile is padded to ceil(H/world) rows so that all
ranks contribute equal-sized slices, as all_gather_into_tensor requires.
"""
import torch
import torch.distributed as dist


def tile_rows(height, world):
    """Rows per rank (the last ranks may own fewer, or no, real rows)."""
    return (height + world - 1) // world


def row_tile(height, rank, world):
    """[begin, end) of the rows rank renders."""
    tr = tile_rows(height, world)
    begin = min(height, rank * tr)
    return begin, min(height, begin + tr)


def alloc_framebuffer(height, width, world, device):
    """Full framebuffer padded to world * tile_rows rows, AoS [rows, W, 3] float32."""
    return torch.zeros((tile_rows(height, world) * world, width, 3), dtype=torch.float32, device=device)


def gather_framebuffer(fb, rank, world, group=None):
    """In-place all-gather of the row tiles held in the padded framebuffer `fb`."""
    if world == 1:
        return fb
    tr = fb.shape[0] // world
    flat = fb.view(-1)
    n = tr * fb.shape[1] * fb.shape[2]
    dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n], group=group)
    return fb


def render_tiled(render_rows, height, width, rank, world, fb=None, device="cpu", group=None):
    """Renders this rank's tile with `render_rows(fb_view, row_begin, row_end)` — which must write rows
    [row_begin,row_end) of the [height, W, 3] view it is given — then gathers.  Returns the [height, W, 3] image."""
    if fb is None:
        fb = alloc_framebuffer(height, width, world, device)
    begin, end = row_tile(height, rank, world)
    if end > begin:
        render_rows(fb[:height], begin, end)
    gather_framebuffer(fb, rank, world, group=group)
    return fb[:height]
