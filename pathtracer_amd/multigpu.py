"""Row-tile data parallelism over the GPUs of one node.

Every (pixel, sample) is independent, the RNG is keyed on the GLOBAL pixel index
and pixel sums are exact, so the image does not depend on how rows are split.
Rank r of `world` renders the rows r, r + world, r + 2*world, ... (interleaved:
rows through the glass spheres cost more than wall rows, so contiguous tiles
would load the GPUs unevenly) into a compact local tile; one all-gather (RCCL
over xGMI under the "nccl" backend) collects the tiles and a strided copy puts
the rows back in image order.  Tiles are padded to ceil(H/world) rows so that
all ranks contribute equal-sized slices, as all_gather_into_tensor requires.
"""
import torch
import torch.distributed as dist


def tile_rows(height, world):
    """Rows per rank in the padded layout (the last ranks may own one real row fewer)."""
    return (height + world - 1) // world


def rank_rows(height, rank, world):
    """The image rows rank renders, as a range."""
    return range(rank, height, world)


def exchange_comm_id(rank, make_id, group=None):
    """The id of the library's own RCCL communicator (pt_comm_unique_id), made by rank 0 with `make_id()` and handed
    to every rank through the torch process group that is already up (one broadcast of 128 bytes; gloo or nccl).
    Every rank then calls pt_comm_init_rank with it (binding.Comm)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ident = bytes(box[0])
    assert len(ident) == 128, len(ident)
    return ident


class TiledFrame:
    """Buffers of one frame in flight: the local compact tile, the gathered tiles and the assembled image.
    comm: a binding.Comm — the gather then goes through the C ABI (pt_gather_framebuffer: ncclAllGather + the
    library's reordering kernel on the current stream) instead of torch's all_gather_into_tensor + strided copy."""

    def __init__(self, height, width, world, device, collective=True, comm=None):
        self.height, self.width, self.world = height, width, world
        self.comm = comm
        self.collective = collective     # False: development runs of one rank's pipeline without a process group
        self.rows = tile_rows(height, world)
        self.local = torch.zeros((self.rows, width, 3), dtype=torch.float32, device=device)
        if world == 1:
            self.gathered = self.local
            self.image = self.local[:height]
        else:
            self.gathered = torch.zeros((world * self.rows, width, 3), dtype=torch.float32, device=device)
            self.image = torch.zeros((height, width, 3), dtype=torch.float32, device=device)

    def gather(self, group=None):
        """All-gather the local tiles and put the rows back in image order."""
        if self.world == 1:
            return self.image
        if self.comm is not None:
            return self.comm.gather(self.local, self.gathered, self.image, self.height, self.width)
        if self.collective and self.local.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsals of the N-rank flow on a box with one GPU: the gather goes through host memory
            host = torch.empty(self.gathered.numel(), dtype=torch.float32)
            dist.all_gather_into_tensor(host, self.local.reshape(-1).cpu(), group=group)
            self.gathered.view(-1).copy_(host)
        elif self.collective:
            dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=group)
        else:
            self.gathered.view(self.world, -1)[0].copy_(self.local.view(-1))
        # gathered[r, k] is image row k*world + r
        ordered = self.gathered.view(self.world, self.rows, self.width, 3).permute(1, 0, 2, 3).reshape(self.world * self.rows, self.width, 3)
        self.image.copy_(ordered[:self.height])
        return self.image


def render_tiled(render_rows, height, width, rank, world, frame=None, device="cpu", group=None):
    """Renders this rank's rows with `render_rows(local, row_begin, row_step)` — which must write image row
    row_begin + k*row_step into local[k] — then gathers.  Returns the [height, W, 3] image."""
    if frame is None:
        frame = TiledFrame(height, width, world, device)
    if rank < height:
        render_rows(frame.local, rank, world)
    return frame.gather(group=group)
