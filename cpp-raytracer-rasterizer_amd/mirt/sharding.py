"""Screen-band sharding of one frame across the GPUs of a node.

Every pixel is independent in both renderers, so the frame is split into `world` contiguous bands of rows;
rank r renders rows [y0, y1) (mirt_raytrace_device / mirt_rasterise_device with the triangle list replicated)
and the XRGB bands are gathered on rank 0 -- the one real exchange step of the path.  The collective is
torch.distributed's gather (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests): each peer sends its
band straight to the root on its own link, recv buffers are views of the full frame so bands land in place.
"""
import torch
import torch.distributed as dist


def band_of(rank, world, H):
    """Rows [y0, y1) of `rank` when H rows are split into `world` contiguous bands (sizes differ by at most 1)."""
    base, rem = divmod(H, world)
    y0 = rank * base + min(rank, rem)
    return y0, y0 + base + (1 if rank < rem else 0)


def all_bands(world, H):
    return [band_of(r, world, H) for r in range(world)]


class BandGather:
    """Gathers per-rank row bands of (H, W) int32 frames on rank `dst`.

    dist.gather needs equally sized tensors, so bands are padded to the tallest band (they differ by at most
    one row); the root copies the valid rows of each slot into the frame.  With H divisible by world and one frame
    per collective the slots ARE the frame rows and no extra copy happens.

    batch > 1: a band buffer holds `batch` consecutive frames' bands, (batch, max_rows, W), and ONE collective moves
    them all -- fewer, larger messages for frames that take less time to render than a collective takes to start
    (the Cornell box at 1080p renders in ~25 us).  The root then holds `frames`, (batch, H, W).
    """

    def __init__(self, H, W, device, dst=0, depth=1, batch=1, via_host=False):
        self.via_host = via_host          # rehearsal on one GPU: the collective runs on host copies (gloo has no device gather)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.H, self.W, self.dst, self.batch = H, W, dst, int(batch)
        self.bands = all_bands(self.world, H)
        self.y0, self.y1 = self.bands[self.rank]
        self.max_rows = max(b - a for a, b in self.bands)
        self.even = all((b - a) == self.max_rows for a, b in self.bands)
        # this rank's band buffer(s) (always max_rows tall so every rank sends the same shape); with depth 2 the
        # render of the next frame(s) can overlap the gather of the previous ones (see bench.py)
        shape = (self.max_rows, W) if self.batch == 1 else (self.batch, self.max_rows, W)
        self.bands_buf = [torch.zeros(shape, dtype=torch.int32, device=device) for _ in range(depth)]
        self.band = self.bands_buf[0]
        self.frame = None
        self.frames = None
        self.slots = None
        self.staging = None
        if self.rank == dst:
            if self.batch == 1:
                self.frame = torch.zeros((H, W), dtype=torch.int32, device=device)
                if self.even:
                    self.slots = [self.frame[a:b] for a, b in self.bands]
                else:
                    self.staging = torch.zeros((self.world, self.max_rows, W), dtype=torch.int32, device=device)
                    self.slots = [self.staging[r] for r in range(self.world)]
            else:
                self.frames = torch.zeros((self.batch, H, W), dtype=torch.int32, device=device)
                self.frame = self.frames[self.batch - 1]
                self.staging = torch.zeros((self.world, self.batch, self.max_rows, W), dtype=torch.int32, device=device)
                self.slots = [self.staging[r] for r in range(self.world)]

    def slot(self, which, b=0):
        """The (max_rows, W) tensor frame `b` of band buffer `which` renders into."""
        return self.bands_buf[which] if self.batch == 1 else self.bands_buf[which][b]

    def gather(self, which=0):
        """Collective: after it returns (stream-ordered for nccl), rank dst's `frame` (batch == 1) or `frames` holds
        the whole image(s)."""
        band = self.bands_buf[which]
        if self.world == 1:
            rows = self.y1 - self.y0
            if self.batch == 1:
                self.frame[self.y0:self.y1].copy_(band[:rows])
            else:
                self.frames[:, self.y0:self.y1].copy_(band[:, :rows])
            return self.frame
        if self.via_host:
            torch.cuda.current_stream().synchronize()
            host = [torch.empty(band.shape, dtype=band.dtype) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(band.cpu(), host, dst=self.dst)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.slots[r].copy_(host[r])
        else:
            dist.gather(band, self.slots if self.rank == self.dst else None, dst=self.dst)
        if self.rank == self.dst:
            if self.batch > 1:
                for r, (a, b) in enumerate(self.bands):
                    self.frames[:, a:b].copy_(self.staging[r, :, : b - a])
            elif not self.even:
                for r, (a, b) in enumerate(self.bands):
                    self.frame[a:b].copy_(self.staging[r, : b - a])
        return self.frame
