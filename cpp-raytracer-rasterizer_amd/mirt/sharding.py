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
    """Gathers per-rank row bands of an (H, W) int32 frame on rank `dst`.

    dist.gather needs equally sized tensors, so bands are padded to the tallest band (they differ by at most
    one row); the root copies the valid rows of each slot into the frame.  With H divisible by world the
    slots ARE the frame rows and no extra copy happens.
    """

    def __init__(self, H, W, device, dst=0, depth=1):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.H, self.W, self.dst = H, W, dst
        self.bands = all_bands(self.world, H)
        self.y0, self.y1 = self.bands[self.rank]
        self.max_rows = max(b - a for a, b in self.bands)
        self.even = all((b - a) == self.max_rows for a, b in self.bands)
        # this rank's band buffer(s) (always max_rows tall so every rank sends the same shape); with depth 2 the
        # render of frame i+1 can overlap the gather of frame i (see bench.py)
        self.bands_buf = [torch.zeros((self.max_rows, W), dtype=torch.int32, device=device) for _ in range(depth)]
        self.band = self.bands_buf[0]
        self.frame = None
        self.slots = None
        if self.rank == dst:
            if self.even:
                self.frame = torch.zeros((H, W), dtype=torch.int32, device=device)
                self.slots = [self.frame[a:b] for a, b in self.bands]
            else:
                self.frame = torch.zeros((H, W), dtype=torch.int32, device=device)
                self.staging = torch.zeros((self.world, self.max_rows, W), dtype=torch.int32, device=device)
                self.slots = [self.staging[r] for r in range(self.world)]

    def gather(self, which=0):
        """Collective: after it returns (stream-ordered for nccl), rank dst's `frame` holds the whole image."""
        band = self.bands_buf[which]
        if self.world == 1:
            self.frame[self.y0:self.y1].copy_(band[: self.y1 - self.y0])
            return self.frame
        dist.gather(band, self.slots if self.rank == self.dst else None, dst=self.dst)
        if self.rank == self.dst and not self.even:
            for r, (a, b) in enumerate(self.bands):
                self.frame[a:b].copy_(self.staging[r, : b - a])
        return self.frame
