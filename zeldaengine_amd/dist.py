"""Screen-tile data parallelism across the GPUs of one node.

The frame is cut into 32x32 tiles; tile t belongs to rank t % world.  Every rank holds the whole scene, culls and bins
against its own tiles only, renders the (1024^2) shadow map redundantly, and lights its tiles into a packed, tile-major
RGBA8 buffer.  ONE collective per frame — an RCCL all-gather of those buffers over xGMI (4 B/pixel in total) — followed by
an untile kernel gives every rank the full frame.  There is no other exchange step.

`pack_tiles` / `untile` are the numpy statement of the packed layout (what k_lighting writes and k_untile reads); the
gloo tests use them, the GPU path uses the kernels.
"""
import numpy as np

TILE = 32


def tile_layout(width, height, world):
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    n = tx * ty
    return {"tiles_x": tx, "tiles_y": ty, "n_tiles": n, "slots_per_rank": (n + world - 1) // world}


def owned_tiles(rank, world, n_tiles):
    return list(range(rank, n_tiles, world))


def pack_tiles(frame, rank, world):
    """frame (H, W, 4) uint8 -> (slots_per_rank, 32, 32, 4): slot k holds tile rank + k * world, zero padded."""
    H, W = frame.shape[:2]
    lay = tile_layout(W, H, world)
    out = np.zeros((lay["slots_per_rank"], TILE, TILE, 4), dtype=np.uint8)
    for k, t in enumerate(owned_tiles(rank, world, lay["n_tiles"])):
        x0, y0 = (t % lay["tiles_x"]) * TILE, (t // lay["tiles_x"]) * TILE
        blk = frame[y0:y0 + TILE, x0:x0 + TILE]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def untile(gathered, width, height):
    """gathered (world, slots_per_rank, 32, 32, 4) -> (H, W, 4)."""
    world = gathered.shape[0]
    lay = tile_layout(width, height, world)
    frame = np.zeros((height, width, 4), dtype=np.uint8)
    for t in range(lay["n_tiles"]):
        x0, y0 = (t % lay["tiles_x"]) * TILE, (t // lay["tiles_x"]) * TILE
        blk = gathered[t % world, t // world]
        h, w = min(TILE, height - y0), min(TILE, width - x0)
        frame[y0:y0 + h, x0:x0 + w] = blk[:h, :w]
    return frame


class _DevBuf:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class DistributedRenderer:
    """One rank of the screen-tile partition: Renderer + the per-frame all-gather + composite.

    world == 1 degenerates to the plain renderer (no collective).  The process group must already be initialised
    (backend "nccl" = RCCL on ROCm) when world > 1.
    """

    def __init__(self, width, height, shadow_dim=1024, device_index=0, rank=0, world=1, flags=0):
        import torch
        from . import engine
        self.torch = torch
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device_index)
        self.r = engine.Renderer(width, height, shadow_dim, device=device_index, tile_rank=rank, tile_world=world, flags=flags)
        # one explicit stream carries render -> all-gather -> composite, so the collective is ordered after the lighting
        # kernel that fills the packed tiles and before the untile kernel that reads the gathered buffer
        self.stream = torch.cuda.Stream(self.device)
        self.r.set_stream(self.stream.cuda_stream)
        self.tiles = self.gathered = None
        if world > 1:
            ptr, nbytes = self.r.tiles_device_buffer()
            self.tiles = torch.as_tensor(_DevBuf(ptr, nbytes), device=self.device)
            self.gathered = torch.empty(nbytes * world, dtype=torch.uint8, device=self.device)

    def frame(self):
        """Enqueue one full frame (asynchronous on self.stream)."""
        with self.torch.cuda.stream(self.stream):
            self.r.render()
            if self.world > 1:
                import torch.distributed as dist
                dist.all_gather_into_tensor(self.gathered, self.tiles)
                self.r.composite(self.gathered.data_ptr())

    def synchronize(self):
        self.stream.synchronize()

    def close(self):
        self.r.close()
