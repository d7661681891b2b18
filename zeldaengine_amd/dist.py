"""Screen-tile data parallelism across the GPUs of one node.

The frame is cut into 32x32 tiles, owned in super-tiles of 4 x 4 tiles (128 x 128 pixels) dealt round-robin with a skew
(`tile_owner` = zr_tile_owner): a meshlet nearly always falls to ONE rank, and only that rank transforms it.  Every rank holds the
whole scene, rejects what cannot reach its region before any vertex work, bins against its own tiles only, and lights its tiles
into a packed, tile-major RGBA8 buffer.  Collectives over xGMI per frame:

* composite: an all-gather of the packed tiles (4 B/pixel in total) followed by an untile kernel gives every rank the frame.  By
  default this is the ONLY collective (every rank renders the whole 1024^2 shadow map);
* `split_shadow="tiles"` adds one: the shadow MAP is owned by light-space super-tiles the way the frame is owned by screen super-tiles; a rank
  draws the casters that reach its tiles and the packed tiles are all-gathered (4 MB in total, nothing reduced: every texel has one owner);
* `split_shadow=True` ("split") adds another kind: rank r rasterises instances i % world == r into its own map and the maps are all-reduced
  with MIN (4 MB; the depth test LESS_OR_EQUAL is a min, so the split is exact), on the collective stream next to the camera passes.

`NativeDistributedRenderer` runs that loop inside the library (zr_dist_frame: RCCL called from C++, nothing of Python or torch in the
frame); `DistributedRenderer` is the same loop spelled with torch.distributed (any backend: the gloo tests use it).

Streams: the render stream (shadow share, lighting) and the library's camera lane produce frame k+1 while the collective
stream gathers and composites frame k (packed and gathered buffers are double-buffered, ordering is by events), so the xGMI
latency is hidden behind rendering.

`pack_tiles` / `untile` are the numpy statement of the packed layout (what k_lighting writes and k_untile reads); the
gloo tests use them, the GPU path uses the kernels.
"""
import numpy as np

TILE = 32
SUPERTILE_SHIFT = 2      # ZR_SUPERTILE_SHIFT: ownership unit = 4 x 4 tiles = 128 x 128 pixels
SUPERTILE_SKEW = 3       # ZR_SUPERTILE_SKEW


def tile_owner(tx, ty, world):
    """zr_tile_owner: super-tiles dealt round-robin along x, skewed per super-tile row."""
    return ((tx >> SUPERTILE_SHIFT) + (ty >> SUPERTILE_SHIFT) * SUPERTILE_SKEW) % world if world > 1 else 0


def owned_tiles(rank, world, tiles_x, tiles_y):
    """Tile indices a rank owns, increasing (= its slot order in the packed buffer)."""
    return [t for t in range(tiles_x * tiles_y) if tile_owner(t % tiles_x, t // tiles_x, world) == rank]


def tile_layout(width, height, world):
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    n = tx * ty
    return {"tiles_x": tx, "tiles_y": ty, "n_tiles": n,
            "slots_per_rank": max(len(owned_tiles(r, world, tx, ty)) for r in range(world))}


def pack_tiles(frame, rank, world, pad=0):
    """frame (H, W, 4) uint8 -> (slots_per_rank, 32, 32, 4): slot k holds the rank's k-th owned tile, padded with `pad`.
    Any plane does: a (D, D) float32 shadow map -> (slots_per_rank, 32, 32) with pad = 1.0 (zr_shadow_pack)."""
    H, W = frame.shape[:2]
    lay = tile_layout(W, H, world)
    out = np.full((lay["slots_per_rank"], TILE, TILE) + frame.shape[2:], pad, dtype=frame.dtype)
    for k, t in enumerate(owned_tiles(rank, world, lay["tiles_x"], lay["tiles_y"])):
        x0, y0 = (t % lay["tiles_x"]) * TILE, (t // lay["tiles_x"]) * TILE
        blk = frame[y0:y0 + TILE, x0:x0 + TILE]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def untile(gathered, width, height):
    """gathered (world, slots_per_rank, 32, 32, 4) -> (H, W, 4)  (or (world, slots, 32, 32) -> (H, W): zr_shadow_unpack)."""
    world = gathered.shape[0]
    lay = tile_layout(width, height, world)
    frame = np.zeros((height, width) + gathered.shape[4:], dtype=gathered.dtype)
    for r in range(world):
        for k, t in enumerate(owned_tiles(r, world, lay["tiles_x"], lay["tiles_y"])):
            x0, y0 = (t % lay["tiles_x"]) * TILE, (t // lay["tiles_x"]) * TILE
            h, w = min(TILE, height - y0), min(TILE, width - x0)
            frame[y0:y0 + h, x0:x0 + w] = gathered[r, k][:h, :w]
    return frame


class DistributedRenderer:
    """One rank of the screen-tile partition: Renderer + the per-frame all-gather + composite.

    world == 1 degenerates to the plain renderer (no collective).  The process group must already be initialised
    (backend "nccl" = RCCL on ROCm) when world > 1.  All buffers RCCL touches are torch allocations; the library renders
    straight into them (zr_set_tiles_buffer).
    """

    def __init__(self, width, height, shadow_dim=1024, device_index=0, rank=0, world=1, flags=0, pipeline=True,
                 split_shadow=False):
        import torch
        from . import engine
        self.torch = torch
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device_index)
        self.r = engine.Renderer(width, height, shadow_dim, device=device_index, tile_rank=rank, tile_world=world, flags=flags)
        self.render_stream = torch.cuda.Stream(self.device)
        self.comm_stream = torch.cuda.Stream(self.device) if (pipeline and world > 1) else self.render_stream
        self.r.set_stream(self.render_stream.cuda_stream)
        self.k = 0
        if world > 1:
            _, nbytes = self.r.tiles_device_buffer()
            self.tiles = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.gathered = [torch.empty(nbytes * world, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.rendered = [torch.cuda.Event() for _ in range(2)]      # tiles[b] is complete
            self.consumed = [torch.cuda.Event() for _ in range(2)]      # tiles[b] has been gathered (may be overwritten)
            for ev in self.consumed:
                ev.record(self.comm_stream)
            from . import abi
            self.shadow_mode = abi.shadow_mode(split_shadow)
            self.split_shadow = self.shadow_mode == "split"
            if self.split_shadow:
                self.shadow = torch.ones(shadow_dim * shadow_dim, dtype=torch.float32, device=self.device)
                self.r.set_shadow_buffer(self.shadow.data_ptr())
                self.r.set_shadow_partition(rank, world)
                self.shadow_done, self.shadow_reduced = torch.cuda.Event(), torch.cuda.Event()
            elif self.shadow_mode == "tiles":
                self.r.set_shadow_tiles(rank, world)
                nb = self.r.shadow_tiles_bytes()
                self.spacked = torch.ones(nb // 4, dtype=torch.float32, device=self.device)      # (unused slots: depth 1.0)
                self.sgathered = torch.empty(nb // 4 * world, dtype=torch.float32, device=self.device)
                self.shadow_reduced = torch.cuda.Event()

    def frame(self):
        """Enqueue one full frame; returns immediately."""
        torch = self.torch
        if self.world == 1:
            with torch.cuda.stream(self.render_stream):
                self.r.render()
            return
        import torch.distributed as dist
        b = self.k & 1
        self.k += 1
        # render stream: wait until buffer b's previous contents were gathered, then render into it
        self.render_stream.wait_event(self.consumed[b])
        self.r.set_tiles_buffer(self.tiles[b].data_ptr())
        with torch.cuda.stream(self.render_stream):
            if self.split_shadow:
                # this rank's share of the shadow casters (library's second stream) next to the camera passes (render stream)
                self.r.render_geometry()
                with torch.cuda.stream(self.comm_stream):       # min-reduce the maps as soon as the share is drawn
                    self.r.stream_wait_shadow(self.comm_stream.cuda_stream)
                    dist.all_reduce(self.shadow, op=dist.ReduceOp.MIN)
                    self.shadow_reduced.record(self.comm_stream)
                self.render_stream.wait_event(self.shadow_reduced)
                self.r.render_lighting()
            elif self.shadow_mode == "tiles":
                # this rank's share of the MAP, its owned tiles packed behind the pass (render stream), gathered on the collective stream,
                # scattered back into the map ahead of the lighting pass
                self.r.render_geometry()
                self.r.shadow_pack(self.spacked.data_ptr())
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_stream(self.render_stream)             # behind the pack
                    dist.all_gather_into_tensor(self.sgathered, self.spacked)
                    self.shadow_reduced.record(self.comm_stream)
                self.render_stream.wait_event(self.shadow_reduced)
                self.r.shadow_unpack(self.sgathered.data_ptr())
                self.r.render_lighting()
            else:
                self.r.render()
            self.rendered[b].record(self.render_stream)
        # collective stream: gather + composite frame k while the render stream starts frame k + 1
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.rendered[b])
            dist.all_gather_into_tensor(self.gathered[b], self.tiles[b])
            self.consumed[b].record(self.comm_stream)
            self.r.set_stream(self.comm_stream.cuda_stream)
            self.r.composite(self.gathered[b].data_ptr())
            self.r.set_stream(self.render_stream.cuda_stream)

    def synchronize(self):
        self.render_stream.synchronize()
        self.comm_stream.synchronize()

    def close(self):
        self.synchronize()
        self.r.close()


def make_distributed(width, height, shadow_dim=1024, device_index=0, rank=0, world=1, flags=0, split_shadow=False, native=False):
    """One rank of the partition.  native=True: the library's own RCCL host (zr_dist_*, no Python or torch in the frame loop).  When
    that host cannot be brought up on EVERY rank (librccl missing, communicator refused), all ranks fall back together to the
    torch.distributed loop and the renderer carries the reason in `.native_fallback`."""
    if native and world > 1:
        dr, why = _try_native(width, height, shadow_dim, device_index, rank, world, flags, split_shadow)
        if dr is not None:
            return dr
        d = DistributedRenderer(width, height, shadow_dim, device_index, rank, world, flags, split_shadow=split_shadow)
        d.native_fallback = why
        return d
    return DistributedRenderer(width, height, shadow_dim, device_index, rank, world, flags, split_shadow=split_shadow)


def _agree(ok):
    """MIN over ranks of a flag, on whatever the process group offers (a CPU tensor when gloo is there: no RCCL communicator of
    torch's is created for it)."""
    import torch
    import torch.distributed as dist
    backend = str(dist.get_backend())
    dev = torch.device("cpu") if "gloo" in backend else torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def _try_native(width, height, shadow_dim, device_index, rank, world, flags, split_shadow):
    """-> (NativeDistributedRenderer, None) or (None, reason); every rank returns the same kind.

    ncclCommInitRank is a collective: a rank that fails BEFORE it (no context, no librccl, no memory) would leave the others blocked
    inside it.  So the bring-up has two agreed steps: (1) everything local - the id on rank 0, the renderer, zr_dist_prepare - then a
    MIN over ranks; only if every rank got that far (2) zr_dist_connect, and a second MIN on its outcome."""
    import torch.distributed as dist
    from . import engine
    uid, err = None, ""
    if rank == 0:
        try:
            uid = engine.dist_unique_id()
        except Exception as e:      # noqa: BLE001
            err = "zr_dist_unique_id: %s" % e
    box = [uid, err]
    dist.broadcast_object_list(box, src=0)
    uid, err0 = box
    dr, err = None, ""
    if uid is not None:
        try:
            dr = NativeDistributedRenderer(width, height, shadow_dim, device_index, rank, world, flags, split_shadow, connect=False)
        except Exception as e:          # noqa: BLE001
            err = "rank %d: %s" % (rank, e)
    if not _agree(dr is not None):
        if dr is not None:
            dr.close()
        return None, err0 or err or "the native RCCL host could not be prepared on another rank"
    try:
        dr.connect(uid)
        ok = True
    except Exception as e:              # noqa: BLE001
        ok, err = False, "rank %d: %s" % (rank, e)
    if not _agree(ok):
        dr.close()
        return None, err or "ncclCommInitRank failed on another rank"
    return dr, None


class NativeDistributedRenderer:
    """One rank of the partition with the library's own RCCL host (zr_dist_*).  torch.distributed (already initialised when
    world > 1) is used ONCE, to hand rank 0's ncclUniqueId to the other ranks; frames are enqueued by one C call each."""

    def __init__(self, width, height, shadow_dim=1024, device_index=0, rank=0, world=1, flags=0, split_shadow=False, unique_id=None,
                 connect=True):
        import torch
        from . import abi, engine
        self.torch = torch
        self.rank, self.world = rank, world
        if world == 1:
            flags |= abi.FLAG_PACKED_TILES
        self.r = engine.Renderer(width, height, shadow_dim, device=device_index, tile_rank=rank, tile_world=world, flags=flags)
        try:
            self.r.dist_prepare(rank, world, split_shadow)          # local: may fail on this rank alone
        except Exception:
            self.r.close()
            raise
        if connect:                                                  # (hosts that agree between the two steps pass connect=False)
            uid = unique_id
            if uid is None:
                uid = engine.dist_unique_id() if rank == 0 else bytes(128)
                if world > 1:
                    import torch.distributed as dist
                    box = [uid]
                    dist.broadcast_object_list(box, src=0)
                    uid = box[0]
            self.connect(uid)

    def connect(self, unique_id):
        try:
            self.r.dist_connect(unique_id)                           # ncclCommInitRank: collective
        except Exception:
            self.r.close()
            raise

    def frame(self):
        self.r.dist_frame()

    def synchronize(self):
        self.r.finish()

    def close(self):
        self.r.close()
