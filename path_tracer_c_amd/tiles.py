"""Row-tile sharding of one frame over the GPUs of a node, and the single
gather that reassembles it.

The reference's unit of parallel work is one image row per task
(include/renderer.h:6-11, src/threadpool.c:84-95).  Across GPUs the frame is
cut into tiles of `tile_rows` consecutive rows dealt round-robin: tile t goes
to rank t mod G, so sky rows and floor rows are spread evenly.  Because the
sample stream is a function of the GLOBAL pixel index (include/ptgpu_rng.h),
the assembled frame does not depend on G or on tile_rows: it equals the
one-GPU frame byte for byte.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI, or
"gloo" in the CPU tests); the only collective is one `gather` of the ranks'
compact row buffers to rank 0 at the end of a frame.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class RowShard:
    """The rows rank `rank` of `world` renders, in ptgpu_params terms."""
    row_begin: int
    num_rows: int
    tile_rows: int
    tile_stride: int

    def global_rows(self):
        """Global row of each local row, in local order."""
        rows = []
        for i in range(self.num_rows):
            tile, within = divmod(i, self.tile_rows)
            rows.append(self.row_begin + tile * self.tile_rows * self.tile_stride + within)
        return rows


def shard_rows(height: int, world: int, rank: int, tile_rows: int) -> RowShard:
    """Interleaved tiles: rank r owns tiles r, r+G, r+2G, ...  The last tile of
    the frame may be short; it is then the last tile of its owner."""
    if not (0 <= rank < world) or tile_rows <= 0 or height <= 0:
        raise ValueError("bad shard request")
    n_tiles = (height + tile_rows - 1) // tile_rows
    mine = range(rank, n_tiles, world)
    rows = 0
    for t in mine:
        rows += min(tile_rows, height - t * tile_rows)
    return RowShard(row_begin=rank * tile_rows, num_rows=rows,
                    tile_rows=tile_rows, tile_stride=world)


def max_shard_rows(height: int, world: int, tile_rows: int) -> int:
    return max(shard_rows(height, world, r, tile_rows).num_rows for r in range(world))


def default_tile_rows(height: int, world: int) -> int:
    """8-row tiles (one 1080p tile = 15 workgroups of 1024 pixels), shrunk for
    small frames so every rank gets work."""
    t = 8
    while t > 1 and height < t * world:
        t //= 2
    return t


def scatter_index(height: int, world: int, tile_rows: int):
    """For the stacked gather buffer of shape (world * pad, W, 3): the frame
    row each buffer row belongs to; padding rows point at a spare row
    `height` that the caller drops."""
    pad = max_shard_rows(height, world, tile_rows)
    index = []
    for r in range(world):
        rows = shard_rows(height, world, r, tile_rows).global_rows()
        index += rows + [height] * (pad - len(rows))
    return index


def gather_frame(local_rows, height: int, width: int, tile_rows: int,
                 group=None, dst: int = 0, scratch=None):
    """Gather the ranks' compact row buffers into the full frame on `dst`.

    `local_rows` is a uint8 tensor of shape (padded_rows, width, 3) where
    padded_rows = max_shard_rows(...) on every rank (equal message sizes for
    the one gather); only the first shard.num_rows rows are meaningful.
    Returns the (height, width, 3) frame on `dst`, None elsewhere.  `scratch`
    (dst only) may carry {"stack", "index", "frame"} tensors between calls.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pad = max_shard_rows(height, world, tile_rows)
    assert local_rows.shape == (pad, width, 3), (local_rows.shape, pad)
    if rank != dst:
        dist.gather(local_rows, gather_list=None, dst=dst, group=group)
        return None
    if scratch is None:
        scratch = {}
    if "stack" not in scratch:
        scratch["stack"] = torch.empty((world, pad, width, 3), dtype=local_rows.dtype,
                                       device=local_rows.device)
        scratch["index"] = torch.tensor(scatter_index(height, world, tile_rows),
                                        dtype=torch.long, device=local_rows.device)
        scratch["frame"] = torch.empty((height + 1, width, 3), dtype=local_rows.dtype,
                                       device=local_rows.device)
    stack = scratch["stack"]
    dist.gather(local_rows, gather_list=[stack[r] for r in range(world)], dst=dst, group=group)
    # one scatter un-interleaves all ranks' tiles (padding rows land in the spare row)
    scratch["frame"].index_copy_(0, scratch["index"], stack.view(world * pad, width, 3))
    return scratch["frame"][:height]


def gather_bytes(height: int, width: int, world: int, tile_rows: int) -> dict:
    """What the one gather of a frame moves: every rank sends its padded compact
    buffer (equal message sizes), rank 0 receives world - 1 of them over xGMI."""
    pad = max_shard_rows(height, world, tile_rows)
    message = pad * width * 3
    return {"message_bytes_per_rank": message, "received_by_root": message * (world - 1),
            "frame_bytes": height * width * 3}


def scaling_report(kernel_ms: float, gather_ms: float, height: int, width: int,
                   tile_rows: int, group=None, device=None) -> dict:
    """Per-rank figures that make a scaling curve explainable (all ranks call
    this; every rank gets the same dict): each rank's kernel time per frame
    (HIP events around its render launches), the time of the gather as each
    rank saw it, the imbalance max / mean of the kernel times, the bytes the
    gather moved, and how many ranks the process group really has.  One
    all_gather of two float64 per rank, on `device` ("cuda" under the nccl
    backend = RCCL, "cpu" under gloo)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        mine = torch.tensor([float(kernel_ms), float(gather_ms)], dtype=torch.float64,
                            device=device or "cpu")
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine, group=group)
        both = [tuple(float(x) for x in t.tolist()) for t in both]
    else:
        both = [(float(kernel_ms), float(gather_ms))]
    kernel = [b[0] for b in both]
    gather = [b[1] for b in both]
    mean = sum(kernel) / len(kernel)
    return {"ranks_seen": world,
            "kernel_ms": kernel,
            "gather_ms": gather,
            "imbalance": (max(kernel) / mean) if mean > 0 else None,
            "gather": gather_bytes(height, width, world, tile_rows)}
