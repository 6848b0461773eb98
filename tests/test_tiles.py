"""Row-tile sharding and the one gather, on CPU: shard arithmetic, and a
world_size-2 gloo run of the N>1 frame path in which each rank's rows are
rendered by the oracle (the GPU renderer's stand-in here; the -m gpu tests run
the same shard parameters through the kernel)."""
import os
import socket

import numpy as np
import pytest

from path_tracer_c_amd import tiles


@pytest.mark.parametrize("height,world,tile", [(1080, 8, 8), (1080, 4, 8), (2160, 8, 8),
                                               (37, 2, 4), (5, 4, 2), (64, 3, 16), (9, 2, 1)])
def test_shards_partition_the_rows(height, world, tile):
    seen = []
    for r in range(world):
        sh = tiles.shard_rows(height, world, r, tile)
        rows = sh.global_rows()
        assert len(rows) == sh.num_rows
        assert all(0 <= y < height for y in rows)
        # tile t belongs to rank t mod world
        assert all((y // tile) % world == r for y in rows)
        seen += rows
    assert sorted(seen) == list(range(height))
    assert tiles.max_shard_rows(height, world, tile) >= (height + world - 1) // world


def test_single_rank_shard_is_the_whole_frame():
    sh = tiles.shard_rows(1080, 1, 0, 8)
    assert (sh.row_begin, sh.num_rows, sh.tile_stride) == (0, 1080, 1)
    assert sh.global_rows() == list(range(1080))


def test_default_tile_rows_gives_every_rank_work():
    for h, w in [(1080, 8), (2160, 8), (16, 8), (7, 4), (3, 2)]:
        t = tiles.default_tile_rows(h, w)
        assert t >= 1
        if h >= w:
            assert all(tiles.shard_rows(h, w, r, t).num_rows > 0 for r in range(w))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, height, width, tile, out_path):
    import torch
    import torch.distributed as dist

    import path_tracer_c_amd as pt
    from oracle import binding as oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = tiles.shard_rows(height, world, rank, tile)
        pad = tiles.max_shard_rows(height, world, tile)
        local = np.zeros((pad, width, 3), dtype=np.uint8)
        with pt.HostScene(1) as scene:
            full = np.zeros((height, width, 3), dtype=np.uint8)
            for i, y in enumerate(sh.global_rows()):
                oracle.render_rows(scene, width, height, y, y + 1, frame=full, spp=4, seed=11)
                local[i] = full[y]
        frame = tiles.gather_frame(torch.from_numpy(local), height, width, tile)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,tile", [(24, 4), (21, 4)])
def test_two_rank_gloo_gather_reassembles_the_frame(tmp_path, pt, oracle, height, tile):
    import torch.multiprocessing as mp

    width, world = 40, 2
    out = tmp_path / "frame.npy"
    mp.spawn(_rank_main, args=(world, _free_port(), height, width, tile, str(out)),
             nprocs=world, join=True)
    got = np.load(out)
    with pt.HostScene(1) as scene:
        ref, _ = oracle.render_rows(scene, width, height, spp=4, seed=11)
    assert np.array_equal(got, ref)
