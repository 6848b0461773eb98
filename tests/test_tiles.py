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


def _report_main(rank, world, port, out_path):
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank r pretends its kernel took 10 + 2r ms and the gather 0.5 + r ms
        rep = tiles.scaling_report(10.0 + 2.0 * rank, 0.5 + rank, 2160, 3840, 8)
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(rep, f)
    finally:
        dist.destroy_process_group()


def test_scaling_report_carries_every_ranks_figures(tmp_path):
    """What `bench.py --gpus N` adds to its line (`ranks`): per-rank kernel and
    gather times, the imbalance and the bytes of the one gather -- here over two
    gloo ranks with made-up timings; every rank ends up with the same report."""
    import json

    import torch.multiprocessing as mp

    world = 2
    out = tmp_path / "report"
    mp.spawn(_report_main, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    reps = [json.load(open(f"{out}.{r}")) for r in range(world)]
    assert reps[0] == reps[1]
    rep = reps[0]
    assert rep["ranks_seen"] == 2
    assert rep["kernel_ms"] == [10.0, 12.0] and rep["gather_ms"] == [0.5, 1.5]
    assert rep["imbalance"] == pytest.approx(12.0 / 11.0)
    # configuration 4 over two ranks: 1080 rows of 3840 RGB pixels each
    assert rep["gather"] == {"message_bytes_per_rank": 1080 * 3840 * 3,
                             "received_by_root": 1080 * 3840 * 3, "frame_bytes": 2160 * 3840 * 3}


def test_scaling_report_without_a_process_group_is_the_one_rank_case():
    rep = tiles.scaling_report(62.5, 0.0, 1080, 1920, 8)
    assert rep["ranks_seen"] == 1 and rep["kernel_ms"] == [62.5] and rep["imbalance"] == 1.0
    assert rep["gather"]["received_by_root"] == 0


def test_gather_bytes_match_the_survey_message_sizes():
    """at configuration 4 each rank sends 12.4 / 6.2 / 3.1 MB at G = 2 / 4 / 8."""
    for world, mb in ((2, 12.4), (4, 6.2), (8, 3.1)):
        b = tiles.gather_bytes(2160, 3840, world, 8)
        assert b["message_bytes_per_rank"] / 1e6 == pytest.approx(mb, abs=0.1)   # (padded to the largest shard)
