"""N > 1 path on CPU: world_size-2 gloo.  The GPU kernels cannot run here, so the oracle (stream mode: the bit-level twin
of the device path) stands in as the renderer of each rank; what is under test is the product's distributed plumbing
(pathtrace_amd/distributed.py): NaiveSpiral tile ownership k mod N covers the film exactly once, and the one sum-reduce
of the RGBA framebuffer reproduces the single-process image BIT FOR BIT (disjoint ownership => the reduce adds zeros)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path

W, H, SPP, TILE = 96, 54, 3, 32


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, balanced):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pt_oracle
    from pathtrace_amd import spiral_tiles
    from pathtrace_amd.distributed import reduce_framebuffer, tiles_for_rank

    osc = pt_oracle.Scene.from_json(scene_path("cornell_box"))
    cfg = pt_oracle.make_config(W, H, SPP)
    fb3 = np.zeros((H, W, 3), np.float32)
    costs = None
    if balanced:   # per-tile ray counts of one sample per pixel, measured by every rank on its own (identical integers)
        costs = []
        scratch = np.zeros((H, W, 3), np.float32)
        for rect in spiral_tiles(W, H, TILE, TILE):
            _, c = osc.render_stream(cfg, seed=0, rect=rect, s0=0, s1=1, threads=2, fb=scratch)
            costs.append(c["rays"] + (rect[2] - rect[0]) * (rect[3] - rect[1]))
    mine = tiles_for_rank(W, H, TILE, TILE, rank, world, costs)
    for (x0, y0, x1, y1) in mine:
        osc.render_stream(cfg, seed=0, rect=(x0, y0, x1, y1), threads=2, fb=fb3)
    fb = torch.zeros((H, W, 4), dtype=torch.float32)
    fb[..., :3] = torch.from_numpy(fb3)
    reduce_framebuffer(fb, dst=0)
    counts = torch.tensor([len(mine)], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        np.save(out_path, fb.numpy())
        assert counts.item() == 6   # 3 x 2 tiles of 32 at 96 x 54
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("balanced", [False, True], ids=["round_robin", "cost_balanced"])
def test_tile_partition_and_reduce_world2(tmp_path, oracle, balanced):
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, balanced), nprocs=2, join=True)
    got = np.load(out)
    ref, _ = oracle.Scene.from_json(scene_path("cornell_box")).render_stream(oracle.make_config(W, H, SPP), seed=0, threads=2)
    assert np.array_equal(got[..., :3].view(np.uint32), ref.view(np.uint32))
    assert (got[..., 3] == 0).all()


def _exchange_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathtrace_amd.distributed import OwnedTileExchange, tiles_for_rank

    w, h, tile = 200, 120, 32   # ragged edge tiles (200 = 6 x 32 + 8, 120 = 3 x 32 + 24)
    lists = [tiles_for_rank(w, h, tile, tile, q, world) for q in range(world)]
    ex = OwnedTileExchange(lists, w, h, rank, world, torch.device("cpu"))
    # every rank fills ITS pixels with a value that names the pixel and the rank; foreign pixels hold garbage that must not travel
    fb = torch.full((h, w, 4), -7.0 - rank, dtype=torch.float32)
    flat = fb.view(-1, 4)
    idx = ex.mine
    flat[idx] = torch.stack([idx.float(), torch.full_like(idx, rank).float(), (idx % 13).float(), torch.zeros(len(idx))], dim=1)
    ex.run(fb)
    if rank == 0:
        np.save(out_path, fb.numpy())
        assert ex.bytes_moved() == sum(ex.counts[1:]) * 16 and sum(ex.counts) == w * h
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_owned_tile_exchange_assembles_the_frame(tmp_path, world):
    # bench.py's N > 1 exchange: every rank sends only the pixels of its tiles (one gather of 1/N-frame buffers), rank 0
    # writes them in place -- the assembled frame holds every pixel's owner's value and nothing else
    from pathtrace_amd.distributed import tiles_for_rank

    out = str(tmp_path / "fb.npy")
    mp.spawn(_exchange_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    w, h, tile = 200, 120, 32
    owner = np.full((h, w), -1, np.int64)
    for q in range(world):
        for (x0, y0, x1, y1) in tiles_for_rank(w, h, tile, tile, q, world):
            owner[y0:y1, x0:x1] = q
    pix = np.arange(w * h, dtype=np.float32).reshape(h, w)
    assert (owner >= 0).all()
    assert np.array_equal(got[..., 0], pix) and np.array_equal(got[..., 1], owner.astype(np.float32))
    assert np.array_equal(got[..., 2], (np.arange(w * h) % 13).astype(np.float32).reshape(h, w)) and (got[..., 3] == 0).all()


def test_tile_ownership_is_a_partition():
    from pathtrace_amd.distributed import tiles_for_rank

    for (w, h, world) in [(1920, 1080, 8), (3840, 2160, 8), (1920, 1080, 3), (200, 200, 2)]:
        cover = np.zeros((h, w), np.int32)
        n = []
        for r in range(world):
            t = tiles_for_rank(w, h, 128, 128, r, world)
            n.append(len(t))
            for x0, y0, x1, y1 in t:
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all() and max(n) - min(n) <= 1
    with pytest.raises(ValueError):
        tiles_for_rank(64, 64, 32, 32, 2, 2)


def test_cost_balanced_ownership():
    from pathtrace_amd import spiral_tiles
    from pathtrace_amd.distributed import balanced_owners, tiles_for_rank

    rng = np.random.default_rng(7)
    for (w, h, world) in [(1920, 1080, 8), (1920, 1080, 4), (3840, 2160, 8), (200, 200, 2), (96, 54, 3)]:
        tiles = spiral_tiles(w, h, 128, 128)
        # a cost field shaped like the Cornell box: expensive centre, cheap border
        cx = np.array([(t[0] + t[2]) / 2 / w - 0.5 for t in tiles])
        cy = np.array([(t[1] + t[3]) / 2 / h - 0.5 for t in tiles])
        costs = (1000 + 9000 * np.exp(-8 * (cx * cx + cy * cy)) + rng.integers(0, 50, len(tiles))).astype(np.int64).tolist()
        owner = balanced_owners(costs, world)
        assert owner == balanced_owners(list(costs), world)            # deterministic
        cover = np.zeros((h, w), np.int32)
        got = []
        for r in range(world):
            t = tiles_for_rank(w, h, 128, 128, r, world, costs)
            assert t == [tiles[k] for k in range(len(tiles)) if owner[k] == r]   # spiral order kept
            got += t
            for x0, y0, x1, y1 in t:
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all() and sorted(got) == sorted(tiles)
        load = lambda own: max(sum(c for c, o in zip(costs, own) if o == r) for r in range(world))
        rr = [k % world for k in range(len(tiles))]
        assert load(owner) <= load(rr)
        if len(tiles) >= 8 * world:
            assert load(owner) <= 1.02 * sum(costs) / world
    assert balanced_owners([], 3) == [] and balanced_owners([5, 5, 5], 1) == [0, 0, 0]
    assert balanced_owners([3, 3, 3, 3], 2) == [0, 1, 0, 1]               # ties: spiral index, then lowest rank
    with pytest.raises(ValueError):
        tiles_for_rank(256, 256, 128, 128, 0, 2, costs=[1, 2, 3])
    with pytest.raises(ValueError):
        balanced_owners([1, -1], 2)
