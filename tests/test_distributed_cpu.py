"""The N>1 path on CPU: world_size-2/3 gloo process groups exercise the tile partition, the gather collective
and the un-tiling exactly as DistributedRVPT uses them (the per-rank radiance comes from the oracle here,
because the HIP path needs a GPU; the GPU-side twin is tests/test_gpu_parity.py::test_gather_untile_on_gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import identity_camera, scene_by_name


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, W, H, full, out_path):
    from rvpt_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tx, ty = D.tile_grid(W, H)
        slot_quads = D.owned_tiles(tx * ty, 0, world) * 256  # every rank sends the largest slot size
        mine = D.tile_numpy(full, rank, world)                # this rank's tile-linear radiance
        assert mine.shape[0] == D.owned_tiles(tx * ty, rank, world) * 256
        slot = np.zeros((slot_quads, 4), np.float32)
        slot[: mine.shape[0]] = mine
        gathered = D.gather_slots(torch.from_numpy(slot.reshape(-1)), rank, world)
        if rank == 0:
            img = D.untile_numpy(gathered.numpy().reshape(world, -1, 4), W, H)
            np.save(out_path, img)
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 96, 64), (3, 100, 37), (2, 16, 16)])
def test_tile_partition_gather_untile_roundtrip(oracle, tmp_path, world, W, H):
    tris, mats, nodes = scene_by_name("default")
    cam = identity_camera(W / H)
    full, _ = oracle.render(oracle.settings_bytes(aa=1), cam, nodes, tris, mats, W, H, oracle.TRAVERSAL_BVH)
    out = tmp_path / "img.npy"
    mp.spawn(_worker, args=(world, _free_port(), W, H, full, str(out)), nprocs=world, join=True)
    assert np.array_equal(np.load(out), full)


def test_partition_covers_every_tile_once():
    from rvpt_amd import distributed as D
    for W, H in [(1920, 1080), (3840, 2160), (17, 33)]:
        tx, ty = D.tile_grid(W, H)
        for world in (1, 2, 3, 4, 8):
            counts = [D.owned_tiles(tx * ty, r, world) for r in range(world)]
            assert sum(counts) == tx * ty and max(counts) - min(counts) <= 1 and counts[0] == max(counts)


def test_tile_untile_numpy_are_inverse():
    from rvpt_amd import distributed as D
    rng = np.random.RandomState(1)
    W, H, world = 70, 45, 4
    img = rng.rand(H, W, 4).astype(np.float32)
    tx, ty = D.tile_grid(W, H)
    slot = D.owned_tiles(tx * ty, 0, world) * 256
    slots = np.zeros((world, slot, 4), np.float32)
    for r in range(world):
        t = D.tile_numpy(img, r, world)
        slots[r, : t.shape[0]] = t
    assert np.array_equal(D.untile_numpy(slots, W, H), img)
