"""One process per GPU: image tile partition + RCCL gather of per-tile radiance.

The reference is single-device; this layer is new.  Every pixel is independent and the
RNG is keyed on the global pixel index (util.glsl:35-36), so rank r of n simply owns the 16x16 tiles
t with t % n == r (interleaved for load balance), keeps their RGBA32F accumulator resident across frames,
and exchanges nothing while rendering.  Only when a full frame is requested do the ranks run ONE collective:
a gather of each rank's tile-linear buffer to rank 0 followed by an un-tiling kernel there.  The collective lives in
the LIBRARY (rvpt_hip_comm_init + rvpt_hip_gather / rvpt_hip_comm_barrier: grouped ncclSend/ncclRecv and a one-float
all-reduce on its own RCCL communicator, every peer on its own xGMI link to the root) and is the ONLY RCCL communicator of
the process: torch.distributed is the control plane — a `gloo` process group carries the 128-byte communicator id from
rank 0 to the others, lets the ranks agree that every one of them can join before any of them blocks in the bootstrap, and
reduces host-side scalars.  Without a library communicator (RVPT_NO_LIBRARY_COMM, RCCL missing, several test ranks
sharing one GPU) the gather is staged through the host over the same gloo group — same frame, not a fast path.  All
rendering goes through the C ABI.
"""
from __future__ import annotations

import os

import numpy as np

from . import native
from .renderer import RVPT


class _DeviceBuffer:
    """Zero-copy view of a raw device allocation for torch.as_tensor."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def owned_tiles(n_tiles: int, rank: int, world: int) -> int:
    return (n_tiles - rank + world - 1) // world if n_tiles > rank else 0


def tile_slot(tx, ty, tiles_x: int):
    """Slot of the tile at (tx, ty): row-major with every row rotated by TILE_SHIFT more tiles than the one above (include/rvpt_hip.h);
    rank = slot % world, local tile = slot // world.  Works on ints and numpy arrays."""
    return ty * tiles_x + (tx + native.TILE_SHIFT * ty) % tiles_x


def tile_grid(width: int, height: int):
    t = native.TILE
    return (width + t - 1) // t, (height + t - 1) // t


def untile_numpy(slots: np.ndarray, width: int, height: int) -> np.ndarray:
    """CPU statement of rvpt_hip_untile (used by the gloo tests and as documentation of the layout):
    slots[rank, local_tile*256 + (y%16)*16 + (x%16), 4] -> row-major [H, W, 4]."""
    world = slots.shape[0]
    tx, ty = tile_grid(width, height)
    slots = slots.reshape(world, -1, 4)
    y, x = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    tile = tile_slot(x // 16, y // 16, tx)
    idx = (tile // world) * 256 + (y % 16) * 16 + (x % 16)
    return slots[tile % world, idx]


def tile_numpy(img: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Inverse of untile_numpy for one rank: row-major image -> that rank's tile-linear slot."""
    height, width = img.shape[:2]
    tx, ty = tile_grid(width, height)
    n = owned_tiles(tx * ty, rank, world)
    out = np.zeros((n * 256, 4), dtype=img.dtype)
    for j in range(n):
        t = j * world + rank
        row = t // tx
        y0, x0 = row * 16, ((t % tx - native.TILE_SHIFT * row) % tx) * 16
        blk = np.zeros((16, 16, 4), dtype=img.dtype)
        sub = img[y0:y0 + 16, x0:x0 + 16]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        out[j * 256:(j + 1) * 256] = blk.reshape(256, 4)
    return out


def _host_staged() -> bool:
    """The process group cannot move device memory (gloo: several ranks sharing one GPU in tests, hosts without RCCL)."""
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() != "nccl"


def gather_slots(local_slot, rank: int, world: int, dst: int = 0):
    """Collective fallback (no library communicator: RVPT_NO_LIBRARY_COMM, RCCL not loadable, a rank that could not join): gather
    equally-sized 1-D tensors to `dst` over the caller's process group — staged through host memory when the group cannot move device
    memory (gloo: the recommended control plane, and ranks sharing one GPU in tests), device to device when the caller made an nccl
    group (then that group's communicator carries it).  Returns [world, n] on dst's device, None elsewhere.  A correctness path."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_slot.reshape(1, -1)
    src = local_slot.cpu() if _host_staged() else local_slot
    if rank == dst:
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.gather(src, parts, dst=dst)
        return torch.stack(parts).to(local_slot.device)
    dist.gather(src, None, dst=dst)
    return None


class DistributedRVPT:
    """RVPT over `world` ranks, one GPU each.  Same host interface as RVPT; read_frame() is collective."""

    def __init__(self, width: int, height: int, traversal: str = "brute", flags: int = 0, rank=None, world=None,
                 device=None):
        import torch
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
        self.device = int(os.environ.get("LOCAL_RANK", 0)) if device is None else device
        torch.cuda.set_device(self.device)
        self.width, self.height = width, height
        self.local = RVPT(width, height, device=self.device, traversal=traversal, tile_rank=self.rank,
                          tile_world=self.world, flags=flags)
        self._slot = None
        self.library_comm = False  # set by initialize(): the gather runs inside the C ABI

    def initialize(self) -> bool:
        ok = self.local.initialize()
        if ok and (self.world > 1 or os.environ.get("RVPT_FORCE_COLLECTIVE")) and not os.environ.get("RVPT_NO_LIBRARY_COMM"):
            self.library_comm = self._init_library_comm()
        return ok

    def _init_library_comm(self) -> bool:
        """Rank 0 makes the RCCL id, torch.distributed (or nothing, for a single rank) carries it, every rank joins."""
        import sys
        import torch

        def all_agree(ok: bool) -> bool:  # every rank takes the same path, or the collectives would not match up
            if self.world == 1:
                return ok
            import torch.distributed as dist
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if _host_staged() else f"cuda:{self.device}")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        why = ""
        try:
            if os.environ.get("RVPT_TEST_FAIL_COMM_RANK") == str(self.rank):  # tests: one rank cannot join
                raise native.NativeError(native.ERR_COMM, "injected failure (RVPT_TEST_FAIL_COMM_RANK)")
            uid = [native.comm_unique_id()]  # on every rank: also tells whether RCCL can be loaded here at all
            usable = True
        except Exception as e:
            uid, usable, why = [None], False, str(e)
        if not all_agree(usable):
            print(f"[rvpt_amd] rank {self.rank}: the library communicator is not available on every rank ({why or 'a peer failed'}); "
                  "gathering through the host over torch.distributed", file=sys.stderr)
            return False
        try:
            if self.world > 1:
                import torch.distributed as dist
                dist.broadcast_object_list(uid, src=0)  # rank 0's id is the one that counts
            self.local.context.comm_init(uid[0])
            joined = True
        except Exception as e:
            joined, why = False, str(e)
        if not all_agree(joined):  # keep rendering: the host-staged gather gives the same frame
            if joined:
                self.local.context.comm_destroy()  # this rank did join: leave again, or its reads would stay collective
            print(f"[rvpt_amd] rank {self.rank}: not every rank joined the library communicator ({why or 'a peer failed'}); "
                  "gathering through the host over torch.distributed", file=sys.stderr)
            return False
        return True

    def __getattr__(self, name):  # add_material / add_triangle(s) / initialize / update / draw / wait / ...
        return getattr(self.local, name)

    def _slot_tensor(self):
        import torch
        if self._slot is None:
            ptr, _, slot_bytes = self.local.context.tile_buffer()
            self._slot = torch.as_tensor(_DeviceBuffer(ptr, slot_bytes // 4), device=f"cuda:{self.device}")
        return self._slot

    def barrier(self) -> None:
        """Every rank's dispatched work has finished.  Through the library communicator (rvpt_hip_comm_barrier: this rank's wait + a
        one-float all-reduce, ~30 us warm); without one, this rank's wait + the process group's barrier."""
        if self.library_comm:
            self.local.context.comm_barrier()
            return
        self.local.wait()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def gather_frame(self):
        """Collective.  Rank 0 returns the full frame as a cuda tensor [H, W, 4] (float32); others None."""
        import torch
        if self.library_comm:  # rvpt_hip_gather: waits for the frames in flight, gathers, un-tiles on rank 0
            out = None
            if self.rank == 0:
                out = torch.empty((self.height, self.width, 4), dtype=torch.float32, device=f"cuda:{self.device}")
            self.local.context.gather(out.data_ptr() if out is not None else None)
            return out
        self.local.wait()  # the library renders on its own stream
        slots = gather_slots(self._slot_tensor(), self.rank, self.world)
        if self.rank != 0:
            return None
        torch.cuda.synchronize(self.device)
        out = torch.empty((self.height, self.width, 4), dtype=torch.float32, device=f"cuda:{self.device}")
        self.local.context.untile(slots.data_ptr(), slots.shape[1] * 4, self.world, out.data_ptr())
        return out

    def read_frame(self):
        out = self.gather_frame()
        return None if out is None else out.cpu().numpy()
