"""Batch-of-frames mode: shard independent frame pairs over the GPUs of one node.

With flags == 0 every Farnebäck pair is an independent computation (reference
transflow/flow/sources/cv.py:478-490: the `flow=` argument is only an output
buffer), so rank r of R owns a contiguous range of pairs plus a one-frame halo
(SURVEY.md §8e).  The remap recurrence is serial per stream, so each rank runs it
over its own range ("independent streams"); no data-path collective is needed.
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU
tests) is used for the rendezvous, the one-off broadcast of shared inputs
(pixmap, masks) and the barrier/max-over-ranks timing -- plumbing only; the
kernels never see a torch type.
"""
from __future__ import annotations

import os


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced split of `total` units: ranks < total % world get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(total), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def frames_needed(pair_range: tuple[int, int]) -> tuple[int, int]:
    """Frame indices a rank must hold for its pairs: pair t uses frames t and t+1."""
    a, b = pair_range
    return (a, b + 1) if b > a else (a, a)


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class Group:
    """Thin wrapper over torch.distributed for the three things the batch mode needs."""

    def __init__(self, backend: str = "nccl", device_index: int | None = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.local_rank, self.world = env_world()
        self.backend = backend
        if backend == "nccl":
            idx = self.local_rank if device_index is None else device_index
            torch.cuda.set_device(idx)
            self.device = torch.device("cuda", idx)
        else:
            self.device = torch.device("cpu")
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = self.device
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)

    def barrier(self):
        if self.backend == "nccl":
            self.dist.barrier(device_ids=[self.device.index])
            self.torch.cuda.synchronize()
        else:
            self.dist.barrier()

    def broadcast_bytes(self, array, src: int = 0):
        """Broadcast a numpy uint8/float array from `src` (RCCL broadcast on GPU tensors)."""
        import numpy as np
        t = self.torch.from_numpy(np.ascontiguousarray(array)).to(self.device)
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def max_over_ranks(self, value: float) -> float:
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_arrays(self, array, dst: int = 0):
        """Gather equal-shaped arrays to `dst` (list on dst, None elsewhere)."""
        import numpy as np
        t = self.torch.from_numpy(np.ascontiguousarray(array)).to(self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(t, out, dst=dst)
        return [o.cpu().numpy() for o in out] if out is not None else None

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
