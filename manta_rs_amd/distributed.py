"""One process per GPU: the exchange step of the range-sharded MSM over torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in CPU tests).

SURVEY.md section 8(e): an MSM shards by contiguous base/scalar range; every rank runs a full Pippenger pass over its
slice and is left with ONE partial point (64-192 B affine). RCCL has no user-defined reduction, so the exchange is an
all_gather of the N partial points followed by an N-term sum on every rank (`mg_points_sum`, host) -- 8 x 96 B on the
wire per 2^20-term BLS12-381 MSM, one hop on the fully connected xGMI mesh. Nothing else of the path needs a
collective: NTT / witness map and whole proofs are replicas (a 2^20 transform is sub-millisecond on one GPU).

The in-process counterpart -- one host process driving several GPUs, what a Rust host linking libmantagpu.so would
use -- is `mg_bases_create_sharded` / `mg_ctx_create_sharded` (api.Bases(devices=...), api.ProvingContext(devices=...)),
where the partial points meet in pinned host memory and no collective exists at all.
"""
from __future__ import annotations

import numpy as np

from . import api


def shard_range(n: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of an n-term MSM owned by `rank` -- the same split the library uses for its
    in-process shards (`mg_bases_create_sharded`)."""
    return n * rank // world, n * (rank + 1) // world


class PartialPointExchange:
    """all_gather + sum of one partial point per rank. Buffers are allocated once (a pinned staging pair and, for the
    nccl backend, two small device tensors): a step costs one H2D of <= 192 B, the collective, one D2H."""

    def __init__(self, curve: int, group: int, process_group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.curve, self.group = curve, group
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.limbs = api.affine_limbs(curve, group)
        self.on_gpu = self.world > 1 and dist.get_backend(process_group) == "nccl"
        if self.world > 1:
            dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if self.on_gpu else "cpu"
            self.send = torch.zeros(self.limbs, dtype=torch.int64, device=dev)
            self.recv = torch.zeros(self.world * self.limbs, dtype=torch.int64, device=dev)
            if self.on_gpu:
                self.h_send = torch.zeros(self.limbs, dtype=torch.int64).pin_memory()
                self.h_recv = torch.zeros(self.world * self.limbs, dtype=torch.int64).pin_memory()

    def all_gather(self, local_point: np.ndarray) -> np.ndarray:
        """-> [world, limbs] uint64: every rank's partial point."""
        if self.world == 1:
            return np.ascontiguousarray(local_point, dtype=np.uint64).reshape(1, -1)
        t = self.torch.from_numpy(np.ascontiguousarray(local_point, dtype=np.uint64).view(np.int64))
        if self.on_gpu:
            self.h_send.copy_(t)
            self.send.copy_(self.h_send, non_blocking=True)
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.pg)  # RCCL over xGMI
            self.h_recv.copy_(self.recv)  # D2H; synchronises with the collective
            out = self.h_recv.numpy()
        else:
            self.send.copy_(t)
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.pg)
            out = self.recv.numpy()
        return out.view(np.uint64).reshape(self.world, self.limbs).copy()

    def sum(self, local_point: np.ndarray) -> np.ndarray:
        """The global point: sum over ranks of their partial points (identical on every rank)."""
        if self.world == 1:
            return np.ascontiguousarray(local_point, dtype=np.uint64)
        return api.points_sum(self.curve, self.group, self.all_gather(local_point))


class ShardedMsmJob:
    def __init__(self, job, exchange):
        self.job, self.exchange = job, exchange

    def finish(self) -> np.ndarray:
        return self.exchange.sum(self.job.finish())


class ShardedMSM:
    """`VariableBaseMSM::multi_scalar_mul` over bases range-sharded across the ranks of a process group: this rank holds
    `local_bases` (its slice, registered with api.Bases on its GPU) and the matching slice of the scalars."""

    def __init__(self, local_bases: api.Bases, process_group=None):
        self.bases = local_bases
        self.exchange = PartialPointExchange(local_bases.curve, local_bases.group, process_group)

    def launch(self, d_local_scalars, n_local, **kw) -> ShardedMsmJob:
        return ShardedMsmJob(api.VariableBaseMSM.launch(self.bases, d_local_scalars, n_local, **kw), self.exchange)

    def multi_scalar_mul(self, local_scalars: np.ndarray) -> np.ndarray:
        return self.exchange.sum(api.VariableBaseMSM.multi_scalar_mul(self.bases, local_scalars))
