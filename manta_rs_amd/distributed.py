"""One process per GPU: the exchange step of the range-sharded MSM and of the range-sharded PROOF over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests and for two ranks on one device).

SURVEY.md section 8(e) / 7.1 C1: an MSM shards by contiguous base/scalar range; every rank runs a full Pippenger pass over
its slice and is left with ONE partial point. RCCL has no user-defined reduction, so the exchange is an all_gather of the N
partial points followed by an N-term sum on every rank. A proof has five MSMs: its exchange is ONE fused all_gather of the
five partial points per rank (<= 1.9 KB), then the usual host assembly (BASELINE configs[3]; caller
manta-accounting/src/transfer/mod.rs:695-715 -> manta-crypto/src/arkworks/groth16.rs:589-600). Nothing else of the path
needs a collective: NTT / witness map are recomputed per rank (1.1 MB of z against 2 MB of h), whole proofs are replicas.

Two transports, same library calls underneath:
  * device path (backend nccl): the library folds the window sums on the GPU and writes the partial point(s) as XYZZ
    coordinates straight into the collective's send buffer (`mg_msm_result_to_device`, `mg_groth16_partials_launch`), the
    stream RCCL runs on is made to wait for them with an event, all_gather_into_tensor runs from device memory, and one
    asynchronous copy brings the gathered points to pinned host memory. No host synchronisation between launch and collective;
    the host waits once, when it needs the result.
  * host path (backend gloo, or bases without precomputed multiples whose fold is a host job): the partial point comes back
    through mg_msm_finish and is gathered from host memory.

The in-process counterpart -- one host process driving several GPUs, what a Rust host linking libmantagpu.so would use -- is
`mg_bases_create_sharded` / `mg_ctx_create_sharded` (api.Bases(devices=...), api.ProvingContext(devices=...)), where the
partial points meet in pinned host memory and no collective exists at all.
"""
from __future__ import annotations

import numpy as np

from . import api


def shard_range(n: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of an n-term MSM owned by `rank` -- the same split the library uses for its
    in-process shards (`mg_bases_create_sharded`, `mg_ctx_create_shard`)."""
    return n * rank // world, n * (rank + 1) // world


class _Slot:
    """one set of exchange buffers: device send / recv tensors, pinned landing area, completion event"""

    def __init__(self, torch, words, world, dev):
        self.send = torch.zeros(words, dtype=torch.int64, device=dev)
        self.recv = torch.zeros(world * words, dtype=torch.int64, device=dev)
        self.h_recv = torch.zeros(world * words, dtype=torch.int64).pin_memory()
        self.done = torch.cuda.Event()
        self.busy = False  # owned by a job from begin() until its wait() has copied h_recv out


class PartialPointExchange:
    """all_gather + sum of `words` u64 per rank (one partial point, or the five of a proof).

    force_collective: run the collective even in a world of one -- that is how the RCCL branch is exercised on a 1-GPU box
    (tests/test_gpu_multi.py); normally a single rank short-circuits.
    ring: sets of buffers used round-robin, so that several MSMs / passes can be in flight (the library writes the next
    partial point while the previous all_gather is still reading its send buffer)."""

    def __init__(self, curve: int, group: int, process_group=None, device=None, force_collective=False, words=None, ring=8):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.curve, self.group = curve, group
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.limbs = api.affine_limbs(curve, group)
        self.collective = dist.is_initialized() and (self.world > 1 or force_collective)
        self.on_gpu = self.collective and dist.get_backend(process_group) == "nccl"
        self.words = int(words) if words else api.xyzz_limbs(curve, group)  # device path: XYZZ points
        self.slots, self.next = [], 0
        self.comm_stream = None
        if self.collective:
            if self.on_gpu:
                dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
                self.slots = [_Slot(torch, self.words, self.world, dev) for _ in range(ring)]
                # The exchange has a NON-blocking stream of its own, never torch's default (= the NULL) stream: the stand-alone MSMs
                # run on blocking streams (hardware queues of their own, csrc/engine.h MsmWorkspace::solo), and legacy-stream
                # ordering would make MSM k+1 wait for the all_gather of MSM k behind the NULL stream -- the pipeline of MSMs in
                # flight would run one at a time.
                self.comm_stream = torch.cuda.Stream(device=dev)
                torch.cuda.current_stream(dev).synchronize()  # (the buffers' zero fills ran on torch's current stream)
            # host path buffers (affine points)
            self.send = torch.zeros(self.limbs, dtype=torch.int64)
            self.recv = torch.zeros(self.world * self.limbs, dtype=torch.int64)

    # ------------------------------------------------------------------ host path
    def all_gather(self, local_point: np.ndarray) -> np.ndarray:
        """-> [world, limbs] uint64: every rank's partial point (affine), gathered from HOST memory."""
        if not self.collective:
            return np.ascontiguousarray(local_point, dtype=np.uint64).reshape(1, -1)
        t = self.torch.from_numpy(np.ascontiguousarray(local_point, dtype=np.uint64).view(np.int64))
        if self.on_gpu:  # nccl cannot gather host tensors: bounce through a device slot (plain bases only take this road)
            sl = self._acquire()
            try:
                with self.torch.cuda.stream(self.comm_stream):
                    sl.send[:self.limbs].copy_(t, non_blocking=True)
                    self.dist.all_gather_into_tensor(sl.recv, sl.send, group=self.pg)
                    # .cpu() waits for the gather on the exchange stream: the slot is ours until the copy below exists
                    out = sl.recv.cpu().numpy().view(np.uint64).reshape(self.world, self.words)[:, :self.limbs].copy()
            finally:
                sl.busy = False
            return out
        self.send.copy_(t)
        self.dist.all_gather_into_tensor(self.recv, self.send, group=self.pg)
        return self.recv.numpy().view(np.uint64).reshape(self.world, self.limbs).copy()

    def sum(self, local_point: np.ndarray) -> np.ndarray:
        """The global point: sum over ranks of their partial points (identical on every rank)."""
        if not self.collective:
            return np.ascontiguousarray(local_point, dtype=np.uint64)
        return api.points_sum(self.curve, self.group, self.all_gather(local_point))

    # ------------------------------------------------------------------ device path
    def _acquire(self) -> _Slot:
        """The next buffer set of the ring. A set belongs to the job that began on it until that job's wait() has copied
        the gathered points out of h_recv: the completion event only says the gather and the copy to pinned memory are done,
        not that their owner has read them, so a set that is still owned is never handed out again -- more than `ring` jobs
        in flight is a caller error and raises instead of overwriting an older job's result."""
        for _ in range(len(self.slots)):
            sl = self.slots[self.next]
            self.next = (self.next + 1) % len(self.slots)
            if not sl.busy:
                sl.busy = True
                return sl
        raise RuntimeError("PartialPointExchange: all %d buffer sets are owned by unfinished jobs -- finish() one before "
                           "launching another (or build the exchange with a larger ring)" % len(self.slots))

    def stream_handle(self) -> int:
        """the raw hipStream_t the collective will be enqueued on (the exchange's own non-blocking stream)"""
        return int((self.comm_stream or self.torch.cuda.current_stream()).cuda_stream)

    def begin(self) -> _Slot:
        """a free buffer set; the producer writes `words` u64 to slot.send (device memory) and makes stream_handle() wait"""
        return self._acquire()

    def gather_async(self, sl: _Slot, words=None):
        """all_gather_into_tensor from device memory on the exchange stream + asynchronous copy to pinned memory. `words` (<= the
        buffer's) = u64 per rank the producer wrote: only those travel -- a pass of k proofs moves k * 5 points per rank over xGMI,
        not the max_batch-sized buffer -- and land packed as [world, words] at the front of recv / h_recv."""
        w = self.words if words is None else int(words)
        if w <= 0 or w > self.words:
            raise ValueError("PartialPointExchange.gather_async: words out of range")
        sl.sent = w
        with self.torch.cuda.stream(self.comm_stream):
            self.dist.all_gather_into_tensor(sl.recv[:self.world * w], sl.send[:w], group=self.pg)  # RCCL over xGMI
            sl.h_recv[:self.world * w].copy_(sl.recv[:self.world * w], non_blocking=True)
            sl.done.record()

    def wait(self, sl: _Slot, words=None) -> np.ndarray:
        """-> [world, words] uint64, once the gather and the copy have completed (the only host wait of the step). `words`
        (<= the buffer's) = how many u64 per rank the producer actually wrote: the tail of a send buffer holds whatever an
        earlier, larger pass left there and must never reach the consumer."""
        sl.done.synchronize()
        sent = getattr(sl, "sent", self.words)  # per-rank length of the gather that filled this slot
        w = sent if words is None else int(words)
        if w < 0 or w > sent:
            sl.busy = False
            raise ValueError("PartialPointExchange.wait: words out of range")
        out = sl.h_recv.numpy().view(np.uint64)[:self.world * sent].reshape(self.world, sent)[:, :w].copy()
        sl.busy = False
        return out


class ShardedMsmJob:
    def __init__(self, job, exchange, slot=None):
        self.job, self.exchange, self.slot = job, exchange, slot

    def finish(self) -> np.ndarray:
        ex = self.exchange
        if self.slot is None:
            return ex.sum(self.job.finish())
        parts = ex.wait(self.slot)  # [world, xyzz limbs]
        self.job.release()
        return api.xyzz_sum(ex.curve, ex.group, parts)


class ShardedMSM:
    """`VariableBaseMSM::multi_scalar_mul` over bases range-sharded across the ranks of a process group: this rank holds
    `local_bases` (its slice, registered with api.Bases on its GPU) and the matching slice of the scalars."""

    def __init__(self, local_bases: api.Bases, process_group=None, force_collective=False):
        self.bases = local_bases
        self.exchange = PartialPointExchange(local_bases.curve, local_bases.group, process_group, force_collective=force_collective)
        # the device-side fold exists for bases with precomputed multiples (one bucket window, no Horner doublings)
        self.device_path = self.exchange.on_gpu and getattr(local_bases, "precompute_window_bits", 0) > 0

    def launch(self, d_local_scalars, n_local, **kw) -> ShardedMsmJob:
        job = api.VariableBaseMSM.launch(self.bases, d_local_scalars, n_local, **kw)
        if not self.device_path:
            return ShardedMsmJob(job, self.exchange)
        ex = self.exchange
        sl = ex.begin()
        job.result_to_device(sl.send.data_ptr(), ex.stream_handle())  # fold on the GPU -> send buffer; the stream waits for it
        ex.gather_async(sl)
        return ShardedMsmJob(job, ex, sl)

    def multi_scalar_mul(self, local_scalars: np.ndarray) -> np.ndarray:
        d = api.DeviceBuffer.from_numpy(np.ascontiguousarray(local_scalars, dtype=np.uint64))
        return self.launch(d, local_scalars.shape[0]).finish()


def task_masks(world: int):
    """Task-parallel placement (last row): which of the five MSMs (bit 0 a, 1 b_g1, 2 b_g2, 3 l, 4 h) each rank
    computes in full. Longest-processing-time greedy over rough weights -- the G2 MSM costs about three G1 MSMs of the same length,
    h is dense and twice as long as the witness MSMs; ranks beyond the fifth get nothing."""
    weights = {2: 3.0, 4: 2.0, 0: 1.0, 1: 1.0, 3: 1.0}
    bins = [[0.0, 0] for _ in range(min(world, 5))]
    for i, w in sorted(weights.items(), key=lambda kv: -kv[1]):
        b = min(bins, key=lambda x: x[0])
        b[0] += w
        b[1] |= 1 << i
    return [b[1] for b in bins] + [0] * (world - len(bins))


class ShardedProofJob:
    def __init__(self, prover, k, rs, ss, slot=None, job=None, parts=None):
        self.prover, self.k, self.rs, self.ss, self.slot, self.job, self.parts = prover, k, rs, ss, slot, job, parts

    def finish(self) -> list:
        p = self.prover
        parts = self.parts
        if parts is None:
            # [world, k * 5 * slot limbs]: only what this pass wrote -- the send buffers are sized for max_batch proofs and
            # the tail beyond k proofs still holds the partial points of whichever larger pass used the set before
            parts = p.exchange.wait(self.slot, self.k * 5 * p.slot)
            p.ctx.partials_finish(self.job)
        return p.ctx.assemble(parts, self.rs, self.ss)


class ShardedProver:
    """`Groth16::prove` with every MSM of the proof range-sharded over the ranks of a process group, one process per GPU
    (BASELINE configs[3]). Every rank gets the whole assignment, recomputes the witness map, runs its five partial MSMs
    (`mg_ctx_create_shard` holds slice rank/world of every query) and contributes ONE fused all_gather of its five partial
    points per proof; every rank then assembles the same proof bytes -- identical to the single-GPU context's.
    max_batch: proofs per pass (<= 32); the exchange buffers are sized for it."""

    def __init__(self, curve, pk, process_group=None, force_collective=False, max_batch=1, ctx=None, placement="range"):
        """placement: "range" (every MSM split into `world` contiguous ranges, the default and what north_star names) or "task"
        (every MSM computed in full by ONE rank, `task_masks`): the exchange and the assembly are the same, a rank that does not
        own an MSM contributes the point at infinity."""
        import torch.distributed as dist
        self.curve = curve
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        if placement not in ("range", "task"):
            raise ValueError("placement is 'range' or 'task'")
        self.placement = placement
        if ctx is not None:
            self.ctx = ctx
        elif placement == "task":
            self.ctx = api.ProvingContext(curve, pk, task_mask=task_masks(world)[rank])
        else:
            self.ctx = api.ProvingContext(curve, pk, shard=(rank, world))
        self.slot = self.ctx.partials_slot_limbs
        self.max_batch = int(max_batch)
        self.exchange = PartialPointExchange(curve, 2, process_group, force_collective=force_collective,
                                             words=self.max_batch * 5 * self.slot)
        self._d_out = None  # host path: a device buffer of our own for the partial results

    def set_r1cs(self, r1cs):
        self.ctx.set_r1cs(r1cs)

    def launch(self, zs, rs, ss) -> ShardedProofJob:
        rs = np.ascontiguousarray(rs, dtype=np.uint64).reshape(-1, 4)
        ss = np.ascontiguousarray(ss, dtype=np.uint64).reshape(-1, 4)
        k = rs.shape[0]
        if k < 1 or k > self.max_batch or ss.shape[0] != k:
            raise ValueError("ShardedProver: between 1 and max_batch proofs per pass, one (r, s) pair each")
        ex = self.exchange
        if ex.on_gpu:  # device path: partial points -> send buffer -> RCCL, no host sync in between
            sl = ex.begin()
            job = self.ctx.partials_launch(zs, k, sl.send.data_ptr(), ex.stream_handle())
            ex.gather_async(sl, k * 5 * self.slot)  # what this pass wrote, not the max_batch-sized buffer
            return ShardedProofJob(self, k, rs, ss, slot=sl, job=job)
        # host path (gloo, or a single rank): the partial points come back through a device buffer of our own
        words = k * 5 * self.slot
        if self._d_out is None:
            self._d_out = api.DeviceBuffer(self.max_batch * 5 * self.slot * 8)
        job = self.ctx.partials_launch(zs, k, self._d_out.ptr, None)
        self.ctx.partials_finish(job)  # waits for the pass
        mine = self._d_out.to_numpy(shape=(self.max_batch * 5 * self.slot,))[:words]
        return ShardedProofJob(self, k, rs, ss, parts=self.gather_host(mine, k))

    def gather_host(self, mine: np.ndarray, k: int) -> np.ndarray:
        """[world, k, 5, slot] from every rank's [k, 5, slot] (host memory; gloo)"""
        ex = self.exchange
        words = k * 5 * self.slot
        mine = np.ascontiguousarray(mine, dtype=np.uint64).reshape(-1)[:words]
        if not ex.collective:
            return mine.reshape(1, k, 5, self.slot)
        t = ex.torch.from_numpy(mine.view(np.int64).copy())
        recv = ex.torch.zeros(ex.world * words, dtype=ex.torch.int64)
        ex.dist.all_gather_into_tensor(recv, t, group=ex.pg)
        return recv.numpy().view(np.uint64).reshape(ex.world, k, 5, self.slot).copy()

    def prove(self, z, r, s) -> bytes:
        return self.launch(z, r, s).finish()[0]

    def prove_batch(self, zs, rs, ss) -> list:
        return self.launch(zs, rs, ss).finish()

    def close(self):
        self.ctx.close()
