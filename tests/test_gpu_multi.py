"""Multi-process layout on ONE GPU (the GPU box has a single device): two ranks, both pinned to device 0, each running
the GPU MSM over its contiguous range; the partial points meet through the product's exchange (gloo here -- RCCL refuses
two ranks on one device; on the 8-GPU node the same code runs with backend nccl = RCCL over xGMI). bench.py checks the
global point against the closed form before timing, so a zero exit code is a parity statement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, port="29541"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MANTA_BENCH_DEVICE="0", MANTA_BENCH_BACKEND="gloo", **extra_env)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", *args],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    # what the driver does with an N-GPU run: the LAST line of stdout is one JSON object of a few KB (round 4's was 22 KB: parsed null)
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8192 and len(out.stdout) < 8192, len(out.stdout)
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_weak_and_strong_scaling(gpu):
    """VERDICT r4 item 7, the pre-flight of the driver's N > 1 runs: `bench.py --gpus 2` end to end as two ranks (weak-scaling MSM
    with the partial-point exchange, strong scaling, the sharded proof in both placements, the proofs half as replicas); the
    rank-0 line parses, is small, and carries every leg; the detail object lands in the file the line names."""
    line = _run({"MANTA_BENCH_LOGN": "16"}, "--steps", "2", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["steps"] == 2
    assert line["metric"] == "G1 MSM Mscalar/s at 2^16"
    assert line["roofline"]["frac"] > 0 and line["config"]["sharding"] != "none"
    ss = line["strong_scaling"]
    assert ss["n_2^20"]["Mscalar_s"] > 0 and ss["n_35174"]["ms_per_msm"] > 0
    assert line["proofs"]["batched"] > 0 and len(line["proofs"]["per_gpu"]) == 2
    sp = line["sharded_proof"]  # BASELINE configs[3]: the proof with its MSMs split over the ranks (gloo exchange here)
    assert sp["sequential_ms"] > 0 and sp["batched"] > 0 and sp["task_parallel_ms"] > 0
    detail = json.load(open(os.path.join(ROOT, line["detail"])))
    assert detail["strong_scaling"]["n_2^20"]["n_per_gpu"] == 1 << 19 and detail["strong_scaling"]["n_35174"]["n_per_gpu"] == 17587
    assert detail["proofs"]["n_gpus"] == 2 and detail["sharded_proof"]["batched"]["proofs_per_pass"] == 32


def test_bench_gpus_2_starts_itself(gpu):
    """exactly `python3 bench.py --gpus 2 --steps 2 --warmup 1` -- NO launcher -- is what a driver that builds
    its N > 1 command like its N = 1 command runs. bench.py becomes the launcher, rank 0 prints the one line as the last line of
    stdout, and the line says what the collective saw. Both ranks sit on device 0 here (one GPU on the box), so the exchange is gloo
    and `devices` holds ONE id; on the 8-GPU node the same path reports nccl and N ids. The parity gate (sharded PrivateTransfer
    proof == single-device proof on every rank, verified) runs before anything is timed."""
    env = dict(os.environ, MANTA_BENCH_DEVICE="0", MANTA_BENCH_LOGN="16")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MANTA_BENCH_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.rstrip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    co = line["collective"]
    assert co == {"backend": "gloo", "ranks": 2, "devices": co["devices"], "launcher": "self", "parity_gate": co["parity_gate"]}
    assert len(co["devices"]) == 1 and co["devices"][0].startswith("cuda:0") and "verified" in co["parity_gate"]
    assert line["sharded_proof"]["batched"] > 0 and line["strong_scaling"]["n_2^20"]["Mscalar_s"] > 0


def test_world2_full_size_msm_only(gpu):
    """the BASELINE-size weak-scaling step (2 x 2^20 terms, closed-form checked) through two ranks"""
    line = _run({}, "--steps", "3", "--warmup", "1", "--quick", port="29542")
    assert line["metric"] == "G1 MSM Mscalar/s at 2^20" and line["n_gpus"] == 2
    assert line["roofline"]["kernel_ms"] > 0


_RCCL_ONE_RANK = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth, distributed
torch.cuda.set_device(0)
api.init(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
# ---- MSM: partial point folded on the GPU -> RCCL all_gather from device memory -> host sum
for curve, group, n, pre in ((1, 1, 5000, 9), (0, 1, 3000, 13), (0, 2, 700, 6), (1, 1, 1 << 15, 16)):
    pts = H.random_points(curve, group, min(n, 2000), seed=3)
    reps = -(-n // pts.shape[0])
    pts = np.concatenate([pts] * reps)[:n]           # repeated bases are fine (and exercise the P + P case)
    sc = synth.msm_scalars(curve, n, "U", seed=4)
    b = api.Bases(curve, group, pts, precompute_window_bits=pre)
    want = api.VariableBaseMSM.multi_scalar_mul(b, sc)
    m = distributed.ShardedMSM(b, force_collective=True)
    assert m.exchange.on_gpu and m.device_path
    d = api.DeviceBuffer.from_numpy(sc)
    jobs = [m.launch(d, n) for _ in range(5)]        # several in flight: the buffer ring
    for j in jobs:
        assert (j.finish() == want).all(), (curve, group, n, pre)
    # plain bases: the fold is a host job, the point is gathered through the device bounce
    pb = api.Bases(curve, group, pts)
    mp = distributed.ShardedMSM(pb, force_collective=True)
    assert mp.exchange.on_gpu and not mp.device_path
    assert (mp.launch(d, n).finish() == want).all()
# ---- proof: five partial points per proof in ONE fused all_gather, assembled on the host
curve = api.BN254
c = synth.make_circuit(curve, 700, 500, 9, seed=11)
pk = O.groth16_setup(c, H.toxic(curve))
ctx = api.ProvingContext(curve, pk)
r1cs = api.R1CS.from_circuit(c)
ctx.set_r1cs(r1cs)
rs = H.rand_fr_mont(curve, 8, seed=5)
sp = distributed.ShardedProver(curve, pk, force_collective=True, max_batch=3)
sp.set_r1cs(r1cs)
assert sp.exchange.on_gpu
for i in range(4):                                  # eager runs, then the captured graphs
    want = api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * i], rs[2 * i + 1])
    assert sp.prove(c.z, rs[2 * i], rs[2 * i + 1]) == want, i
assert O.groth16_verify(curve, pk, c.z[1:c.P], want) == 1
zs = np.stack([c.z] * 3)
got = sp.prove_batch(zs, rs[0:3], rs[3:6])
assert got == api.Groth16.prove_batch(ctx, zs, rs[0:3], rs[3:6])
# (advisor r3, high) a pass SMALLER than max_batch after larger ones have used every buffer set of the ring: the tail of the
# send buffers still holds their partial points and must not reach the assembly
for i in range(len(sp.exchange.slots)):
    assert sp.prove_batch(zs, rs[0:3], rs[3:6]) == got
for i in range(len(sp.exchange.slots) + 2):
    assert sp.prove(c.z, rs[0], rs[1]) == api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]), i
assert sp.prove_batch(zs[:2], rs[0:2], rs[3:5]) == got[:2]
# (advisor r3, low) a buffer set belongs to its job until finish(): more jobs than sets in flight raises, nothing is overwritten
jobs = [sp.launch(c.z, rs[0], rs[1]) for _ in range(len(sp.exchange.slots))]
try:
    sp.launch(c.z, rs[0], rs[1])
    raise SystemExit("a ninth job took a buffer set that was still owned")
except RuntimeError:
    pass
want1 = api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
assert all(j.finish()[0] == want1 for j in jobs)
# (advisor r3, medium) a lone range shard of more than one cannot prove on its own
import pytest
half = api.ProvingContext(curve, pk, shard=(0, 2))
half.set_r1cs(r1cs)
with pytest.raises(api.MantaGpuError) as ei:
    api.Groth16.prove_with_randomness(half, c.z, rs[0], rs[1])
assert ei.value.status == 5
with pytest.raises(api.MantaGpuError):
    api.Groth16.prove_batch(half, zs, rs[0:3], rs[3:6])
half.close()
# ---- RCCL INSIDE the library (mg_ctx_opts.exchange = MG_EXCHANGE_RCCL, what the Rust host reaches through mg_ctx_create_ex): a
# one-device list -> ncclCommInitAll of one rank, partial points folded on the GPU, grouped ncclAllGather, assembly; the
# process already holds torch's librccl, which the library must pick up instead of loading a second one
rc = api.ProvingContext(curve, pk, devices=[0], exchange=api.EXCHANGE_RCCL)
rc.set_r1cs(r1cs)
for i in range(4):
    assert api.Groth16.prove_with_randomness(rc, c.z, rs[2 * i], rs[2 * i + 1]) == api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * i], rs[2 * i + 1]), i
assert api.Groth16.prove_batch(rc, zs, rs[0:3], rs[3:6]) == got
z40 = np.stack([c.z] * 40)  # longer than a pass: streamed as passes in flight, each with its own exchange buffers
rs40 = H.rand_fr_mont(curve, 80, seed=7)
assert api.Groth16.prove_batch(rc, z40, rs40[:40], rs40[40:]) == api.Groth16.prove_batch(ctx, z40, rs40[:40], rs40[40:])
rc.close()
with pytest.raises(api.MantaGpuError):  # RCCL refuses one device twice in a clique: the host exchange serves such lists
    api.ProvingContext(curve, pk, devices=[0, 0], exchange=api.EXCHANGE_RCCL)
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ok")
'''


def test_rccl_branch_with_a_one_rank_group(gpu):
    """The RCCL branch of the exchange, executed on the 1-GPU box: a process group of ONE rank with backend nccl and the
    world-of-one short-circuit disabled. MSM: fold on the device -> all_gather_into_tensor from device memory -> pinned copy
    -> host sum, five jobs in flight; proof: distributed.ShardedProver, one fused gather of the five partial points, bytes
    equal to the single-GPU context's (single proofs through eager and graph replay, and a batch of three)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK.format(root=ROOT)], capture_output=True, text=True, env=env,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "rccl one-rank ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


_TWO_RANK_PROOF = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth, distributed
api.init(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
curve = api.BN254
shape = os.environ.get("SHAPE")
c = synth.make_shape(curve, shape) if shape else synth.make_circuit(curve, 900, 650, 11, seed=21)
if shape:
    from manta_rs_amd import keygen
    rng = synth.XorShift(0x4D414E5441_0002)
    pk = keygen.generate(c, [rng.field(synth.FR_MODULUS[curve]) for _ in range(5)])
else:
    pk = O.groth16_setup(c, H.toxic(curve))
r1cs = api.R1CS.from_circuit(c)
rs = H.rand_fr_mont(curve, 8, seed=6)
sp = distributed.ShardedProver(curve, pk, max_batch=2)   # this rank: slice rank/world of every query
sp.set_r1cs(r1cs)
assert sp.ctx.num_shards == world and not sp.exchange.on_gpu
ctx = api.ProvingContext(curve, pk)                      # the single-GPU reference, same device
ctx.set_r1cs(r1cs)
for i in range(3):
    want = api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * i], rs[2 * i + 1])
    assert sp.prove(c.z, rs[2 * i], rs[2 * i + 1]) == want, i
zs = np.stack([c.z] * 2)
assert sp.prove_batch(zs, rs[0:2], rs[2:4]) == api.Groth16.prove_batch(ctx, zs, rs[0:2], rs[2:4])
# task placement (SURVEY.md 8(e), last row): rank 0 computes b_g2 and b_g1, rank 1 a, l and the witness map + h; same exchange, same bytes
assert distributed.task_masks(2) == [0b00110, 0b11001] and sorted(distributed.task_masks(8))[-5:] == [1, 2, 4, 8, 16]
tp = distributed.ShardedProver(curve, pk, max_batch=2, placement="task")
tp.set_r1cs(r1cs)
for i in range(3):
    assert tp.prove(c.z, rs[2 * i], rs[2 * i + 1]) == api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * i], rs[2 * i + 1]), i
assert tp.prove_batch(zs, rs[0:2], rs[2:4]) == api.Groth16.prove_batch(ctx, zs, rs[0:2], rs[2:4])
import pytest
with pytest.raises(api.MantaGpuError):      # a task context cannot prove on its own
    api.Groth16.prove_with_randomness(tp.ctx, c.z, rs[0], rs[1])
dist.barrier()
print("rank", rank, "sharded proof ok")
'''


@pytest.mark.parametrize("shape", ["", "private_transfer"])
def test_sharded_proof_two_ranks_on_one_gpu(gpu, shape):
    """BASELINE configs[3] in its process-per-GPU form, two ranks pinned to device 0 (gloo: RCCL refuses two ranks on one
    device): each holds half of every query, proves its five partial MSMs, the ten partial points are gathered in one
    collective, both ranks assemble -- byte-identical to the single-GPU proof (small circuit and the PrivateTransfer shape)."""
    script = _TWO_RANK_PROOF.format(root=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SHAPE=shape)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29548", "--no-python", sys.executable, "-c", script],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count("sharded proof ok") == 2, out.stdout[-3000:] + out.stderr[-3000:]
