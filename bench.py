#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Groth16 hot path.

Metric (BASELINE.json): "G1 MSM Mscalar/s at 2^20" on configs[1] = "2^20 BLS12-381 G1 variable-base
MSM, synthetic scalars/bases, 1 MI355X". A step = one full MSM (digits -> sort -> bucket accumulate ->
merge -> bucket reduce -> host fold to one affine point) over n = 2^20 scalars that are already
resident in HBM, against bases registered (resident, with their 2^(c w) multiples) before timing.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

N > 1 (weak scaling): the global MSM has N * 2^20 terms, sharded by contiguous base/scalar range, one
process per GPU; every step ends with an RCCL all_gather of the N partial points (96 B each, xGMI) and
the local N-term sum -- SURVEY.md section 8(e). value = N * 2^20 * K / t with t the max over ranks.

One JSON line on rank 0. `roofline` is for the dominant kernel (bucket accumulate), timed live with HIP
events on its own stream; `cpu_baseline` is the arkworks-algorithm CPU restatement (oracle/, 1 thread
like the reference) on a bounded prefix of the same inputs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LOG_N = int(os.environ.get("MANTA_BENCH_LOGN", "20"))  # 20 = the BASELINE config; smaller n only for scaling studies
CURVE = 1  # BLS12-381
WINDOW_BITS = int(os.environ.get("MANTA_BENCH_C", "16"))
DEPTH = int(os.environ.get("MANTA_BENCH_DEPTH", "3"))  # MSMs in flight (each on its own stream + workspace)
ALGO_BYTES_PER_SCALAR = 128  # 32 B scalar + 96 B affine G1 base (BLS12-381)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec

BLS_G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
          0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["msm", "prove"], default="msm",
                    help="msm = BASELINE configs[1] (default, the headline line); prove = whole Groth16 proofs of a "
                         "manta-pay circuit shape (configs[0]/[3]/[4] shapes, BN254)")
    ap.add_argument("--shape", default="private_transfer", choices=["to_private", "to_public", "private_transfer"])
    ap.add_argument("--threads", type=int, default=2, help="prove workload: host threads issuing proofs concurrently")
    ap.add_argument("--batch", type=int, default=1,
                    help="prove workload: proofs per mg_groth16_prove_batch call (a step is still ONE proof)")
    args = ap.parse_args()
    if args.workload == "prove":
        return prove_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"

    import torch
    import torch.distributed as dist
    from manta_rs_amd import api, synth

    dev = _device_index(local_rank)
    backend = os.environ.get("MANTA_BENCH_BACKEND", "nccl")  # "gloo": single-GPU functional check of the N>1 path
    torch.cuda.set_device(dev)
    api.init(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = 1 << LOG_N
    p = synth.FR_MODULUS[CURVE]
    q = synth.FQ_MODULUS[CURVE]
    G = synth.to_mont(list(BLS_G1), q, 6).reshape(-1)

    # ---- synthetic inputs (untimed). Bases: arithmetic progression P_i = [s0 + i*s1]G over the GLOBAL
    # index range, built on the GPU by the library's fixed-base batch multiply; rank r owns [r*n, (r+1)*n).
    s0, s1 = 0x243F6A8885A308D313198A2E03707344, 0x9E3779B97F4A7C15F39CC0605CEDC835
    base_idx = np.arange(rank * n, (rank + 1) * n, dtype=object)
    ks = [(s0 + int(i) * s1) % p for i in base_idx]
    d_ks = api.DeviceBuffer.from_numpy(synth.ints_to_limbs(ks, 4))
    d_pts = api.fixed_base_mul(CURVE, 1, G, d_ks, n)
    bases = api.Bases(CURVE, 1, (d_pts.ptr, n), precompute_window_bits=WINDOW_BITS, on_device=True)
    scalars = synth.msm_scalars(CURVE, n, "U", seed=0x4D414E54 + rank)  # uniform: the h-query MSM's case
    d_sc = api.DeviceBuffer.from_numpy(scalars)
    api.synchronize()

    acc_ms = []

    def gather_sum(local_pt):
        if world == 1:
            return local_pt
        t = torch.from_numpy(local_pt.view(np.int64))
        if backend == "nccl":
            t = t.cuda()
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)  # RCCL over xGMI: 96 B per rank
        pts = torch.stack(outs).cpu().numpy().view(np.uint64)
        return api.points_sum(CURVE, 1, pts)

    def barrier():
        api.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run(steps):
        """DEPTH MSMs in flight (each on its own HIP stream + workspace): the serial tail of step i
        (bucket reduce, host fold, partial-point exchange) overlaps the accumulate kernel of step i+1."""
        res = None
        pending = []

        def finish_one():
            pt = pending.pop(0).finish()
            acc_ms.append(api.last_accumulate_ms())  # HIP events around the accumulate kernel, on its own stream
            return gather_sum(pt)
        for _ in range(steps):
            pending.append(api.VariableBaseMSM.launch(bases, d_sc, n))
            if len(pending) == DEPTH:
                res = finish_one()
        while pending:
            res = finish_one()
        return res

    # ---- correctness gate before any timing counts: closed form sum_i k_i (s0 + i s1) mod r, one scalar mult
    result = run(1)
    sc_int = synth.limbs_to_ints(scalars)
    part = sum(k * b for k, b in zip(sc_int, ks)) % p
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, part)
        part = sum(parts) % p
    d_one = api.DeviceBuffer.from_numpy(synth.ints_to_limbs([part], 4))
    expect = api.fixed_base_mul(CURVE, 1, G, d_one, 1).to_numpy()
    assert (result == expect).all(), "MSM result does not match the closed-form expectation"

    run(args.warmup)
    api.set_kernel_timing(True)  # two hipEventRecord per MSM: the dominant kernel is timed live, in the timed region
    barrier()
    acc_ms.clear()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    api.set_kernel_timing(False)
    timed_acc_ms = list(acc_ms)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel from the HIP-event durations collected in the timed region (rank 0)
    roofline = cpu = None
    if rank == 0:
        k_ms = float(np.mean(timed_acc_ms))  # average launch duration over the K timed steps (MSMs overlap: DEPTH in flight)
        achieved = n * ALGO_BYTES_PER_SCALAR / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_accumulate.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "accumulate_chunks<FpR<Bls381Fq>>", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                    "traffic": traffic, "kernel_ms": round(k_ms, 4),
                    "algorithmic_bytes_per_launch": n * ALGO_BYTES_PER_SCALAR,
                    "note": "integer-multiply bound, not HBM bound (SURVEY.md F7); int_mad gives the bound that governs",
                    # v_mad_u64_u32 per launch: ~15.06 mixed adds per scalar (16 signed 16-bit windows, top one nearly
                    # empty) x (8 mul x 406 + 2 sqr x 315) mads; peak = 1024 SIMDs x 64 lanes / 4.2 cycles (measured issue
                    # rate, profiles/r01_ubench2_mad_u64_u32.txt) x 2.4 GHz
                    "int_mad": {"mads_per_launch": int(n * 15.06 * 3878), "achieved_Tmad_s": round(n * 15.06 * 3878 / (k_ms * 1e-3) / 1e12, 2),
                                "peak_Tmad_s": round(1024 * 64 / 4.2 * 2.4e9 / 1e12, 2),
                                "frac": round(n * 15.06 * 3878 / (k_ms * 1e-3) / (1024 * 64 / 4.2 * 2.4e9), 3)} if LOG_N == 20 and WINDOW_BITS == 16 else None}
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O  # the checker, here only as the timed CPU baseline
            ns = min(n, 1 << 20)  # one whole 2^20 MSM: ~15 s of CPU work on one thread
            host_pts = d_pts.to_numpy(shape=(n, 12))[:ns].copy()
            tcpu, out = O.time_msm(CURVE, 1, host_pts, scalars[:ns])
            if ns == n:  # and a free parity check: the CPU restatement's point equals the GPU's
                assert (np.asarray(out).reshape(-1) == np.asarray(result).reshape(-1)).all(), "CPU and GPU MSM results differ"
            cpu = {"value": round(ns / tcpu / 1e6, 5), "unit": "Mscalar/s", "cores": 1, "kind": "port",
                   "sample": f"{'the same' if ns == n else 'first'} {ns} bases/scalars of the workload, arkworks-0.3 Pippenger restatement, "
                             f"{tcpu:.1f} s on 1 thread (the reference ships arkworks without `parallel`)"}

    if rank == 0:
        total = world * n * args.steps
        line = {
            "metric": "G1 MSM Mscalar/s at 2^%d" % LOG_N, "value": round(total / dt / 1e6, 3), "unit": "Mscalar/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "2^%d BLS12-381 G1 variable-base MSM per GPU, uniform scalars resident in HBM" % LOG_N,
                       "curve": "BLS12-381", "log_n": LOG_N, "window_bits": WINDOW_BITS,
                       "precomputed_base_multiples": True, "bases_hbm_bytes": bases.device_bytes(),
                       "sharding": "contiguous base/scalar ranges, all_gather of partial points" if world > 1 else "none"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _device_index(local_rank):
    """LOCAL_RANK, unless MANTA_BENCH_DEVICE pins every rank to one device (functional test of the
    multi-process path on a 1-GPU box)."""
    return int(os.environ.get("MANTA_BENCH_DEVICE", local_rank))


def prove_main(args):
    """Whole proofs of a shape-exact synthetic manta-pay circuit (BN254, the curve manta-pay uses).
    A step = one `Groth16::prove` call: H2D of z, witness map (3 SpMV, 7 NTT), 5 MSMs, host assembly, 128 proof
    bytes out. `--threads` host threads share ONE ProvingContext (the reference's signer does the same,
    manta-pay/src/simulation/mod.rs:75-79); replicas across GPUs need no collective."""
    import threading
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus
    import torch
    import torch.distributed as dist
    from manta_rs_amd import api, synth, keygen
    dev = _device_index(local_rank)
    backend = os.environ.get("MANTA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev)
    api.init(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    curve = synth.BN254
    p = synth.FR_MODULUS[curve]
    t0 = time.perf_counter()
    c = synth.make_shape(curve, args.shape)
    rng = synth.XorShift(0x4D414E5441_0002)
    toxic = [rng.field(p) for _ in range(5)]
    pk = keygen.generate(c, toxic)
    ctx = api.ProvingContext(curve, pk)
    r1cs = api.R1CS.from_circuit(c)
    ctx.set_r1cs(r1cs)
    setup_s = time.perf_counter() - t0
    nrs = max(args.steps, args.warmup, 1)
    rs = synth.to_mont([rng.field(p) for _ in range(2 * nrs)], p, 4).reshape(nrs, 2, 4)
    first = api.Groth16.prove_with_randomness(ctx, c.z, rs[0][0], rs[0][1])

    K = max(1, args.batch)
    # the assignment lives in page-locked memory (mg_host_alloc), as a host integration would keep it: the
    # library then DMAs it from there instead of staging a copy first
    z1_pin = api.PinnedArray.like(c.z)
    z1 = z1_pin.array
    zK_pin = api.PinnedArray.like(np.stack([c.z] * K)) if K > 1 else None
    zK = zK_pin.array if K > 1 else None

    def run(steps, threads):
        idx = iter(range(0, steps, K))
        lock = threading.Lock()
        out = [None] * steps

        def worker():
            while True:
                with lock:
                    i = next(idx, None)
                if i is None:
                    return
                if K == 1:
                    out[i] = api.Groth16.prove_with_randomness(ctx, z1, rs[i % nrs][0], rs[i % nrs][1])
                else:  # one pass of the GPU pipeline for proofs i .. i+K-1 (the last batch wraps around)
                    sel = [(i + q) % nrs for q in range(K)]
                    got = api.Groth16.prove_batch(ctx, zK, rs[sel, 0], rs[sel, 1])
                    for q in range(min(K, steps - i)):
                        out[i + q] = got[q]
        ts = [threading.Thread(target=worker) for _ in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return out

    def barrier():
        api.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run(args.warmup if K == 1 else max(args.warmup, 4 * K * args.threads), args.threads)  # slots capture their graphs on the 3rd call
    # sequential latency (one proof at a time)
    nlat = min(5, args.steps)
    for i in range(4):  # whichever slot serves a lone caller has captured its graphs after three passes
        api.Groth16.prove_with_randomness(ctx, z1, rs[i % nrs][0], rs[i % nrs][1])
    t0 = time.perf_counter()
    for i in range(nlat):
        api.Groth16.prove_with_randomness(ctx, z1, rs[i][0], rs[i][1])
    lat_ms = (time.perf_counter() - t0) / nlat * 1e3
    barrier()
    t0 = time.perf_counter()
    proofs = run(args.steps, args.threads)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert proofs[0] == first
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O  # checker, here as the timed CPU baseline (and a free byte-parity check)
        ncpu = min(8, args.steps, len(proofs))  # ~10 s of CPU work on one thread for PrivateTransfer
        t1 = time.perf_counter()
        want = [O.groth16_prove(c, pk, rs[i % nrs][0], rs[i % nrs][1], msm_algo=1) for i in range(ncpu)]
        tcpu = time.perf_counter() - t1
        for i in range(ncpu):
            assert want[i] == proofs[i], "GPU proof bytes differ from the CPU restatement"
        ok = O.groth16_verify(curve, pk, c.z[1:c.P], first)
        assert ok == 1, "proof does not satisfy the pairing equation"
        cpu = {"value": round(ncpu / tcpu, 4), "unit": "proofs/s", "cores": 1, "kind": "port",
               "sample": f"{ncpu} proofs of the same circuit/key/witness (the timed run's first {ncpu} (r, s) pairs), "
                         f"arkworks-0.3 algorithm restatement, {tcpu:.2f} s on 1 thread; bytes equal the GPU proofs; "
                         "proof pairing-verified"}
    if rank == 0:
        D, V, P = synth.SHAPES[args.shape]
        algo_bytes = 7 * 64 * D + 32 * V + 32 * D + 64 * (3 * V - P + D) + 128 * V
        line = {"metric": f"Groth16 proofs/sec (manta-pay {args.shape} shape)", "value": round(world * args.steps / dt, 3),
                "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": f"Groth16 prove, shape-exact synthetic {args.shape} circuit (D={D}, V={V}, P={P}), BN254",
                           "host_threads": args.threads, "proofs_per_call": K, "assignment_memory": "page-locked (mg_host_alloc)", "sequential_latency_ms": round(lat_ms, 3),
                           "setup_s": round(setup_s, 2)},
                "roofline": {"bound": "hbm", "achieved": round(algo_bytes * args.steps / dt / 1e9, 3), "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": round(algo_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBPS, 6),
                             "traffic": None, "algorithmic_bytes_per_proof": algo_bytes,
                             "note": "whole-proof algorithmic bytes (SURVEY.md 8(d), SpMV term excluded); integer-multiply and latency bound"},
                "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
