"""CPU tests (no GPU) of the host side: the C-ABI library loads and exports every symbol the header
declares, host-only entry points work, synthetic circuits have the real manta-pay shapes, and the
multi-GPU partial-point reduction is exercised with world_size-2 gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from manta_rs_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from manta_rs_amd import api
    hdr = open(os.path.join(ROOT, "include", "mantagpu.h")).read()
    declared = set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(api.LIB, name), f"{name} declared in mantagpu.h but not exported"
    assert declared == set(api.EXPORTS)


def test_no_gpu_is_an_error_not_a_fallback():
    from manta_rs_amd import api
    try:
        n = api.device_count()
    except api.MantaGpuError:
        n = 0
    if n == 0:
        with pytest.raises(api.MantaGpuError):
            api.Bases(0, 1, np.ones((4, 8), dtype=np.uint64))
        with pytest.raises(api.MantaGpuError):
            api.Radix2EvaluationDomain(0, 8).fft(np.zeros((8, 4), dtype=np.uint64))


def test_host_point_sum_and_serialize_match_oracle():
    """mg_points_sum / mg_point_serialize are host-only code paths of the product (the multi-GPU partial
    point reduction and the proof encoder)."""
    from manta_rs_amd import api
    import helpers as H
    for curve, group in ((0, 1), (1, 1), (0, 2), (1, 2)):
        pts = H.random_points(curve, group, 9, seed=curve * 10 + group)
        pts[4] = 0
        assert (api.points_sum(curve, group, pts) == O.g_sum(curve, group, pts)).all()
        both = np.stack([pts[0], O.g_mul(curve, group, pts[0], synth.ints_to_limbs([synth.FR_MODULUS[curve] - 1], 4)[0])])
        assert not api.points_sum(curve, group, both).any()  # P + (-P) = infinity
        assert (api.points_sum(curve, group, np.stack([pts[1], pts[1]])) ==
                O.g_mul(curve, group, pts[1], synth.ints_to_limbs([2], 4)[0])).all()
        for p in (pts[0], pts[4]):
            for comp in (True, False):
                assert api.point_serialize(curve, group, p, comp) == O.serialize(curve, group, p, comp)


@pytest.mark.parametrize("name", ["to_private", "to_public", "private_transfer"])
def test_synthetic_shapes_match_manta_pay(name):
    D, V, P = synth.SHAPES[name]
    # shape arithmetic only: pk bytes = 624 + 320 V + 64 D for MPC keys
    sizes = {"to_private": 3690160, "to_public": 11040176, "private_transfer": 15450928}
    assert 624 + 320 * V + 64 * D == sizes[name]


def test_small_synthetic_circuit_is_satisfied_and_deterministic():
    a = synth.make_circuit(0, 500, 300, 13, seed=1)
    b = synth.make_circuit(0, 500, 300, 13, seed=1)
    assert synth.check_satisfied(a)
    assert (a.z == b.z).all() and (a.A.val == b.A.val).all()
    assert a.D == 1024
    zeros_ones = sum(1 for v in a.z_int if v in (0, 1))
    assert zeros_ones > 0.3 * a.V  # boolean-heavy like real witnesses


def test_msm_range_sharding_with_gloo_world2(tmp_path):
    """N > 1 path on CPU through the PRODUCT's exchange (manta_rs_amd/distributed.py: all_gather of the ranks' partial
    points -- gloo here, RCCL on the GPUs -- and the library's host sum). Without a GPU the per-rank partial MSM is
    stood in for by the oracle; tests/test_gpu_multi.py runs the same two-rank layout through the GPU MSM."""
    script = tmp_path / "w.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth, distributed
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 129                                  # odd: the two ranges differ in length
pts = H.random_points(1, 1, n, seed=3)
sc = synth.msm_scalars(1, n, "W", seed=4)
lo, hi = distributed.shard_range(n, rank, world)
assert (lo, hi) == ((0, 64) if rank == 0 else (64, 129))
part = O.msm(1, 1, pts[lo:hi], sc[lo:hi])   # stand-in for this rank's GPU MSM
ex = distributed.PartialPointExchange(1, 1)
allp = ex.all_gather(part)
assert allp.shape == (world, 12) and (allp[rank] == part).all()
for _ in range(3):                          # buffers are reused across steps
    total = ex.sum(part)
    assert (total == O.msm(1, 1, pts, sc)).all()
# G2 points (192 B) through the same exchange
p2 = H.random_points(0, 2, world, seed=5)
assert (distributed.PartialPointExchange(0, 2).sum(p2[rank]) == O.g_sum(0, 2, p2)).all()
dist.barrier()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_sharded_proof_exchange_with_gloo_world2(tmp_path):
    """The fused exchange of a sharded PROOF on CPU (distributed.ShardedProver.gather_host, gloo world 2): every rank
    contributes [k][5][slot] partial points -- stood in for by oracle MSMs over its slices of five queries -- in ONE
    all_gather; the gathered slots, summed per (proof, query), equal the five full MSMs. (The byte-identical-proof statement
    needs the GPU library: tests/test_gpu_multi.py runs it with two ranks on one device and with a one-rank RCCL group.)"""
    script = tmp_path / "w.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
import numpy as np, torch.distributed as dist
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth, distributed
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
curve, n, k = 0, 61, 2
slot = api.xyzz_limbs(curve, 2)
one = synth.to_mont([1], synth.FQ_MODULUS[curve], synth.FQ_LIMBS[curve]).reshape(-1)
def xyzz(pt, group):                       # affine -> X | Y | ZZ | ZZZ with ZZ = ZZZ = 1, padded to the G2 slot
    L = api.affine_limbs(curve, group) // 2
    out = np.zeros(slot, dtype=np.uint64)
    if pt.any():
        out[:2 * L] = pt
        out[2 * L:2 * L + len(one)] = one
        out[3 * L:3 * L + len(one)] = one
    return out
groups = (1, 1, 2, 1, 1)                   # a, b_g1, b_g2, l, h
queries = [H.random_points(curve, g, n, seed=30 + i) for i, g in enumerate(groups)]
scalars = [[synth.msm_scalars(curve, n, "W", seed=40 + 5 * q + i) for i in range(5)] for q in range(k)]
lo, hi = distributed.shard_range(n, rank, world)
mine = np.stack([np.stack([xyzz(O.msm(curve, groups[i], queries[i][lo:hi], scalars[q][i][lo:hi]), groups[i]) for i in range(5)])
                 for q in range(k)])
class StubCtx:                              # what ShardedProver needs from a context when nothing runs on a GPU
    partials_slot_limbs = slot
sp = distributed.ShardedProver(curve, None, max_batch=k, ctx=StubCtx())
parts = sp.gather_host(mine, k)
assert parts.shape == (world, k, 5, slot) and (parts[rank] == mine).all()
for q in range(k):
    for i in range(5):
        L = api.xyzz_limbs(curve, groups[i])
        total = api.xyzz_sum(curve, groups[i], parts[:, q, i, :L])
        assert (total == O.msm(curve, groups[i], queries[i], scalars[q][i])).all(), (q, i)
dist.barrier()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29534", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_proof_decode_rejects_what_ark_serialize_rejects():
    """ark-serialize 0.3 `SWFlags::from_u8` has no value for "infinity AND positive-y" (both top bits of the last byte),
    and `deserialize_with_flags` rejects a field element >= p even under the infinity flag: mg_proof_decode (host code, no
    GPU needed) must refuse both, or proofs are malleable where the reference's are not."""
    from manta_rs_amd import api
    for curve, fb in ((0, 32), (1, 48)):
        n = 4 * fb
        inf = bytearray(n)
        for end in (fb, 3 * fb, 4 * fb):
            inf[end - 1] = 0x40
        assert not api.proof_decode(curve, bytes(inf)).any()          # three points at infinity decode (to zeros)
        for end in (fb, 3 * fb, 4 * fb):                                # a, b, c in turn
            both = bytearray(inf)
            both[end - 1] = 0xC0
            with pytest.raises(api.MantaGpuError):
                api.proof_decode(curve, bytes(both))
            big = bytearray(inf)                                        # x = 2^(8 fb - 2) - 1 >= p under the infinity flag
            for i in range(end - fb, end - 1):
                big[i] = 0xFF
            big[end - 1] = 0x7F
            with pytest.raises(api.MantaGpuError):
                api.proof_decode(curve, bytes(big))
        c0_big = bytearray(inf)                                         # b's first Fq2 coefficient (no flag bits) >= p
        for i in range(fb, 2 * fb):
            c0_big[i] = 0xFF
        with pytest.raises(api.MantaGpuError):
            api.proof_decode(curve, bytes(c0_big))


def test_rust_patch_has_no_panic_on_upload_failure():
    """mantagpu.h promises status codes ("no exceptions, no abort") and the reference's `ProvingContext::new` cannot fail:
    the patch to groth16.rs must not `.expect(...)` the upload -- it defers it to the first `prove`, whose `Result` carries
    the failure as the module's opaque `Error`."""
    patch = open(os.path.join(ROOT, "rust", "manta-crypto.patch")).read()
    added = "\n".join(ln[1:] for ln in patch.splitlines() if ln.startswith("+") and not ln.startswith("+++"))
    assert ".expect(" not in added and ".unwrap()" not in added and "panic!" not in added
    assert "gpu: Arc::new(Mutex::new(None))" in added              # `new` uploads nothing
    assert ".gpu()?" in added and "GpuProvingContext::new(&self.proving_key, None).map_err(|_| Error)?" in added


def test_capture_fixture_container_round_trips_and_matches_the_rust_writer():
    """tests/fixture_io.py mirrors rust/capture/src/lib.rs (the writer that runs next to the reference): same section list,
    and encode -> decode gives back every field; truncation and trailing bytes are refused."""
    import fixture_io as FX
    rust = open(os.path.join(ROOT, "rust", "capture", "src", "lib.rs")).read()
    assert 'pub const SECTIONS: &str = "%s";' % FX.SECTIONS in rust and 'b"MGFX0001"' in rust
    c = synth.make_circuit(0, 40, 30, 4, seed=5)
    r, s = np.arange(4, dtype=np.uint64) + 7, np.arange(4, dtype=np.uint64) + 11
    blob = FX.encode(0, c.A, c.B, c.C, c.m, c.P, c.z, r, s, b"\x01" * 100, b"\x02" * 128)
    fx = FX.decode(blob)
    assert (fx.curve, fx.m, fx.P, fx.V) == (0, c.m, c.P, c.V)
    for got, want in ((fx.A, c.A), (fx.B, c.B), (fx.C, c.C)):
        assert (got.row_ptr == want.row_ptr).all() and (got.col == want.col).all() and (got.val == want.val).all()
    assert (fx.z == c.z).all() and (fx.r == r).all() and (fx.s == s).all() and fx.pk_bytes == b"\x01" * 100 and fx.proof == b"\x02" * 128
    for bad in (blob[:-1], blob + b"\x00", b"XXXXXXXX" + blob[8:]):
        with pytest.raises(ValueError):
            FX.decode(bad)


def test_release_library_has_no_calibration_switch():
    """The gather-only calibration twin of the accumulate kernel (wrong results by design) and the environment variable
    that selected it exist only in -DMG_CALIBRATION builds: nothing in the shipped library reads MANTA_ACC_GATHER_ONLY."""
    from manta_rs_amd import api
    blob = open(api.LIB_PATH, "rb").read()
    assert b"GATHER_ONLY" not in blob and b"gather_only_chunks" not in blob


def test_shard_ranges_tile_the_index_space():
    from manta_rs_amd import distributed
    for n in (1, 7, 35174, 1 << 20):
        for world in (1, 2, 3, 8):
            if world > n:
                continue
            r = [distributed.shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[g][1] == r[g + 1][0] for g in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_rust_sys_crate_declares_exactly_the_header():
    """rust/mantagpu-sys/src/lib.rs (source only: no Rust toolchain here) must bind every entry point of mantagpu.h
    and nothing else, and mirror the structs that cross the ABI field for field (mg_tuning and mg_ctx_opts since round 6)."""
    hdr = open(os.path.join(ROOT, "include", "mantagpu.h")).read()
    rs = open(os.path.join(ROOT, "rust", "mantagpu-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", hdr))
    bound = set(re.findall(r"pub fn (mg_[a-z0-9_]+)\s*\(", rs))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    for struct, n_fields in (("mg_pk_view", 13), ("mg_csr", 4), ("mg_pk_out", 12), ("mg_tuning", 15), ("mg_ctx_opts", 9)):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, rs, re.S).group(1)
        assert len(re.findall(r"pub \w+:", body)) == n_fields, struct
        end = hdr.index("} %s;" % struct)
        cbody = hdr[hdr.rindex("typedef struct", 0, end):end].split("{", 1)[1]
        cnames = re.findall(r"\*?(\w+)\s*[;,]", re.sub(r"/\*.*?\*/", "", cbody, flags=re.S))
        rnames = re.findall(r"pub (\w+):", body)
        assert rnames == cnames, (struct, rnames, cnames)


def _canned_bench_result(n_gpus=1):
    """a full detail object of a real run (round 4's 22 KB line, kept as a fixture) -- with N > 1 the legs of that path bolted on"""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    if n_gpus > 1:
        d["n_gpus"] = n_gpus
        d["strong_scaling"] = {"n_2^20": {"n_total": 1 << 20, "n_per_gpu": (1 << 20) // n_gpus, "window_bits": 17, "ms_per_msm_latency_mode": 1.9,
                                          "ms_per_msm_pipelined": 1.5, "Mscalar_s_pipelined": 700.0},
                               "n_35174": {"n_total": 35174, "n_per_gpu": 35174 // n_gpus, "window_bits": 8, "ms_per_msm_latency_mode": 0.4,
                                           "ms_per_msm_pipelined": 0.2, "Mscalar_s_pipelined": 170.0}}
        d["sharded_proof"] = {"workload": "x" * 300, "exchange": "y" * 300, "sequential": {"ms_per_proof": 1.2, "proofs_per_s": 830.0},
                              "batched": {"proofs_per_pass": 32, "passes_in_flight": 3, "proofs_per_s": 2900.0, "ms_per_proof": 0.34},
                              "task_parallel": {"placement": "z" * 200, "sequential": {"ms_per_proof": 1.3, "proofs_per_s": 770.0}}}
        d["proofs"]["per_gpu_proofs_per_s"] = [3300.0] * n_gpus
        d["collective"] = {"backend": "nccl", "ranks": n_gpus, "devices": ["cuda:%d@0000:%02x:00" % (i, 5 + 16 * i) for i in range(n_gpus)],
                           "launcher": "self", "parity_gate": {"sharded_proof_equals_single_device_on_all_ranks": True, "verified": True,
                                                               "exchange_on_gpu": True, "seconds": 9.1}}
        for k in ("ntt", "config2", "hbm_reference"):
            d.pop(k)
    return d


@pytest.mark.parametrize("n_gpus", [1, 2, 8])
def test_bench_line_is_small_enough_for_the_driver(n_gpus):
    """the driver keeps ~8 KB of stdout; round 4's 22 KB line came back `parsed: null`. The printed line is
    built by bench.compact_line from the detail object: < 8 KB (target 4 KB), valid JSON, with the contract's keys, `roofline`
    and `cpu_baseline` inside."""
    import json
    import bench
    full = _canned_bench_result(n_gpus)
    assert len(json.dumps(full)) > 20000  # the object that broke the driver
    s = bench.compact_line(full)
    assert "\n" not in s and len(s) < bench.LINE_TARGET < bench.LINE_HARD_CAP == 8192
    d = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["n_gpus"] == n_gpus
    assert d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] > r["algorithmic_bytes_per_launch"] and 0 < r["int_mad"]["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["all_cores"]["cores"] >= 1 and c["sample"]
    p = d["proofs"]
    assert p["value"] == p["batched"] and set(p["witness_profiles"]) == {"sparse", "W", "dense"}
    assert all(v["batched"] > 0 for v in p["witness_profiles"].values())
    if n_gpus > 1:
        assert d["sharded_proof"]["batched"] and d["strong_scaling"]["n_2^20"]["Mscalar_s"]
        co = d["collective"]  # what the collective saw
        assert co["backend"] == "nccl" and co["ranks"] == n_gpus and len(co["devices"]) == n_gpus and co["parity_gate"]
    else:
        assert d["ntt"]["ms"] and d["config2"]["ms"] and d["verify"]["batch_per_s"]


def test_bench_line_survives_missing_legs():
    """--quick / a failed optional leg: the headline still prints, errors are named, nothing raises"""
    import json
    import bench
    full = _canned_bench_result(1)
    for k in ("proofs", "ntt", "config2", "hbm_reference"):
        full[k] = None
    full["errors"] = {"proofs": "RuntimeError: " + "x" * 5000}
    full["roofline"]["int_mad"] = None
    full["cpu_baseline"] = None
    d = json.loads(bench.compact_line(full))
    assert d["value"] == full["value"] and "proofs" not in d and len(d["errors"]["proofs"]) <= 160 and d["roofline"]["frac"]


def test_bench_launches_its_own_ranks_when_no_launcher_did():
    """`python3 bench.py --gpus 8` (the shape of the driver's N = 1 command) must not die on WORLD_SIZE != N:
    with no launcher around it bench.py re-executes itself as N ranks; under a launcher (WORLD_SIZE set) or at N = 1 it is a rank."""
    import bench
    assert bench.launch_plan(1, {}, ["--gpus", "1"]) is None
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, ["--gpus", "8"]) is None  # the driver's torch.distributed.run form
    cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]  # the ranks see exactly the caller's flags


def test_bench_rank_refuses_a_launcher_that_disagrees(tmp_path):
    """WORLD_SIZE=2 with --gpus 4 is a launcher error: a one-line SystemExit naming both, not an AssertionError traceback, and
    before any device is touched (this runs without a GPU)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in out.stderr and "AssertionError" not in out.stderr


def test_tuning_crosses_the_abi_and_the_library_reads_fifteen_variables():
    """what a deployment decides lives in `mg_tuning` (C header == Python mirror == Rust mirror, checked above);
    the shipped library reads the environment in ONE place -- the table behind mg_tuning_env_names (<= 15 names) -- and every
    other knob of the measurement campaigns is compiled in: the shipped binary holds no other MANTA_* name, the diagnosis twin
    (every unit built with -DMG_DIAG) does. mg_set_tuning validates as a whole."""
    import ctypes
    import subprocess
    from manta_rs_amd import api
    names = api.tuning_env_names()
    assert len(names) == len(set(names)) <= 15 and "MANTA_RCCL_LIB" in names and "MANTA_GRAPH" in names
    blob = open(api.LIB_PATH, "rb").read()
    found = set(m.decode() for m in re.findall(rb"MANTA_[A-Z0-9_]{2,}", blob))
    assert found <= set(names) | {"MANTA_"}, sorted(found - set(names))  # nothing else is even a string of the release library
    src = ""
    for f in os.listdir(os.path.join(ROOT, "manta_rs_amd", "csrc")):
        if f.endswith((".cpp", ".h", ".hip")) and f != "tuning.h":
            src += open(os.path.join(ROOT, "manta_rs_amd", "csrc", f)).read()
    sites = [ln for ln in src.splitlines() if re.search(r"\bgetenv\(", ln) and "MG_CALIBRATION" not in ln]
    assert len(sites) <= 3, sites  # the tuning table's loop, MANTA_RCCL_LIB, and nothing that is not named above
    diag = os.path.join(os.path.dirname(api.LIB_PATH), "libmantagpu_diag.so")
    assert b"MANTA_RED_MIN" in open(diag, "rb").read()  # the A/B knobs live on in the diagnosis twin
    # defaults, get, set (in a child: process-wide state), refusal of an out-of-range field
    d = api.tuning_defaults()
    assert d.as_dict() == dict(graph_mode=1, graph_mode_batch=-1, prove_streams=6, linear_chains=3, coalesce_inflight=2, coalesce_gather_us=100,
                               batch_inflight=3, queue_aware=1, msm_dedicated_queues=1, window_bits_narrow=0, window_bits_wide=0, window_bits_h=0,
                               window_bits_g2=0, full_table_bytes=-1)
    code = ("import sys; sys.path.insert(0, %r)\nfrom manta_rs_amd import api\n"
            "t = api.get_tuning(); assert t.graph_mode == 2 and t.coalesce_inflight == 0 and t.full_table_bytes == 3500000000 and t.prove_streams == 6, t.as_dict()\n"
            "api.set_tuning(graph_mode=0, batch_inflight=2); t = api.get_tuning(); assert t.graph_mode == 0 and t.batch_inflight == 2 and t.coalesce_inflight == 0\n"
            "try:\n    api.set_tuning(prove_streams=7); raise SystemExit('accepted prove_streams=7')\nexcept api.MantaGpuError: pass\n"
            "assert api.get_tuning().prove_streams == 6\nprint('tuning ok')\n" % ROOT)
    env = dict(os.environ, MANTA_GRAPH="split", MANTA_COALESCE="0", MANTA_FULL_TABLE_GB="3.5", MANTA_PROVE_STREAMS="9")  # 9: out of range, ignored
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "tuning ok" in out.stdout, out.stdout + out.stderr
