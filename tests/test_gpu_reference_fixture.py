"""The one parity lever left: hot-path vectors produced by the REFERENCE. rust/capture (source only here: no Rust
toolchain in the build image) hooks `Groth16::prove` (manta-crypto/src/arkworks/groth16.rs:589-600) while the reference's own
helpers prove real manta-pay transfers (manta-pay/src/test/payment.rs:52-83,222-273,364-413, the functions
manta-benchmark/benches/{to_private,private_transfer,to_public}.rs time) and writes (A, B, C, z, r, s, proving key, proof
bytes) per proof into tests/golden/arkworks/*.bin. This test feeds every such file to the GPU library through the key's wire
format (mg_ctx_create_from_bytes) and mg_groth16_prove and demands the reference's proof bytes, bit for bit.

Until somebody with `cargo` runs the capture the directory is empty and the reference half SKIPS, loudly; the consumer path
itself is exercised on files in the identical container written by the CPU oracle (tests/fixture_io.py)."""
import glob
import os

import numpy as np
import pytest

import fixture_io as FX
import helpers as H
import oracle_lib as O
from manta_rs_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CAPTURED = sorted(glob.glob(os.path.join(HERE, "golden", "arkworks", "*.bin")))


def consume(gpu, data: bytes):
    """fixture bytes -> the GPU prover's proof bytes for the captured (key, circuit, assignment, randomness)"""
    fx = FX.decode(data)
    ctx = gpu.ProvingContext.decode(fx.curve, fx.pk_bytes)
    assert ctx.num_variables == fx.V and ctx.num_inputs == fx.P
    ctx.set_r1cs(gpu.R1CS(fx.curve, fx.A, fx.B, fx.C, fx.m, fx.P, fx.z))
    got = gpu.Groth16.prove_with_randomness(ctx, fx.z, fx.r, fx.s)
    again = gpu.Groth16.prove_batch(ctx, np.stack([fx.z] * 2), np.stack([fx.r] * 2), np.stack([fx.s] * 2))
    assert again == [got, got]
    pts = gpu.proof_decode(fx.curve, got)  # also: the proof is a valid encoding of three subgroup points
    assert pts.any()
    ctx.close()
    return fx, got


@pytest.mark.skipif(not CAPTURED, reason="NO REFERENCE-PRODUCED VECTORS: tests/golden/arkworks/*.bin is empty -- run rust/capture "
                                         "(cargo test, see rust/capture/tests/capture.rs) on a machine with a Rust toolchain and commit "
                                         "the files; until then MSM / NTT / proof parity rests on the CPU oracle (parity unpinned)")
@pytest.mark.parametrize("path", CAPTURED or ["<none>"])
def test_gpu_proof_equals_the_reference_capture(gpu, path):
    fx, got = consume(gpu, open(path, "rb").read())
    assert got == fx.proof, "GPU proof differs from the arkworks proof captured from the reference: " + os.path.basename(path)
    # the capture also writes the witness histogram of the real transfer (rust/capture `histogram_json`): it must be the histogram
    # of the z in the fixture -- and it is the number that tells which synthetic profile (synth.py: sparse / W / dense) is closest
    hist = path[:-4] + ".hist.json"
    if os.path.exists(hist):
        import json
        h = json.load(open(hist))
        z_int = synth.from_mont(fx.z, synth.FR_MODULUS[fx.curve])
        mine = synth.histogram(z_int)
        assert h["n"] == mine["n"] == fx.V
        for k in ("zero", "one", "small", "dense"):
            assert h[k] == round(mine[k] * mine["n"]), k
        print("witness histogram of %s: %s" % (os.path.basename(path), {k: round(h[k] / h["n"], 4) for k in ("zero", "one", "small", "dense")}))


@pytest.mark.parametrize("curve", [0, 1])
def test_fixture_consumer_on_an_oracle_written_capture(gpu, curve, tmp_path):
    """the same consumer on a file in the same container, written here by the CPU oracle: key through the wire format,
    matrices / assignment / randomness as the capture lays them out, expected bytes = the oracle's proof"""
    c = synth.make_circuit(curve, 600, 450, 8, seed=0xF1C5)
    pk = O.groth16_setup(c, H.toxic(curve, seed=9))
    rs = H.rand_fr_mont(curve, 2, seed=10)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    path = tmp_path / "oracle-0000.bin"
    path.write_bytes(FX.encode(curve, c.A, c.B, c.C, c.m, c.P, c.z, rs[0], rs[1], FX.pk_bytes(O, curve, pk), want))
    fx, got = consume(gpu, path.read_bytes())
    assert got == fx.proof == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], got) == 1
