"""dev tool (round 5): the wrong-C defect of linear part-A graphs (profiles/r04_z3_helper_thread.txt), bisected.

driver:  python tools/diag_linear.py            -> runs the child under a list of (library, environment) configurations
child:   python tools/diag_linear.py --child    -> create ctx, set_r1cs, prove x4, then a SECOND context of the same key is created,
         given the circuit and proves three times; after every step the first context proves again and its proof is split A | B | C
         against the oracle. With a -DMG_DIAG library (mg_diag_slot_sums) every buffer of the slot is summed after each step and the
         first buffer whose sum differs from the good state is named.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["z", "h(a)", "b", "c", "nonzero(a|b|c)", "first_nonzero"] + ["%s.%s" % (m, b) for m in ("msm0", "msm_g2", "msm_h")
                                   for b in ("count", "keys_in", "vals_in", "keys_out", "vals_out", "pkeys0", "ppts0", "ppts1", "buckets", "redS")]


def child():
    import numpy as np  # noqa: F401
    import oracle_lib as O
    import helpers as H
    from manta_rs_amd import api, synth
    api.init(0)
    shape = os.environ.get("DIAG_SHAPE", "small")
    curve = api.BN254
    if shape == "small":
        c = synth.make_circuit(curve, 700, 500, 9, seed=11)
    elif shape == "pt":
        c = synth.make_shape(curve, "private_transfer", profile="W")
    else:
        curve = api.BLS12_381
        c = synth.make_circuit(curve, (1 << 15) - 16, 1 << 15, 16, seed=12, profile="W")
    O.set_threads(O.usable_cpus())
    pk = O.groth16_setup(c, H.toxic(curve))
    budget = os.environ.get("DIAG_FULL_TABLE_BYTES")
    kw = {} if budget is None else {"full_table_bytes": int(budget)}
    ctx = api.ProvingContext(curve, pk, **kw)
    r1cs = api.R1CS.from_circuit(c)
    ctx.set_r1cs(r1cs)
    rs = H.rand_fr_mont(curve, 8, seed=5)
    truth = O.groth16_prove(c, pk, rs[0], rs[1])
    g1b, g2b = api.PROOF_BYTES[curve] // 4, api.PROOF_BYTES[curve] // 2
    diag = getattr(api.LIB, "mg_diag_slot_sums", None) if hasattr(api.LIB, "mg_diag_slot_sums") else None
    if diag is not None:
        diag.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int]
        diag.restype = ctypes.c_int
    good = [None]
    nbad = [0]

    def sums(mode=0):
        if diag is None:
            return None
        buf = (ctypes.c_uint64 * 64)()
        n = diag(ctx.handle, buf, 64, mode)
        return list(buf[:n]) if n > 0 else None

    def check(tag, ref=False):
        p = api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
        parts = "".join("ok " if p[a:b] == truth[a:b] else "BAD " for a, b in ((0, g1b), (g1b, g1b + g2b), (g1b + g2b, 2 * g1b + g2b)))
        nbad[0] += "BAD" in parts
        extra = ""
        s = sums()
        if s is not None:
            if ref:
                good[0] = s
            elif good[0] is not None:
                diff = [NAMES[i] for i in range(min(len(s), len(NAMES))) if s[i] != good[0][i]]
                extra = " | buffers differing from the good state: " + (", ".join(diff) if diff else "none")
            extra += " | non-zero words of a|b|c: %d (first at %d)" % (s[4], s[5] if s[5] < (1 << 63) else -1)
        print(f"{tag:<30} A B C: {parts}{extra}", flush=True)

    for i in range(4):
        check(f"run {i}", ref=i == 2)  # run 2 = the first replay of the freshly captured graphs
    check("run 4 (same state)")
    c2 = api.ProvingContext(curve, pk, **kw)
    check("after create(ctx2)")
    c2.set_r1cs(r1cs)
    check("after set_r1cs(ctx2)")
    if diag is not None and os.environ.get("DIAG_EAGER"):
        sums(1)  # this slot launches eagerly from now on: same buffers, plain launches
        check("ctx1 slot eager now")
        check("ctx1 slot eager again")
        sums(2)
        check("ctx1 slot re-captured")
        check("ctx1 slot replayed")
    for i in range(3):
        api.Groth16.prove_with_randomness(c2, c.z, rs[0], rs[1])
        check(f"after prove(ctx2) {i + 1}")
    # batched + threads on both contexts for good measure
    zs = np.stack([c.z] * 8)
    got = api.Groth16.prove_batch(c2, zs, np.stack([rs[0]] * 8), np.stack([rs[1]] * 8))
    if not os.environ.get("MG_DIAG_WM_STOP"):
        assert all(g == truth for g in got), "batched proofs of ctx2 differ"
    check("after batch(ctx2)")
    print("RESULT nbad=%d" % nbad[0], flush=True)


def driver():
    lib_diag = os.path.join(ROOT, "manta_rs_amd", "lib", "libmantagpu_diag.so")
    S1 = {"MANTA_PROVE_STREAMS": "1", "MANTA_LIB": lib_diag}
    MS = dict(S1, MG_DIAG_MEMSET="1")  # the round-4 memset node back in front of the SpMV
    configs = [
        ("shipped default", {}),
        ("shipped, MANTA_PROVE_STREAMS=1 (fenced: must behave like the default)", {"MANTA_PROVE_STREAMS": "1"}),
        ("shipped, MANTA_GRAPH=split (six linear graphs)", {"MANTA_GRAPH": "split"}),
        ("shipped, no full tables (linear G2 graph with sort + reduce)", {"DIAG_FULL_TABLE_BYTES": "0"}),
        ("shipped, split, no full tables", {"MANTA_GRAPH": "split", "DIAG_FULL_TABLE_BYTES": "0"}),
        ("diag lib, default topology", {"MANTA_LIB": lib_diag}),
        ("diag lib, linear part A", dict(S1)),
        ("diag lib, linear part A, MANTA_Z3=0", dict(S1, MANTA_Z3="0")),
        ("diag lib, linear part A, no full tables", dict(S1, DIAG_FULL_TABLE_BYTES="0")),
        ("diag lib, linear part A + memset node (NEGATIVE CONTROL: must go BAD)", dict(MS)),
        ("diag lib, split + memset node (NEGATIVE CONTROL: must go BAD)", {"MANTA_LIB": lib_diag, "MANTA_GRAPH": "split", "MG_DIAG_MEMSET": "1"}),
        ("diag lib, default topology + memset node (forked graph: stays right)", {"MANTA_LIB": lib_diag, "MG_DIAG_MEMSET": "1"}),
        ("diag lib, linear + memset node, witness map cut after the memset", dict(MS, MG_DIAG_WM_STOP="1")),
        ("diag lib, linear + memset node, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0", dict(MS, DEBUG_CLR_GRAPH_PACKET_CAPTURE="0")),
        ("shipped default, PrivateTransfer shape", {"DIAG_SHAPE": "pt"}),
        ("shipped split, PrivateTransfer shape, no full tables", {"DIAG_SHAPE": "pt", "MANTA_GRAPH": "split", "DIAG_FULL_TABLE_BYTES": "0"}),
        ("diag lib, linear part A, PrivateTransfer shape, no full tables", dict(S1, DIAG_SHAPE="pt", DIAG_FULL_TABLE_BYTES="0")),
        ("shipped default, PrivateTransfer shape, no full tables", {"DIAG_SHAPE": "pt", "DIAG_FULL_TABLE_BYTES": "0"}),
        ("shipped default, BLS12-381 2^15, no full tables", {"DIAG_SHAPE": "bls", "DIAG_FULL_TABLE_BYTES": "0"}),
        ("shipped split, BLS12-381 2^15, no full tables", {"DIAG_SHAPE": "bls", "MANTA_GRAPH": "split", "DIAG_FULL_TABLE_BYTES": "0"}),
        ("diag lib, linear part A, BLS12-381 2^15", dict(S1, DIAG_SHAPE="bls")),
    ]
    only = os.environ.get("DIAG_ONLY")
    for name, env in configs:
        if only and only not in name:
            continue
        print("## " + name + "   " + " ".join("%s=%s" % kv for kv in env.items() if kv[0] != "MANTA_LIB"), flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=900)
        print(r.stdout.rstrip(), flush=True)
        if r.returncode:
            print("   child exit code %d: %s" % (r.returncode, r.stderr[-1500:]), flush=True)


if __name__ == "__main__":
    child() if "--child" in sys.argv else driver()
