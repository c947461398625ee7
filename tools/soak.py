"""Soak run: what a signer process does for hours, compressed into minutes.

Three ProvingContexts (ToPrivate / ToPublic / PrivateTransfer shapes: `MultiProvingContext`, manta-accounting/src/transfer/
canonical.rs:561-588) shared by six host threads (manta-pay/src/bin/simulation.rs:36-38) that mix single calls (coalesced by the library
when they overlap) with batches of 2 / 5 / 32 proofs; a recycler thread replaces one context every few seconds (create -> set_r1cs ->
swap in -> drain the old one -> destroy), so graphs are captured, replayed and destroyed next to other contexts' passes the whole time.
EVERY proof is byte-compared with the CPU oracle's proof of the same (circuit, key, assignment, r, s), precomputed before the run; the
pool entries have DISTINCT assignments.

usage: python tools/soak.py [seconds=600] [threads=6] [recycle_every_s=4]      -> a report on stdout, exit code 1 on any mismatch / error
"""
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def soak(seconds=600.0, threads=6, recycle_every=4.0, pool=6, shapes=("to_private", "to_public", "private_transfer"), log=print,
         mix=(1, 1, 1, 1, 2, 5, 32)):
    import numpy as np
    import oracle_lib as O
    import helpers as H
    from manta_rs_amd import api, synth, keygen
    api.init(0)
    curve = api.BN254
    O.set_threads(O.usable_cpus())
    t_setup = time.perf_counter()
    S = {}
    for si, name in enumerate(shapes):
        c = synth.make_shape(curve, name, profile="W")
        pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=40 + si), synth.FR_MODULUS[curve]))
        rs = H.rand_fr_mont(curve, 2 * pool, seed=500 + si)
        rs[1][:] = 0  # one pair with r = 0 (b_g1 unused: ark-groth16 skips it)
        # every pool entry has its OWN satisfying assignment (fresh public inputs and witnesses): consecutive passes of a slot see
        # different MSM results, so a pass that read a previous pass's staged data would be caught (with one z for all it could not)
        R = synth.Reassigner(c)
        zs = [c.z] + [R.assign(0x50AC0000 + 16 * si + j).z for j in range(1, pool)]
        want = [O.groth16_prove(c, pk, rs[j], rs[pool + j], z=zs[j]) for j in range(pool)]
        S[name] = {"c": c, "pk": pk, "rs": rs, "zs": zs, "want": want, "r1cs": api.R1CS.from_circuit(c), "box": None, "gen": 0}
    O.set_threads(1)
    reg = threading.Lock()
    drained = threading.Condition(reg)
    slock = threading.Lock()

    # Round 6: the contexts that come and go are not all alike -- in rotation: a signer's budget of full tables;
    # NO full tables (bucket tables, sort + bucket reduce on every chain); every MSM range-sharded over devices [0, 0] (two shards on
    # one GPU, partial points through pinned host memory, no coalescing); a context with its own mg_tuning (split graphs, one pass of
    # a batch in flight, narrow windows of 7 bits)
    variants = (dict(full_table_bytes=12 << 30), dict(full_table_bytes=0), dict(devices=[0, 0], full_table_bytes=4 << 30),
                dict(full_table_bytes=6 << 30, tuning=dict(graph_mode=api.GRAPH_SPLIT, batch_inflight=1, coalesce_inflight=1)))
    only = os.environ.get("SOAK_VARIANTS")  # diagnosis: e.g. "0" or "0,1" restricts the rotation
    if only:
        variants = tuple(variants[int(x)] for x in only.split(","))
    made = {"n": 0, "by_variant": [0] * len(variants)}

    def make_ctx(name):
        s = S[name]
        with slock:
            v = made["n"] % len(variants)
            made["n"] += 1
            made["by_variant"][v] += 1
        ctx = api.ProvingContext(curve, s["pk"], **variants[v])
        ctx.set_r1cs(s["r1cs"])
        return ctx

    class Box:  # a context and the calls in flight on it
        def __init__(self, ctx):
            self.ctx, self.users = ctx, 0

    for name in shapes:
        S[name]["box"] = Box(make_ctx(name))
    log("soak: %d shapes, %d oracle proofs each, setup %.1f s; running %.0f s with %d threads, a context recycled every %.1f s"
        % (len(shapes), pool, time.perf_counter() - t_setup, seconds, threads, recycle_every))
    stats = {"proofs": 0, "single_calls": 0, "batch_calls": 0, "msms": 0, "mismatches": 0, "errors": 0, "contexts_created": len(shapes), "first_bad": None}
    deadline = time.perf_counter() + seconds
    stop = threading.Event()

    def acquire(name):
        with reg:
            b = S[name]["box"]
            b.users += 1
            return b

    def release(b):
        with reg:
            b.users -= 1
            drained.notify_all()

    def worker(tid):
        rnd = random.Random(1000 + tid)
        while time.perf_counter() < deadline and not stop.is_set():
            name = shapes[rnd.randrange(len(shapes))]
            s = S[name]
            box = acquire(name)
            ctx = box.ctx
            try:
                k = rnd.choice(mix) if name != "private_transfer" or rnd.random() < 0.5 else 1
                js = [rnd.randrange(pool) for _ in range(k)]
                rs = s["rs"]
                if k == 1:
                    got = [api.Groth16.prove_with_randomness(ctx, s["zs"][js[0]], rs[js[0]], rs[pool + js[0]])]
                else:
                    got = api.Groth16.prove_batch(ctx, np.stack([s["zs"][j] for j in js]), rs[js], rs[[pool + j for j in js]])
                bad = [(name, j) for j, g in zip(js, got) if g != s["want"][j]]
                with slock:
                    stats["proofs"] += k
                    stats["single_calls" if k == 1 else "batch_calls"] += 1
                    if bad:
                        stats["mismatches"] += len(bad)
                        g, w = got[js.index(bad[0][1])], s["want"][bad[0][1]]
                        ev = {"shape": name, "k": k, "j": bad[0][1], "thread": tid, "gen": s["gen"], "t": round(seconds - (deadline - time.perf_counter()), 2),
                              "elements_equal_A_B_C": [g[a:b] == w[a:b] for a, b in ((0, 32), (32, 96), (96, 128))],
                              "equals_another_pool_proof": [jj for jj in range(pool) if g == s["want"][jj]],
                              "A_equals_pool_A": [jj for jj in range(pool) if g[:32] == s["want"][jj][:32]]}
                        stats.setdefault("bad_events", [])
                        if len(stats["bad_events"]) < 12:
                            stats["bad_events"].append(ev)
                        if stats["first_bad"] is None:
                            stats["first_bad"] = ev
            except Exception as e:  # noqa: BLE001
                with slock:
                    stats["errors"] += 1
                    if stats["first_bad"] is None:
                        stats["first_bad"] = {"shape": name, "error": repr(e)[:300]}
            finally:
                release(box)

    # stand-alone MSMs beside the proofs (round 6: they run on streams with hardware queues of their own -- blocking streams --
    # next to captures, replays and context creation): three in flight, BN254 G1 and G2, every result against the oracle's
    msm_sets = []
    for group, n, pre in ((1, 1 << 14, 11), (2, 3000, 8)):
        pts = H.random_points(curve, group, 1500, seed=900 + group)
        pts = np.concatenate([pts] * (-(-n // 1500)))[:n]
        sc = synth.msm_scalars(curve, n, "W", seed=910 + group)
        msm_sets.append((api.Bases(curve, group, pts, precompute_window_bits=pre), api.DeviceBuffer.from_numpy(sc), n, O.msm(curve, group, pts, sc, algo=1)))

    if os.environ.get("SOAK_TOUCH_MSM"):  # diagnosis: ONE stand-alone MSM before the run (creates its stream), none during it
        b0, d0, n0, w0 = msm_sets[0]
        assert (api.VariableBaseMSM.launch(b0, d0, n0, sparse=True).finish() == w0).all()

    def msm_worker():
        while time.perf_counter() < deadline and not stop.is_set() and not os.environ.get("SOAK_NO_MSM"):
            for b, d, n, want in msm_sets:
                try:
                    t_a = time.perf_counter()
                    jobs = [api.VariableBaseMSM.launch(b, d, n, sparse=True) for _ in range(3)]
                    t_b = time.perf_counter()
                    bad = sum(0 if (j.finish() == want).all() else 1 for j in jobs)
                    with slock:
                        stats["msm_launch_s"] = stats.get("msm_launch_s", 0.0) + (t_b - t_a)
                        stats["msm_finish_s"] = stats.get("msm_finish_s", 0.0) + (time.perf_counter() - t_b)
                except Exception as e:  # noqa: BLE001
                    with slock:
                        stats["errors"] += 1
                        stats["first_bad"] = stats["first_bad"] or {"msm": True, "error": repr(e)[:300]}
                    continue
                with slock:
                    stats["msms"] += 3
                    if bad:
                        stats["mismatches"] += bad
                        stats["first_bad"] = stats["first_bad"] or {"msm": True, "group": b.group, "bad": bad}
            time.sleep(float(os.environ.get("SOAK_MSM_SLEEP", "0.002")))

    def recycler():
        rnd = random.Random(77)
        while not stop.wait(recycle_every) and time.perf_counter() < deadline:
            name = shapes[rnd.randrange(len(shapes))]
            try:
                fresh = make_ctx(name)  # built while the old one keeps serving
            except Exception as e:  # noqa: BLE001
                with slock:
                    stats["errors"] += 1
                    stats["first_bad"] = stats["first_bad"] or {"shape": name, "error": "recycle: " + repr(e)[:300]}
                continue
            with reg:
                s = S[name]
                old, s["box"] = s["box"], Box(fresh)
                s["gen"] += 1
                while old.users > 0:  # calls that took the old context before the swap: let them finish, then destroy it
                    drained.wait(0.05)
            old.ctx.close()
            with slock:
                stats["contexts_created"] += 1

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(threads)] + [threading.Thread(target=msm_worker), threading.Thread(target=recycler)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts[:-1]]
    stop.set()
    ts[-1].join()
    dt = time.perf_counter() - t0
    for name in shapes:
        S[name]["box"].ctx.close()
    stats["contexts_by_variant"] = dict(zip(("full_tables", "no_full_tables", "sharded_0_0", "own_tuning"), made["by_variant"])) if not only else made["by_variant"]
    stats["seconds"] = round(dt, 1)
    stats["proofs_per_s"] = round(stats["proofs"] / dt, 1)
    return stats


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    thr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rec = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
    kw = {}
    if os.environ.get("SOAK_SHAPES"):
        kw["shapes"] = tuple(os.environ["SOAK_SHAPES"].split(","))
    if os.environ.get("SOAK_MIX"):
        kw["mix"] = tuple(int(x) for x in os.environ["SOAK_MIX"].split(","))
    st = soak(secs, thr, rec, **kw)
    print("soak result:", st, flush=True)
    sys.exit(1 if (st["mismatches"] or st["errors"]) else 0)
