"""The 45-second version of tools/soak.py in the suite (the 10-minute run is profiles/r06_soak.txt): three contexts, six threads,
single and batched calls mixed, stand-alone MSMs three in flight beside them, a context replaced every three seconds -- in rotation
one with full tables, one without, one sharded over devices [0, 0], one with its own mg_tuning --, every proof and every MSM result
byte-compared with the oracle's."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_soak_45_seconds(gpu):
    import soak
    st = soak.soak(seconds=45.0, threads=6, recycle_every=2.5, pool=4, log=lambda *_: None)
    assert st["errors"] == 0 and st["mismatches"] == 0, st
    assert st["proofs"] > 4000 and st["contexts_created"] >= 10 and st["batch_calls"] > 50 and st["single_calls"] > 400 and st["msms"] > 300, st
    assert min(st["contexts_by_variant"].values()) >= 2, st
