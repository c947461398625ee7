"""Makes the reference-held known answers travel (the GPU box has no /root/reference). Run in the build container:
    python tests/golden/gen_reference_fixtures.py
1. copies the three ARCHIVED testnet verifying keys (manta-parameters/data/archive/testnet/verifying/*.dat, BLAKE3 in
   manta-parameters/data.checkfile) next to the current ones as tests/golden/testnet-*.dat -- binary data files the
   reference's own tests load (manta-parameters/src/lib.rs), byte-for-byte;
2. extracts the NUMBERS of the reference's BLS12-381 Fr Poseidon known answers
   (manta-pay/src/crypto/poseidon/{parameters_hardcoded_test/lfsr_values, mds_hardcoded_tests/width3,
   permutation_hardcoded_test/width3}; used by round_constants.rs:84-93, mds.rs:363-385, hash.rs:249-258) into
   tests/golden/poseidon_bls381_fr.json -- decimal field elements only, no source text."""
import json, os, re, shutil
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
for f in ("to-private", "to-public", "private-transfer"):
    shutil.copyfile(f"{REF}/manta-parameters/data/archive/testnet/verifying/{f}.dat", f"{HERE}/testnet-{f}.dat")
base = f"{REF}/manta-pay/src/crypto/poseidon/"
num = lambda path: [x for x in re.findall(r'"(\d+)"', open(base + path).read())]
rc, mds, out = num("parameters_hardcoded_test/lfsr_values"), num("mds_hardcoded_tests/width3"), num("permutation_hardcoded_test/width3")
assert (len(rc), len(mds), len(out)) == (189, 9, 3)
json.dump({"field": "BLS12-381 Fr", "width": 3, "full_rounds": 8, "partial_rounds": 55, "sbox_exponent": 5,
           "input": ["3", "1", "2"], "round_constants": rc, "mds": [mds[0:3], mds[3:6], mds[6:9]], "output": out,
           "sources": {"round_constants": "manta-pay/src/crypto/poseidon/parameters_hardcoded_test/lfsr_values (round_constants.rs:84-93)",
                       "mds": "manta-pay/src/crypto/poseidon/mds_hardcoded_tests/width3 (mds.rs:363-385)",
                       "output": "manta-pay/src/crypto/poseidon/permutation_hardcoded_test/width3 (hash.rs:249-258): "
                                 "Poseidon permutation of (domain tag 2^2 - 1, 1, 2)"}},
          open(f"{HERE}/poseidon_bls381_fr.json", "w"), indent=0)
print("ok")
