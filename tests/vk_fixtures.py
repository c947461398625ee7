"""Parser of the reference's committed verifying-key files (tests/golden/*.dat; `VerifyingContext` wire layout,
manta-crypto/src/arkworks/groth16.rs:337-361: alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | Vec gamma_abc_g1 |
alpha_g1_beta_g2 (Fq12, 384 B) | two G2Prepared). Shared by the CPU and GPU pin tests."""
import os
import struct

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
VK_FILES = {"to-private": 13, "private-transfer": 27, "to-public": 19,
            "testnet-to-private": 13, "testnet-private-transfer": 27, "testnet-to-public": 19}


class VK:
    def __init__(self, name):
        d = self.raw = open(os.path.join(HERE, "golden", name + ".dat"), "rb").read()
        self.alpha_bytes = d[0:32]
        self.g2_bytes = [d[32:96], d[96:160], d[160:224]]
        (self.P,) = struct.unpack("<Q", d[224:232])
        self.abc_bytes = [d[232 + 32 * i:264 + 32 * i] for i in range(self.P)]
        off = 232 + 32 * self.P
        self.alpha_beta_bytes = d[off:off + 384]
        dec = lambda g, b: O.deserialize(0, g, b)
        ok, self.alpha = dec(1, self.alpha_bytes)
        oks = [ok]
        self.g2 = []
        for b in self.g2_bytes:
            ok, pt = dec(2, b)
            oks.append(ok)
            self.g2.append(pt)
        self.beta, self.gamma, self.delta = self.g2
        self.abc = []
        for b in self.abc_bytes:
            ok, pt = dec(1, b)
            oks.append(ok)
            self.abc.append(pt)
        assert all(oks)
