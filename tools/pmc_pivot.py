"""dev tool: pivot tools/pmc_kernels.py lines (one counter per line) into one row per kernel, next to the durations of
tools/rocprof_summary.py. usage: python tools/pmc_pivot.py <pmc lines> <durations> [kernels per unit of work]"""
import re, sys
from collections import OrderedDict
rows = OrderedDict()
for ln in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+(\w+)\s+launches=(\d+)\s+sum=(\S+)\s+per_launch=(\S+)", ln)
    if not m:
        continue
    k = re.sub(r"^void ", "", m.group(1)).replace("(anonymous namespace)::", "").replace("mg::", "")
    k = re.sub(r"\(.*", "", k)[:58]
    rows.setdefault(k, {})[m.group(2)] = (int(m.group(3)), float(m.group(5)))
dur = {}
for ln in open(sys.argv[2]):
    m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
    if m:
        k = re.sub(r"\(.*", "", m.group(1).replace("(anonymous namespace)::", "").replace("mg::", ""))[:58]
        dur[k] = (int(m.group(2)), float(m.group(4)))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
        "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "FETCH_SIZE", "WRITE_SIZE"]
hdr = ["waves", "VALU", "SALU", "LDS", "LDSconfl", "VMEMrd", "VMEMwr", "busy_cyc", "act_VALU", "wait_any", "FETCH_KiB", "WRITE_KiB"]
print(f"{'kernel (per launch)':<58} {'launches':>8} {'avg_us':>8} " + " ".join(f"{h:>9}" for h in hdr))
for k, d in rows.items():
    n = max(v[0] for v in d.values())
    du = next((v for kk, v in dur.items() if kk.startswith(k[:50]) or k.startswith(kk[:50])), (0, 0.0))
    print(f"{k:<58} {n:>8} {du[1]:>8.1f} " + " ".join(f"{d[c][1]:>9.3g}" if c in d else f"{'-':>9}" for c in cols))
