"""dev tool: ntt_pass_kernel durations grouped by (variant, grid, workgroup) from a rocprofv3 kernel-trace db."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
acc = collections.defaultdict(list)
for name, s, e, g, gy, gz, w in db.execute(f"select name, start, end, {gx}, {gx.replace('x','y')}, {gx.replace('x','z')}, {wx} from kernels where name like '%ntt_pass%'"):
    acc[("DIF" if "true" in name else "DIT", g, gy, gz, w)].append((e - s) / 1e3)
for k, v in sorted(acc.items()):
    print(k, f"n={len(v)} avg={sum(v)/len(v):.1f} us min={min(v):.1f}")
