"""dev tool: registers / spills / scratch of every kernel in one .hip unit (cross-compiles here, no GPU needed).
usage: python tools/kinfo.py msm_bls381_g1 [regex] [extra hipcc flags...]"""
import os, re, subprocess, sys
unit = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "."
extra = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/kinfo", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", *extra, "-c",
                unit + ".hip", "-o", f"/tmp/kinfo/{unit}.o", "--save-temps=obj"], cwd=os.path.join(root, "manta_rs_amd", "csrc"),
               stderr=subprocess.DEVNULL, check=True)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f"/tmp/kinfo/{unit}-hip-amdgcn-amd-amdhsa-gfx950.out"],
                       capture_output=True, text=True).stdout
cur, out = {}, []
for ln in notes.splitlines():
    m = re.match(r"\s+-?\s*\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size|agpr_count):\s+(\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "agpr_count" and cur.get("name"):
        out.append(cur)
        cur = {}
    cur[k] = v
if cur.get("name"):
    out.append(cur)
for d in out:
    nm = subprocess.run(["c++filt", d["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("mg::", "")
    if re.search(filt, nm):
        print("%-62s vgpr %s agpr %s spill %s scratch %s" % (nm[:62], d.get("vgpr_count"), d.get("agpr_count"), d.get("vgpr_spill_count"),
                                                            d.get("private_segment_fixed_size")))
