#!/bin/bash
# window-width / reduce-knob sweep of the 2^20 BLS12-381 G1 MSM (bench.py --quick): one summary line per setting.
# usage: source tools/red_sweep.sh <outdir>; run <name> VAR=value ...
out=${1:-gpurun_out/red_sweep}
mkdir -p $out
run() {
  name=$1; shift
  env "$@" python bench.py --quick --no-cpu-baseline --steps 12 > $out/$name.json 2> $out/$name.err
  python - "$name" "$out/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "pipelined", d["value"], "ms", d["ms_per_step"], "latency", d["config"]["latency_mode"]["ms_per_msm"], "acc_ms", d["roofline"]["kernel_ms"], flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
}
