#!/bin/bash
# dev tool: latency-mode 2^20 MSM under different tail knobs (each run prints latency_mode ms and the pipelined value)
for env in "X=1" "MANTA_COOP_TILES=512" "MANTA_COOP_TILES=1024" "MANTA_MERGE_G=8" "MANTA_MERGE_G=8 MANTA_COOP_TILES=512" "MANTA_MERGE_G=2 MANTA_COOP_WAVES=4096" "MANTA_COOP_WAVES=2048 MANTA_COOP_TILES=512"; do
  echo -n "$env : "
  env $env python bench.py --quick --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['latency_mode'], d['roofline']['kernel_ms'])"
done
