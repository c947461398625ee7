#!/bin/bash
# dev tool (round 4): single-proof latency and batched rate per witness profile under the table choices
out=gpurun_out/r4b; mkdir -p $out
run() { name=$1; shift; (env "$@" CHECK=0 timeout 300 python tools/profile_proofs.py sparse,W,dense > $out/$name.txt 2>&1; echo rc=$? >> $out/$name.txt); }
run full_default
run full_80gb MANTA_FULL_TABLE_GB=80
run bucket_narrow MANTA_FULL_TABLE_GB=0
run bucket_wide11 MANTA_FULL_TABLE_GB=0 MANTA_WIDE_MIN=1
run bucket_wide13 MANTA_FULL_TABLE_GB=0 MANTA_WIDE_MIN=1 MANTA_PROVE_CW=13
run bucket_wide9 MANTA_FULL_TABLE_GB=0 MANTA_WIDE_MIN=1 MANTA_PROVE_CW=9
grep -H "==\|sequential\|batched\|pass of\|rc=" $out/*.txt
