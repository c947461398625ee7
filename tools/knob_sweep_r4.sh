#!/bin/bash
# dev tool (round 4): merge / chunk knobs against single-proof latency on W and dense witnesses
out=gpurun_out/${1:-r4d}; mkdir -p $out
run() { name=$1; shift; (env "$@" CHECK=0 SEQ_ONLY=1 timeout 200 python tools/profile_proofs.py sparse,W,dense > $out/$name.txt 2>&1; echo rc=$? >> $out/$name.txt); echo "$name: $(grep sequential $out/$name.txt | awk '{printf "%s ", $2}')"; }
run base
run g4 MANTA_MERGE_G=4
run g8 MANTA_MERGE_G=8
run g2 MANTA_MERGE_G=2
run g4_coop2k MANTA_MERGE_G=4 MANTA_COOP_WAVES=2048
run g8_coop2k MANTA_MERGE_G=8 MANTA_COOP_WAVES=2048
run rw1 MANTA_ACC_ROUND_WAVES=1
run rw2 MANTA_ACC_ROUND_WAVES=2
run rw1_g4 MANTA_ACC_ROUND_WAVES=1 MANTA_MERGE_G=4
run rw2_g4 MANTA_ACC_ROUND_WAVES=2 MANTA_MERGE_G=4
run rw2_g8 MANTA_ACC_ROUND_WAVES=2 MANTA_MERGE_G=8
run l6 MANTA_MSM_L=6
run l6_g4 MANTA_MSM_L=6 MANTA_MERGE_G=4
run base_again
