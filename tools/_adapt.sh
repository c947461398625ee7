R=$PWD
t() { echo "== $*"; env "$@" python $R/tools/prove_profile.py 2>/dev/null | tail -1; env "$@" python $R/tools/prove_batch_profile.py 32 12 2>/dev/null | tail -1; }
for rep in 1 2; do
t MANTA_ACC_ROUND_WAVES=0
t MANTA_ACC_ROUND_WAVES=-1
t MANTA_ACC_ROUND_WAVES=-1 MANTA_MERGE_G=8
t MANTA_ACC_ROUND_WAVES=-1 MANTA_MERGE_G=12
t MANTA_ACC_ROUND_WAVES=1
t MANTA_ACC_ROUND_WAVES=2
t MANTA_ACC_ROUND_WAVES=3
done
