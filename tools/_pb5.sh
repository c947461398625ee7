V=$PWD/manta_rs_amd/lib/libmantagpu_g2w2.so
for rep in 1 2 3; do
echo "== base"; timeout 100 python tools/prove_profile.py 2>/dev/null | tail -1; timeout 200 python tools/batch_threads_sweep.py 1024 2>/dev/null | grep -E "K=" | tail -1
echo "== g2 accumulate at 2 waves"; MANTA_LIB=$V timeout 100 python tools/prove_profile.py 2>/dev/null | tail -1;  MANTA_LIB=$V timeout 200 python tools/batch_threads_sweep.py 1024 2>/dev/null | grep -E "K=" | tail -1
done
