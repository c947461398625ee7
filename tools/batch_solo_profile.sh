R=$PWD; cd /tmp; export TMPDIR=/tmp
for st in 1 6; do
rm -rf /tmp/ab; MANTA_PROVE_STREAMS=$st rocprofv3 --kernel-trace --stats -d /tmp/ab -o a -- python $R/tools/prove_batch_profile.py 32 6 > /tmp/ab.txt 2>/dev/null
echo "== streams=$st"; grep -E "ms per pass" /tmp/ab.txt
python $R/tools/rocprof_summary.py $(find /tmp/ab -name "*.db" | head -1) | head -32
done
