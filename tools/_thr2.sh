for rep in 1 2; do
echo "== exact sizes"; timeout 300 python tools/single_threads_sweep.py 2>&1 | grep -E "threads= (1|2|3|4|6)|threads=12"
echo "== MANTA_COALESCE_POW2=1"; MANTA_COALESCE_POW2=1 timeout 300 python tools/single_threads_sweep.py 2>&1 | grep -E "threads= (1|2|3|4|6)|threads=12"
done
echo "== exact, MANTA_GRAPH=off"; MANTA_GRAPH=off timeout 300 python tools/single_threads_sweep.py 2>&1 | grep -E "threads= (1|2|3|4|6)|threads=12"
