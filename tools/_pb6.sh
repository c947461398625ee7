for rep in 1 2; do for w in 0 -1; do
echo "== ROUND_WAVES=$w"
MANTA_ACC_ROUND_WAVES=$w python bench.py --workload prove --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('proofs', d)
print({k:p[k].get('proofs_per_s') for k in ('sequential','two_threads','six_threads','batched') if k in p})"
done; done
