for t in 0 512 256; do echo "== MANTA_NTT_THREADS=$t"; MANTA_NTT_THREADS=$t timeout 200 python tools/_ntt.py | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(d[k]['device_ms'], d[k]['us_per_pass']) for k in ('fft','ifft','coset_fft','coset_ifft')})"; MANTA_NTT_THREADS=$t timeout 100 python tools/prove_profile.py 2>/dev/null | tail -1; MANTA_NTT_THREADS=$t timeout 200 python tools/batch_threads_sweep.py 1024 2>/dev/null | grep -E "K=" | tail -1; done
