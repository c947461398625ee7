echo "== scan only"; timeout 300 python tools/plain_sweep.py 0 14 15 16 17 2>&1 | grep plain
echo "== MANTA_RED_S=3"; MANTA_RED_S=3 timeout 300 python tools/plain_sweep.py 14 15 16 17 18 2>&1 | grep plain
echo "== MANTA_RED_S=3 MANTA_RED_SIDE=0"; MANTA_RED_S=3 MANTA_RED_SIDE=0 timeout 300 python tools/plain_sweep.py 15 16 17 2>&1 | grep plain
echo "== MANTA_RED_S=3 MANTA_RED_MIN=4096"; MANTA_RED_S=3 MANTA_RED_MIN=4096 MANTA_RED_SIDE=0 timeout 300 python tools/plain_sweep.py 14 15 16 2>&1 | grep plain
