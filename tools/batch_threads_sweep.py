"""dev tool: batched PrivateTransfer proving throughput against host threads x proofs per call (bench.py's ProveSetup).
Calls of more than 32 proofs are streamed by the library as passes of 32, MANTA_BATCH_INFLIGHT (3) in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ps = bench.ProveSetup("private_transfer")
CASES = ((32, (1, 2, 3)), (96, (1, 2)), (256, (1, 2)), (1024, (1,)))
if len(sys.argv) > 1:
    CASES = ((int(sys.argv[1]), (2,)),)
for K, ths in CASES:
    for th in ths:
        n = max(1024, K * th * 2)
        ps.run(max(4 * K * th, 8), th, K)
        t = time.perf_counter(); ps.run(n, th, K); dt = time.perf_counter() - t
        print(f"K={K:4d} host threads={th}: {n/dt:8.1f} proofs/s", flush=True)
