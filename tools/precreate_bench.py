"""dev tool (round 5): bench.py in a process where K other normal-priority streams exist before the library creates its own (another
library's streams: torch here) -- the order of stream creation decides which hardware queues streams share.
usage: python tools/precreate_bench.py K [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
k = int(sys.argv[1])
keep = [torch.cuda.Stream() for _ in range(k)]
for s in keep:
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda").add_(1)
torch.cuda.synchronize()
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
