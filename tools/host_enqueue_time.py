"""dev tool: host time to ENQUEUE one bench frame (no synchronisation) against the GPU time of the same frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
seq = bench.Sequence(cfg, dev)
seq.fuse_first = False
if seq.D <= 2048:
    seq.enable_pcg_shadow()
for f in range(20):
    seq.frame(f)
torch.cuda.synchronize()
for n in (20, 50, 100):
    t0 = time.perf_counter()
    for f in range(n):
        seq.frame(20 + f)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%3d frames: host enqueue %.1f us/frame, until the GPU is done %.1f us/frame" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# where the host time goes
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for f in range(100):
    seq.frame(200 + f)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
