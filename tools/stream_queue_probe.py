"""Which HIP streams really run beside the current one: a long fill on the current stream, a one-element kernel on each of ten
fresh streams, twice.  First pass: a stream's hardware queue is created at its first use (milliseconds).  Second pass: the
kernel finishes in ~0.04 ms on a stream with a queue of its own and in the fill's 0.25 ms on one that shares the current
stream's queue (MI355X, ROCm 7.0: about one stream in eight).  bench.py's concurrent_stream() is built on this."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
main = torch.cuda.current_stream(dev)
big = torch.empty(96 << 20, dtype=torch.float32, device=dev)
one = torch.zeros(1, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(10)]
for rep in range(2):
    for i, cand in enumerate(streams):
        e0, e_busy, e_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        e0.record(main)
        for _ in range(4):
            big.fill_(1.0)
        e_busy.record(main)
        cand.wait_event(e0)
        with torch.cuda.stream(cand):
            one.add_(1.0)
            e_side.record(cand)
        torch.cuda.synchronize(dev)
        print(rep, i, "busy %.3f ms side %.3f ms" % (e0.elapsed_time(e_busy), e0.elapsed_time(e_side)), flush=True)
