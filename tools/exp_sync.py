"""What a host pays to learn that a short dependent launch chain has finished: k tiny launches + one stream synchronisation, per call;
and the same with the host busy-polling a pinned word the last kernel writes (through torch: a copy of one element into pinned memory)."""
import sys, time
import torch
dev = torch.device('cuda', 0)
x = torch.zeros(64, device=dev)
pin = torch.zeros(1).pin_memory()
flag = torch.ones(1, device=dev)
for k in (1, 10):
    for _ in range(200):
        for _ in range(k):
            x.add_(1.0)
        torch.cuda.current_stream().synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        for _ in range(k):
            x.add_(1.0)
        torch.cuda.current_stream().synchronize()
    print("%2d launches + stream sync: %.1f us per call" % (k, (time.perf_counter() - t0) / n * 1e6))
    t0 = time.perf_counter()
    for i in range(n):
        for _ in range(k):
            x.add_(1.0)
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
    print("%2d launches + event query spin: %.1f us per call" % (k, (time.perf_counter() - t0) / n * 1e6))
