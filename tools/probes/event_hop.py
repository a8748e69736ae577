"""Latency of a cross-stream dependency (event record on one stream, wait on another) vs the same kernels on one stream."""
import os, time, torch
x = torch.zeros(1024, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 2000
def one_stream():
    with torch.cuda.stream(s1):
        for _ in range(N):
            x.add_(1); x.add_(1)
def ping_pong():
    for _ in range(N):
        with torch.cuda.stream(s1):
            x.add_(1)
        e = torch.cuda.Event(); e.record(s1); s2.wait_event(e)
        with torch.cuda.stream(s2):
            x.add_(1)
        e2 = torch.cuda.Event(); e2.record(s2); s1.wait_event(e2)
for name, fn in (("one stream", one_stream), ("ping-pong", ping_pong), ("one stream", one_stream), ("ping-pong", ping_pong)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("GPU_MAX_HW_QUEUES=%s %-10s: %.1f us per pair of kernels" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), name, dt / N * 1e6))
