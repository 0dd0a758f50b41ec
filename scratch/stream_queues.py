"""Which of torch's pooled streams run CONCURRENTLY with the default stream (and with each other)?  Two spin
kernels, one per stream: elapsed ~ one kernel = separate hardware queues, ~ two = serialised."""
import time
import torch
dev = torch.device("cuda", 0)
main = torch.cuda.current_stream()
def spin(n):
    torch.cuda._sleep(n)
def pair(a, b, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        spin(n)
    with torch.cuda.stream(b):
        spin(n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
n = 1000000
spin(n); torch.cuda.synchronize()
t0 = time.perf_counter(); spin(n); torch.cuda.synchronize(); one = time.perf_counter() - t0
print("one spin kernel: %.3f ms" % (one * 1e3), flush=True)
streams = [torch.cuda.Stream(device=dev) for _ in range(20)]
for i, s in enumerate(streams):
    r = pair(main, s, n) / one
    r2 = pair(streams[i - 1], s, n) / one if i else 0.0
    print("stream %2d handle %x: with main %.2f, with the previous one %.2f" % (i, s.cuda_stream, r, r2), flush=True)
