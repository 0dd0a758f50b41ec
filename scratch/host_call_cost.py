"""Host cost of one C-ABI call from a tape: ctypes marshalling alone (m = 0: the entry returns before
any launch) against the same call with its launch."""
import sys, time, torch, ctypes
sys.path.insert(0, '.')
from graphsage_amd import _lib, ops
dev = torch.device('cuda:0')
dz = torch.randn(512, 256, device=dev); w = torch.randn(256, 256, device=dev)
o1 = torch.zeros(512, 256, device=dev); o2 = torch.zeros(512, 256, device=dev)
z = torch.randn(512, 256, device=dev); y = torch.empty_like(z); inv = torch.empty(512, device=dev)
for m, tag in ((0, 'no launch (m = 0)'), (512, 'with launches')):
    with _lib.recording() as tape:
        ops.linear_dinput2(dz, w, o1, w, o2, m=m)
    with _lib.recording() as tape2:
        ops.l2norm_relu_fwd(z, False, m=max(m, 1) if m else 0, y=y, inv_norm=inv)
    for t, name in ((tape, 'linear_dinput2 (15 args)'), (tape2, 'l2norm_relu_fwd')):
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            _lib.replay(t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print('%-28s %-18s %.2f us per replayed call' % (name, tag, (t1 - t0) / n * 1e6))
s1, s2 = torch.cuda.Stream(), torch.cuda.current_stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    s1.wait_stream(s2)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('torch wait_stream: %.2f us' % ((t1 - t0) / 2000 * 1e6))
