"""Times the contractions of the products-scale unsupervised step in isolation."""
import torch, os, sys
sys.path.insert(0, '.')
from graphsage_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
N, F, H, M, B = 2_400_000, 100, 256, 13312, 512
x = torch.randn(N, F, device=dev)
agg = torch.randn(M, F, device=dev)
wl = torch.randn(H, F, device=dev) * 0.1
wr = torch.randn(H, F, device=dev) * 0.1
rows = torch.randint(0, N, (M,), device=dev)
rows_seq = torch.arange(M, device=dev)
z = torch.zeros(M, H, device=dev)
mdev = torch.tensor([M], device=dev)

def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print('fwd K=100+100 rows random  accum : %.1f us' % t(lambda: ops.linear_concat_fwd(agg, wl, x, wr, m=M, m_dev=mdev, rows2=rows, out=z, accumulate=True)))
print('fwd K=100+100 rows random  store : %.1f us' % t(lambda: ops.linear_concat_fwd(agg, wl, x, wr, m=M, m_dev=mdev, rows2=rows, out=z)))
print('fwd K=100+100 rows seq     accum : %.1f us' % t(lambda: ops.linear_concat_fwd(agg, wl, x, wr, m=M, m_dev=mdev, rows2=rows_seq, out=z, accumulate=True)))
print('fwd K=100+100 no rows      accum : %.1f us' % t(lambda: ops.linear_concat_fwd(agg, wl, x, wr, m=M, m_dev=mdev, out=z, accumulate=True)))
print('fwd K=100 only             accum : %.1f us' % t(lambda: ops.linear_concat_fwd(agg, wl, m=M, m_dev=mdev, out=z, accumulate=True)))
# layer 1 of the step: 512 rows, K = 256 + 256, N = 256
h = torch.randn(M, H, device=dev); a1 = torch.randn(B, H, device=dev)
w1l = torch.randn(H, H, device=dev) * 0.05; w1r = torch.randn(H, H, device=dev) * 0.05
z1 = torch.zeros(B, H, device=dev); bdev = torch.tensor([B], device=dev)
print('layer-1 fwd 512x(256+256)x256    : %.1f us' % t(lambda: ops.linear_concat_fwd(a1, w1l, h, w1r, m=B, m_dev=bdev, out=z1)))
dz = torch.randn(B, H, device=dev); o1 = torch.zeros(M, H, device=dev); o2 = torch.zeros(B, H, device=dev)
print('layer-1 dinput2 512x256x256 (x2) : %.1f us' % t(lambda: ops.linear_dinput2(dz, w1r, o1, w1l, o2, m=B, m_dev=bdev)))
g1 = torch.zeros(H, H, device=dev); g2 = torch.zeros(H, H, device=dev)
print('layer-1 dweight2 256x(256|256)   : %.1f us' % t(lambda: ops.linear_dweight2(dz, a1, g1, h, g2, m=B, m_dev=bdev)))
dz0 = torch.randn(M, H, device=dev); g0l = torch.zeros(H, F, device=dev); g0r = torch.zeros(H, F, device=dev)
print('layer-0 dweight2 rows random     : %.1f us' % t(lambda: ops.linear_dweight2(dz0, agg, g0l, x, g0r, m=M, m_dev=mdev, rows2=rows)))
print('layer-0 dweight2 rows seq        : %.1f us' % t(lambda: ops.linear_dweight2(dz0, agg, g0l, x, g0r, m=M, m_dev=mdev, rows2=rows_seq)))
