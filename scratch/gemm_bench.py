import sys, torch
sys.path.insert(0, '.')
from graphsage_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
M, N, K, NT = 13100, 256, 602, 232965
ld = 608
table = torch.randn(NT, ld, device=dev)[:, :K]
agg = torch.randn(13312, ld, device=dev)[:, :K]
rows = torch.randint(0, NT, (13312,), device=dev)
wl = torch.randn(N, ld, device=dev).mul_(K ** -0.5)[:, :K]
wr = torch.randn(N, ld, device=dev).mul_(K ** -0.5)[:, :K]
out = torch.empty(13312, N, device=dev)
mdev = torch.tensor([M], device=dev)
dz = torch.randn(13312, N, device=dev)
dwl = torch.zeros(N, ld, device=dev)[:, :K]
dwr = torch.zeros(N, ld, device=dev)[:, :K]

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

fwd = lambda: ops.linear_concat_fwd(agg, wl, table, wr, m=13312, m_dev=mdev, rows2=rows, out=out)
t = timeit(fwd); fl = 2 * M * 2 * K * N
print(f"fwd concat      {t:8.1f} us  {fl / t / 1e6:6.1f} TF/s")
xg = table[rows[:M]].contiguous(); ac = agg[:M].contiguous()
cat = torch.cat([ac, xg], 1); wcat = torch.cat([wl, wr], 1).contiguous()
ref = cat @ wcat.t()
print("fwd max err", float((out[:M] - ref).abs().max()))
t = timeit(lambda: torch.matmul(cat, wcat.t())); print(f"rocBLAS fwd     {t:8.1f} us  {fl / t / 1e6:6.1f} TF/s")
dw = lambda: ops.linear_dweight2(dz, agg, dwl, table, dwr, m=13312, m_dev=mdev, rows2=rows)
t = timeit(dw); print(f"dweight2        {t:8.1f} us  {fl / t / 1e6:6.1f} TF/s")
dwl.zero_(); dwr.zero_(); dw()
refl = dz[:M].t() @ ac; refr = dz[:M].t() @ xg
print("dw max err", float((dwl - refl).abs().max()), float((dwr - refr).abs().max()), float(refl.abs().max()))
t = timeit(lambda: torch.matmul(dz[:M].t(), cat)); print(f"rocBLAS dW      {t:8.1f} us  {fl / t / 1e6:6.1f} TF/s")
