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
for _ in range(5):
    ops.linear_concat_fwd(agg, wl, table, wr, m=13312, m_dev=mdev, rows2=rows, out=out)
    ops.linear_dweight2(dz, agg, dwl, table, dwr, m=13312, m_dev=mdev, rows2=rows)
torch.cuda.synchronize()
