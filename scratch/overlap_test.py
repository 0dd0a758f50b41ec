"""Do an HBM-bound gather and an MFMA-bound GEMM overlap when launched on two streams?"""
import sys, torch
sys.path.insert(0, '.')
from graphsage_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
M, N, K, NT = 13100, 256, 602, 232965
ld = 608
table = torch.randn(NT, ld, device=dev)[:, :K]
agg = torch.randn(13312, ld, device=dev)[:, :K]
agg2 = torch.randn(13312, ld, device=dev)[:, :K]
rows = torch.randint(0, NT, (13312,), device=dev)
wl = torch.randn(N, ld, device=dev).mul_(K ** -0.5)[:, :K]
wr = torch.randn(N, ld, device=dev).mul_(K ** -0.5)[:, :K]
out = torch.empty(13312, N, device=dev)
mdev = torch.tensor([M], device=dev)
dz = torch.randn(13312, N, device=dev)
dwl = torch.zeros(N, ld, device=dev)[:, :K]
dwr = torch.zeros(N, ld, device=dev)[:, :K]
# gather problem: 13100 targets x 10 neighbours out of 81000 batch nodes
nb = 81000
node = torch.randperm(NT, device=dev)[:nb].contiguous()
deg = 10
seg = (torch.arange(13313, device=dev) * deg).clamp(max=M * deg).contiguous()
src = torch.randint(0, nb, (13312 * deg,), device=dev)

fwd = lambda: ops.linear_concat_fwd(agg, wl, table, wr, m=13312, m_dev=mdev, rows2=rows, out=out)
dw = lambda: ops.linear_dweight2(dz, agg, dwl, table, dwr, m=13312, m_dev=mdev, rows2=rows)
import ctypes
_region = 2 * N * ld
_ws = torch.zeros(32 * _region, device=dev)
_v1, _v2 = _ws[:N * ld].view(N, ld)[:, :K], _ws[N * ld:_region].view(N, ld)[:, :K]
_holder = ctypes.c_int(0)
dwp = lambda: ops.linear_dweight2_parts(dz, agg, _v1, table, _v2, _region, 32, _holder, m=13312, m_dev=mdev, rows2=rows)
gat = lambda: ops.gather_reduce_fwd(table, seg, src, 13312, 'mean', node_map=node, num_targets_dev=mdev,
                                    out=agg2, want_argmax=False)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for prio in (None,):
    if prio is None:
        main, side = torch.cuda.Stream(), torch.cuda.Stream()
    elif prio == 'side_low':
        main, side = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=1)
    else:
        main, side = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
    print("priorities", prio, main.priority, side.priority)
    for name, g in (("fwd", fwd), ("dweight2", dw), ("dw slices", dwp)):
        with torch.cuda.stream(main):
            tg = timeit(g); tq = timeit(gat)
            def serial():
                g(); gat()
            ts = timeit(serial)
            def both():
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    gat()
                g()
                main.wait_stream(side)
            def both2():
                side.wait_stream(main)
                g()
                with torch.cuda.stream(side):
                    gat()
                main.wait_stream(side)
            tb = timeit(both); tb2 = timeit(both2)
        print(f"{name:9s} alone {tg:6.1f}  gather alone {tq:6.1f}  serial {ts:6.1f}  overlapped(gather first) {tb:6.1f}  (gemm first) {tb2:6.1f}")
