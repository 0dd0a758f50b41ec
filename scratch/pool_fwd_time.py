"""Pool Linear at the Reddit shape (81k rows x 602 -> 602): row-mapped random table rows (what the
engine does) against the first 81k rows of the table read in place, and against sorted row ids."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphsage_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N, F, m = 232965, 602, 81400
x = ops.empty_rows(N, F, dev); x.copy_(torch.randn(N, F, generator=g, device=dev))
w = ops.empty_rows(F, F, dev); w.copy_(torch.randn(F, F, generator=g, device=dev) * 0.05)
b = torch.randn(F, generator=g, device=dev)
out = ops.empty_rows(m, F, dev); dpre = ops.empty_rows(m, F, dev)
rows = torch.randperm(N, generator=g, device=dev)[:m].contiguous()
cases = {"random rows": rows, "sorted rows": rows.sort().values.contiguous(), "in place": None}
for name, r in cases.items():
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.pool_linear_fwd(x, w, b, m=m, rows=r, out=out, dpre=dpre)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(name, " ".join("%.0f" % t for t in ts[2:]))
dout = ops.empty_rows(m, F, dev); dout.copy_(torch.randn(m, F, generator=g, device=dev))
dw = torch.zeros(F, 608, device=dev)[:, :F]
for name, r in cases.items():
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear_dweight(dout, x, m=m, rows=r, dw=dw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("dW", name, " ".join("%.0f" % t for t in ts[2:]))
