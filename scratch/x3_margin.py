"""How far inside its bound test_linear_concat_fwd_bf16x3_mode / test_linear_dweight_bf16x3_mode run:
max error against float64 of the bf16x3 mode and of the exact-fp32 path on the same inputs."""
import os, sys
import torch
sys.path.insert(0, '.')
from graphsage_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
def pad16(x):
    ld = (x.size(1) + 15) // 16 * 16
    buf = torch.zeros((x.size(0), ld), device=dev); buf[:, :x.size(1)] = x.to(dev); return buf[:, :x.size(1)]
for (m, n, k) in [(13100, 256, 602), (26000, 200, 130), (9000, 256, 256)]:
    a, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) * k ** -0.5
    want = a.double() @ w.double().t()
    errs = {}
    for mode in ("0", "1"):
        os.environ["SAGE_GEMM_X3"] = mode
        for rep in range(3):
            out = torch.zeros((m, n), device=dev)
            ops.linear_concat_fwd(pad16(a), pad16(w), None, None, m=m, out=out, accumulate=True)
            errs.setdefault(mode, []).append(float((out.double().cpu() - want).abs().max()))
    print("fwd", (m, n, k), "f32", ["%.2e" % e for e in errs["0"]], "x3", ["%.2e" % e for e in errs["1"]])
    d = torch.randn(m, n, generator=g) * 1e-2
    want = d.double().t() @ a.double()
    errs = {}
    for mode in ("0", "1"):
        os.environ["SAGE_GEMM_X3"] = mode
        for rep in range(3):
            dw = torch.zeros((n, k), device=dev)
            ops.linear_dweight(pad16(d), pad16(a), m=m, dw=dw)
            errs.setdefault(mode, []).append(float((dw.double().cpu() - want).abs().max()))
    print("dw ", (m, n, k), "f32", ["%.2e" % e for e in errs["0"]], "x3", ["%.2e" % e for e in errs["1"]])
