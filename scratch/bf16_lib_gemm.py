"""What does the library bf16 GEMM (hipBLASLt / rocBLAS behind torch.mm) reach on the lstm / pool shapes?
Timing only: decides whether six plane products through the library beat the fused bf16x3 kernels."""
import time
import torch
dev = torch.device("cuda", 0)


def bench(m, n, k, reps=10):
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c = a @ b.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"NT m={m} n={n} k={k}: {dt*1e3:.3f} ms = {2*m*n*k/dt/1e12:.0f} TF/s", flush=True)
    # TN (weight gradient): [n, m] x [m, k]
    d = torch.randn(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        w = d.t() @ a
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        w = d.t() @ a
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"TN n={n} k={k} over m={m}: {dt*1e3:.3f} ms = {2*m*n*k/dt/1e12:.0f} TF/s", flush=True)


bench(12400, 1806, 15050)
bench(12400, 1808, 15056)
bench(81000, 602, 602)
bench(81000, 608, 608)
bench(13100, 256, 1204)
bench(13100, 256, 1216)
