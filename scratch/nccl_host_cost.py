"""Host-side cost of one gradient all-reduce call on the real backend (1 rank on RCCL)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.zeros(332288, device=dev)
pg = dist.group.WORLD
def bench(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    return host / n * 1e6, tot / n * 1e6
def a():
    w = dist.all_reduce(t, async_op=True); w.wait()
def b():
    dist.all_reduce(t)
opts = dist.AllreduceOptions()
def c():
    w = pg.allreduce([t], opts); w.wait()
def d():
    t.add_(1.0)
for name, fn in (("dist.all_reduce async + wait", a), ("dist.all_reduce sync", b), ("pg.allreduce + wait", c), ("one torch op (reference)", d)):
    h, tot = bench(fn)
    print(f"{name:32s} host {h:7.1f} us/call   wall {tot:7.1f} us/call", flush=True)
dist.destroy_process_group()
