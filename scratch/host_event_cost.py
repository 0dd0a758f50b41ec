"""Host cost of the cross-stream primitives: torch wait_stream, torch Event.record + wait_event,
hipEventRecord + hipStreamWaitEvent through ctypes."""
import sys, time, torch, ctypes
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.current_stream()
n = 5000
def t(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print('torch wait_stream               : %.2f us' % t(lambda: s1.wait_stream(s2)))
ev = torch.cuda.Event()
ev.record(s2)
print('torch ev.record(stream)         : %.2f us' % t(lambda: ev.record(s2)))
print('torch stream.wait_event(ev)     : %.2f us' % t(lambda: s1.wait_event(ev)))
hip = ctypes.CDLL('libamdhip64.so')
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
e, a, b = ctypes.c_void_p(ev.cuda_event), ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
print('hipEventRecord via ctypes       : %.2f us' % t(lambda: hip.hipEventRecord(e, b)))
print('hipStreamWaitEvent via ctypes   : %.2f us' % t(lambda: hip.hipStreamWaitEvent(a, e, 0)))
