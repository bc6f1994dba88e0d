"""Probe: external event record / wait nodes inside captured hipGraphs (hipEventRecordWithFlags(hipEventRecordExternal),
hipStreamWaitEvent(hipEventWaitExternal)) through ctypes - torch 2.10+rocm refuses torch.cuda.Event(external=True).  Would let each
stream's whole chain of the training step be ONE graph (a graph boundary costs 13-19 us, a cross-stream hand-off ~22 us)."""
import ctypes, time, torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = ctypes.CDLL(path)
print(path, hasattr(hip, "hipEventRecordWithFlags"))
a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 22, device=dev); y = torch.zeros(1 << 22, device=dev)
ev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 2) == 0          # hipEventDisableTiming
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(ga, stream=a, capture_error_mode="thread_local"):
    for _ in range(20): x.add_(1.0)
    rc = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(a.cuda_stream), 1)
    print("record external rc", rc)
    for _ in range(200): x.add_(1.0)
with torch.cuda.graph(gb, stream=b, capture_error_mode="thread_local"):
    rc = hip.hipStreamWaitEvent(ctypes.c_void_p(b.cuda_stream), ev, 1)
    print("wait external rc", rc)
    y.copy_(x)
torch.cuda.synchronize()
for it in range(4):
    x.zero_(); y.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(a): ga.replay()
    with torch.cuda.stream(b): gb.replay()
    torch.cuda.synchronize()
    print("iter", it, "x", float(x[0]), "y (20 <= y < 220 if the wait node orders it behind the record node)", float(y[0]))
# order reversed at enqueue: b first - the wait node must still see THIS replay's record
for it in range(2):
    x.zero_(); y.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(b): gb.replay()
    time.sleep(0.01)
    with torch.cuda.stream(a): ga.replay()
    torch.cuda.synchronize()
    print("reversed", it, "x", float(x[0]), "y", float(y[0]))
