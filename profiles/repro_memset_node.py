"""Minimal reproduction of the hipGraph memset-node problem hdrsky_zero ran into (ROCm 7.2, gfx950): a graph = one
hipMemsetAsync(0) over n floats + one add of 1.0, replayed four times - every replay must leave exactly 1.0 everywhere.
Observed: correct for n = 1 (4 bytes) and n = 2^20 (4 MB); for n = 7 and n = 4096 the FIRST replay is correct and every later
one leaves inf in part of the buffer, on the capturing stream and on a side stream alike.  The library therefore clears its
accumulators with a kernel.      usage: python profiles/repro_memset_node.py"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
for n in (1, 7, 4096, 1 << 20):
    for side in (False, True):
        x = torch.ones(n, device=dev)
        s = torch.cuda.Stream() if side else None
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            rc = hip.hipMemsetAsync(x.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
            x.add_(1.0)
        vals = []
        for it in range(4):
            g.replay(); torch.cuda.synchronize(); vals.append((float(x.min()), float(x.max())))
        print("n=%d side_stream=%s rc=%d ->" % (n, side, rc), vals)
