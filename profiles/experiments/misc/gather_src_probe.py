import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
def gtime(fn, iters=10):
    for _ in range(2): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (B,H,W,C) in ((8,128,512,64),(8,64,256,128),(8,32,128,128),(32,8,32,128)):
    x = torch.randn(B,H,W,C,device=dev); xb = x.to(torch.bfloat16)
    offs = K.da_offsets_device(H, W, 3, device=dev)
    table = K.da_transpose_table(H, W, 3, device=dev)
    t32 = gtime(lambda: K.da_gather_bf16(x, offs, ksize=3)); t16 = gtime(lambda: K.da_gather_bf16(xb, offs, ksize=3))
    tt = gtime(lambda: K.da_gather_bf16(xb, table=table, ksize=3))
    mb = B*H*W*9*C*2/1e6
    print("%s G %.0f MB: fp32 src %.1f us (%.2f TB/s of G), bf16 src %.1f us (%.2f), bf16 src transposed table %.1f" % ((B,H,W,C), mb, t32, mb/t32/1e6*1e6/1e6, t16, mb/t16, tt))
