"""LDS canary: runs each suspect op beside workgroups that watch their own LDS for foreign writes."""
import sys, importlib, ctypes, torch
sys.path.insert(0, "/root/repo")
import hdrsky_amd as hs
params, synth, trainer, K, L = (importlib.import_module(hs.__name__ + "." + m) for m in ("params", "synth", "trainer", "kernels", "_lib"))
dev = torch.device("cuda:0")
lib = L.load()
lib.hdrsky_debug_lds_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = synth.make_batch(B, seed=1234)
tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=(B == 2), compute=K.BF16X3 if B == 2 else K.BF16)
ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
tr._bind(ldr, hdr, gt)
names = [n for n, *_ in tr._segs if n != "apply"]
tr._execute(names); torch.cuda.synchronize()
sc, sw = torch.cuda.Stream(), torch.cuda.Stream()
for target in names:
    fn = [f for n, si, d, f in tr._segs if n == target][0]
    if fn is None: continue
    rep = torch.zeros(64, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(sc):
        lib.hdrsky_debug_lds_canary(4096, 3000, rep.data_ptr(), torch.cuda.current_stream().cuda_stream)
    with torch.cuda.stream(sw):
        for _ in range(5):
            fn()
            if target.startswith("bwd") or target == "disc_step": tr._take_wgrads()
    torch.cuda.synchronize()
    r = rep.tolist()
    print("%-12s foreign LDS writes seen: %d %s" % (target, r[0], [(r[1 + 2 * i], hex(r[2 + 2 * i] & 0xffffffff)) for i in range(min(r[0], 4))]), flush=True)
