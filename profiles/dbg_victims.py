"""Debug: which launches stop being reproducible beside the 128 px x 128 ch conv tile (HDRSKY_TILE_WIDE=2,4,4,2,32,1)?"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0"); torch.manual_seed(0)
side = torch.cuda.Stream()
xn = torch.randn(16, 64, 256, 64, device=dev); pwn = K.PackedConv(torch.randn(4, 4, 64, 128, device=dev) * 0.03, False); bn = torch.zeros(128, device=dev)
def neighbour(n=6):
    with torch.cuda.stream(side):
        for _ in range(n): K.conv2d(xn, pwn, bn, stride=2)
B, H, W = 8, 128, 512
victims = {}
x = torch.randn(B, H // 2, W // 2, 64, device=dev); pw = K.PackedConv(torch.randn(3, 3, 64, 64, device=dev) / 24, False); b0 = torch.zeros(64, device=dev)
victims["conv 3x3 64->64 @64x256 (128 px x 64 ch tile)"] = lambda: K.conv2d(x, pw, b0)[0]
x32 = torch.randn(B, H, W, 32, device=dev); pw7 = K.PackedConv(torch.randn(7, 7, 32, 3, device=dev) / 40, False); b3 = torch.zeros(3, device=dev)
victims["conv 7x7 32->3 @128x512 (512 px x 16 ch direct-B)"] = lambda: K.conv2d(x32, pw7, b3)[0]
offs = K.da_offsets_device(H // 4, W // 4, 3, 1, True, dev)
xd = torch.randn(B, H // 4, W // 4, 128, device=dev); pwd = K.PackedConv(torch.randn(3, 3, 128, 128, device=dev) / 34, False); bd = torch.zeros(128, device=dev)
victims["da_conv2d forward 3x3 128->128 @32x128"] = lambda: K.da_conv2d(xd, pwd, bd, offs, K.BF16)[0] if isinstance(K.da_conv2d(xd, pwd, bd, offs, K.BF16), tuple) else K.da_conv2d(xd, pwd, bd, offs, K.BF16)
table = K.da_transpose_table(H, W, 3, 1, True, dev); dd2 = torch.randn(B, H, W, 32, device=dev)
pwT = K.PackedConv(torch.randn(3, 3, 64, 32, device=dev) / 24, False, transpose_flip=True)
victims["da_conv2d_dgrad 3x3 32->64 @128x512"] = lambda: K.da_conv2d_dgrad(dd2, pwT, table, 3, K.BF16)
xb = torch.randn(32, 8, 32, 128, device=dev).to(torch.bfloat16)
rc = importlib.import_module(PKG + ".kernels")
up = torch.randn(B, H, W, 64, device=dev)
victims["up2x_bwd @128x512x64"] = lambda: K.up2x_bwd(up)
torch.cuda.synchronize()
for name, fn in victims.items():
    out = []
    for contend in (False, True):
        ref, nbad = None, 0
        for it in range(8):
            if contend: neighbour()
            y = fn(); torch.cuda.synchronize()
            y = y[0] if isinstance(y, tuple) else y
            if ref is None: ref = y.clone(); continue
            nbad += int(not torch.equal(ref, y))
        out.append(nbad)
    print("%-55s differing runs of 7: alone %d, beside the conv %d" % (name, out[0], out[1]), flush=True)
