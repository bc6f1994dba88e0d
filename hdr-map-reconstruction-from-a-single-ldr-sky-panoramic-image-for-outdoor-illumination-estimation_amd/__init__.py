"""MI355X-native LDR->HDR sky-panorama reconstruction hot path (generator / sun-pose /
sun-radiance forward + Grad-CAM, distortion-aware conv, training step) on libhdrsky.so.

Host code is Python (like the reference); every FLOP of the path runs in hand-written HIP
kernels for gfx950 behind the C ABI in include/hdrsky.h.  Sub-modules:
  params   parameter inventories / Keras-equivalent initialisers (host numpy)
  synth    seeded synthetic batches shaped like the reference's training data
  _lib     ctypes binding of libhdrsky.so (no fallback: fails loudly when missing)
  kernels  torch-tensor front end of each C entry point
  engine   fused execution plan of the generator graph
"""
PACKAGE = __name__
