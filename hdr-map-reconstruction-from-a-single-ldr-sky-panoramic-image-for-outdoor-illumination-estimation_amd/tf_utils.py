"""Mirror of the live `tf_utils` helpers on libhdrsky (tf_utils.py:61-73, 85-93, 263-280)."""
import torch

from . import kernels as K

_S1 = (1.2262735, 1.5450078, 1.9465878, 2.452547)      # tf_utils.py:68
_S2 = (1.5450078, 1.9465878, 2.452547, 3.0900156)      # tf_utils.py:69


def hdr_logCompression(x, validDR=10.):
    """log(1 + validDR*x) / log(1 + validDR)  (tf_utils.py:263-271)."""
    if float(validDR) != 10.0:
        raise ValueError("only validDR = 10 (the reference's only call value) is built")
    return K.tonemap(x, False)


def hdr_logDecompression(x, validDR=10.):
    """(exp(x*log(1 + validDR)) - 1) / validDR  (tf_utils.py:273-280)."""
    if float(validDR) != 10.0:
        raise ValueError("only validDR = 10 (the reference's only call value) is built")
    return K.tonemap(x, True)


def rgb2bgr(x):
    """Channel reversal (tf_utils.py:85-88): one hdrsky_flip_rgb launch.  (The CLIs reverse the channels while staging
    the decoded image on the host, inference.py:load_ldr; the training step's inputs are BGR already.)"""
    return K.flip_rgb(x.contiguous())


bgr2rgb = rgb2bgr   # tf_utils.py:90-93


def DoG(img, kernel_size=3, sigma=1.2489996, num_intervals=3, assumed_blur=0.5, image_border_width=5):
    """Difference-of-Gaussian pyramid (tf_utils.py:61-73): 2x bilinear upsample, 3x3 Gaussian base (REFLECT), then
    four differences of 3x3 Gaussians of the base.  Returns the four [B,2H,2W,C] tensors.  (The training step uses
    the fused `dog_loss`, which never materialises them.)"""
    if kernel_size != 3:
        raise ValueError("only the 3x3 filter of the reference is built")
    base = K.blur3(K.up2x(img), float(sigma))
    blurred = {s: K.blur3(base, s) for s in sorted(set(_S1 + _S2))}
    return tuple(K.axpby(blurred[b], 1.0, blurred[a], -1.0) for a, b in zip(_S1, _S2))
