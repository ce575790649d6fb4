"""Pixel unpack / pack at native width -- mirror of the host loops on either side of the path:

    extract_image_data  <- encoder.extractImageData  (encoder.go:79-213, + the Options.Precision rescale)
    create_image        <- decoder.createImage       (decoder.go:417-588)

Pixel buffers are Go image.* `Pix` layouts: numpy uint8 [h, stride] (16-bit samples big-endian)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PIX_GRAY8, PIX_GRAY16, PIX_RGBA8, PIX_RGBA64, PIX_NRGBA8, PIX_NRGBA64  # noqa: F401
from .context import default_context

_BPP = {PIX_GRAY8: 1, PIX_GRAY16: 2, PIX_RGBA8: 4, PIX_RGBA64: 8, PIX_NRGBA8: 4, PIX_NRGBA64: 8}


def components(fmt):
    return int(_lib.lib().j2k_pixels_components(int(fmt)))


def precision(fmt):
    return int(_lib.lib().j2k_pixels_precision(int(fmt)))


def extract_image_data(pix, fmt, w, h, target_precision=0, ctx=None):
    """pix: uint8 array of h rows x stride bytes.  Returns [component planes] (int32, h x w) like e.componentData."""
    ctx = ctx or default_context()
    pix = np.ascontiguousarray(pix, dtype=np.uint8).reshape(h, -1) if h else np.zeros((0, 0), np.uint8)
    stride = pix.shape[1] if h else w * _BPP[fmt]
    nc = components(fmt)
    planes = [np.zeros((h, w), dtype=np.int32) for _ in range(nc)]
    arr = (C.c_void_p * nc)(*[p.ctypes.data for p in planes])
    ctx.check(ctx.L.j2k_extract_image_data(ctx.h, int(fmt), pix.ctypes.data_as(C.c_void_p), C.c_size_t(stride), int(w), int(h),
                                           int(target_precision), arr))
    return planes


def create_image(planes, prec, stride=None, ctx=None):
    """planes: list of int32 [h, w] arrays.  Returns the uint8 Pix buffer [h, stride] createImage would fill."""
    ctx = ctx or default_context()
    nc = len(planes)
    h, w = planes[0].shape
    bpp = (1 if nc == 1 else 4) * (2 if prec > 8 else 1)
    stride = stride or w * bpp
    planes = [np.ascontiguousarray(p, dtype=np.int32) for p in planes]
    pix = np.zeros((h, stride), dtype=np.uint8)
    arr = (C.c_void_p * nc)(*[p.ctypes.data for p in planes])
    ctx.check(ctx.L.j2k_create_image(ctx.h, arr, nc, int(prec), int(w), int(h), pix.ctypes.data_as(C.c_void_p), C.c_size_t(stride)))
    return pix
