"""Decode-side colour conversions to sRGB -- mirror of colorspace.go (getColorConversion and the convert*ToRGB it
selects, colorspace.go:54-480).  The constants are the reference's ColorSpace values (jpeg2000.go:124-197)."""
import ctypes as C

import numpy as np

from .context import default_context

(ColorSpaceUnknown, ColorSpaceUnspecified, ColorSpaceSRGB, ColorSpaceGray, ColorSpaceSYCC, ColorSpaceEYCC, ColorSpaceCMYK,
 ColorSpaceBilevel, ColorSpaceYCbCr2, ColorSpaceYCbCr3, ColorSpacePhotoYCC, ColorSpaceCMY, ColorSpaceYCCK, ColorSpaceCIELab,
 ColorSpaceCIEJab, ColorSpaceESRGB, ColorSpaceROMMRGB, ColorSpaceYPbPr60, ColorSpaceYPbPr50) = range(-1, 18)


def convert(componentData, cs, precision, ctx=None):
    """In place on a list of int32 numpy planes, like getColorConversion(cs)(componentData, precision)."""
    ctx = ctx or default_context()
    nc = len(componentData)
    if nc == 0:
        return componentData
    for p in componentData:
        assert p.dtype == np.int32 and p.flags["C_CONTIGUOUS"]
    arr = (C.c_void_p * nc)(*[p.ctypes.data for p in componentData])
    ctx.check(ctx.L.j2k_convert_colorspace(ctx.h, int(cs), arr, nc, C.c_size_t(componentData[0].size), int(precision)))
    return componentData
