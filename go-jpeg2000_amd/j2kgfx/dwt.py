"""internal/dwt mirror (dwt.go).  `data` is a flat C-contiguous ndarray standing for the
Go slice; it is transformed in place.  `length`/`width`/`height` as in the reference."""
import ctypes as C

from .context import default_context
from .mct import _f64, _i32


def _need(data, n):
    if data.size < n:
        raise ValueError("slice shorter than width*height")  # Go: slice bounds out of range


def Forward53(data, length, ctx=None):                   # dwt.go:73-118
    ctx = ctx or default_context(); _need(data, length)
    ctx.check(ctx.L.j2k_forward53(ctx.h, _i32(data), int(length)))


def Inverse53(data, length, ctx=None):                   # dwt.go:122-147
    ctx = ctx or default_context(); _need(data, length)
    ctx.check(ctx.L.j2k_inverse53(ctx.h, _i32(data), int(length)))


def Forward97(data, length, ctx=None):                   # dwt.go:161-210
    ctx = ctx or default_context(); _need(data, length)
    ctx.check(ctx.L.j2k_forward97(ctx.h, _f64(data), int(length)))


def Inverse97(data, length, ctx=None):                   # dwt.go:213-262
    ctx = ctx or default_context(); _need(data, length)
    ctx.check(ctx.L.j2k_inverse97(ctx.h, _f64(data), int(length)))


def Forward2D53(data, width, height, ctx=None):          # dwt.go:356-407
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_forward2d53(ctx.h, _i32(data), int(width), int(height)))


def Inverse2D53(data, width, height, ctx=None):          # dwt.go:410-429
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_inverse2d53(ctx.h, _i32(data), int(width), int(height)))


def Forward2D97(data, width, height, ctx=None):          # dwt.go:432-451
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_forward2d97(ctx.h, _f64(data), int(width), int(height)))


def Inverse2D97(data, width, height, ctx=None):          # dwt.go:454-473
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_inverse2d97(ctx.h, _f64(data), int(width), int(height)))


def DecomposeMultiLevel53(data, width, height, levels, ctx=None):     # dwt.go:524-531
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_decompose_multilevel53(ctx.h, _i32(data), int(width), int(height), int(levels)))


def ReconstructMultiLevel53(data, width, height, levels, ctx=None):   # dwt.go:534-548
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_reconstruct_multilevel53(ctx.h, _i32(data), int(width), int(height), int(levels)))


def DecomposeMultiLevel97(data, width, height, levels, ctx=None):     # dwt.go:551-558
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_decompose_multilevel97(ctx.h, _f64(data), int(width), int(height), int(levels)))


def ReconstructMultiLevel97(data, width, height, levels, ctx=None):   # dwt.go:561-573
    ctx = ctx or default_context(); _need(data, width * height)
    ctx.check(ctx.L.j2k_reconstruct_multilevel97(ctx.h, _f64(data), int(width), int(height), int(levels)))
