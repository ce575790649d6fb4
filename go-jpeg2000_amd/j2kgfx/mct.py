"""internal/mct mirror (mct.go).  Arrays are modified in place, as the Go slices are."""
import ctypes as C

import numpy as np

from .context import default_context


def _i32(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous):
        raise TypeError("expected a C-contiguous int32 ndarray (stands for []int32)")
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
        raise TypeError("expected a C-contiguous float64 ndarray (stands for []float64)")
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _same_len(*arrs):
    n = arrs[0].size
    for a in arrs:
        if a.size != n:
            raise ValueError("slices of different length")  # Go would panic (index out of range)
    return n


def DCLevelShiftForward(data, precision, ctx=None):      # mct.go:96-101
    ctx = ctx or default_context()
    ctx.check(ctx.L.j2k_dc_level_shift_forward(ctx.h, _i32(data), C.c_size_t(data.size), int(precision)))


def DCLevelShiftInverse(data, precision, ctx=None):      # mct.go:113-118
    ctx = ctx or default_context()
    ctx.check(ctx.L.j2k_dc_level_shift_inverse(ctx.h, _i32(data), C.c_size_t(data.size), int(precision)))


def ForwardRCT(r, g, b, ctx=None):                       # mct.go:28-38
    ctx = ctx or default_context()
    ctx.check(ctx.L.j2k_forward_rct(ctx.h, _i32(r), _i32(g), _i32(b), C.c_size_t(_same_len(r, g, b))))


def InverseRCT(y, u, v, ctx=None):                       # mct.go:56-66
    ctx = ctx or default_context()
    ctx.check(ctx.L.j2k_inverse_rct(ctx.h, _i32(y), _i32(u), _i32(v), C.c_size_t(_same_len(y, u, v))))


def ForwardICT(r, g, b, ctx=None):                       # mct.go:14-24
    ctx = ctx or default_context()
    ctx.check(ctx.L.j2k_forward_ict(ctx.h, _f64(r), _f64(g), _f64(b), C.c_size_t(_same_len(r, g, b))))


def InverseICT(y, cb, cr, ctx=None):                     # mct.go:43-53
    ctx = ctx or default_context()
    ctx.check(ctx.L.j2k_inverse_ict(ctx.h, _f64(y), _f64(cb), _f64(cr), C.c_size_t(_same_len(y, cb, cr))))
