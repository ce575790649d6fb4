"""ctypes binding of the C oracle (oracle/libj2koracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("J2K_ORACLE_LIB") or os.path.join(_HERE, "libj2koracle.so")   # J2K_ORACLE_LIB: the sanitised build (tests/test_sanitized_host_build.py)
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_mq_encode.restype = C.c_long
        _LIB.orc_raw_encode.restype = C.c_long
        _LIB.orc_t1_encode.restype = C.c_long
        _LIB.orc_ht_encode.restype = C.c_long
        _LIB.orc_ht_bound.restype = C.c_size_t
        _LIB.orc_ht_bound.argtypes = [C.c_int, C.c_int]
        _LIB.orc_enumerate_blocks.restype = C.c_size_t
        _LIB.orc_encode_tile_blocks.restype = C.c_long
        _LIB.orc_enumerate_blocks2.restype = C.c_size_t
        _LIB.orc_encode_tile_blocks2.restype = C.c_long
    return _LIB


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _own_i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a.copy()


def _own_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64).copy()


# ---- mct ---------------------------------------------------------------------
def dc_shift_fwd(d, precision):
    d = _own_i32(d); lib().orc_dc_shift_fwd(_i32(d), C.c_size_t(d.size), int(precision)); return d


def dc_shift_inv(d, precision):
    d = _own_i32(d); lib().orc_dc_shift_inv(_i32(d), C.c_size_t(d.size), int(precision)); return d


def rct_fwd(r, g, b):
    r, g, b = _own_i32(r), _own_i32(g), _own_i32(b)
    lib().orc_rct_fwd(_i32(r), _i32(g), _i32(b), C.c_size_t(r.size)); return r, g, b


def rct_inv(y, u, v):
    y, u, v = _own_i32(y), _own_i32(u), _own_i32(v)
    lib().orc_rct_inv(_i32(y), _i32(u), _i32(v), C.c_size_t(y.size)); return y, u, v


def ict_fwd(r, g, b):
    r, g, b = _own_f64(r), _own_f64(g), _own_f64(b)
    lib().orc_ict_fwd(_f64(r), _f64(g), _f64(b), C.c_size_t(r.size)); return r, g, b


def ict_inv(y, cb, cr):
    y, cb, cr = _own_f64(y), _own_f64(cb), _own_f64(cr)
    lib().orc_ict_inv(_f64(y), _f64(cb), _f64(cr), C.c_size_t(y.size)); return y, cb, cr


# ---- dwt ---------------------------------------------------------------------
def fwd53_1d(d):
    d = _own_i32(d); lib().orc_fwd53_1d(_i32(d), int(d.size)); return d


def inv53_1d(d):
    d = _own_i32(d); lib().orc_inv53_1d(_i32(d), int(d.size)); return d


def fwd97_1d(d):
    d = _own_f64(d); lib().orc_fwd97_1d(_f64(d), int(d.size)); return d


def inv97_1d(d):
    d = _own_f64(d); lib().orc_inv97_1d(_f64(d), int(d.size)); return d


def _2d(fn, d, w, h, conv, ptr, *extra):
    d = conv(d).reshape(-1)
    assert d.size == w * h
    fn(ptr(d), int(w), int(h), *extra)
    return d.reshape(h, w)


def fwd53_2d(d, w, h): return _2d(lib().orc_fwd53_2d, d, w, h, _own_i32, _i32)
def inv53_2d(d, w, h): return _2d(lib().orc_inv53_2d, d, w, h, _own_i32, _i32)
def fwd97_2d(d, w, h): return _2d(lib().orc_fwd97_2d, d, w, h, _own_f64, _f64)
def inv97_2d(d, w, h): return _2d(lib().orc_inv97_2d, d, w, h, _own_f64, _f64)
def decompose53(d, w, h, levels): return _2d(lib().orc_decompose53, d, w, h, _own_i32, _i32, int(levels))
def reconstruct53(d, w, h, levels): return _2d(lib().orc_reconstruct53, d, w, h, _own_i32, _i32, int(levels))
def decompose97(d, w, h, levels): return _2d(lib().orc_decompose97, d, w, h, _own_f64, _f64, int(levels))
def reconstruct97(d, w, h, levels): return _2d(lib().orc_reconstruct97, d, w, h, _own_f64, _f64, int(levels))


def tcd_forward_dwt(d, w, h, levels, reversible):
    return _2d(lib().orc_tcd_forward_dwt, d, w, h, _own_i32, _i32, int(levels), int(reversible))


def tcd_inverse_dwt(d, w, h, levels, reversible):
    return _2d(lib().orc_tcd_inverse_dwt, d, w, h, _own_i32, _i32, int(levels), int(reversible))


def _plane_ptrs(planes):
    arr = (C.POINTER(C.c_int32) * len(planes))()
    for i, p in enumerate(planes):
        arr[i] = _i32(p)
    return arr


def preprocess(planes, w, h, precision, lossless, num_resolutions, quality=0):
    """encoder.preprocess on a list of C planes (each h*w int32). Returns new planes."""
    planes = [_own_i32(p).reshape(-1) for p in planes]
    lib().orc_preprocess(_plane_ptrs(planes), len(planes), int(w), int(h), int(precision),
                         int(bool(lossless)), int(num_resolutions), int(quality))
    return [p.reshape(h, w) for p in planes]


def postprocess(planes, precision, reversible, mct=True, is_signed=False):
    shape = np.asarray(planes[0]).shape
    planes = [_own_i32(p).reshape(-1) for p in planes]
    lib().orc_postprocess(_plane_ptrs(planes), len(planes), C.c_size_t(planes[0].size), int(precision),
                          int(bool(reversible)), int(bool(mct)), int(bool(is_signed)))
    return [p.reshape(shape) for p in planes]


# ---- MQ ----------------------------------------------------------------------
def extract_image_data(pix, fmt, w, h, target_precision=0):
    """encoder.extractImageData (encoder.go:79-213) on a Go Pix buffer (uint8 [h, stride])."""
    pix = np.ascontiguousarray(pix, dtype=np.uint8).reshape(h, -1)
    nc = [1, 1, 3, 3, 4, 4][fmt]
    planes = [np.zeros(h * w, dtype=np.int32) for _ in range(nc)]
    r = lib().orc_extract_image_data(int(fmt), _u8(pix), C.c_size_t(pix.shape[1]), int(w), int(h), int(target_precision),
                                     _plane_ptrs(planes))
    assert r == nc
    return [p.reshape(h, w) for p in planes]


def create_image(planes, precision, stride=None):
    """decoder.createImage (decoder.go:417-588): returns the uint8 Pix buffer [h, stride]."""
    nc = len(planes)
    h, w = np.asarray(planes[0]).shape
    bpp = (1 if nc == 1 else 4) * (2 if precision > 8 else 1)
    stride = stride or w * bpp
    planes = [_own_i32(p).reshape(-1) for p in planes]
    pix = np.zeros((h, stride), dtype=np.uint8)
    r = lib().orc_create_image(_plane_ptrs(planes), nc, int(precision), int(w), int(h), _u8(pix), C.c_size_t(stride))
    assert r == nc
    return pix


def mq_encode(ctx, dec):
    ctx = np.ascontiguousarray(ctx, dtype=np.uint8); dec = np.ascontiguousarray(dec, dtype=np.uint8)
    out = np.zeros(ctx.size * 2 + 64, dtype=np.uint8)
    n = lib().orc_mq_encode(_u8(ctx), _u8(dec), C.c_size_t(ctx.size), _u8(out), C.c_size_t(out.size))
    assert n >= 0
    return out[:n].copy()


def mq_decode(data, ctx):
    data = np.ascontiguousarray(data, dtype=np.uint8); ctx = np.ascontiguousarray(ctx, dtype=np.uint8)
    out = np.zeros(ctx.size, dtype=np.uint8)
    lib().orc_mq_decode(_u8(data), C.c_size_t(data.size), _u8(ctx), C.c_size_t(ctx.size), _u8(out))
    return out


def convert_colorspace(planes, cs, precision):
    """getColorConversion(cs)(componentData, precision) (colorspace.go); returns new planes."""
    shape = np.asarray(planes[0]).shape
    planes = [_own_i32(p).reshape(-1) for p in planes]
    lib().orc_convert_colorspace(int(cs), _plane_ptrs(planes), len(planes), C.c_size_t(planes[0].size), int(precision))
    return [p.reshape(shape) for p in planes]


def raw_encode(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(bits.size // 7 + 16, dtype=np.uint8)
    n = lib().orc_raw_encode(_u8(bits), C.c_size_t(bits.size), _u8(out), C.c_size_t(out.size))
    assert n >= 0
    return out[:n].copy()


def raw_decode(data, n):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    lib().orc_raw_decode(_u8(data), C.c_size_t(data.size), C.c_size_t(n), _u8(out))
    return out[:n]


def mq_table():
    qe = np.zeros(94, np.uint32); nm = np.zeros(94, np.uint8); nl = np.zeros(94, np.uint8)
    lib().orc_mq_table(qe.ctypes.data_as(C.POINTER(C.c_uint32)), _u8(nm), _u8(nl))
    return qe, nm, nl


def t1_luts():
    zc = np.zeros(1024, np.uint8); sc = np.zeros(256, np.uint8); sp = np.zeros(256, np.uint8)
    lib().orc_t1_luts(_u8(zc), _u8(sc), _u8(sp))
    return zc, sc, sp


# ---- T1 / HT -----------------------------------------------------------------
def t1_encode(data, w, h, band):
    """Returns (bytes ndarray (len 0 == Go nil), numBPS)."""
    d = _own_i32(data).reshape(-1); assert d.size == w * h
    out = np.zeros(w * h * 2 + 16384, dtype=np.uint8)
    nb = C.c_int(0)
    n = lib().orc_t1_encode(_i32(d), int(w), int(h), int(band), _u8(out), C.c_size_t(out.size), C.byref(nb))
    assert n >= 0
    return out[:n].copy(), nb.value


def t1_decode(data, numbps, band, w, h):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(w * h, dtype=np.int32)
    lib().orc_t1_decode(_u8(data), C.c_size_t(data.size), int(numbps), int(band), int(w), int(h), _i32(out))
    return out.reshape(h, w)


def ht_bound(w, h):
    return int(lib().orc_ht_bound(int(w), int(h)))


def ht_encode(data, w, h, band=0):
    """Returns bytes ndarray (len 0 == Go nil) or raises on the Go-panic domain."""
    d = _own_i32(data).reshape(-1); assert d.size == w * h
    out = np.zeros(ht_bound(w, h), dtype=np.uint8)
    n = lib().orc_ht_encode(_i32(d), int(w), int(h), int(band), _u8(out), C.c_size_t(out.size))
    if n < 0:
        raise ValueError("orc_ht_encode status %d" % n)
    return out[:n].copy()


def ht_decode(data, w, h, num_bitplanes=0, band=0):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(w * h, dtype=np.int32)
    r = lib().orc_ht_decode(_u8(data), C.c_size_t(data.size), int(num_bitplanes), int(band), int(w), int(h), _i32(out))
    assert r == 0
    return out.reshape(h, w)


# ---- encodeTile --------------------------------------------------------------
class Block(C.Structure):
    _fields_ = [("comp", C.c_int32), ("res", C.c_int32), ("band", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


BLOCK_DTYPE = np.dtype([("comp", "<i4"), ("res", "<i4"), ("band", "<i4"),
                        ("x0", "<i4"), ("y0", "<i4"), ("w", "<i4"), ("h", "<i4")])


def enumerate_blocks(ncomp, w, h, num_resolutions, cb_w, cb_h, windows=0):
    """windows = 0: the reference's top-left windows (encoder.go:597-673); 1: the closed-loop mode's Mallat rectangles
    (this library's, not the reference's -- see orc_enumerate_blocks2)."""
    n = lib().orc_enumerate_blocks2(int(ncomp), int(w), int(h), int(num_resolutions), int(cb_w), int(cb_h), int(windows),
                                    None, C.c_size_t(0))
    out = np.zeros(n, dtype=BLOCK_DTYPE)
    lib().orc_enumerate_blocks2(int(ncomp), int(w), int(h), int(num_resolutions), int(cb_w), int(cb_h), int(windows),
                                out.ctypes.data_as(C.POINTER(Block)), C.c_size_t(n))
    return out


def encode_tile_blocks(planes, w, h, num_resolutions, cb_w, cb_h, coder, windows=0):
    """Sequential encodeTile body. Returns (bytes, lens[u32], numbps[u8])."""
    planes = [_own_i32(p).reshape(-1) for p in planes]
    jobs = enumerate_blocks(len(planes), w, h, num_resolutions, cb_w, cb_h, windows)
    nj = len(jobs)
    cap = 0
    for b in jobs:
        bw, bh = int(b["w"]), int(b["h"])
        cap += max(ht_bound(bw, bh), bw * bh * 2 + 16384)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    lens = np.zeros(max(nj, 1), dtype=np.uint32); nbps = np.zeros(max(nj, 1), dtype=np.uint8)
    n = lib().orc_encode_tile_blocks2(_plane_ptrs(planes), len(planes), int(w), int(h), int(num_resolutions),
                                      int(cb_w), int(cb_h), int(coder), int(windows), _u8(out), C.c_size_t(out.size),
                                      lens.ctypes.data_as(C.POINTER(C.c_uint32)), _u8(nbps))
    if n < 0:
        raise ValueError("orc_encode_tile_blocks status %d" % n)
    return out[:n].copy(), lens[:nj].copy(), nbps[:nj].copy()


def decode_tile_blocks(data, lens, numbps, ncomp, w, h, num_resolutions, cb_w, cb_h, coder, windows=1):
    """The decode body decoder.decodeTile leaves out (decoder.go:375-411): tcd.DecodeCodeBlock (tcd.go:393-413) for every job
    of the list, each block put back at its window of the zeroed component planes.  Returns the planes [(h, w) int32]."""
    data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32); numbps = np.ascontiguousarray(numbps, dtype=np.uint8)
    planes = [np.zeros(h * w, dtype=np.int32) for _ in range(ncomp)]
    pad = np.concatenate([data, np.zeros(8, np.uint8)])
    r = lib().orc_decode_tile_blocks(_u8(pad), lens.ctypes.data_as(C.POINTER(C.c_uint32)), _u8(numbps), int(ncomp), int(w), int(h),
                                     int(num_resolutions), int(cb_w), int(cb_h), int(coder), int(windows), _plane_ptrs(planes))
    if r != 0:
        raise ValueError("orc_decode_tile_blocks status %d" % r)
    return [p.reshape(h, w) for p in planes]


def create_tile_header(tile_idx, tile_data):
    """encoder.createTileHeader (encoder.go:746-760)"""
    d = np.frombuffer(bytes(tile_data), dtype=np.uint8)
    out = np.zeros(14 + d.size, np.uint8)
    lib().orc_create_tile_header.restype = C.c_size_t
    n = lib().orc_create_tile_header(int(tile_idx), _u8(d) if d.size else None, C.c_size_t(d.size), _u8(out))
    return out[:n].tobytes()
