"""internal/entropy mirror: T1 (MQ block coder), HTEncoder / HTDecoder, band constants.

Object-per-block API with the reference's names (t1.go:94-133, 292, 918, 1261; ht.go:77-93,
922-942) on top of the batched C-ABI calls j2k_encode_blocks / j2k_decode_blocks, plus
encode_blocks()/decode_blocks() for whole batches (the form the codec drivers use).
"""
import ctypes as C

import numpy as np

from . import _lib
from .context import default_context

BandLL, BandHL, BandLH, BandHH = 0, 1, 2, 3      # t1.go:125-130

BLOCK_DTYPE = np.dtype([("plane", "<i4"), ("band", "<i4"), ("x0", "<i4"), ("y0", "<i4"), ("w", "<i4"), ("h", "<i4")])


def block_bound(coder, w, h):
    return int(_lib.lib().j2k_block_bound(int(coder), int(w), int(h)))


def encode_blocks(coder, planes, blocks, ctx=None):
    """planes: list of 2-D C-contiguous int32 arrays; blocks: ndarray of BLOCK_DTYPE.
    Returns (stream u8, offs u64, lens u32, numbps u8)."""
    ctx = ctx or default_context()
    planes = [np.ascontiguousarray(p, dtype=np.int32) for p in planes]
    blocks = np.ascontiguousarray(blocks, dtype=BLOCK_DTYPE)
    n = blocks.size
    pp = (C.POINTER(C.c_int32) * len(planes))(*[p.ctypes.data_as(C.POINTER(C.c_int32)) for p in planes])
    pw = np.array([p.shape[1] for p in planes], dtype=np.int32)
    ph = np.array([p.shape[0] for p in planes], dtype=np.int32)
    cap = sum(block_bound(coder, int(b["w"]), int(b["h"])) for b in blocks) + 16
    out = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(max(n, 1), dtype=np.uint64); lens = np.zeros(max(n, 1), dtype=np.uint32)
    nb = np.zeros(max(n, 1), dtype=np.uint8)
    total = C.c_size_t(0)
    ctx.check(ctx.L.j2k_encode_blocks(
        ctx.h, int(coder), pp, pw.ctypes.data_as(C.c_void_p), ph.ctypes.data_as(C.c_void_p), len(planes),
        blocks.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap),
        offs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), nb.ctypes.data_as(C.c_void_p),
        C.byref(total)))
    return out[:total.value].copy(), offs[:n], lens[:n], nb[:n]


def decode_blocks(coder, stream, offs, lens, numbps, blocks, ctx=None):
    """Returns a list of 2-D int32 arrays (h, w), one per block."""
    ctx = ctx or default_context()
    blocks = np.ascontiguousarray(blocks, dtype=BLOCK_DTYPE)
    n = blocks.size
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64); lens = np.ascontiguousarray(lens, dtype=np.uint32)
    numbps = np.ascontiguousarray(numbps, dtype=np.uint8)
    sizes = blocks["w"].astype(np.int64) * blocks["h"].astype(np.int64)
    coff = np.zeros(max(n, 1), dtype=np.uint64)
    coff[:n] = np.concatenate(([0], np.cumsum(sizes)[:-1])) if n else []
    coeffs = np.zeros(max(int(sizes.sum()), 1), dtype=np.int32)
    sp = stream.ctypes.data_as(C.c_void_p) if stream.size else None
    ctx.check(ctx.L.j2k_decode_blocks(
        ctx.h, int(coder), sp, offs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
        numbps.ctypes.data_as(C.c_void_p), blocks.ctypes.data_as(C.c_void_p), C.c_size_t(n),
        coeffs.ctypes.data_as(C.c_void_p), coff.ctypes.data_as(C.c_void_p)))
    return [coeffs[int(coff[i]):int(coff[i]) + int(sizes[i])].reshape(int(blocks[i]["h"]), int(blocks[i]["w"]))
            for i in range(n)]


def _one_block(w, h, band):
    b = np.zeros(1, dtype=BLOCK_DTYPE)
    b[0] = (0, band, 0, 0, w, h)
    return b


class T1:
    """entropy.T1 (t1.go:94-133).  NewT1(w,h) / GetT1(w,h) -> SetData -> Encode(band) ; Decode(bytes,numBPS,band)."""

    def __init__(self, width, height, ctx=None):
        self.width, self.height = int(width), int(height)
        self.ctx = ctx or default_context()
        self.data = np.zeros((self.height, self.width), dtype=np.int32)
        self.numBPS = 0

    def Resize(self, width, height):                 # t1.go:45-47
        self.__init__(width, height, self.ctx)

    def SetData(self, data):                          # t1.go:292-304
        flat = np.asarray(data, dtype=np.int32).reshape(-1)
        n = min(flat.size, self.data.size)            # Go copy(): min(len(dst), len(src))
        self.data.reshape(-1)[:n] = flat[:n]

    def Encode(self, bandType):                       # t1.go:918 -> t1_fast5.go:10
        if self.width <= 0 or self.height <= 0:
            return None
        s, _, lens, nb = encode_blocks(_lib.CODER_MQ, [self.data], _one_block(self.width, self.height, bandType), self.ctx)
        self.numBPS = int(nb[0])
        return bytes(s) if lens[0] else None          # nil for an all-zero block

    def Decode(self, data, numBPS, bandType):          # t1.go:1261-1292
        data = np.frombuffer(bytes(data or b""), dtype=np.uint8)
        out = decode_blocks(_lib.CODER_MQ, data, [0], [data.size], [numBPS], _one_block(self.width, self.height, bandType), self.ctx)
        return out[0].reshape(-1)


def NewT1(width, height):
    return T1(width, height)


GetT1 = NewT1                                          # the pool (t1.go:15-41) has no meaning across the ABI


def PutT1(t):
    pass


class HTEncoder:
    """entropy.HTEncoder (ht.go:872-1045)."""

    def __init__(self, width, height, ctx=None):
        self.width, self.height = int(width), int(height)
        self.ctx = ctx or default_context()
        self.data = np.zeros((self.height, self.width), dtype=np.int32)

    def SetData(self, data):                          # ht.go:935-937
        flat = np.asarray(data, dtype=np.int32).reshape(-1)
        n = min(flat.size, self.data.size)
        self.data.reshape(-1)[:n] = flat[:n]

    def Encode(self, bandType):                       # ht.go:942-1045
        s, _, lens, _ = encode_blocks(_lib.CODER_HT, [self.data], _one_block(self.width, self.height, bandType), self.ctx)
        return bytes(s) if lens[0] else None


class HTDecoder:
    """entropy.HTDecoder (ht.go:14-150); every Decode behaves like a fresh NewHTDecoder (zeroed output)."""

    def __init__(self, width, height, ctx=None):
        self.width, self.height = int(width), int(height)
        self.ctx = ctx or default_context()

    def Decode(self, data, numBitplanes, bandType):    # ht.go:93-150
        data = np.frombuffer(bytes(data or b""), dtype=np.uint8)
        out = decode_blocks(_lib.CODER_HT, data, [0], [data.size], [numBitplanes], _one_block(self.width, self.height, bandType), self.ctx)
        return out[0].reshape(-1)


NewHTEncoder, GetHTEncoder = HTEncoder, HTEncoder
NewHTDecoder, GetHTDecoder = HTDecoder, HTDecoder


def PutHTEncoder(e):
    pass


def PutHTDecoder(d):
    pass


# ---- stand-alone coders (internal/entropy/mqc.go) ---------------------------------------------------
def _buf(a):
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
    return a, a.ctypes.data_as(C.c_void_p)


def mq_encode(ctxs, decisions, ctx=None):
    """NewMQEncoder(); for i: Encode(ctxs[i], decisions[i]); Flush()  (mqc.go:169-349).  Returns bytes (b"" for nil)."""
    ctx = ctx or default_context()
    cs, pc = _buf(ctxs)
    ds, pd = _buf(decisions)
    assert cs.size == ds.size
    out = np.zeros(cs.size * 2 + 64, dtype=np.uint8)
    n = C.c_size_t(0)
    ctx.check(ctx.L.j2k_mq_encode(ctx.h, pc, pd, C.c_size_t(cs.size), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(n)))
    return out[:n.value].tobytes()


def mq_decode(data, ctxs, ctx=None):
    """NewMQDecoder(data); [Decode(c) for c in ctxs]  (mqc.go:352-497)."""
    ctx = ctx or default_context()
    dat, pdat = _buf(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    cs, pc = _buf(ctxs)
    out = np.zeros(max(cs.size, 1), dtype=np.uint8)
    ctx.check(ctx.L.j2k_mq_decode(ctx.h, pdat, C.c_size_t(dat.size), pc, C.c_size_t(cs.size), out.ctypes.data_as(C.c_void_p)))
    return out[:cs.size]


def raw_encode(bits, ctx=None):
    """NewRawEncoder(); EncodeBit(b) for b in bits; Flush()  (mqc.go:560-600)."""
    ctx = ctx or default_context()
    bs, pb = _buf(bits)
    out = np.zeros(bs.size // 7 + 16, dtype=np.uint8)
    n = C.c_size_t(0)
    ctx.check(ctx.L.j2k_raw_encode(ctx.h, pb, C.c_size_t(bs.size), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(n)))
    return out[:n.value].tobytes()


def raw_decode(data, n, ctx=None):
    """NewRawDecoder(data); [DecodeBit() for _ in range(n)]  (mqc.go:516-557)."""
    ctx = ctx or default_context()
    dat, pdat = _buf(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    ctx.check(ctx.L.j2k_raw_decode(ctx.h, pdat, C.c_size_t(dat.size), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
    return out[:n]


class MQEncoder:
    """Same call sequence as the Go type; the symbols are coded on the device at Flush()."""

    def __init__(self):
        self._c, self._d = [], []

    def Reset(self):
        self._c, self._d = [], []

    def Encode(self, ctx, decision):
        self._c.append(int(ctx)); self._d.append(int(decision) & 0xFF)

    def Flush(self):
        return mq_encode(self._c, self._d)


def NewMQEncoder():
    return MQEncoder()
