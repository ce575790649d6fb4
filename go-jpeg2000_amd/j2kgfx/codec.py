"""Batched frame pipeline over the plan calls of the C ABI: the device-side mirror of
encoder.preprocess (encoder.go:216-281), encoder.encodeTile's job loop (encoder.go:597-688),
tcd.TileDecoder.DecodeCodeBlock / ApplyInverseDWT (tcd.go:393-437) and the tail of
decoder.decodeTiles (decoder.go:321-348).

torch is used only to own device memory (tensors on `cuda:<ctx.device>`); all kernels are
launched by libj2kgfx on the context's own (non-blocking) HIP stream, which torch's caching allocator
knows nothing about.  With `track_streams=True` (the default) every stage call therefore
  * makes the library stream wait for what torch has queued on its current stream (inputs written by torch ops), and
  * keeps a reference to every tensor handed over until the next `ctx.sync()` (or `ctx.close()`), so that a temporary
    dropped before then -- `plan.decode_blocks(*plan.encode_stream(plan.forward(x)))` -- is not given to another tensor
    while kernels still use it.  (Not `Tensor.record_stream`: the allocator would record an event on the library stream
    when such a tensor is finally freed, possibly after its context -- and the stream -- are gone: a crash at teardown.)
Results are read after `ctx.sync()` (or after ordering torch's stream behind `torch.cuda.ExternalStream(ctx.stream)`).
bench.py passes `track_streams=False`: its buffers live for the whole run and the launching thread has no slack.
"""
import functools
import ctypes as C
import sys

import numpy as np

from . import _lib
from .context import default_context

BLOCK_DTYPE = np.dtype([("plane", "<i4"), ("band", "<i4"), ("x0", "<i4"), ("y0", "<i4"), ("w", "<i4"), ("h", "<i4")])


def _torch():
    import torch
    return torch


def _stage(fn):
    """A stage call: order the library stream behind torch's current stream first (see the module docstring)."""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if self.track_streams:
            t = _torch()
            self._ext().wait_stream(t.cuda.current_stream(self.device))
        return fn(self, *a, **k)
    return wrapped


class FramePlan:
    def __init__(self, width, height, ncomp, precision=8, lossless=True, quality=0, num_resolutions=6,
                 cb=(64, 64), tile=(0, 0), coder=_lib.CODER_MQ, is_signed=False, tile_first=0, tile_count=0,
                 ctx=None, track_streams=True, frame_rows=0, closed_loop=False):
        self.ctx = ctx or default_context()
        self.track_streams = bool(track_streams)
        self._ext_stream = None
        L = self.ctx.L
        self.params = _lib.Params(width=width, height=height, ncomp=ncomp, precision=precision,
                                  is_signed=int(bool(is_signed)), lossless=int(bool(lossless)), quality=quality,
                                  num_resolutions=num_resolutions, cb_w=cb[0], cb_h=cb[1], tile_w=tile[0],
                                  tile_h=tile[1], coder=coder, tile_first=tile_first, tile_count=tile_count,
                                  frame_rows=frame_rows,    # frame_rows > 0: a batch of height / frame_rows frames stacked vertically
                                  closed_loop=int(bool(closed_loop)))   # this library's closed-loop mode (not the reference): windows that partition the plane, readable packets
        h = C.c_void_p()
        self.ctx.check(L.j2k_plan_create(self.ctx.h, C.byref(self.params), C.byref(h)))
        self.h = h
        info = _lib.PlanInfo()
        self.ctx.check(L.j2k_plan_get_info(self.h, C.byref(info)))
        self.info = info
        self.width, self.height, self.ncomp = width, height, ncomp
        self.device = "cuda:%d" % self.ctx.device
        from .context import register_plan
        register_plan(self)                     # closed by the package's atexit hook if the caller never does

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):    # (a plan whose context is gone was destroyed with it)
                self.ctx.L.j2k_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- geometry ---------------------------------------------------------------
    def blocks(self):
        n = int(self.info.blocks)
        out = np.zeros(n, dtype=BLOCK_DTYPE)
        if n:
            self.ctx.check(self.ctx.L.j2k_plan_get_blocks(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out

    def planes(self):
        """(n,7) int64: tile, comp, x0, y0, w, h, coefficient offset."""
        n = int(self.info.planes)
        out = np.zeros((n, 7), dtype=np.int64)
        self.ctx.check(self.ctx.L.j2k_plan_get_planes(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out

    def decoded_offsets(self):
        n = int(self.info.blocks)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self.ctx.check(self.ctx.L.j2k_plan_get_decoded_offsets(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out[:n]

    # ---- device buffers -----------------------------------------------------------
    def empty(self, n, dtype):
        t = _torch()
        return t.empty(max(int(n), 4), dtype=dtype, device=self.device)

    def alloc_coeff(self):
        return self.empty(self.info.coeff_elems, _torch().int32)

    def alloc_frame(self):
        t = _torch()
        return t.empty((self.ncomp, self.height, self.width), dtype=t.int32, device=self.device)

    def _ext(self):
        if self._ext_stream is None:
            self._ext_stream = _torch().cuda.ExternalStream(self.ctx.stream, device=self.device)
        return self._ext_stream

    def _p(self, t):
        if self.track_streams:
            self.ctx.hold(t)                 # alive until the library stream has been synchronised
        return C.c_void_p(t.data_ptr())

    # ---- stages (asynchronous on the ctx stream) ------------------------------------
    @_stage
    def forward(self, frame, coeff=None):
        """encoder.preprocess for every tile-component: frame int32 [C,H,W] -> coefficient buffer."""
        coeff = coeff if coeff is not None else self.alloc_coeff()
        assert frame.is_contiguous() and frame.numel() == self.ncomp * self.height * self.width
        self.ctx.check(self.ctx.L.j2k_plan_forward(self.h, self._p(frame), self._p(coeff)))
        return coeff

    @_stage
    def inverse(self, coeff, frame=None):
        frame = frame if frame is not None else self.alloc_frame()
        self.ctx.check(self.ctx.L.j2k_plan_inverse(self.h, self._p(coeff), self._p(frame)))
        return frame

    @_stage
    def forward_rgba8(self, pix, coeff=None):
        """extractImageData + preprocess fused: pix = device uint8 [H, stride] packed RGBA (image.RGBA.Pix)."""
        coeff = coeff if coeff is not None else self.alloc_coeff()
        assert pix.dim() == 2 and pix.shape[0] == self.height and pix.is_contiguous()
        self.ctx.check(self.ctx.L.j2k_plan_forward_rgba8(self.h, self._p(pix), C.c_size_t(int(pix.shape[1])), self._p(coeff)))
        return coeff

    @_stage
    def inverse_rgba8(self, coeff, pix=None):
        """inverse path + createImage (3 components, 8 bit) fused: returns device uint8 [H, W*4] packed RGBA."""
        t = _torch()
        if pix is None:
            pix = t.empty((self.height, self.width * 4), dtype=t.uint8, device=self.device)
        self.ctx.check(self.ctx.L.j2k_plan_inverse_rgba8(self.h, self._p(coeff), self._p(pix), C.c_size_t(int(pix.shape[1]))))
        return pix

    @_stage
    def encode_blocks(self, coeff, slots=None, lens=None, numbps=None):
        t = _torch()
        n = int(self.info.blocks)
        slots = slots if slots is not None else self.empty(self.info.bytes_cap, t.uint8)
        lens = lens if lens is not None else self.empty(n, t.int32)
        numbps = numbps if numbps is not None else self.empty(n, t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_encode_blocks(self.h, self._p(coeff), self._p(slots), self._p(lens),
                                                         self._p(numbps)))
        return slots, lens, numbps

    def set_decode_coded_rows_only(self, on=True):
        """HT coder: decode_blocks leaves the rows the reference's decoder never writes (y % 4 != 0) untouched -- the pooled
        HTDecoder's behaviour; the caller owns a buffer it zeroed once (j2k_plan_set_decode_coded_rows_only)."""
        self.ctx.check(self.ctx.L.j2k_plan_set_decode_coded_rows_only(self.h, int(bool(on))))

    @_stage
    def forward_pixels(self, fmt, pix, coeff=None):
        """extractImageData (+ rescale to the plan's precision) + preprocess: pix = device uint8 [H, stride] in a Go Pix layout."""
        coeff = coeff if coeff is not None else self.alloc_coeff()
        assert pix.dim() == 2 and pix.shape[0] == self.height and pix.is_contiguous()
        self.ctx.check(self.ctx.L.j2k_plan_forward_pixels(self.h, int(fmt), self._p(pix), C.c_size_t(int(pix.shape[1])), self._p(coeff)))
        return coeff

    def pixels_fused(self, fmt, pix, inverse=False):
        """would forward_pixels / inverse_pixels read / write these pixels in the level-0 kernels (True) or stage them (False)?"""
        r = self.ctx.L.j2k_plan_pixels_fused(self.h, int(fmt), self._p(pix), C.c_size_t(int(pix.shape[1])), 1 if inverse else 0)
        if r < 0:
            self.ctx.check(r)
        return bool(r)

    @_stage
    def inverse_pixels(self, coeff, pix):
        """inverse path + createImage for the plan's component count and precision into pix (device uint8 [H, stride])."""
        self.ctx.check(self.ctx.L.j2k_plan_inverse_pixels(self.h, self._p(coeff), self._p(pix), C.c_size_t(int(pix.shape[1]))))
        return pix

    @_stage
    def encode_stream(self, coeff, stream=None, offs=None, lens=None, numbps=None):
        """encode_blocks + compact in one call (one kernel for HT blocks up to 64x64): returns (stream, offs, lens, numbps)."""
        t = _torch()
        n = int(self.info.blocks)
        stream = stream if stream is not None else self.empty(self.info.bytes_cap, t.uint8)
        offs = offs if offs is not None else self.empty(n + 1, t.int64)
        lens = lens if lens is not None else self.empty(n, t.int32)
        numbps = numbps if numbps is not None else self.empty(n, t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_encode_stream(self.h, self._p(coeff), self._p(stream), self._p(offs), self._p(lens),
                                                         self._p(numbps)))
        return stream, offs, lens, numbps

    def t2_packets(self, layer=0):
        """one packet per (tile-component, resolution) of the plan in job order: numpy table (t2.DEV_PACKET_DTYPE)"""
        from . import t2
        n = C.c_size_t(0)
        st = self.ctx.L.j2k_plan_t2_packets(self.h, int(layer), None, C.c_size_t(0), C.byref(n))
        if st not in (0, -4):
            self.ctx.check(st)
        out = np.zeros(n.value, t2.DEV_PACKET_DTYPE)
        self.ctx.check(self.ctx.L.j2k_plan_t2_packets(self.h, int(layer), out.ctypes.data_as(C.c_void_p), C.c_size_t(n.value), C.byref(n)))
        return out

    @_stage
    def t2_fill_cbs(self, mb, offs, lens, numbps, cbs=None):
        """the block coder's outputs as the packet coder's code-block table (device uint8 [blocks * 24])"""
        t = _torch()
        cbs = cbs if cbs is not None else self.empty(int(self.info.blocks) * 24, t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_t2_fill_cbs(self.h, int(mb), self._p(offs), self._p(lens), self._p(numbps), self._p(cbs)))
        return cbs

    # ---- the closed-loop frame codec (plans made with closed_loop=True) ----------------------------------------------
    def frame_bound(self):
        self.ctx.L.j2k_plan_frame_bound.restype = C.c_size_t
        return int(self.ctx.L.j2k_plan_frame_bound(self.h))

    @_stage
    def encode_tile_parts(self, stream, offs, lens, numbps, sop=False, eph=False, out=None, tile_offs=None):
        """the block coder's outputs -> SOT | SOD | packets per tile, end to end: (out uint8, tile_offs int64[tiles + 1])"""
        t = _torch()
        out = out if out is not None else self.empty(self.frame_bound(), t.uint8)
        tile_offs = tile_offs if tile_offs is not None else self.empty(int(self.info.tiles) + 1, t.int64)[:int(self.info.tiles) + 1]
        self.ctx.check(self.ctx.L.j2k_plan_encode_tile_parts(self.h, self._p(stream), self._p(offs), self._p(lens), self._p(numbps), int(bool(sop)),
                                                             int(bool(eph)), self._p(out), C.c_size_t(int(out.numel())), self._p(tile_offs)))
        return out, tile_offs

    @_stage
    def decode_tile_parts(self, cs, length, tile_offs=None, sop=False, eph=False, offs=None, lens=None, numbps=None):
        """tile-parts cs[:length] -> (offs, lens, numbps) as decode_blocks takes them (offsets into cs)"""
        t = _torch()
        n = int(self.info.blocks)
        offs = offs if offs is not None else self.empty(n + 1, t.int64)
        lens = lens if lens is not None else self.empty(n, t.int32)
        numbps = numbps if numbps is not None else self.empty(n, t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_decode_tile_parts(self.h, self._p(cs), C.c_size_t(int(length)), self._p(tile_offs) if tile_offs is not None else None,
                                                             int(bool(sop)), int(bool(eph)), self._p(offs), self._p(lens), self._p(numbps)))
        return offs, lens, numbps

    @_stage
    def place_blocks(self, decoded, coeff=None):
        coeff = coeff if coeff is not None else self.alloc_coeff()
        self.ctx.check(self.ctx.L.j2k_plan_place_blocks(self.h, self._p(decoded), self._p(coeff)))
        return coeff

    def frame_status(self):
        """synchronises; raises what the asynchronous frame calls found since the last call (capacity, malformed input)"""
        self.ctx.check(self.ctx.L.j2k_plan_frame_status(self.h))
        self.ctx.sync()

    def frame_parallel_tiles(self):
        """synchronises; tiles whose packets the decode calls since the last query parsed side by side (SOP + EPH streams)"""
        import ctypes
        n = ctypes.c_long(0)
        self.ctx.check(self.ctx.L.j2k_plan_frame_parallel_tiles(self.h, ctypes.byref(n)))
        return int(n.value)

    @_stage
    def encode_frame_pixels(self, fmt, pix, sop=False, eph=False, out=None, tile_offs=None):
        """pixels (a Go Pix layout, device uint8 [H, stride]) -> tile-parts: (out uint8, tile_offs int64[tiles + 1])"""
        t = _torch()
        out = out if out is not None else self.empty(self.frame_bound(), t.uint8)
        tile_offs = tile_offs if tile_offs is not None else self.empty(int(self.info.tiles) + 1, t.int64)[:int(self.info.tiles) + 1]
        self.ctx.check(self.ctx.L.j2k_plan_encode_frame_pixels(self.h, int(fmt), self._p(pix), C.c_size_t(int(pix.shape[1])), int(bool(sop)), int(bool(eph)),
                                                               self._p(out), C.c_size_t(int(out.numel())), self._p(tile_offs)))
        return out, tile_offs

    @_stage
    def decode_frame_pixels(self, cs, length, pix, tile_offs=None, sop=False, eph=False):
        """tile-parts cs[:length] -> pixels into pix (device uint8 [H, stride])"""
        self.ctx.check(self.ctx.L.j2k_plan_decode_frame_pixels(self.h, self._p(cs), C.c_size_t(int(length)), self._p(tile_offs) if tile_offs is not None else None,
                                                               int(bool(sop)), int(bool(eph)), self._p(pix), C.c_size_t(int(pix.shape[1]))))
        return pix

    # ---- host memory in, host memory out: the one-call forms (j2k_encode_pixels_host / j2k_decode_pixels_host) -------------------
    def encode_pixels_host(self, fmt, pix, sop=False, eph=False, cap=None):
        """pix: numpy uint8 [H, stride] (a Go Pix layout) -> dict(bytes, tile_offs, lens, numbps): every tile as a tile-part"""
        L = self.ctx.L
        n, nt = int(self.info.blocks), int(self.info.tiles)
        L.j2k_plan_tile_parts_bound.restype = C.c_size_t
        cap = int(cap) if cap is not None else max(self.frame_bound(), int(L.j2k_plan_tile_parts_bound(self.h))) + 64
        pix = np.ascontiguousarray(pix, dtype=np.uint8)
        out = np.zeros(max(cap, 1), np.uint8)
        toffs = np.zeros(nt + 1, np.uint64); lens = np.zeros(max(n, 1), np.uint32); nbps = np.zeros(max(n, 1), np.uint8)
        olen = C.c_size_t(0)
        st = L.j2k_encode_pixels_host(self.h, int(fmt), pix.ctypes.data_as(C.c_void_p), C.c_size_t(int(pix.shape[1])), int(bool(sop)), int(bool(eph)),
                                      out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(olen), toffs.ctypes.data_as(C.c_void_p),
                                      lens.ctypes.data_as(C.c_void_p), nbps.ctypes.data_as(C.c_void_p))
        self.encoded_len = olen.value
        self.ctx.check(st)
        return dict(bytes=out[:olen.value].copy(), tile_offs=toffs, lens=lens[:n].copy(), numbps=nbps[:n].copy())

    def decode_pixels_host(self, cs, shape, sop=False, eph=False):
        """closed-loop plans: tile-parts (bytes / numpy uint8) -> numpy uint8 pixels of `shape` = (H, stride)"""
        cs = np.ascontiguousarray(np.frombuffer(bytes(cs), np.uint8) if not isinstance(cs, np.ndarray) else cs, dtype=np.uint8)
        pix = np.zeros(shape, np.uint8)
        self.ctx.check(self.ctx.L.j2k_decode_pixels_host(self.h, cs.ctypes.data_as(C.c_void_p), C.c_size_t(cs.size), int(bool(sop)), int(bool(eph)),
                                                         pix.ctypes.data_as(C.c_void_p), C.c_size_t(int(shape[1]))))
        return pix

    def pack_bound(self):
        return int(self.ctx.L.j2k_plan_pack_bound(self.h))

    @_stage
    def pack_stream(self, stream, offs, lens, numbps, pack=None):
        """Transport form (blocks without the reference's MEL zero runs + the per-block arrays) of the stream the LAST
        encode_stream call on this plan produced; the first int64 of the pack is its length in bytes."""
        t = _torch()
        pack = pack if pack is not None else self.empty(self.pack_bound(), t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_pack_stream(self.h, self._p(stream), self._p(offs), self._p(lens), self._p(numbps),
                                                       self._p(pack)))
        return pack

    @_stage
    def assemble_tiles(self, stream, offs, out=None, out_len=None):
        """encoder.createTileHeader for every tile of this plan's shard, on the device (j2k_plan_assemble_tiles_device):
        (out uint8, out_len int64[1]); out[:out_len] = SOT ... SOD data of tile tile_first, tile_first + 1, ... end to end."""
        t = _torch()
        L = self.ctx.L
        L.j2k_plan_tile_parts_bound.restype = C.c_size_t
        out = out if out is not None else self.empty(int(L.j2k_plan_tile_parts_bound(self.h)), t.uint8)
        out_len = out_len if out_len is not None else self.empty(1, t.int64)
        self.ctx.check(L.j2k_plan_assemble_tiles_device(self.h, self._p(stream), self._p(offs), self._p(out), self._p(out_len)))
        return out, out_len

    @_stage
    def unpack_stream(self, pack, stream=None, offs=None, lens=None, numbps=None):
        """Root side of the gather: pack -> (stream, offs, lens, numbps), byte for byte what encode_stream produced."""
        t = _torch()
        n = int(self.info.blocks)
        stream = stream if stream is not None else self.empty(self.info.bytes_cap, t.uint8)
        offs = offs if offs is not None else self.empty(n + 1, t.int64)
        lens = lens if lens is not None else self.empty(n, t.int32)
        numbps = numbps if numbps is not None else self.empty(n, t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_unpack_stream(self.h, self._p(pack), C.c_size_t(int(pack.numel())), self._p(stream), self._p(offs),
                                                         self._p(lens), self._p(numbps)))
        return stream, offs, lens, numbps

    @_stage
    def unpack_streams(self, packs, outs):
        """unpack_stream for several packs of this geometry in ONE launch: packs = list of uint8 tensors, outs = list of
        (stream, offs, lens, numbps) tuples to fill.  A pack is foreign input: its tensor length is passed along and
        nothing outside it is read."""
        k = len(packs)
        assert k == len(outs)
        VP = C.c_void_p * k
        cols = list(zip(*outs)) if k else [(), (), (), ()]
        arrs = [VP(*[self._p(t_).value for t_ in col]) for col in ([*packs],) + tuple(cols)]
        sizes = (C.c_size_t * k)(*[int(p_.numel()) for p_ in packs])
        self.ctx.check(self.ctx.L.j2k_plan_unpack_streams(self.h, C.c_int(k), arrs[0], sizes, *arrs[1:]))
        return outs

    @_stage
    def compact(self, slots, lens, offs=None, stream=None):
        t = _torch()
        n = int(self.info.blocks)
        offs = offs if offs is not None else self.empty(n + 1, t.int64)
        stream = stream if stream is not None else self.empty(self.info.bytes_cap, t.uint8)
        self.ctx.check(self.ctx.L.j2k_plan_compact(self.h, self._p(slots), self._p(lens), self._p(offs),
                                                   self._p(stream)))
        return offs, stream

    @_stage
    def decode_blocks(self, stream, offs, lens, numbps, decoded=None):
        t = _torch()
        decoded = decoded if decoded is not None else self.empty(self.info.decoded_elems, t.int32)
        self.ctx.check(self.ctx.L.j2k_plan_decode_blocks(self.h, self._p(stream), self._p(offs), self._p(lens),
                                                         self._p(numbps), self._p(decoded)))
        return decoded

    # ---- host planes in, bytes out (encoder.preprocess + encodeTile) --------------------
    def encode_frame(self, planes):
        """planes: list of C int32 ndarrays (H*W).  Single tile: planes are overwritten with the
        coefficients like e.componentData.  Returns dict(bytes, lens, numbps, tile_offs, coeff)."""
        n = int(self.info.blocks)
        arr = (C.POINTER(C.c_int32) * len(planes))()
        for i, p in enumerate(planes):
            assert p.dtype == np.int32 and p.flags.c_contiguous and p.size == self.width * self.height
            arr[i] = p.ctypes.data_as(C.POINTER(C.c_int32))
        out = np.zeros(max(int(self.info.bytes_cap), 16), dtype=np.uint8)
        coeff = np.zeros(max(int(self.info.coeff_elems), 4), dtype=np.int32)
        lens = np.zeros(max(n, 1), dtype=np.uint32)
        nbps = np.zeros(max(n, 1), dtype=np.uint8)
        toffs = np.zeros(int(self.info.tiles) + 1, dtype=np.uint64)
        olen = C.c_size_t(0)
        self.ctx.check(self.ctx.L.j2k_encode_frame(
            self.h, arr, coeff.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
            C.byref(olen), toffs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
            nbps.ctypes.data_as(C.c_void_p)))
        return dict(bytes=out[:olen.value].copy(), lens=lens[:n].copy(), numbps=nbps[:n].copy(), tile_offs=toffs,
                    coeff=coeff)
