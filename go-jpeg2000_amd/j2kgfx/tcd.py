"""Mirror of the hot-path methods of internal/tcd (the reference's tile-component API):

    TileEncoder.ApplyForwardDWT   tcd.go:508-534     TileDecoder.ApplyInverseDWT   tcd.go:416-437
    TileEncoder.EncodeCodeBlock   tcd.go:537-553     TileDecoder.DecodeCodeBlock   tcd.go:393-413

Same argument meaning; `TileComponent.Data` / `CodeBlock.Data` are numpy arrays / bytes mutated in place like the Go
fields.  Only what the path needs of the Go structs is modelled."""
import ctypes as C

import numpy as np

from . import entropy
from .context import default_context


class TileComponent:
    def __init__(self, x0, y0, x1, y1, data=None):
        self.X0, self.Y0, self.X1, self.Y1 = int(x0), int(y0), int(x1), int(y1)
        n = (self.X1 - self.X0) * (self.Y1 - self.Y0)
        self.Data = np.zeros(n, dtype=np.int32) if data is None else np.ascontiguousarray(data, dtype=np.int32).reshape(-1)


class CodeBlock:
    def __init__(self, x0, y0, x1, y1):
        self.X0, self.Y0, self.X1, self.Y1 = int(x0), int(y0), int(x1), int(y1)
        self.Data = None               # encoded bytes
        self.TotalBitPlanes = 0
        self.Coefficients = None


class _Tile:
    def __init__(self, num_decompositions, reversible=True, htj2k=False, ctx=None):
        self.numLevels = int(num_decompositions)            # header.CodingStyle.NumDecompositions
        self.reversible = bool(reversible)                  # header.CodingStyle.WaveletTransform == 1
        self.htj2k = bool(htj2k)
        self.ctx = ctx or default_context()

    def _dwt(self, fn, tc):
        w, h = tc.X1 - tc.X0, tc.Y1 - tc.Y0
        assert tc.Data.size >= w * h
        self.ctx.check(fn(self.ctx.h, tc.Data.ctypes.data_as(C.c_void_p), int(w), int(h), self.numLevels, int(self.reversible)))


class TileEncoder(_Tile):
    def ApplyForwardDWT(self, tc):                          # tcd.go:508-534
        self._dwt(self.ctx.L.j2k_tcd_apply_forward_dwt, tc)

    def EncodeCodeBlock(self, cb, data, bandType):          # tcd.go:537-553
        w, h = cb.X1 - cb.X0, cb.Y1 - cb.Y0
        if self.htj2k:
            enc = entropy.GetHTEncoder(w, h)
            enc.SetData(data)
            cb.Data = enc.Encode(bandType)
        else:
            t1 = entropy.NewT1(w, h)
            t1.SetData(data)
            cb.Data = t1.Encode(bandType)
            cb.TotalBitPlanes = t1.numBPS                   # the Go field is filled by the caller; kept here for Decode


class TileDecoder(_Tile):
    def ApplyInverseDWT(self, tc):                          # tcd.go:416-437
        self._dwt(self.ctx.L.j2k_tcd_apply_inverse_dwt, tc)

    def DecodeCodeBlock(self, cb, bandType):                # tcd.go:393-413
        if not cb.Data:
            return None
        w, h = cb.X1 - cb.X0, cb.Y1 - cb.Y0
        dec = entropy.GetHTDecoder(w, h) if self.htj2k else entropy.NewT1(w, h)
        cb.Coefficients = dec.Decode(cb.Data, cb.TotalBitPlanes, bandType)
        return None
