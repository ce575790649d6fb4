"""j2kgfx -- host-side mirror of go-jpeg2000's hot-path packages over the C ABI of
libj2kgfx.so (HIP, gfx950).  Same names and argument meaning as the reference:

    j2kgfx.mct      <- internal/mct      (ForwardRCT, InverseRCT, ForwardICT, ..., DCLevelShiftForward, ...)
    j2kgfx.dwt      <- internal/dwt      (Forward53, ..., DecomposeMultiLevel53, ReconstructMultiLevel97, ...)
    j2kgfx.entropy  <- internal/entropy  (T1, HTEncoder, HTDecoder, BandLL..BandHH)
    j2kgfx.tcd      <- internal/tcd      (TileEncoder/TileDecoder: ApplyForwardDWT, EncodeCodeBlock, ...)
    j2kgfx.colorspace <- colorspace.go (decode-side conversions to sRGB)
    j2kgfx.pixels   <- encoder.extractImageData / decoder.createImage (pixel buffers at native width)
    j2kgfx.codec    <- encoder.preprocess / encodeTile / decoder.decodeTiles tail, batched per frame
    j2kgfx.dist     <- tile / frame sharding over ranks + gather of compressed blocks (torch.distributed)

numpy arrays stand in for Go slices and are mutated IN PLACE like them.  There is no
CPU fallback: without libj2kgfx.so and a HIP device every call raises J2KError.
"""
from ._lib import (J2KError, BAND_LL, BAND_HL, BAND_LH, BAND_HH, CODER_MQ, CODER_HT, lib)  # noqa: F401
from .context import Context, default_context  # noqa: F401
