"""CPU: every expectation of the reference's colorspace_spec_test.go (:14-420) reproduced against the oracle's restatement of
colorspace.go (orc_convert_colorspace) -- the vectors with their tolerances, ranges and neutrality checks, the 16-bit cases,
the CMY round trip and the gamma functions.  (colorspace_test.go's expectations: tests/test_oracle_reference_identities.py;
the GPU kernels against the oracle on random planes: tests/test_gpu_pixels.py::test_colorspace_conversions.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

# jpeg2000.go:124-197 ColorSpace constants (include/j2kgfx.h J2K_CS_*) -> the conversion getColorConversion picks (colorspace.go:54-89)
SYCC, EYCC, CMYK, YCBCR2, YCBCR3, PHOTOYCC, CMY, CIELAB, ESRGB, ROMM, YPBPR60, YPBPR50 = 3, 4, 5, 7, 8, 9, 10, 12, 14, 15, 16, 17


def conv(cs, vals, precision=8):
    return [int(p[0]) for p in orc.convert_colorspace([np.array([v], np.int32) for v in vals], cs, precision)]


@pytest.mark.parametrize("name,cs,ycc,rgb,tol", [               # TestSpecYCbCrToRGB :14-75
    ("sYCC_gray", SYCC, (128, 128, 128), (128, 128, 128), 2),
    ("sYCC_black", SYCC, (0, 128, 128), (0, 0, 0), 2),
    ("sYCC_white", SYCC, (255, 128, 128), (255, 255, 255), 2),
    ("sYCC_red", SYCC, (54, 99, 255), (255, 0, 0), 15),
    ("sYCC_green", SYCC, (182, 30, 12), (0, 255, 0), 15),
    ("sYCC_blue", SYCC, (18, 255, 116), (0, 0, 255), 15),
    ("BT601_gray", YCBCR2, (128, 128, 128), (128, 128, 128), 2),
    ("BT601_black", YCBCR2, (0, 128, 128), (0, 0, 0), 2),
    ("BT601_white", YCBCR2, (255, 128, 128), (255, 255, 255), 2),
    ("BT601_gray_525", YCBCR3, (128, 128, 128), (128, 128, 128), 2),    # colorspace.go:60-61: the same conversion
])
def test_spec_ycbcr_to_rgb(name, cs, ycc, rgb, tol):
    got = conv(cs, ycc)
    assert all(abs(g - w) <= tol for g, w in zip(got, rgb)), (name, got)


@pytest.mark.parametrize("cmy,rgb", [                            # TestSpecCMYToRGB :78-121, exact
    ((0, 0, 0), (255, 255, 255)), ((255, 255, 255), (0, 0, 0)), ((0, 255, 255), (255, 0, 0)), ((255, 0, 255), (0, 255, 0)),
    ((255, 255, 0), (0, 0, 255)), ((255, 0, 0), (0, 255, 255)), ((0, 255, 0), (255, 0, 255)), ((0, 0, 255), (255, 255, 0)),
    ((128, 128, 128), (127, 127, 127)),
])
def test_spec_cmy_to_rgb(cmy, rgb):
    assert tuple(conv(CMY, cmy)) == rgb


@pytest.mark.parametrize("cmyk,rgb,tol", [                       # TestSpecCMYKToRGB :124-153
    ((0, 0, 0, 0), (255, 255, 255), 1), ((0, 0, 0, 255), (0, 0, 0), 1), ((255, 255, 255, 0), (0, 0, 0), 1),
    ((0, 255, 255, 0), (255, 0, 0), 1), ((0, 0, 0, 128), (127, 127, 127), 2), ((0, 255, 255, 128), (127, 0, 0), 2),
])
def test_spec_cmyk_to_rgb(cmyk, rgb, tol):
    got = conv(CMYK, cmyk)[:3]
    assert all(abs(g - w) <= tol for g, w in zip(got, rgb)), got


@pytest.mark.parametrize("lab,lo,hi,gray", [                     # TestSpecCIELabToRGB :156-204
    ((0, 128, 128), 0, 5, True), ((255, 128, 128), 250, 255, False), ((128, 128, 128), 80, 140, True),
    ((128, 200, 128), 100, 255, False), ((128, 128, 50), 0, 200, False),
])
def test_spec_cielab_to_rgb(lab, lo, hi, gray):
    r, g, b = conv(CIELAB, lab)
    assert lo <= r <= hi
    if gray:
        assert abs(r - g) <= 25 and abs(g - b) <= 25


@pytest.mark.parametrize("cs", [YPBPR60, YPBPR50])
@pytest.mark.parametrize("ypp,rgb", [((128, 128, 128), (128, 128, 128)), ((0, 128, 128), (0, 0, 0)), ((255, 128, 128), (255, 255, 255))])
def test_spec_ypbpr_to_rgb(cs, ypp, rgb):                        # TestSpecYPbPrToRGB :207-240
    assert all(abs(g - w) <= 2 for g, w in zip(conv(cs, ypp), rgb))


@pytest.mark.parametrize("ycc,lo,hi", [((128, 156, 156), 100, 160), ((0, 156, 156), 0, 20)])
def test_spec_photoycc_to_rgb(ycc, lo, hi):                      # TestSpecPhotoYCCToRGB :243-274
    r, g, b = conv(PHOTOYCC, ycc)
    assert lo <= r <= hi and abs(r - g) <= 20 and abs(g - b) <= 20


@pytest.mark.parametrize("rgb", [(0, 0, 0), (255, 255, 255), (128, 128, 128)])
def test_spec_romm_rgb_to_rgb(rgb):                              # TestSpecROMMRGBToRGB :277-303
    assert all(0 <= v <= 255 for v in conv(ROMM, rgb))


def test_spec_extended_colorspaces():                            # TestSpecExtendedColorspaces :306-329
    assert all(50 <= v <= 200 for v in conv(ESRGB, (128, 128, 128)))
    assert all(abs(v - 128) <= 5 for v in conv(EYCC, (128, 128, 128)))


def test_spec_16_bit_precision():                                # TestSpec16BitPrecision :332-366
    assert all(abs(v - 32768) <= 200 for v in conv(SYCC, (32768, 32768, 32768), 16))
    assert conv(CMY, (0, 0, 0), 16) == [65535, 65535, 65535]
    assert conv(CMYK, (0, 0, 0, 65535), 16)[:3] == [0, 0, 0]


def test_spec_cmy_round_trip():                                  # TestSpecRoundTrip :369-393
    for c in range(0, 256, 51):
        for m in range(0, 256, 51):
            for y in range(0, 256, 51):
                r, g, b = conv(CMY, (c, m, y))
                assert (255 - r, 255 - g, 255 - b) == (c, m, y)


def test_spec_gamma_functions():                                 # TestSpecGammaFunctions :396-417
    L = orc.lib()
    L.orc_pin_srgb_gamma.restype = C.c_double
    L.orc_pin_srgb_inverse_gamma.restype = C.c_double
    L.orc_pin_srgb_gamma.argtypes = [C.c_double]
    L.orc_pin_srgb_inverse_gamma.argtypes = [C.c_double]
    for i in range(101):
        lin = i / 100.0
        assert abs(L.orc_pin_srgb_inverse_gamma(L.orc_pin_srgb_gamma(lin)) - lin) <= 0.0001
    assert abs(L.orc_pin_srgb_gamma(0.0031308) - 12.92 * 0.0031308) <= 0.0001
