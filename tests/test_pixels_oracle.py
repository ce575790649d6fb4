"""CPU: the oracle's extractImageData / createImage restatement (oracle/j2k_oracle.c) against an independent numpy
restatement of the same reference loops (encoder.go:79-213, decoder.go:417-588), including the int32 wraparound of the
precision rescale (65535 * 65535 overflows; Go wraps, then divides truncating toward zero)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

COMPS = [1, 1, 3, 3, 4, 4]
PREC = [8, 16, 8, 16, 8, 16]
BPP = [1, 2, 4, 8, 4, 8]


def go_muldiv(v, a, b):
    """int32(v) * int32(a) with wraparound, then Go's truncating division by b"""
    p = (v.astype(np.int64) * a) & 0xFFFFFFFF
    p = np.where(p >= 1 << 31, p - (1 << 32), p)
    return (np.sign(p) * (np.abs(p) // b)).astype(np.int32)


def np_extract(pix, fmt, w, h, target):
    rows = pix[:, :w * BPP[fmt]]
    if PREC[fmt] == 8:
        s = rows.reshape(h, w, BPP[fmt]).astype(np.int32)
    else:
        b = rows.reshape(h, w, BPP[fmt] // 2, 2).astype(np.int32)
        s = b[..., 0] * 256 + b[..., 1]
    planes = [s[..., c].copy() for c in range(COMPS[fmt])]
    if 0 < target <= 16 and target != PREC[fmt]:
        planes = [go_muldiv(p, (1 << target) - 1, (1 << PREC[fmt]) - 1) for p in planes]
    return planes


def np_create(planes, prec, stride):
    nc = len(planes)
    h, w = planes[0].shape
    mx = (1 << prec) - 1
    vals = []
    for p in planes:
        t = np.clip(p.astype(np.int64), 0, mx).astype(np.int32)
        if prec > 8:
            t = go_muldiv(t, 65535, mx)
        elif prec != 8:
            t = go_muldiv(t, 255, mx)
        vals.append(t)
    pix = np.zeros((h, stride), dtype=np.uint8)
    if nc == 1:
        if prec <= 8:
            pix[:, :w] = vals[0].astype(np.uint8)
        else:
            u = vals[0].astype(np.uint16)
            pix[:, 0:2 * w:2] = (u >> 8).astype(np.uint8); pix[:, 1:2 * w:2] = u.astype(np.uint8)
    else:
        four = vals[:3] + [vals[3] if nc == 4 else np.full((h, w), 65535 if prec > 8 else 255, np.int32)]
        for c, v in enumerate(four):
            if prec <= 8:
                pix[:, c:4 * w:4] = v.astype(np.uint8)
            else:
                u = v.astype(np.uint16)
                pix[:, 2 * c:8 * w:8] = (u >> 8).astype(np.uint8); pix[:, 2 * c + 1:8 * w:8] = u.astype(np.uint8)
    return pix


@pytest.mark.parametrize("fmt", range(6))
@pytest.mark.parametrize("target", [0, 8, 10, 12, 16, 5])
def test_extract_image_data_matches_numpy(fmt, target):
    import oracle as orc
    rng = np.random.default_rng(fmt * 31 + target)
    w, h = 37, 11
    stride = w * BPP[fmt] + 12
    pix = rng.integers(0, 256, (h, stride)).astype(np.uint8)
    got = orc.extract_image_data(pix, fmt, w, h, target)
    want = np_extract(pix, fmt, w, h, target)
    assert len(got) == COMPS[fmt]
    for g, wnt in zip(got, want):
        assert np.array_equal(g, wnt)


def test_extract_hand_values():
    import oracle as orc
    # RGBA: alpha ignored; Gray16 big-endian; 16 -> 16 is a no-op, 16 -> 12 of 65535: 65535*4095 wraps in int32
    pix = np.array([[10, 20, 30, 99, 255, 0, 1, 7]], dtype=np.uint8)
    r, g, b = orc.extract_image_data(pix, 2, 2, 1)
    assert r.tolist() == [[10, 255]] and g.tolist() == [[20, 0]] and b.tolist() == [[30, 1]]
    g16 = orc.extract_image_data(np.array([[0x12, 0x34, 0xFF, 0xFF]], dtype=np.uint8), 1, 2, 1)[0]
    assert g16.tolist() == [[0x1234, 0xFFFF]]
    g12 = orc.extract_image_data(np.array([[0xFF, 0xFF]], dtype=np.uint8), 1, 1, 1, 12)[0]
    wrapped = (65535 * 4095) & 0xFFFFFFFF
    wrapped = wrapped - (1 << 32) if wrapped >= 1 << 31 else wrapped
    assert g12.tolist() == [[int(abs(wrapped) // 65535) * (1 if wrapped >= 0 else -1)]]
    assert orc.extract_image_data(np.array([[200]], dtype=np.uint8), 0, 1, 1, 10)[0].tolist() == [[200 * 1023 // 255]]


@pytest.mark.parametrize("nc", [1, 3, 4])
@pytest.mark.parametrize("prec", [1, 5, 8, 10, 12, 16])
def test_create_image_matches_numpy(nc, prec):
    import oracle as orc
    rng = np.random.default_rng(nc * 17 + prec)
    w, h = 29, 9
    mx = (1 << prec) - 1
    planes = [rng.integers(-40, mx + 40, (h, w)).astype(np.int32) for _ in range(nc)]
    planes[0][0, 0] = mx; planes[0][0, 1] = 0; planes[0][0, 2] = -2147483648; planes[0][0, 3] = 2147483647
    bpp = (1 if nc == 1 else 4) * (2 if prec > 8 else 1)
    stride = w * bpp + 8
    got = orc.create_image(planes, prec, stride)
    want = np_create(planes, prec, stride)
    assert np.array_equal(got, want)


def test_colorspace_oracle_hand_values():
    """colorspace.go by hand: sYCC neutral grey stays grey; CMY is the integer complement; CMYK with K = max is black;
    BT.601 red; an unknown space and a 2-component image are untouched."""
    import oracle as orc
    g = [np.array([100], np.int32), np.array([128], np.int32), np.array([128], np.int32)]
    assert [int(p[0]) for p in orc.convert_colorspace(g, 3, 8)] == [100, 100, 100]
    c = [np.array([0, 255, 10], np.int32), np.array([255, 0, 20], np.int32), np.array([5, 5, 30], np.int32)]
    out = orc.convert_colorspace(c, 10, 8)
    assert out[0].tolist() == [255, 0, 245] and out[1].tolist() == [0, 255, 235] and out[2].tolist() == [250, 250, 225]
    k = [np.array([10], np.int32), np.array([20], np.int32), np.array([30], np.int32), np.array([255], np.int32)]
    assert [int(p[0]) for p in orc.convert_colorspace(k, 5, 8)[:3]] == [0, 0, 0]
    y = [np.array([76], np.int32), np.array([85], np.int32), np.array([255], np.int32)]      # Y, Cb, Cr of pure red (BT.601)
    r, gg, b = [int(p[0]) for p in orc.convert_colorspace(y, 7, 8)]
    assert r == 254 and gg == 0 and b == 0      # 76 + 1.402*127 = 254.05 -> 254; g = 76 + 14.8 - 90.7 = 0.1 -> 0; b = 76 - 76.2 -> clamp 0
    same = [np.array([1, 2, 3], np.int32)] * 3
    assert orc.convert_colorspace([p.copy() for p in same], 1, 8)[0].tolist() == [1, 2, 3]
    assert orc.convert_colorspace([np.array([9], np.int32)] * 2, 3, 8)[0].tolist() == [9]
