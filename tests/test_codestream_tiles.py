"""CPU (host calls of the C ABI, no device): multi-tile assembly = encoder.createTileHeader per tile (encoder.go:746-760) and
its reader = codestream.Parser.ReadTilePartHeader (internal/codestream/parser.go:894-983), against the oracle restatement and
the vectors of the reference's own parser tests (internal/codestream/parser_test.go:940-1068)."""
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cs():
    from j2kgfx import codestream
    return codestream


def test_create_tile_header_bytes(cs, oracle):
    for idx, data in [(0, b""), (0, b"\x01\x02\x03"), (7, bytes(range(200))), (65535, b"x" * 5), (65536 + 3, b"wrap")]:
        got = cs.create_tile_header(idx, data)
        # by hand from encoder.go:746-760
        want = struct.pack(">HHHIBBH", 0xFF90, 10, idx & 0xFFFF, 14 + len(data), 0, 1, 0xFF93) + data
        assert got == want == oracle.create_tile_header(idx, data)


def test_assemble_and_parse_round_trip(cs, oracle):
    rng = np.random.default_rng(3)
    lens = [0, 1, 513, 4096, 0, 77]
    stream = rng.integers(0, 256, sum(lens), dtype=np.uint8).tobytes()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    got = cs.assemble_tiles(stream, offs, tile_first=5)
    want = b"".join(oracle.create_tile_header(5 + t, stream[int(offs[t]):int(offs[t + 1])]) for t in range(len(lens)))
    assert got == want
    parts = cs.parse_tile_parts(got + b"\xFF\xD9")                      # EOC ends the run
    assert [(p.TileIndex, p.TilePartIndex, p.NumTileParts, p.TilePartLength, p.header_markers) for p, _ in parts] == \
        [(5 + t, 0, 1, 14 + lens[t], 0) for t in range(len(lens))]
    assert [d for _, d in parts] == [stream[int(offs[t]):int(offs[t + 1])] for t in range(len(lens))]


def test_reference_parser_vectors(cs):
    """parser_test.go: SOT(10, 0, 1000, 0, 1) SOD -> TileIndex 0, TilePartLength 1000, TilePartIndex 0, NumTileParts 1 (:962-991);
    a tile-part COD (:993-1034) / QCD (:1036-1067) between SOT and SOD is stepped over."""
    sot = struct.pack(">HHHIBB", 0xFF90, 10, 0, 1000, 0, 1)
    buf = sot + struct.pack(">H", 0xFF93) + bytes(1000 - 14)
    tp = cs.read_tile_part_header(buf)
    assert (tp.TileIndex, tp.TilePartLength, tp.TilePartIndex, tp.NumTileParts) == (0, 1000, 0, 1)
    assert (tp.data_off, tp.data_len, tp.header_markers) == (14, 986, 0)
    cod = struct.pack(">HHBBHBBBBBB", 0xFF52, 12, 0, 1, 2, 1, 4, 3, 3, 0, 0)
    qcd = struct.pack(">HHBH", 0xFF5C, 5, 0x60 | 1, 0x6000)
    sot2 = struct.pack(">HHHIBB", 0xFF90, 10, 0, 2000, 0, 1)
    buf = sot2 + cod + qcd + struct.pack(">H", 0xFF93) + bytes(2000 - 12 - len(cod) - len(qcd) - 2)
    tp = cs.read_tile_part_header(buf)
    assert (tp.TilePartLength, tp.header_markers, tp.header_off) == (2000, 2, 12)
    assert tp.data_off == 12 + len(cod) + len(qcd) + 2 and tp.data_off + tp.data_len == 2000


def test_parser_errors(cs):
    from j2kgfx import J2KError
    ok = struct.pack(">HHHIBBH", 0xFF90, 10, 0, 14, 0, 1, 0xFF93)
    assert cs.read_tile_part_header(ok).data_len == 0
    for bad in (struct.pack(">HHHIBBH", 0xFF90, 9, 0, 14, 0, 1, 0xFF93),            # "invalid SOT length"
                ok[:9],                                                             # unexpected EOF
                struct.pack(">HHHIBB", 0xFF90, 10, 0, 0, 0, 1) + struct.pack(">HH", 0xFF52, 1),   # segment length < 2
                struct.pack(">HHHIBBH", 0xFF90, 10, 0, 13, 0, 1, 0xFF93),           # Psot shorter than the header
                struct.pack(">HHHIBBH", 0xFF90, 10, 0, 99, 0, 1, 0xFF93),           # Psot past the end
                b"\xFF\x4F" + ok):                                                  # not at a SOT marker
        with pytest.raises(J2KError):
            cs.read_tile_part_header(bad)
    # Psot = 0: the data runs to the end of the codestream
    tp = cs.read_tile_part_header(struct.pack(">HHHIBBH", 0xFF90, 10, 3, 0, 0, 1, 0xFF93) + b"abcdef")
    assert (tp.TileIndex, tp.data_len) == (3, 6)
