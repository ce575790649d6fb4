"""Mirror of the tile-part layer between the block coder and the Go-side codestream writer:

    encoder.createTileHeader              encoder.go:746-760      -> create_tile_header(tile_idx, tile_data)
    the per-tile loop of generateTiles    encoder.go:568-579      -> assemble_tiles(stream, tile_offs, tile_first)
    codestream.Parser.ReadTilePartHeader  parser.go:894-983       -> read_tile_part_header(cs, pos) / parse_tile_parts(cs)

Host calls of the C ABI (no device, no context): bytes in, bytes out."""
import ctypes as C

import numpy as np

from . import _lib


class TilePart(C.Structure):
    _fields_ = [("TileIndex", C.c_uint16), ("TilePartIndex", C.c_uint8), ("NumTileParts", C.c_uint8), ("TilePartLength", C.c_uint32),
                ("header_markers", C.c_uint32), ("pad_", C.c_uint32), ("header_off", C.c_uint64), ("data_off", C.c_uint64),
                ("data_len", C.c_uint64)]


def _check(st, what):
    if st != _lib.OK:
        raise _lib.J2KError(st, "%s: %s" % (what, _lib.lib().j2k_status_string(st).decode()))


def _u8(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def create_tile_header(tile_idx, tile_data):
    """e.createTileHeader(tileIdx, tileData): the 14-byte SOT ... SOD header followed by the data."""
    d, dp = _u8(tile_data)
    out = np.zeros(14 + d.size, np.uint8)
    n = C.c_size_t(0)
    _check(_lib.lib().j2k_create_tile_header(int(tile_idx), dp, C.c_size_t(d.size), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
                                             C.byref(n)), "create_tile_header")
    return out[:n.value].tobytes()


def assemble_tiles(stream, tile_offs, tile_first=0):
    """Tile-parts of every tile of a (gathered) stream, laid end to end in tile order."""
    s, sp = _u8(stream)
    offs = np.ascontiguousarray(tile_offs, dtype=np.uint64)
    nt = offs.size - 1
    L = _lib.lib()
    L.j2k_tile_part_bound.restype = C.c_size_t
    cap = int(L.j2k_tile_part_bound(offs.ctypes.data_as(C.c_void_p), int(nt)))
    out = np.zeros(max(cap, 1), np.uint8)
    n = C.c_size_t(0)
    _check(L.j2k_assemble_tiles(sp, offs.ctypes.data_as(C.c_void_p), int(tile_first), int(nt), out.ctypes.data_as(C.c_void_p),
                                C.c_size_t(cap), C.byref(n)), "assemble_tiles")
    return out[:n.value].tobytes()


def read_tile_part_header(cs, pos=0):
    """Parser.ReadTilePartHeader for the tile-part whose SOT marker is at cs[pos]."""
    d, dp = _u8(cs)
    tp = TilePart()
    _check(_lib.lib().j2k_read_tile_part_header(dp, C.c_size_t(d.size), C.c_size_t(int(pos)), C.byref(tp)), "read_tile_part_header")
    return tp


def parse_tile_parts(cs):
    """[(TilePart, data bytes)] for a run of tile-parts (up to EOC or the end)."""
    d, dp = _u8(cs)
    L = _lib.lib()
    n = C.c_size_t(0)
    st = L.j2k_parse_tile_parts(dp, C.c_size_t(d.size), None, C.c_size_t(0), C.byref(n))
    if st not in (_lib.OK, -4):
        _check(st, "parse_tile_parts")
    parts = (TilePart * max(n.value, 1))()
    _check(L.j2k_parse_tile_parts(dp, C.c_size_t(d.size), parts, C.c_size_t(n.value), C.byref(n)), "parse_tile_parts")
    return [(parts[i], d[int(parts[i].data_off):int(parts[i].data_off + parts[i].data_len)].tobytes()) for i in range(n.value)]
