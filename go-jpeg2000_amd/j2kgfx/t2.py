"""Mirror of the reference's Tier-2 layer and tile geometry, host calls of the C ABI (csrc/t2.cpp):

    tcd.NewPacketIterator / Next / Reset            internal/tcd/t2.go:41-238    -> PacketIterator
    tcd.NewPacketEncoder / EncodePacket             internal/tcd/t2.go:241-438   -> PacketEncoder
    tcd.NewPacketDecoder / DecodePacket / Position  internal/tcd/t2.go:439-652   -> PacketDecoder
    tcd.NewTagTree                                  internal/tcd/tcd.go:168-197  -> TagTree
    tcd.TileDecoder.InitTile                        internal/tcd/tcd.go:240-390  -> init_tile

Same names, argument meaning and error behaviour as the Go objects (an error return is a J2KError; an input on which the Go
code panics is J2KError with status ERR_GO_PANIC).  No device, no context."""
import ctypes as C

import numpy as np

from . import _lib

LRCP, RLCP, RPCL, PCRL, CPRL = 0, 1, 2, 3, 4          # codestream/markers.go:177-188


class _Packet(C.Structure):
    _fields_ = [("layer", C.c_int32), ("resolution", C.c_int32), ("component", C.c_int32), ("precinct", C.c_int32)]


class _Cb(C.Structure):
    _fields_ = [("included_in_layers", C.c_int32), ("zero_bit_planes", C.c_int32), ("num_passes", C.c_int32),
                ("data_len", C.c_uint32), ("data_cap", C.c_uint32), ("pad_", C.c_uint32), ("data", C.c_void_p)]


class _Precinct(C.Structure):
    _fields_ = [("nbands", C.c_int32), ("incl_tree_w", C.c_int32), ("imsb_tree_w", C.c_int32), ("pad_", C.c_int32),
                ("band_ncb", C.c_void_p), ("cbs", C.c_void_p)]


class _DecState(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("rpos", C.c_uint64), ("buf", C.c_uint8), ("cnt", C.c_uint8), ("saw_ff", C.c_uint8),
                ("pad_", C.c_uint8 * 5)]


class _Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]

    def t(self):
        return (self.x0, self.y0, self.x1, self.y1)


class _Band(C.Structure):
    _fields_ = [("comp", C.c_int32), ("res", C.c_int32), ("type", C.c_int32), ("cbx", C.c_int32), ("cby", C.c_int32),
                ("pad_", C.c_int32), ("r", _Rect), ("cb0", C.c_uint64)]


class _Header(C.Structure):
    _fields_ = [("image_w", C.c_uint32), ("image_h", C.c_uint32), ("image_x0", C.c_uint32), ("image_y0", C.c_uint32),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32),
                ("num_tiles_x", C.c_uint32), ("ncomp", C.c_int32), ("subsampling", C.c_void_p),
                ("num_decompositions", C.c_uint8), ("cb_w_exp", C.c_uint8), ("cb_h_exp", C.c_uint8), ("pad_", C.c_uint8 * 5)]


def _check(st, what):
    if st != _lib.OK:
        raise _lib.J2KError(st, "%s: %s" % (what, _lib.lib().j2k_status_string(st).decode()))


class TagTree:
    """tcd.NewTagTree(width, height): the coder only ever uses .width as a divisor (t2.go:328,352)."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        lv = C.c_int32(0)
        sizes = (C.c_int64 * 64)()
        _check(_lib.lib().j2k_tagtree_shape(self.width, self.height, C.byref(lv), sizes, C.c_size_t(64)), "NewTagTree")
        self.levels = lv.value
        self.level_sizes = [int(sizes[i]) for i in range(self.levels)]


class CodeBlock:
    """the tcd.CodeBlock fields the packet coder touches (tcd.go:103-128); Passes is a count or a list"""

    def __init__(self, Data=None, IncludedInLayers=0, ZeroBitPlanes=0, Passes=0, Index=0):
        self.Index = Index
        self.Data = None if Data is None else bytes(Data)
        self.IncludedInLayers, self.ZeroBitPlanes = int(IncludedInLayers), int(ZeroBitPlanes)
        self.Passes = len(Passes) if hasattr(Passes, "__len__") else int(Passes)


class Precinct:
    def __init__(self, CodeBlocks, InclusionTree=None, IMSBTree=None):
        self.CodeBlocks = CodeBlocks                    # [band][code-block]
        self.InclusionTree = InclusionTree if InclusionTree is not None else TagTree(1, 1)
        self.IMSBTree = IMSBTree if IMSBTree is not None else TagTree(1, 1)


def _marshal(precinct, data_cap=0):
    """ctypes view of a Precinct: (struct, keep-alive list, flat list of the CodeBlock objects, their data buffers)"""
    flat = [cb for band in precinct.CodeBlocks for cb in band]
    ncb = np.array([len(b) for b in precinct.CodeBlocks], np.int32)
    cbs = (_Cb * max(len(flat), 1))()
    bufs = []
    for i, cb in enumerate(flat):
        dl = 0 if cb.Data is None else len(cb.Data)
        buf = np.zeros(max(dl, data_cap, 1), np.uint8)
        if dl:
            buf[:dl] = np.frombuffer(cb.Data, np.uint8)
        bufs.append(buf)
        cbs[i] = _Cb(cb.IncludedInLayers, cb.ZeroBitPlanes, cb.Passes, dl, buf.size if (data_cap or dl) else 0, 0,
                     buf.ctypes.data if (data_cap or dl) else None)
    p = _Precinct(len(precinct.CodeBlocks), precinct.InclusionTree.width, precinct.IMSBTree.width, 0,
                  ncb.ctypes.data if ncb.size else None, C.addressof(cbs))
    return p, (ncb, cbs, bufs), flat, bufs


class PacketIterator:
    """tcd.NewPacketIterator(numComponents, numResolutions, numLayers, precincts, order); Next() -> (Packet, ok)"""

    def __init__(self, numComponents, numResolutions, numLayers, precincts, order):
        counts = np.array([int(r[0]) for c in precincts for r in c], np.int32)
        nres = np.array([len(c) for c in precincts], np.int32)
        L = _lib.lib()
        n = C.c_size_t(0)
        self._keep = (counts, nres)
        args = (int(numComponents), int(numResolutions), int(numLayers), counts.ctypes.data_as(C.c_void_p) if counts.size else None,
                nres.ctypes.data_as(C.c_void_p) if nres.size else None, len(precincts), int(order))
        st = L.j2k_t2_packet_sequence(*args, None, C.c_size_t(0), C.byref(n))
        if st not in (_lib.OK, _lib.ERR_CAPACITY):
            _check(st, "PacketIterator")
        self._seq = (_Packet * max(n.value, 1))()
        _check(L.j2k_t2_packet_sequence(*args, self._seq, C.c_size_t(n.value), C.byref(n)), "PacketIterator")
        self._n, self._i = n.value, 0

    def Next(self):
        if self._i >= self._n:
            return None, False
        p = self._seq[self._i]
        self._i += 1
        return (p.layer, p.resolution, p.component, p.precinct), True

    def Reset(self):
        self._i = 0


class PacketEncoder:
    """tcd.NewPacketEncoder(w): the bytes written to w accumulate in .buf"""

    def __init__(self):
        self.buf = bytearray()
        self._delay = C.c_uint8(0)

    def EncodePacket(self, precinct, layer, enableSOP, enableEPH):
        L = _lib.lib()
        p, keep, _, _ = _marshal(precinct)
        L.j2k_t2_packet_bound.restype = C.c_size_t
        cap = int(L.j2k_t2_packet_bound(C.byref(p))) + 16
        out = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        _check(L.j2k_t2_encode_packet(C.byref(p), int(layer), int(bool(enableSOP)), int(bool(enableEPH)), C.byref(self._delay),
                                      out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n)), "EncodePacket")
        self.buf += out[:n.value].tobytes()


class _DevCb(C.Structure):
    _fields_ = [("included_in_layers", C.c_int32), ("zero_bit_planes", C.c_int32), ("num_passes", C.c_int32), ("data_len", C.c_uint32),
                ("data_off", C.c_uint64)]


class _DevPacket(C.Structure):
    _fields_ = [("layer", C.c_int32), ("incl_tree_w", C.c_int32), ("imsb_tree_w", C.c_int32), ("flags", C.c_int32), ("cb0", C.c_int64),
                ("ncb", C.c_int64)]


DEV_CB_DTYPE = np.dtype([("included_in_layers", "<i4"), ("zero_bit_planes", "<i4"), ("num_passes", "<i4"), ("data_len", "<u4"), ("data_off", "<u8")])
DEV_PACKET_DTYPE = np.dtype([("layer", "<i4"), ("incl_tree_w", "<i4"), ("imsb_tree_w", "<i4"), ("flags", "<i4"), ("cb0", "<i8"), ("ncb", "<i8")])


class DevicePacketEncoder:
    """One tcd.PacketEncoder whose EncodePacket calls are made a RUN at a time on device buffers (csrc/t2dev.hip,
    j2k_t2_encode_packets_device): the byte-stuffing writer's state carries from run to run as it does from packet to packet."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._delay = C.c_uint8(0)

    def tables(self, runs, flags=0):
        """numpy tables for a run given as [(Precinct, layer), ...]: (packets, cbs, data) with every block's bytes appended to data.
        flags: J2K_T2_* of the closed-loop mode for every packet (an int, or one per packet)"""
        packets = np.zeros(len(runs), DEV_PACKET_DTYPE)
        cbs, data = [], bytearray()
        for i, (pr, layer) in enumerate(runs):
            flat = [cb for band in pr.CodeBlocks for cb in band]
            packets[i] = (int(layer), pr.InclusionTree.width, pr.IMSBTree.width, int(flags if np.isscalar(flags) else flags[i]), len(cbs), len(flat))
            for cb in flat:
                d = cb.Data or b""
                cbs.append((cb.IncludedInLayers, cb.ZeroBitPlanes, cb.Passes, len(d), len(data)))
                data += d
        return packets, np.array(cbs, DEV_CB_DTYPE) if cbs else np.zeros(0, DEV_CB_DTYPE), np.frombuffer(bytes(data), np.uint8)

    def encode(self, d_packets, npackets, d_cbs, d_data, enableSOP, enableEPH, d_out, d_offs):
        """device tensors in (uint8 views of the tables), packets out into d_out at d_offs (int64 / uint64 [npackets + 1]); returns the total"""
        L = self.ctx.L
        total = C.c_size_t(0)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None      # noqa: E731
        st = L.j2k_t2_encode_packets_device(self.ctx.h, ptr(d_packets), C.c_size_t(int(npackets)), ptr(d_cbs),
                                            C.c_size_t(int(d_cbs.numel()) // 24 if d_cbs is not None else 0), ptr(d_data), int(bool(enableSOP)),
                                            int(bool(enableEPH)), C.byref(self._delay), ptr(d_out), C.c_size_t(int(d_out.numel()) if d_out is not None else 0),
                                            ptr(d_offs), C.byref(total))
        self.total = total.value
        self.ctx.check(st)
        return total.value


class DevicePacketDecoder:
    """One tcd.PacketDecoder whose DecodePacket calls are made a RUN at a time on a device buffer (csrc/t2dec.hip,
    j2k_t2_decode_packets_device): the decoder's state (Position(), the header bit reader) carries from run to run."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._st = _DecState()
        self.done = 0

    def decode(self, d_packets, npackets, d_cbs, d_data, sopEnabled, ephEnabled):
        """device tensors (uint8 views of the packet table, the code-block table -- read and written -- and the buffer the
        decoder was made on); returns the packets decoded; raises the first failing packet's J2KError after setting .done"""
        L = self.ctx.L
        done = C.c_size_t(0)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None      # noqa: E731
        st = L.j2k_t2_decode_packets_device(self.ctx.h, ptr(d_packets), C.c_size_t(int(npackets)), ptr(d_cbs),
                                            C.c_size_t(int(d_cbs.numel()) // 24 if d_cbs is not None else 0), ptr(d_data),
                                            C.c_size_t(int(d_data.numel()) if d_data is not None else 0), int(bool(sopEnabled)),
                                            int(bool(ephEnabled)), C.byref(self._st), C.byref(done))
        self.done = done.value
        self.ctx.check(st)
        return done.value

    def Position(self):
        return int(self._st.pos)


class PacketDecoder:
    """tcd.NewPacketDecoder(data)"""

    def __init__(self, data):
        self._data = np.frombuffer(bytes(data), np.uint8).copy()
        self._st = _DecState()

    def DecodePacket(self, precinct, layer, sopEnabled, ephEnabled):
        p, keep, flat, bufs = _marshal(precinct, data_cap=256)       # a decoded length has at most 7 bits (t2.go:633-648)
        st = _lib.lib().j2k_t2_decode_packet(self._data.ctypes.data_as(C.c_void_p) if self._data.size else None,
                                             C.c_size_t(self._data.size), C.byref(self._st), C.byref(p), int(layer),
                                             int(bool(sopEnabled)), int(bool(ephEnabled)))
        cbs = keep[1]
        for i, cb in enumerate(flat):                               # what was decoded before an error stays, as in Go
            cb.IncludedInLayers, cb.ZeroBitPlanes, cb.Passes = cbs[i].included_in_layers, cbs[i].zero_bit_planes, cbs[i].num_passes
            had = cb.Data is not None
            if cbs[i].data_len or had:
                cb.Data = bufs[i][:cbs[i].data_len].tobytes()
        _check(st, "DecodePacket")

    def Position(self):
        return int(self._st.pos)


def init_tile(header, tile_index):
    """TileDecoder.InitTile(tileIndex) for the header fields it reads (dict: ImageWidth, ImageHeight, ImageXOffset, ImageYOffset,
    TileWidth, TileHeight, TileXOffset, TileYOffset, NumTilesX, NumDecompositions, CodeBlockWidthExp, CodeBlockHeightExp,
    Subsampling = [(sx, sy)] per component).  Returns (tile bounds, components): component = (bounds, resolutions),
    resolution = (level, bounds, bands), band = (type, bounds, CodeBlocksX, CodeBlocksY, [code-block bounds])."""
    sub = np.array([v for s in header["Subsampling"] for v in s], np.uint8)
    nc = len(header["Subsampling"])
    nd = int(header["NumDecompositions"])
    if not 0 <= nd <= 255:
        raise ValueError("NumDecompositions is a uint8")
    h = _Header(header["ImageWidth"], header["ImageHeight"], header["ImageXOffset"], header["ImageYOffset"], header["TileWidth"],
                header["TileHeight"], header["TileXOffset"], header["TileYOffset"], header["NumTilesX"], nc,
                sub.ctypes.data if sub.size else None, nd, header["CodeBlockWidthExp"], header["CodeBlockHeightExp"])
    L = _lib.lib()
    tile = _Rect()
    comps = (_Rect * max(nc, 1))()
    ress = (_Rect * max(nc * (min(nd, 32) + 1), 1))()
    nb, ncb = C.c_size_t(0), C.c_size_t(0)
    st = L.j2k_tcd_init_tile(C.byref(h), int(tile_index), C.byref(tile), comps, ress, None, C.c_size_t(0), C.byref(nb), None,
                             C.c_size_t(0), C.byref(ncb))
    if st not in (_lib.OK, _lib.ERR_CAPACITY):
        _check(st, "InitTile")
    bands = (_Band * max(nb.value, 1))()
    cbs = (_Rect * max(ncb.value, 1))()
    _check(L.j2k_tcd_init_tile(C.byref(h), int(tile_index), C.byref(tile), comps, ress, bands, C.c_size_t(nb.value), C.byref(nb), cbs,
                               C.c_size_t(ncb.value), C.byref(ncb)), "InitTile")
    out = []
    bi = 0
    for c in range(nc):
        rl = []
        for r in range(nd + 1):
            bl = []
            for _ in range(1 if r == 0 else 3):
                b = bands[bi]
                bi += 1
                bl.append((b.type, b.r.t(), b.cbx, b.cby, [cbs[b.cb0 + i].t() for i in range(b.cbx * b.cby)]))
            rl.append((r, ress[c * (nd + 1) + r].t(), bl))
        out.append((comps[c].t(), rl))
    return tile.t(), out
