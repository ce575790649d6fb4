// t2.cpp -- Tier-2 packet coding and tile geometry behind the C ABI (host only).
//
// Replaces (reference, mrjoshuak/go-jpeg2000), as written there:
//   tcd.PacketIterator / NewPacketIterator / Next / Reset        internal/tcd/t2.go:18-238
//   tcd.PacketEncoder.EncodePacket + header / value coders        internal/tcd/t2.go:241-438
//   tcd.PacketDecoder.DecodePacket + header / value decoders      internal/tcd/t2.go:439-652
//   bio.ByteStuffingWriter / ByteStuffingReader                   internal/bio/bio.go:105-226
//   tcd.NewTagTree (shape)                                        internal/tcd/tcd.go:168-197
//   tcd.TileDecoder.InitTile / initResolution / initBand          internal/tcd/tcd.go:240-390
//
// Organised for batch use rather than as the reference's objects: the iterator is a closed-form walk over counters that
// emits the whole sequence, the bit writer is a byte sink with the 7-bit rule folded into one "room in this byte"
// counter, the tile geometry comes back as flat tables.  The reference's quirks are kept (see include/j2kgfx.h): they
// are what parity means here.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/j2kgfx.h"

namespace {

// ---- bit sink / source with byte stuffing (bio.go:105-226) ----------------------------------------------------------
struct StuffedSink {
    uint8_t *out; size_t cap, n = 0;
    bool overflow = false;
    unsigned acc = 0, have = 0;   // bits gathered for the current byte
    bool after_ff;                // the byte in progress may hold only 7 bits
    StuffedSink(uint8_t *o, size_t c, bool delay) : out(o), cap(c), after_ff(delay) {}
    void emit(uint8_t b) { if (n < cap) out[n] = b; else overflow = true; n++; }
    void raw(const uint8_t *p, size_t len) {
        if (len && n + len <= cap) memcpy(out + n, p, len); else if (len) overflow = true;
        n += len;
    }
    void byte_done() { emit((uint8_t)acc); after_ff = (acc & 0xFF) == 0xFF; acc = 0; have = 0; }
    void bit(unsigned b) {
        acc = ((acc << 1) | (b & 1)) & 0xFF;
        if (++have == (after_ff ? 7u : 8u)) byte_done();
    }
    void bits(uint32_t v, unsigned count) {        // MSB first; positions above bit 31 are zero (uint32 shifted out)
        for (unsigned i = count; i > 0; i--) bit(i - 1 < 32 ? (v >> (i - 1)) & 1 : 0);
    }
    void unary(int value) {                        // "simplified tag tree": value zeros, then a one (t2.go:368-377)
        for (int i = 0; i < value; i++) bit(0);
        bit(1);
    }
    void flush() {                                 // pad the byte in progress with zeros (bio.go:211-221)
        if (have) { acc = (acc << ((after_ff ? 7u : 8u) - have)) & 0xFF; byte_done(); }
    }
};

struct StuffedSource {
    const uint8_t *data; size_t len;
    j2k_t2_dec_state &st;
    bool eof = false;
    StuffedSource(const uint8_t *d, size_t l, j2k_t2_dec_state &s) : data(d), len(l), st(s) {}
    unsigned bit() {
        if (st.cnt == 0) {
            if (st.rpos >= len) { eof = true; return 0; }
            const uint8_t b = data[st.rpos++];
            st.cnt = st.saw_ff ? 7 : 8;
            st.saw_ff = b == 0xFF;
            st.buf = b;
        }
        st.cnt--;
        return (st.buf >> st.cnt) & 1;
    }
    uint32_t bits(unsigned count) {
        uint32_t r = 0;
        for (unsigned i = 0; i < count && !eof; i++) r = (r << 1) | bit();
        return r;
    }
    int unary() {
        int v = 0;
        while (!eof && bit() == 0 && !eof) v++;
        return v;
    }
};

inline bool contributes(const j2k_t2_cb &cb, int layer) { return cb.included_in_layers <= layer && cb.data_len > 0; }

size_t total_cbs(const j2k_t2_precinct *p) {
    size_t n = 0;
    for (int b = 0; b < p->nbands; b++) n += (size_t)(p->band_ncb[b] > 0 ? p->band_ncb[b] : 0);
    return n;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// packet sequence
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int j2k_t2_packet_sequence(int ncomp, int nres, int nlayers, const int32_t *prec_counts, const int32_t *prec_nres,
                                      int prec_ncomp, int order, j2k_packet *out, size_t cap, size_t *count) {
    if (!count || (cap && !out) || prec_ncomp < 0 || (prec_ncomp > 0 && !prec_nres)) return J2K_ERR_INVALID_ARG;
    *count = 0;
    if (order < 0 || order > 4) return J2K_OK;                     // hasMore's default branch: no packet at all
    // row starts of the ragged precinct table
    std::vector<size_t> row((size_t)prec_ncomp + 1, 0);
    for (int c = 0; c < prec_ncomp; c++) {
        if (prec_nres[c] < 0) return J2K_ERR_INVALID_ARG;
        row[c + 1] = row[c] + (size_t)prec_nres[c];
    }
    if (row[prec_ncomp] > 0 && !prec_counts) return J2K_ERR_INVALID_ARG;
    auto nprec = [&](int c, int r) -> int {                        // t2.go:133-136 and its four copies
        if (c >= 0 && c < prec_ncomp && r >= 0 && r < prec_nres[c]) return prec_counts[row[c] + r];
        return 1;
    };
    int maxp = 0;                                                  // maxPrecincts (t2.go:102-114): over the nominal grid only
    for (int c = 0; c < ncomp; c++)
        for (int r = 0; r < nres; r++)
            if (c < prec_ncomp && r < prec_nres[c] && prec_counts[row[c] + r] > maxp) maxp = prec_counts[row[c] + r];
    int L = 0, R = 0, C = 0, P = 0;                                // the iterator's position; all starts are 0 (unexported)
    size_t n = 0;
    // Every order is the same odometer with a different digit order and a different "is there more" digit; the precinct
    // digit's range is looked up when it is incremented, at the (component, resolution) current at that moment.
    for (;;) {
        bool more;
        switch (order) {
        case 0: more = L < nlayers; break;
        case 1: case 2: more = R < nres; break;
        case 3: more = P < maxp; break;
        default: more = C < ncomp; break;
        }
        if (!more) break;
        if (n < cap) out[n] = j2k_packet{L, R, C, P};
        n++;
        if (n > ((size_t)1 << 32)) return J2K_ERR_INVALID_ARG;     // a degenerate input that never terminates in the reference either
        switch (order) {
        case 0:   // LRCP: precinct, component, resolution, layer
            if (++P >= nprec(C, R)) { P = 0; if (++C >= ncomp) { C = 0; if (++R >= nres) { R = 0; L++; } } }
            break;
        case 1:   // RLCP: precinct, component, layer, resolution
            if (++P >= nprec(C, R)) { P = 0; if (++C >= ncomp) { C = 0; if (++L >= nlayers) { L = 0; R++; } } }
            break;
        case 2:   // RPCL: layer, component, precinct, resolution
            if (++L >= nlayers) { L = 0; if (++C >= ncomp) { C = 0; if (++P >= nprec(C, R)) { P = 0; R++; } } }
            break;
        case 3:   // PCRL: layer, resolution, component, precinct (unbounded here: hasMore stops it)
            if (++L >= nlayers) { L = 0; if (++R >= nres) { R = 0; if (++C >= ncomp) { C = 0; P++; } } }
            break;
        default:  // CPRL: layer, resolution, precinct, component
            if (++L >= nlayers) { L = 0; if (++R >= nres) { R = 0; if (++P >= nprec(C, R)) { P = 0; C++; } } }
            break;
        }
    }
    *count = n;
    return n > cap ? J2K_ERR_CAPACITY : J2K_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// packet encoder
// ---------------------------------------------------------------------------------------------------------------------
extern "C" size_t j2k_t2_packet_bound(const j2k_t2_precinct *p) {
    if (!p || p->nbands < 0 || (p->nbands && (!p->band_ncb || !p->cbs))) return 0;
    // markers 8; header: presence bit + per block (inclusion + zero bit-planes in unary) + 16 pass bits + 3 + 32 length bits,
    // 7 payload bits per byte in the worst case; bodies
    size_t bits = 1, body = 0;
    const size_t n = total_cbs(p);
    for (size_t i = 0; i < n; i++) {
        const j2k_t2_cb &cb = p->cbs[i];
        bits += (size_t)(cb.included_in_layers > 0 ? cb.included_in_layers : 0) + (size_t)(cb.zero_bit_planes > 0 ? cb.zero_bit_planes : 0) + 2 + 16 + 35;
        body += cb.data_len;
    }
    return 8 + bits / 7 + 2 + body;
}

extern "C" int j2k_t2_encode_packet(const j2k_t2_precinct *p, int layer, int sop, int eph, uint8_t *bio_delay, uint8_t *out,
                                    size_t cap, size_t *len) {
    if (!p || !len || !bio_delay || (cap && !out) || p->nbands < 0 || (p->nbands && !p->band_ncb)) return J2K_ERR_INVALID_ARG;
    const size_t n = total_cbs(p);
    if (n && !p->cbs) return J2K_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; i++)
        if (p->cbs[i].data_len && !p->cbs[i].data) return J2K_ERR_INVALID_ARG;
    StuffedSink w(out, cap, *bio_delay != 0);
    if (sop) {                                                     // t2.go:257-263
        const uint8_t m[6] = {0xFF, 0x91, 0x00, 0x04, (uint8_t)((unsigned)layer >> 8), (uint8_t)layer};
        w.raw(m, 6);
    }
    bool any = false;
    for (size_t i = 0; i < n && !any; i++) any = contributes(p->cbs[i], layer);
    if (!any) {
        w.bit(0);                                                  // empty packet: one zero bit, padded (t2.go:314-319)
    } else {
        w.bit(1);
        size_t i = 0;
        for (int b = 0; b < p->nbands; b++)
            for (int k = 0; k < p->band_ncb[b]; k++, i++) {
                const j2k_t2_cb &cb = p->cbs[i];
                const bool inc = contributes(cb, layer);
                if (layer == 0) {
                    if (p->incl_tree_w == 0) return J2K_ERR_GO_PANIC;     // cbIdx % width
                    w.unary(cb.included_in_layers);                        // written whether or not the block is included
                } else {
                    w.bit(inc ? 1 : 0);
                }
                if (!inc) continue;
                if (cb.included_in_layers == layer) {
                    if (p->imsb_tree_w == 0) return J2K_ERR_GO_PANIC;
                    w.unary(cb.zero_bit_planes);
                }
                const int np = cb.num_passes;                              // t2.go:379-406
                if (np == 1) w.bit(0);
                else if (np == 2) w.bits(2, 2);
                else if (np <= 5) { w.bits(3, 2); w.bits((uint32_t)(np - 3), 2); }
                else if (np <= 36) { w.bits(15, 4); w.bits((uint32_t)(np - 6), 5); }
                else { w.bits(0x1FF, 9); w.bits((uint32_t)(np - 37), 7); }
                unsigned nb = 0;                                           // t2.go:408-437: bit length in 3 bits (it wraps), then the length
                for (uint32_t t = cb.data_len; t; t >>= 1) nb++;
                w.bits(nb, 3);
                w.bits(cb.data_len, nb);
            }
    }
    w.flush();
    *bio_delay = w.after_ff ? 1 : 0;
    if (eph) { const uint8_t m[2] = {0xFF, 0x92}; w.raw(m, 2); }
    for (size_t i = 0; i < n; i++)
        if (contributes(p->cbs[i], layer)) w.raw(p->cbs[i].data, p->cbs[i].data_len);
    *len = w.n;
    return w.overflow ? J2K_ERR_CAPACITY : J2K_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// packet decoder
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int j2k_t2_decode_packet(const uint8_t *data, size_t len, j2k_t2_dec_state *st, j2k_t2_precinct *p, int layer, int sop,
                                    int eph) {
    if (!st || !p || (len && !data) || p->nbands < 0 || (p->nbands && !p->band_ncb)) return J2K_ERR_INVALID_ARG;
    const size_t n = total_cbs(p);
    if (n && !p->cbs) return J2K_ERR_INVALID_ARG;
    if (sop && st->pos + 6 <= len && data[st->pos] == 0xFF && data[st->pos + 1] == 0x91) st->pos += 6;   // t2.go:470-474
    StuffedSource r(data, len, *st);
    if (r.bit() == 1 && !r.eof) {
        size_t i = 0;
        for (int b = 0; b < p->nbands; b++)
            for (int k = 0; k < p->band_ncb[b]; k++, i++) {
                j2k_t2_cb &cb = p->cbs[i];
                bool inc;
                if (layer == 0) {
                    if (p->incl_tree_w == 0) return J2K_ERR_GO_PANIC;
                    const int v = r.unary();
                    if (r.eof) return J2K_ERR_INVALID_ARG;
                    inc = v == layer;
                    cb.included_in_layers = v;
                } else {
                    inc = r.bit() == 1;
                    if (r.eof) return J2K_ERR_INVALID_ARG;
                    if (inc) cb.included_in_layers = layer;
                }
                if (!inc) continue;
                if (cb.included_in_layers == layer) {
                    if (p->imsb_tree_w == 0) return J2K_ERR_GO_PANIC;
                    const int v = r.unary();
                    if (r.eof) return J2K_ERR_INVALID_ARG;
                    cb.zero_bit_planes = v;
                }
                int np;                                                    // t2.go:592-631
                if (r.bit() == 0) np = 1;
                else if (r.bit() == 0) np = 2;
                else {
                    uint32_t v = r.bits(2);
                    if (v < 3) np = (int)v + 3;
                    else {
                        v = r.bits(5);
                        if (v < 31) np = (int)v + 6;
                        else np = (int)r.bits(7) + 37;
                    }
                }
                if (r.eof) return J2K_ERR_INVALID_ARG;
                const uint32_t nb = r.bits(3);                             // t2.go:633-648
                const uint32_t length = nb ? r.bits(nb) : 0;
                if (r.eof) return J2K_ERR_INVALID_ARG;
                cb.num_passes = np;
                if (length > cb.data_cap || (length && !cb.data)) return J2K_ERR_CAPACITY;
                cb.data_len = length;                                      // make([]byte, length): zeros until the body is copied
                if (length) memset(cb.data, 0, length);
            }
    } else if (r.eof) {
        return J2K_ERR_INVALID_ARG;                                        // no presence bit to read
    }
    if (eph && st->pos + 2 <= len && data[st->pos] == 0xFF && data[st->pos + 1] == 0x92) st->pos += 2;
    for (size_t i = 0; i < n; i++) {                                       // bodies from Position(), not from behind the header
        j2k_t2_cb &cb = p->cbs[i];
        if (cb.included_in_layers == layer && cb.data_len > 0) {
            if (st->pos + cb.data_len > len) return J2K_ERR_INVALID_ARG;
            if (!cb.data) return J2K_ERR_INVALID_ARG;
            memcpy(cb.data, data + st->pos, cb.data_len);
            st->pos += cb.data_len;
        }
    }
    return J2K_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// tag tree shape, tile geometry
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int j2k_tagtree_shape(int width, int height, int32_t *levels, int64_t *level_sizes, size_t cap) {
    if (!levels) return J2K_ERR_INVALID_ARG;
    int64_t w = width, h = height;
    int nl = 1;
    for (int64_t a = w, b = h; a > 1 || b > 1; a = (a + 1) / 2, b = (b + 1) / 2) nl++;
    *levels = nl;
    if ((size_t)nl > cap) return level_sizes || cap ? J2K_ERR_CAPACITY : J2K_OK;
    for (int l = 0; l < nl; l++) {
        if (w * h < 0) return J2K_ERR_GO_PANIC;                            // make([]tagNode, negative)
        level_sizes[l] = w * h;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    return J2K_OK;
}

extern "C" int j2k_tcd_init_tile(const j2k_tcd_header *h, int tile_index, j2k_tcd_rect *tile, j2k_tcd_rect *comps, j2k_tcd_rect *ress,
                                 j2k_tcd_band *bands, size_t band_cap, size_t *nbands, j2k_tcd_rect *cbs, size_t cb_cap, size_t *ncbs) {
    if (!h || !tile || !nbands || !ncbs || h->ncomp < 0 || tile_index < 0 || (h->ncomp && (!h->subsampling || !comps || !ress)))
        return J2K_ERR_INVALID_ARG;
    if (h->num_tiles_x == 0) return J2K_ERR_GO_PANIC;
    if (h->num_decompositions > 32 || h->cb_w_exp > 28 || h->cb_h_exp > 28) return J2K_ERR_GO_PANIC;
    auto cdiv = [](int64_t a, int64_t b) { return (a + b - 1) / b; };
    const int64_t tx = tile_index % (int64_t)h->num_tiles_x, ty = tile_index / (int64_t)h->num_tiles_x;
    const int64_t x0 = std::max<int64_t>((int64_t)h->tile_x0 + tx * h->tile_w, h->image_x0), y0 = std::max<int64_t>((int64_t)h->tile_y0 + ty * h->tile_h, h->image_y0);
    const int64_t x1 = std::min<int64_t>((int64_t)h->tile_x0 + (tx + 1) * h->tile_w, h->image_w), y1 = std::min<int64_t>((int64_t)h->tile_y0 + (ty + 1) * h->tile_h, h->image_h);
    auto fits = [](int64_t v) { return v >= INT32_MIN && v <= INT32_MAX; };
    if (!fits(x0) || !fits(y0) || !fits(x1) || !fits(y1)) return J2K_ERR_UNSUPPORTED;
    *tile = j2k_tcd_rect{(int32_t)x0, (int32_t)y0, (int32_t)x1, (int32_t)y1};
    const int nd = h->num_decompositions;
    const int64_t cbw = (int64_t)1 << (h->cb_w_exp + 2), cbh = (int64_t)1 << (h->cb_h_exp + 2);
    size_t nb = 0, nc = 0;
    for (int c = 0; c < h->ncomp; c++) {
        const int64_t sx = h->subsampling[2 * c], sy = h->subsampling[2 * c + 1];
        if (sx == 0 || sy == 0) return J2K_ERR_GO_PANIC;
        const int64_t cx0 = cdiv(x0, sx), cy0 = cdiv(y0, sy), cx1 = cdiv(x1, sx), cy1 = cdiv(y1, sy);
        if ((cx1 - cx0) * (cy1 - cy0) < 0) return J2K_ERR_GO_PANIC;       // make([]int32, negative)
        comps[c] = j2k_tcd_rect{(int32_t)cx0, (int32_t)cy0, (int32_t)cx1, (int32_t)cy1};
        for (int r = 0; r <= nd; r++) {
            const int64_t scale = (int64_t)1 << (nd - r);
            const int64_t rx0 = cdiv(cx0, scale), ry0 = cdiv(cy0, scale), rx1 = cdiv(cx1, scale), ry1 = cdiv(cy1, scale);
            ress[(size_t)c * (nd + 1) + r] = j2k_tcd_rect{(int32_t)rx0, (int32_t)ry0, (int32_t)rx1, (int32_t)ry1};
            const int64_t mx = (rx0 + rx1) / 2, my = (ry0 + ry1) / 2;
            for (int bt = (r == 0 ? 0 : 1); bt <= (r == 0 ? 0 : 3); bt++) {
                int64_t b0x = rx0, b0y = ry0, b1x = rx1, b1y = ry1;        // tcd.go:343-361, as written
                if (bt == 1) b1y = my;
                else if (bt == 2) b1x = mx;
                else if (bt == 3) { b0x = mx; b0y = my; }
                const int64_t gx = cdiv(b1x - b0x, cbw), gy = cdiv(b1y - b0y, cbh);
                if (gx * gy < 0) return J2K_ERR_GO_PANIC;
                if (nb < band_cap) {
                    bands[nb] = j2k_tcd_band{c, r, bt, (int32_t)gx, (int32_t)gy, 0,
                                             j2k_tcd_rect{(int32_t)b0x, (int32_t)b0y, (int32_t)b1x, (int32_t)b1y}, (uint64_t)nc};
                }
                nb++;
                for (int64_t i = 0; i < gx * gy; i++, nc++) {
                    if (nc >= cb_cap) continue;
                    const int64_t ix = i % gx, iy = i / gx;
                    cbs[nc] = j2k_tcd_rect{(int32_t)(b0x + ix * cbw), (int32_t)(b0y + iy * cbh),
                                           (int32_t)std::min(b0x + (ix + 1) * cbw, b1x), (int32_t)std::min(b0y + (iy + 1) * cbh, b1y)};
                }
            }
        }
    }
    *nbands = nb; *ncbs = nc;
    return (nb > band_cap || nc > cb_cap) ? J2K_ERR_CAPACITY : J2K_OK;
}
