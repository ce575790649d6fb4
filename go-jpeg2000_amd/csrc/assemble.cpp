// assemble.cpp -- multi-tile codestream assembly around the gathered block bytes.  Host code only.
//
// Replaces (reference, mrjoshuak/go-jpeg2000):
//   encoder.createTileHeader                 encoder.go:746-760   (SOT Lsot=10 Isot Psot=14+len TPsot=0 TNsot=1, SOD, data)
//   the tile loop encoder.generateTiles would need for more than one tile (encoder.go:568-579 codes tile 0 only)
//   codestream.Parser.ReadTilePartHeader     internal/codestream/parser.go:894-983 (+ skipMarkerSegment :180-190)
// The reference encoder writes ONE tile (the whole image); the plan calls of this library code many tiles per frame and,
// on N > 1 GPUs, gather them on rank 0 -- j2k_assemble_tiles is where the gathered packets become tile-parts, each exactly
// what createTileHeader(tileIdx, tileData) returns, in tile order.
#include <stdint.h>
#include <string.h>
#include "../../include/j2kgfx.h"

namespace {
enum : uint16_t { M_SOT = 0xFF90, M_SOD = 0xFF93, M_EOC = 0xFFD9 };     // internal/codestream/markers.go:8-11
inline void be16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
inline void be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }
inline uint32_t rd32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
}  // namespace

extern "C" size_t j2k_tile_part_bound(const uint64_t *tile_offs, int ntiles) {
    if (!tile_offs || ntiles <= 0) return 0;
    return (size_t)(tile_offs[ntiles] - tile_offs[0]) + (size_t)14 * (size_t)ntiles;
}

extern "C" int j2k_create_tile_header(int tile_idx, const uint8_t *tile_data, size_t len, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out || (len && !tile_data) || !out_len) return J2K_ERR_INVALID_ARG;
    *out_len = 14 + len;
    if (cap < 14 + len) return J2K_ERR_CAPACITY;
    be16(out + 0, M_SOT);
    be16(out + 2, 10);                                  // sotLength
    be16(out + 4, (uint16_t)tile_idx);                  // Go: uint16(tileIdx) truncates
    be32(out + 6, (uint32_t)(14 + len));                // Go: uint32(14 + len(tileData)) truncates
    out[10] = 0;                                        // tile-part index
    out[11] = 1;                                        // number of tile-parts
    be16(out + 12, M_SOD);
    if (len) memcpy(out + 14, tile_data, len);
    return J2K_OK;
}

extern "C" int j2k_assemble_tiles(const uint8_t *stream, const uint64_t *tile_offs, int tile_first, int ntiles, uint8_t *out, size_t cap,
                                  size_t *out_len) {
    if (!tile_offs || ntiles < 0 || !out_len || (ntiles && (!stream || !out))) return J2K_ERR_INVALID_ARG;
    for (int t = 0; t < ntiles; t++)
        if (tile_offs[t + 1] < tile_offs[t]) return J2K_ERR_INVALID_ARG;
    const size_t need = j2k_tile_part_bound(tile_offs, ntiles);
    *out_len = need;
    if (cap < need) return J2K_ERR_CAPACITY;
    size_t pos = 0;
    for (int t = 0; t < ntiles; t++) {
        size_t n = 0;
        const size_t len = (size_t)(tile_offs[t + 1] - tile_offs[t]);
        int r = j2k_create_tile_header(tile_first + t, stream + tile_offs[t], len, out + pos, cap - pos, &n);
        if (r != J2K_OK) return r;
        pos += n;
    }
    return J2K_OK;
}

// One tile-part: ReadTilePartHeader's fields (parser.go:894-983) plus where its data lies.  `pos` points AT the SOT marker.
extern "C" int j2k_read_tile_part_header(const uint8_t *cs, size_t len, size_t pos, j2k_tile_part *tp) {
    if (!cs || !tp) return J2K_ERR_INVALID_ARG;
    memset(tp, 0, sizeof *tp);
    if (pos + 2 > len || rd16(cs + pos) != M_SOT) return J2K_ERR_INVALID_ARG;
    size_t p = pos + 2;
    if (p + 10 > len) return J2K_ERR_INVALID_ARG;                     // io.ReadFull fails: unexpected EOF
    if (rd16(cs + p) != 10) return J2K_ERR_INVALID_ARG;                // "invalid SOT length"
    tp->tile_index = rd16(cs + p + 2);
    tp->tile_part_length = rd32(cs + p + 4);
    tp->tile_part_index = cs[p + 8];
    tp->num_tile_parts = cs[p + 9];
    p += 10;
    tp->header_off = (uint64_t)p;                                      // first marker after the SOT segment
    for (;;) {                                                         // tile-part header markers until SOD (parser.go:933-982)
        if (p + 2 > len) return J2K_ERR_INVALID_ARG;
        const uint16_t m = rd16(cs + p);
        p += 2;
        if (m == M_SOD) break;
        // COD / COC / QCD / QCC / POC / PPT and everything else: a length-prefixed segment (skipMarkerSegment, parser.go:180-190);
        // their contents stay with the Go-side parser -- this call only locates the data
        if (p + 2 > len) return J2K_ERR_INVALID_ARG;
        const uint16_t seg = rd16(cs + p);
        if (seg < 2 || p + seg > len) return J2K_ERR_INVALID_ARG;     // "invalid marker segment length" / EOF
        p += seg;
        tp->header_markers++;
    }
    tp->data_off = (uint64_t)p;
    const uint64_t hdr = (uint64_t)(p - pos);
    if (tp->tile_part_length == 0) tp->data_len = (uint64_t)(len - p);             // Psot = 0: to the end of the codestream
    else if (tp->tile_part_length < hdr || pos + tp->tile_part_length > len) return J2K_ERR_INVALID_ARG;
    else tp->data_len = tp->tile_part_length - hdr;
    return J2K_OK;
}

extern "C" int j2k_parse_tile_parts(const uint8_t *cs, size_t len, j2k_tile_part *parts, size_t cap, size_t *nparts) {
    if (!cs || !nparts || (cap && !parts)) return J2K_ERR_INVALID_ARG;
    size_t pos = 0, n = 0;
    while (pos < len) {
        if (pos + 2 <= len && rd16(cs + pos) == M_EOC) break;
        j2k_tile_part tp;
        int r = j2k_read_tile_part_header(cs, len, pos, &tp);
        if (r != J2K_OK) { *nparts = n; return r; }
        if (n < cap) parts[n] = tp;
        n++;
        pos = (size_t)(tp.data_off + tp.data_len);
    }
    *nparts = n;
    return n > cap ? J2K_ERR_CAPACITY : J2K_OK;
}
