// t2dec.hip -- Tier-2 packet DECODING on device buffers and the rest of the decode body:
//   t2_decode_kernel      PacketDecoder.DecodePacket (internal/tcd/t2.go:463-652) for chains of packets, one decoder object (one
//                         wavefront) per chain -- the reference's decoder as it is written, or, with the J2K_T2_* flags of the
//                         closed-loop mode (include/j2kgfx.h), one that can read what the encoder wrote
//   t2_tile_chains_kernel the tile-parts of a frame (SOT ... SOD, codestream.Parser.ReadTilePartHeader, parser.go:894-983) as chains
//   t2_blocks_kernel      the decoded code-block fields as the block decoder's inputs (offsets, lengths, bit-plane counts)
//   place_blocks_kernel   decoded blocks -> their windows of the coefficient planes (the step decoder.decodeTile leaves out,
//                         decoder.go:375-411)
// What is serial and what is not.  A packet header is a prefix code: the place of every field depends on every field before it,
// and the place of the NEXT packet on the lengths this header carries -- one chain per decoder object, i.e. per tile in the
// closed-loop mode (a new PacketDecoder per tile, as the encoder side takes a new PacketEncoder) and per run in the reference's
// mode (its header reader runs over the whole buffer on its own).  A chain is parsed by one wavefront with every lane holding the
// same state (wave-uniform: the instruction stream of one lane, and all 64 there for the refills of the LDS window the bytes are
// read through); the chains of a frame run side by side.  Nothing is copied: a block's body stays where it is and the block
// decoder reads it from there.
#include "j2k_internal.h"

namespace j2k {

#define T2D_WIN 1024                    // bytes of the chain that the LDS window holds (a header of a C2 tile: 10 ... 250 bytes)

// bio.ByteStuffingReader (bio.go:105-155) over a window of the buffer: a byte is taken when a bit is asked for and none is left
// -- never earlier, so that (rpos, cnt, buf, sawFF) are the reference's at every field boundary.  Wave-uniform.
struct T2Rd {
    const uint8_t *data;                // d_data; positions are offsets from it
    uint64_t end;                       // the decoder's buffer is data[0, end)
    uint64_t *win;                      // LDS, T2D_WIN / 8 words
    uint64_t wbase;                     // ADDRESS (8-byte aligned) of win[0]; ~0 = nothing loaded
    uint64_t cw, cwa;                   // the window word the reader is in, and its address (~0: none): one LDS read per eight bytes
    uint64_t rpos;                      // next byte the reader takes
    uint64_t acc;                       // its low `have` bits are unread, oldest on top
    uint32_t have;                      // < 8 at every field boundary (bio's cnt)
    uint32_t lastb;                     // the byte they come from (bio's buf)
    bool saw_ff, eof;
    int lane;

    __device__ __forceinline__ void refill(uint64_t addr) {
        __syncthreads();                                            // (one wavefront: everybody is done with the old window)
        wbase = addr & ~7ull;
        const uint64_t lim = ((uint64_t)(uintptr_t)data + end + 7) & ~7ull;      // words that overlap data[0, end) only
        for (int k = lane; k < T2D_WIN / 8; k += 64) {
            const uint64_t a = wbase + 8ull * (uint64_t)k;
            win[k] = a < lim ? *reinterpret_cast<const uint64_t *>((uintptr_t)a) : 0ull;
        }
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t byte_at(uint64_t pos) {    // pos < end
        const uint64_t a = (uint64_t)(uintptr_t)data + pos;
        if ((a & ~7ull) != cwa) {
            if (a - wbase >= T2D_WIN) refill(a);                    // (also true for a < wbase: the difference wraps)
            cwa = a & ~7ull;
            cw = win[(a - wbase) >> 3];
        }
        return (uint32_t)(cw >> (8u * ((uint32_t)a & 7u))) & 0xFFu;
    }
    __device__ __forceinline__ bool need(uint32_t n) {              // n <= 32
        while (have < n) {
            if (rpos >= end) { eof = true; return false; }
            const uint32_t b = byte_at(rpos);
            rpos++;
            const uint32_t nb = saw_ff ? 7u : 8u;                   // bio.go:127-131: behind a 0xFF byte the next one holds seven bits
            acc = (acc << nb) | (uint64_t)(b & (0xFFu >> (8u - nb)));
            have += nb;
            saw_ff = b == 0xFFu;
            lastb = b;
        }
        return true;
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {           // ReadBits(n), 0 <= n <= 32; 0 at the end of the buffer (eof set)
        if (n == 0) return 0;
        if (!need(n)) return 0;
        have -= n;
        return (uint32_t)(acc >> have) & (n < 32 ? (1u << n) - 1u : 0xFFFFFFFFu);
    }
    __device__ __forceinline__ uint32_t unary() {                   // decodeTagTreeValue (t2.go:574-590): zeros up to the first one
        uint32_t v = 0;
        for (;;) {
            if (!need(1)) return v;
            const uint64_t w = acc & ((1ull << have) - 1ull);
            if (w == 0) { v += have; have = 0; continue; }
            const uint32_t top = 63u - (uint32_t)__builtin_clzll(w);        // position of the first one among the unread bits
            v += have - 1u - top;
            have = top;
            return v;
        }
    }
};

// one decoder object's packets: [packet0, packet0 + npackets) on data[start, end); the state in and out as j2k_t2_dec_state
struct T2Chain {
    uint64_t start, end;                // the decoder's buffer within d_data (positions in the state are relative to `start`)
    int64_t packet0, npackets;
    j2k_t2_dec_state st;
    int32_t status, done;               // out: J2K_OK or the first failing packet's status; packets decoded before it
    int32_t skip, pad_;                 // != 0: the chain was found malformed before it started (status set): nothing to do
};

__global__ __launch_bounds__(64) void t2_decode_kernel(T2Chain *__restrict__ chains, const j2k_t2_dev_packet *__restrict__ packets, long npackets_all,
                                                       j2k_t2_dev_cb *__restrict__ cbs, uint64_t ncbs, const uint8_t *__restrict__ data, int sop, int eph, int clean,
                                                       uint64_t *__restrict__ body_base, int *__restrict__ frame_status) {
    __shared__ uint64_t win[T2D_WIN / 8];
    const int lane = threadIdx.x;
    T2Chain &Cn = chains[blockIdx.x];
    if (Cn.skip) { if (lane == 0 && frame_status && Cn.status) atomicMin(frame_status, Cn.status); return; }
    const uint64_t base = Cn.start;
    T2Rd r;
    r.data = data + base; r.end = Cn.end - base; r.win = win; r.wbase = ~0ull; r.cwa = ~0ull; r.cw = 0; r.lane = lane;
    r.rpos = Cn.st.rpos; r.have = Cn.st.cnt; r.lastb = Cn.st.buf; r.acc = Cn.st.buf; r.saw_ff = Cn.st.saw_ff != 0; r.eof = false;
    uint64_t pos = Cn.st.pos;
    const uint64_t end = r.end;
    int status = J2K_OK;
    int64_t done = 0;
    if (Cn.packet0 < 0 || Cn.npackets < 0 || Cn.packet0 + Cn.npackets > npackets_all) status = J2K_ERR_INVALID_ARG;
    for (int64_t k = 0; status == J2K_OK && k < Cn.npackets; k++) {
        const int64_t pk = Cn.packet0 + k;
        const j2k_t2_dev_packet P = packets[pk];
        if (P.ncb < 0 || P.cb0 < 0 || (uint64_t)P.cb0 + (uint64_t)P.ncb > ncbs) { status = J2K_ERR_INVALID_ARG; break; }
        const int layer = P.layer;
        const uint32_t lenbits = (P.flags & J2K_T2_WIDE_LEN) ? 5u : 3u;
        if (P.flags & J2K_T2_FRESH) { r.saw_ff = false; r.have = 0; }                           // NewPacketDecoder
        if (sop && pos + 6 <= end && r.byte_at(pos) == 0xFFu && r.byte_at(pos + 1) == 0x91u) pos += 6;     // t2.go:470-474
        if (P.flags & J2K_T2_SEATED) { r.rpos = pos; r.have = 0; }                              // closed-loop mode: the header starts at Position()
        uint64_t body = 0;                                                                       // bytes of the bodies this packet carries
        const uint32_t present = r.get(1);
        if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
        j2k_t2_dev_cb *pc = cbs + P.cb0;
        // The body loop (t2.go:489-499) takes bytes for every block with IncludedInLayers == layer and data -- also one this
        // header did not touch but whose fields say so from before (a table the caller filled, or an earlier packet of the same
        // layer).  `clean`: the caller zeroed the table and decodes layer 0 only, so there is no such block and nothing is read.
        const bool old_matters = !clean || layer != 0;
        if (present) {
            for (int64_t i = 0; i < P.ncb; i++) {                                                // t2.go:516-571
                j2k_t2_dev_cb old{};
                if (old_matters) old = pc[i];
                bool inc;
                int incl_layers;
                if (layer == 0) {
                    if (P.incl_tree_w == 0) { status = J2K_ERR_GO_PANIC; break; }
                    const uint32_t v = r.unary();
                    if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                    inc = v == 0;
                    incl_layers = (int)v;
                    if (lane == 0) pc[i].included_in_layers = incl_layers;
                } else {
                    inc = r.get(1) == 1;
                    if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                    incl_layers = inc ? layer : old.included_in_layers;
                    if (inc && lane == 0) pc[i].included_in_layers = layer;
                }
                if (!inc) {
                    if (incl_layers == layer && old.data_len > 0) { if (lane == 0) pc[i].data_off = body; body += old.data_len; }
                    continue;
                }
                if (incl_layers == layer) {
                    if (P.imsb_tree_w == 0) { status = J2K_ERR_GO_PANIC; break; }
                    const uint32_t v = r.unary();
                    if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                    if (lane == 0) pc[i].zero_bit_planes = (int)v;
                }
                int np;                                                                          // t2.go:592-631
                if (r.get(1) == 0) np = 1;
                else if (r.get(1) == 0) np = 2;
                else {
                    uint32_t v = r.get(2);
                    if (v < 3) np = (int)v + 3;
                    else {
                        v = r.get(5);
                        if (v < 31) np = (int)v + 6;
                        else np = (int)r.get(7) + 37;
                    }
                }
                if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                const uint32_t nb = r.get(lenbits);                                              // t2.go:633-648
                const uint32_t length = r.get(nb);
                if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                if (lane == 0) { pc[i].num_passes = np; pc[i].data_len = length; pc[i].data_off = body; }   // (offset within the packet's bodies; t2_bodies_kernel adds where they start)
                body += length;
            }
            if (status != J2K_OK) break;
        } else if (old_matters) {
            for (int64_t i = 0; i < P.ncb; i++) {
                const j2k_t2_dev_cb old = pc[i];
                if (old.included_in_layers == layer && old.data_len > 0) { if (lane == 0) pc[i].data_off = body; body += old.data_len; }
            }
        }
        if (P.flags & J2K_T2_SEATED) pos = r.rpos;                                               // ... and Position() moves past the header
        if (eph && pos + 2 <= end && r.byte_at(pos) == 0xFFu && r.byte_at(pos + 1) == 0x92u) pos += 2;     // t2.go:481-486
        if (pos + body > end) { status = J2K_ERR_INVALID_ARG; break; }                           // "unexpected end of packet data"
        if (lane == 0) body_base[pk] = base + pos;
        pos += body;
        done = k + 1;
    }
    if (lane == 0) {
        for (int64_t k = done; k < Cn.npackets && Cn.packet0 + k < npackets_all && Cn.packet0 + k >= 0; k++) body_base[Cn.packet0 + k] = ~0ull;
        Cn.st.pos = pos; Cn.st.rpos = r.rpos; Cn.st.buf = (uint8_t)r.lastb; Cn.st.cnt = (uint8_t)r.have; Cn.st.saw_ff = r.saw_ff ? 1 : 0;
        Cn.status = status; Cn.done = (int32_t)done;
        if (frame_status && status) atomicMin(frame_status, status);
    }
}

// data_off of the blocks a packet includes: from "within the packet's bodies" to "within d_data" (one workgroup per packet)
__global__ __launch_bounds__(256) void t2_bodies_kernel(const j2k_t2_dev_packet *__restrict__ packets, j2k_t2_dev_cb *__restrict__ cbs, uint64_t ncbs,
                                                        const uint64_t *__restrict__ body_base) {
    const j2k_t2_dev_packet P = packets[blockIdx.x];
    const uint64_t b = body_base[blockIdx.x];
    if (b == ~0ull || P.ncb < 0 || P.cb0 < 0 || (uint64_t)P.cb0 + (uint64_t)P.ncb > ncbs) return;
    for (int64_t i = threadIdx.x; i < P.ncb; i += 256) {
        j2k_t2_dev_cb &cb = cbs[P.cb0 + i];
        if (cb.included_in_layers == P.layer && cb.data_len > 0) cb.data_off += b;
    }
}

// ---- a frame's tile-parts as chains ----------------------------------------------------------------------------------------
// Tile-part t of the shard: SOT (FF90, Lsot = 10, Isot, Psot, TPsot, TNsot), marker segments stepped over by their lengths,
// SOD (FF93), then the packets up to Psot bytes from the SOT marker (Psot = 0: to the end) -- ReadTilePartHeader,
// parser.go:894-983.  With tile_offs the tile-parts are looked at side by side (a thread each); without, thread 0 walks them.
__device__ int t2_tile_chain(const uint8_t *cs, uint64_t len, uint64_t at, int want_index, T2Chain &Cn, uint64_t &next) {
    if (at + 12 > len || cs[at] != 0xFF || cs[at + 1] != 0x90) return J2K_ERR_INVALID_ARG;
    const uint32_t lsot = (uint32_t)cs[at + 2] << 8 | cs[at + 3], isot = (uint32_t)cs[at + 4] << 8 | cs[at + 5];
    const uint32_t psot = (uint32_t)cs[at + 6] << 24 | (uint32_t)cs[at + 7] << 16 | (uint32_t)cs[at + 8] << 8 | cs[at + 9];
    if (lsot != 10 || isot != ((uint32_t)want_index & 0xFFFFu)) return J2K_ERR_INVALID_ARG;
    const uint64_t stop = psot ? at + psot : len;
    if (stop > len || stop < at + 14) return J2K_ERR_INVALID_ARG;
    uint64_t p = at + 12;
    for (;;) {                                                      // parser.go:180-190: segments by length until SOD
        if (p + 2 > stop || cs[p] != 0xFF) return J2K_ERR_INVALID_ARG;
        if (cs[p + 1] == 0x93) { p += 2; break; }
        if (p + 4 > stop) return J2K_ERR_INVALID_ARG;
        const uint32_t l = (uint32_t)cs[p + 2] << 8 | cs[p + 3];
        if (l < 2) return J2K_ERR_INVALID_ARG;
        p += 2 + l;
    }
    Cn.start = p; Cn.end = stop;
    next = stop;
    return J2K_OK;
}
__global__ __launch_bounds__(64) void t2_tile_chains_kernel(const uint8_t *__restrict__ cs, uint64_t len, const uint64_t *__restrict__ tile_offs, int ntiles,
                                                            int tile_first, const int *__restrict__ tile_packet0, T2Chain *__restrict__ chains) {
    const int t0 = blockIdx.x * 64 + threadIdx.x;
    if (tile_offs) {
        if (t0 >= ntiles) return;
        T2Chain Cn{};
        uint64_t next = 0;
        Cn.packet0 = tile_packet0[t0]; Cn.npackets = tile_packet0[t0 + 1] - tile_packet0[t0];
        Cn.status = t2_tile_chain(cs, len, tile_offs[t0], tile_first + t0, Cn, next);
        Cn.skip = Cn.status != J2K_OK;
        chains[t0] = Cn;
        return;
    }
    if (t0 != 0) return;
    uint64_t at = 0;
    int bad = J2K_OK;
    for (int t = 0; t < ntiles; t++) {
        T2Chain Cn{};
        Cn.packet0 = tile_packet0[t]; Cn.npackets = tile_packet0[t + 1] - tile_packet0[t];
        if (bad == J2K_OK) bad = t2_tile_chain(cs, len, at, tile_first + t, Cn, at);
        Cn.status = bad; Cn.skip = bad != J2K_OK;                   // (behind a malformed tile-part nothing can be found)
        chains[t] = Cn;
    }
}

// Code-block fields -> what j2k_plan_decode_blocks takes.  A block the packets did not include (or a chain that failed before it)
// has no data: length 0, no bit planes -- tcd.DecodeCodeBlock leaves its coefficients alone (tcd.go:394-396).  Bit planes: the MQ
// coder's pass count is 3 * numBPS - 2 (t1_fast5.go:66-70); an HT block carries one pass and the decoder does not use the count,
// so it is mb - ZeroBitPlanes there (what j2k_plan_t2_fill_cbs wrote).
__global__ __launch_bounds__(256) void t2_blocks_kernel(long n, const j2k_t2_dev_cb *__restrict__ cbs, int ht, int mb, uint64_t total, uint64_t *__restrict__ offs,
                                                        uint32_t *__restrict__ lens, uint8_t *__restrict__ numbps) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const j2k_t2_dev_cb cb = cbs[j];
    // (a body outside the buffer can only come from a chain that failed half way -- the frame's status says so; nothing is read there)
    const bool has = cb.included_in_layers == 0 && cb.data_len > 0 && cb.num_passes > 0 && cb.data_off <= total && cb.data_len <= total - cb.data_off;
    int nb = 0;
    if (has) nb = ht ? (mb > cb.zero_bit_planes ? mb - cb.zero_bit_planes : 0) : (cb.num_passes + 2) / 3;
    offs[j] = has ? cb.data_off : 0;
    lens[j] = has ? cb.data_len : 0u;
    numbps[j] = (uint8_t)(nb > 255 ? 255 : nb);
}

// decoded block j (dense w x h at D.out_off) -> its window of the coefficient planes (S.src_off, row stride S.stride); one
// workgroup per block and 64 rows
__global__ __launch_bounds__(256) void place_blocks_kernel(const BlockJob *__restrict__ src_jobs, const BlockJob *__restrict__ dec_jobs,
                                                           const int32_t *__restrict__ decoded, int32_t *__restrict__ coeff) {
    const BlockJob S = src_jobs[blockIdx.x];
    const int64_t doff = dec_jobs[blockIdx.x].out_off;
    const int w = S.w, h = S.h;
    const int y0 = blockIdx.y * 64, y1 = min(h, y0 + 64);
    if (y0 >= h) return;
    const int32_t *src = decoded + doff;
    int32_t *dst = coeff + S.src_off;
    if (!(w & 3) && !(doff & 3) && !(S.src_off & 3) && !(S.stride & 3)) {
        const int wq = w >> 2;
        for (int i = threadIdx.x; i < (y1 - y0) * wq; i += 256) {
            const int y = y0 + i / wq, x = (i - (i / wq) * wq) << 2;
            *reinterpret_cast<int4 *>(dst + (int64_t)y * S.stride + x) = *reinterpret_cast<const int4 *>(src + (int64_t)y * w + x);
        }
    } else {
        for (int i = threadIdx.x; i < (y1 - y0) * w; i += 256) {
            const int y = y0 + i / w, x = i - (i / w) * w;
            dst[(int64_t)y * S.stride + x] = src[(int64_t)y * w + x];
        }
    }
}

size_t t2_chain_bytes() { return sizeof(T2Chain); }

hipError_t launch_t2_tile_chains(hipStream_t s, const uint8_t *cs, uint64_t len, const uint64_t *tile_offs, int ntiles, int tile_first,
                                 const int *tile_packet0, void *chains) {
    if (ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(t2_tile_chains_kernel, dim3(tile_offs ? (unsigned)((ntiles + 63) / 64) : 1u), dim3(64), 0, s, cs, len, tile_offs, ntiles, tile_first,
                       tile_packet0, reinterpret_cast<T2Chain *>(chains));
    return hipGetLastError();
}
// chains: nchains x T2Chain (device); body_base: npackets x u64 scratch; frame_status: optional sticky word (min of the chains' statuses)
hipError_t launch_t2_decode_packets(hipStream_t s, void *chains, int nchains, const j2k_t2_dev_packet *packets, long npackets, j2k_t2_dev_cb *cbs, uint64_t ncbs,
                                    const uint8_t *data, int sop, int eph, int clean, uint64_t *body_base, int *frame_status) {
    if (nchains <= 0 || npackets <= 0) return hipSuccess;
    hipLaunchKernelGGL(t2_decode_kernel, dim3((unsigned)nchains), dim3(64), 0, s, reinterpret_cast<T2Chain *>(chains), packets, npackets, cbs, ncbs, data, sop, eph,
                       clean, body_base, frame_status);
    hipLaunchKernelGGL(t2_bodies_kernel, dim3((unsigned)npackets), dim3(256), 0, s, packets, cbs, ncbs, body_base);
    return hipGetLastError();
}
// the generic call's one chain, made on the host
void t2_make_chain(void *dst, uint64_t len, long npackets, const j2k_t2_dec_state &st) {
    T2Chain Cn{};
    Cn.start = 0; Cn.end = len; Cn.packet0 = 0; Cn.npackets = npackets; Cn.st = st;
    __builtin_memcpy(dst, &Cn, sizeof Cn);
}
void t2_read_chain(const void *src, j2k_t2_dec_state &st, int &status, long &done) {
    T2Chain Cn;
    __builtin_memcpy(&Cn, src, sizeof Cn);
    st = Cn.st; status = Cn.status; done = Cn.done;
}
hipError_t launch_t2_blocks(hipStream_t s, long n, const j2k_t2_dev_cb *cbs, int ht, int mb, uint64_t total, uint64_t *offs, uint32_t *lens, uint8_t *numbps) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(t2_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, cbs, ht, mb, total, offs, lens, numbps);
    return hipGetLastError();
}
hipError_t launch_place_blocks(hipStream_t s, const BlockJob *src_jobs, const BlockJob *dec_jobs, int njobs, int max_h, const int32_t *decoded, int32_t *coeff) {
    if (njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(place_blocks_kernel, dim3((unsigned)njobs, (unsigned)((max_h + 63) / 64)), dim3(256), 0, s, src_jobs, dec_jobs, decoded, coeff);
    return hipGetLastError();
}

}  // namespace j2k
