// t2dev.hip -- Tier-2 packet ENCODING on device buffers: PacketEncoder.EncodePacket (internal/tcd/t2.go:250-438)
// for a whole run of packets, over code-block bytes that never leave HBM.  csrc/t2.cpp is the same coder on host buffers, one packet
// per call; this is the batch form for the end of the encode pipeline (block coder -> compaction -> packets).
//
// What is serial in the reference and what is not:
//   * a packet's header is a bit string through bio.ByteStuffingWriter (bio.go:157-226: after a 0xFF byte the next byte holds seven
//     bits), flushed at its end: bit-serial, but independent of every other packet EXCEPT for one flag -- whether the last header byte
//     the encoder wrote was 0xFF, which the writer keeps across packets (the Flush leaves it).  So every packet is sized for both
//     values of the flag (t2_size_kernel, a wavefront per packet: the lanes form the code-blocks' header fields, two lanes string them
//     together for the two entry states), a scan composes the packets' flag -> flag maps and sums the lengths (t2_scan_kernel, one
//     workgroup), and the packets are then written in parallel, each at its final offset with its flag known (t2_header_kernel, a
//     wavefront per packet; t2_body_kernel, eight wavefronts per packet copy the code-blocks' bytes);
//   * the "tag tree" values are unary (t2.go:368-377) -- a run of zeros is added in bulk, so a large value costs its bytes, not its bits;
//     everything else goes into the writer a field at a time, not a bit at a time.
// The reference's quirks stay (include/j2kgfx.h, "Tier-2"): inclusion is written for every block of layer 0 whether or not it is
// included, the length-of-length field has three bits and wraps, a tree width of 0 is Go's divide panic.
#include "j2k_internal.h"

namespace j2k {

// bio.ByteStuffingWriter, fed several bits at a time; out == nullptr: count only.  A byte holds 8 bits, or 7 behind a 0xFF byte (and a
// 7-bit byte is never 0xFF itself).
struct T2Sink {
    uint8_t *out;
    uint64_t n;
    uint64_t acc;         // the low `have` bits are pending, oldest on top
    unsigned have;        // < 8 between calls
    bool after_ff;
    __device__ __forceinline__ void drain() {
        for (;;) {
            const unsigned room = after_ff ? 7u : 8u;
            if (have < room) break;
            const unsigned b = (unsigned)(acc >> (have - room)) & ((1u << room) - 1u);
            if (out) out[n] = (uint8_t)b;
            n++;
            have -= room;
            after_ff = b == 0xFF;
        }
    }
    __device__ __forceinline__ void put(uint32_t v, unsigned count) {       // count <= 32, MSB first (bio.go:196-205)
        if (!count) return;
        acc = (acc << count) | (uint64_t)(count < 32 ? v & ((1u << count) - 1u) : v);
        have += count;
        drain();
    }
    __device__ void zeros(int64_t z) {                      // a unary value's run (t2.go:368-377): in bulk once a byte boundary is reached
        while (z > 0) {
            if (have == 0 && !after_ff && z >= 8) {
                const int64_t whole = z >> 3;               // whole zero bytes: eight bits each, none of them 0xFF
                if (out) for (int64_t i = 0; i < whole; i++) out[n + i] = 0;
                n += (uint64_t)whole;
                z &= 7;
            } else {
                const unsigned room = (after_ff ? 7u : 8u) - have;
                const unsigned c = (unsigned)(z < (int64_t)room ? z : (int64_t)room);
                put(0, c);
                z -= c;
            }
        }
    }
    __device__ void flush() {                               // bio.go:211-221: pad the byte in progress with zeros
        if (have) put(0, (after_ff ? 7u : 8u) - have);
    }
};

__device__ __forceinline__ bool t2_contributes(const j2k_t2_dev_cb &cb, int layer) { return cb.included_in_layers <= layer && cb.data_len > 0; }

// One code-block's share of a packet header (t2.go:320-364) as two unary runs and two bit strings: zeros z1, bits b1 (n1), zeros z2,
// bits b2 (n2 <= 54: the closing one of the zero-bit-plane value, the pass code of t2.go:379-406, the bit length of the length in a
// 3-bit field -- it wraps; five bits with J2K_T2_WIDE_LEN -- and the length in all its bits, t2.go:408-437).
struct T2Fields { uint32_t z1, z2, b1, n1, n2, pad_; uint64_t b2; };
__device__ __forceinline__ T2Fields t2_fields(const j2k_t2_dev_cb &cb, int layer, int flags) {
    T2Fields F{0, 0, 0, 0, 0, 0, 0};
    const bool inc = t2_contributes(cb, layer);
    if (layer == 0) { F.z1 = cb.included_in_layers > 0 ? (uint32_t)cb.included_in_layers : 0u; F.b1 = 1; F.n1 = 1; }   // written whether or not the block is included
    else { F.b1 = inc ? 1u : 0u; F.n1 = 1; }
    if (!inc) return F;
    uint64_t b = 0;
    unsigned n = 0;
    auto add = [&](uint32_t v, unsigned c) { b = (b << c) | ((uint64_t)v & ((1ull << c) - 1ull)); n += c; };
    if (cb.included_in_layers == layer) { F.z2 = cb.zero_bit_planes > 0 ? (uint32_t)cb.zero_bit_planes : 0u; add(1, 1); }
    const int np = cb.num_passes;
    if (np == 1) add(0, 1);
    else if (np == 2) add(2, 2);
    else if (np <= 5) { add(3, 2); add((uint32_t)(np - 3), 2); }
    else if (np <= 36) { add(15, 4); add((uint32_t)(np - 6), 5); }
    else { add(0x1FF, 9); add((uint32_t)(np - 37), 7); }
    unsigned nb = 0;
    for (uint32_t t = cb.data_len; t; t >>= 1) nb++;
    add(nb, (flags & J2K_T2_WIDE_LEN) ? 5 : 3);            // (closed-loop mode: five bits, so that the field holds every length's bit count)
    if (nb) add(cb.data_len, nb);
    F.b2 = b; F.n2 = n;
    return F;
}

// encodePacketHeader (t2.go:293-366) by one wavefront: 64 code-blocks at a time, every lane forms the fields of one, then the lanes
// below `nsinks` string them together, each into its own sink (the size pass runs the two entry states of the writer side by side).
// Returns false on the reference's divide panic (a tree of width 0 that the coder consults) or a table out of range.
// does the packet hold data (t2.go:295-309), and does its header divide by a tree width of 0 (false)?
__device__ bool t2_header_checks(const j2k_t2_dev_packet &P, const j2k_t2_dev_cb *__restrict__ cbs, int lane, bool &any) {
    const int layer = P.layer;
    bool need_imsb = false;
    any = false;
    for (int64_t i0 = 0; i0 < P.ncb; i0 += 64) {
        const int64_t i = i0 + lane;
        bool c = false, m = false;
        if (i < P.ncb) { const j2k_t2_dev_cb cb = cbs[i]; c = t2_contributes(cb, layer); m = c && cb.included_in_layers == layer; }
        any |= __any(c) != 0;
        need_imsb |= __any(m) != 0;
    }
    return !(any && ((layer == 0 && P.incl_tree_w == 0) || (need_imsb && P.imsb_tree_w == 0)));
}
__device__ bool t2_header_wave(T2Sink &w, int nsinks, const j2k_t2_dev_packet &P, const j2k_t2_dev_cb *__restrict__ cbs, T2Fields *fld, int lane) {
    const int layer = P.layer;
    bool any = false;
    if (!t2_header_checks(P, cbs, lane, any)) return false;
    const bool mine = lane < nsinks;
    if (!any) {
        if (mine) w.put(0, 1);
    } else {
        if (mine) w.put(1, 1);
        for (int64_t i0 = 0; i0 < P.ncb; i0 += 64) {
            const int64_t i = i0 + lane;
            __syncthreads();                                // (one wavefront: orders the LDS traffic of the two phases)
            if (i < P.ncb) fld[lane] = t2_fields(cbs[i], layer, P.flags);
            __syncthreads();
            const int cnt = (int)(P.ncb - i0 < 64 ? P.ncb - i0 : 64);
            if (mine)
                for (int j = 0; j < cnt; j++) {
                    const T2Fields F = fld[j];
                    w.zeros(F.z1); w.put(F.b1, F.n1);
                    w.zeros(F.z2);
                    if (F.n2 > 32) { w.put((uint32_t)(F.b2 >> 32), F.n2 - 32); w.put((uint32_t)F.b2, 32); } else w.put((uint32_t)F.b2, F.n2);
                }
        }
    }
    if (mine) w.flush();
    return true;
}

// Headers of up to 16 384 bits in parallel: without the stuffing a header is just its fields end to end.  The wavefront lays the
// fields of all code-blocks out in an LDS bit buffer (bit offsets by a scan over the lanes, the bits OR-ed in; most significant bit first)
// and returns the number of bits -- or 0 when the header does not fit the buffer and the serial writer has to take it.
#define T2_FAST_WORDS 512
__device__ __forceinline__ void t2_or_bits(uint32_t *bits, uint64_t pos, uint32_t v, unsigned n) {      // n <= 32
    if (!n) return;
    const uint32_t w = (uint32_t)(pos >> 5), sh = (uint32_t)pos & 31u;
    if (n < 32) v &= (1u << n) - 1u;
    if (sh + n <= 32) atomicOr(&bits[w], v << (32 - sh - n));
    else { const unsigned lo = sh + n - 32; atomicOr(&bits[w], v >> lo); atomicOr(&bits[w + 1], v << (32 - lo)); }
}
__device__ uint64_t t2_fast_build(const j2k_t2_dev_packet &P, const j2k_t2_dev_cb *__restrict__ cbs, bool any, uint32_t *bits, int lane) {
    for (int i = lane; i < T2_FAST_WORDS + 2; i += 64) bits[i] = 0;
    __syncthreads();
    if (!any) return 1;                                     // the presence bit, 0
    if (lane == 0) bits[0] = 0x80000000u;                   // the presence bit, 1
    uint64_t base = 1;
    for (int64_t i0 = 0; i0 < P.ncb; i0 += 64) {
        const int64_t i = i0 + lane;
        T2Fields F{0, 0, 0, 0, 0, 0, 0};
        if (i < P.ncb) F = t2_fields(cbs[i], P.layer, P.flags);
        const uint64_t tb = (uint64_t)F.z1 + F.n1 + F.z2 + F.n2;
        uint64_t incl = tb;
        for (int d = 1; d < 64; d <<= 1) { const uint64_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        const uint64_t tot = __shfl(incl, 63);
        if (base + tot > (uint64_t)T2_FAST_WORDS * 32) return 0;
        const uint64_t off = base + incl - tb;
        t2_or_bits(bits, off + F.z1, F.b1, F.n1);
        const uint64_t p2 = off + F.z1 + F.n1 + F.z2;
        if (F.n2 > 32) { t2_or_bits(bits, p2, (uint32_t)(F.b2 >> 32), F.n2 - 32); t2_or_bits(bits, p2 + F.n2 - 32, (uint32_t)F.b2, 32); }
        else t2_or_bits(bits, p2, (uint32_t)F.b2, F.n2);
        base += tot;
    }
    __syncthreads();
    return base;
}
// The byte-stuffing writer over that bit buffer, by the whole wavefront: as long as no byte is 0xFF the bytes are the bits cut every
// eight (the first one every seven when the writer enters behind a 0xFF), so the lanes cut 64 bytes at a time and look for the first
// 0xFF; everything up to it stands, the byte behind it holds seven bits, and the cutting starts again from there -- one round per 0xFF
// byte in the header.  out == nullptr: count only.  Returns the bytes; ff = the writer's state afterwards.
__device__ uint64_t t2_fast_emit(const uint32_t *bits, uint64_t total, int s, uint8_t *out, int lane, bool &ff) {
    uint64_t pos = 0, nout = 0;
    bool after_ff = s != 0;
    while (pos < total) {
        const unsigned first_nb = after_ff ? 7u : 8u;
        const uint64_t rest = total - pos;
        const uint64_t nby = rest <= first_nb ? 1 : 1 + (rest - first_nb + 7) / 8;
        int64_t found = -1;
        for (uint64_t k0 = 0; k0 < nby && found < 0; k0 += 64) {
            const uint64_t k = k0 + lane;
            uint32_t v = 0;
            if (k < nby) {
                const uint64_t bp = pos + (k ? first_nb + 8 * (k - 1) : 0);
                const unsigned nb = k ? 8u : first_nb;
                const uint32_t w = (uint32_t)(bp >> 5), sh = (uint32_t)bp & 31u;
                const uint64_t two = (uint64_t)bits[w] << 32 | bits[w + 1];
                v = (uint32_t)(two >> (64 - sh - nb)) & ((1u << nb) - 1u);
            }
            const uint64_t m = __ballot(k < nby && v == 0xFF);
            const int first = m ? __builtin_ctzll(m) : 64;
            if (out && k < nby && lane <= first) out[nout + k] = (uint8_t)v;
            if (m) found = (int64_t)k0 + first;
        }
        if (found < 0) { nout += nby; after_ff = false; break; }
        nout += (uint64_t)found + 1;
        pos += first_nb + 8 * (uint64_t)found;
        after_ff = true;
    }
    ff = after_ff;
    return nout;
}

// per packet: header bytes for the writer's flag clear / set on entry, the flag on exit, the body bytes
struct T2Size { uint64_t hlen[2]; uint64_t body; uint32_t ff_out; uint32_t pad_; };

__global__ __launch_bounds__(64) void t2_size_kernel(const j2k_t2_dev_packet *__restrict__ packets, long npackets, const j2k_t2_dev_cb *__restrict__ cbs,
                                                     uint64_t ncbs, T2Size *__restrict__ sizes, uint64_t *__restrict__ result) {
    __shared__ T2Fields fld[64];
    __shared__ uint32_t bits[T2_FAST_WORDS + 2];
    const long p = blockIdx.x;
    const int lane = threadIdx.x;
    const j2k_t2_dev_packet P = packets[p];
    T2Sink w{nullptr, 0, 0, 0, lane == 1};
    bool bad = P.ncb < 0 || P.cb0 < 0 || (uint64_t)P.cb0 + (uint64_t)P.ncb > ncbs;
    bool fast = false;
    if (!bad) {
        bool any = false;
        bad = !t2_header_checks(P, cbs + P.cb0, lane, any);
        if (!bad) {
            const uint64_t total = t2_fast_build(P, cbs + P.cb0, any, bits, lane);
            if (total) {
                bool f0, f1;
                const uint64_t n0 = t2_fast_emit(bits, total, 0, nullptr, lane, f0), n1 = t2_fast_emit(bits, total, 1, nullptr, lane, f1);
                fast = true;
                w.n = lane == 1 ? n1 : n0;                              // (lanes 0 and 1 carry the two entry states, as below)
                w.after_ff = lane == 1 ? f1 : f0;
            }
        }
    }
    if (bad || (!fast && !t2_header_wave(w, 2, P, cbs + P.cb0, fld, lane))) {
        if (lane == 0) { atomicMax((unsigned long long *)&result[2], 1ull); sizes[p] = T2Size{}; }
        return;
    }
    uint64_t body = 0;
    for (int64_t i = lane; i < P.ncb; i += 64) { const j2k_t2_dev_cb cb = cbs[P.cb0 + i]; if (t2_contributes(cb, P.layer)) body += cb.data_len; }
    for (int d = 32; d > 0; d >>= 1) body += __shfl_xor(body, d);
    const uint64_t h1 = __shfl(w.n, 1);
    const uint32_t f1 = __shfl((uint32_t)w.after_ff, 1);
    if (lane == 0) {
        T2Size S{};
        S.hlen[0] = w.n; S.hlen[1] = h1; S.body = body;
        S.ff_out = (w.after_ff ? 1u : 0u) | f1 << 1;
        if (P.flags & J2K_T2_FRESH) { S.hlen[1] = S.hlen[0]; S.ff_out = (S.ff_out & 1u) * 3u; }     // a new encoder object: whatever came before, its writer starts clear
        sizes[p] = S;
    }
}

// offs[p] = where packet p starts, var[p] = the writer's flag on entry; result = {total bytes, flag after the last packet, fault}
struct T2Heads {                        // frame calls: the tile-part heads written by the scan's own workgroup once the offsets stand (ntiles == 0: none)
    const int *tile_packet0; int ntiles, tile_first; uint8_t *out; uint64_t cap; uint64_t *tile_offs; int *status;
};
__device__ void t2_tile_heads(const uint64_t *__restrict__ offs, long npackets, const T2Heads &Hd, const uint64_t *__restrict__ result, int t0, int tstep);
__global__ __launch_bounds__(256) void t2_scan_kernel(const T2Size *__restrict__ sizes, long npackets, int fixed, int delay_in, uint64_t *__restrict__ offs,
                                                      uint8_t *__restrict__ var, uint64_t *__restrict__ result, T2Heads Hd) {
    __shared__ uint64_t len_s[256][2], base_s[256];
    __shared__ uint8_t st_s[256][2], in_s[256];
    const int t = threadIdx.x;
    const long chunk = (npackets + 255) / 256, p0 = (long)t * chunk < npackets ? (long)t * chunk : npackets, p1 = p0 + chunk < npackets ? p0 + chunk : npackets;
    for (int s = 0; s < 2; s++) {
        uint64_t len = 0;
        int st = s;
        for (long p = p0; p < p1; p++) { const T2Size S = sizes[p]; len += (uint64_t)fixed + S.hlen[st] + S.body; st = (S.ff_out >> st) & 1; }
        len_s[t][s] = len; st_s[t][s] = (uint8_t)st;
    }
    __syncthreads();
    // inclusive scan of the chunks' maps (entry state -> bytes, exit state) under composition, Hillis-Steele over the 256 threads
    for (int d = 1; d < 256; d <<= 1) {
        uint64_t l0 = 0, l1 = 0;
        uint8_t s0 = 0, s1 = 1;
        const bool on = t >= d;
        if (on) {                                            // (chunk t - d ... first, then mine)
            const uint8_t a0 = st_s[t - d][0], a1 = st_s[t - d][1];
            l0 = len_s[t - d][0] + len_s[t][a0]; s0 = st_s[t][a0];
            l1 = len_s[t - d][1] + len_s[t][a1]; s1 = st_s[t][a1];
        }
        __syncthreads();
        if (on) { len_s[t][0] = l0; len_s[t][1] = l1; st_s[t][0] = s0; st_s[t][1] = s1; }
        __syncthreads();
    }
    {
        const int s = delay_in ? 1 : 0;
        base_s[t] = t ? len_s[t - 1][s] : 0;
        in_s[t] = t ? st_s[t - 1][s] : (uint8_t)s;
        if (t == 255) { result[0] = len_s[255][s]; result[1] = (uint64_t)st_s[255][s]; offs[npackets] = len_s[255][s]; }
    }
    __syncthreads();
    uint64_t off = base_s[t];
    int st = in_s[t];
    for (long p = p0; p < p1; p++) {
        const T2Size S = sizes[p];
        offs[p] = off; var[p] = (uint8_t)st;
        off += (uint64_t)fixed + S.hlen[st] + S.body;
        st = (S.ff_out >> st) & 1;
    }
    if (Hd.ntiles > 0) {                                            // (one workgroup wrote every offset: visible to all of it behind the barrier)
        __threadfence_block();                                    // (workgroup scope is all it takes -- a device-scope release writes the L2 back, docs/KERNEL_NOTES.md 4p)
        __syncthreads();
        t2_tile_heads(offs, npackets, Hd, result, t, 256);
    }
}

// markers + header of packet p at its final place, the writer's entry state known (t2.go:257-276)
__global__ __launch_bounds__(64) void t2_header_kernel(const j2k_t2_dev_packet *__restrict__ packets, long npackets, const j2k_t2_dev_cb *__restrict__ cbs,
                                                       const T2Size *__restrict__ sizes, const uint64_t *__restrict__ offs, const uint8_t *__restrict__ var,
                                                       int sop, int eph, uint8_t *__restrict__ out, uint64_t cap, const uint64_t *__restrict__ result,
                                                       const int32_t *__restrict__ ptile, uint64_t extra) {
    __shared__ T2Fields fld[64];
    __shared__ uint32_t bits[T2_FAST_WORDS + 2];
    const long p = blockIdx.x;
    const int lane = threadIdx.x;
    if (result[2] != 0 || offs[npackets] + extra > cap) return;      // (a fault or too little room: nothing is written, the host reports it)
    const j2k_t2_dev_packet P = packets[p];
    const int v = (P.flags & J2K_T2_FRESH) ? 0 : var[p];
    // (frame calls: straight into the tile-parts -- packet p of tile t lies 14 (t + 1) bytes further, behind t + 1 SOT | SOD heads)
    uint8_t *o = out + offs[p] + (ptile ? 14ull * (uint64_t)(ptile[p] + 1) : 0ull);
    if (lane == 0 && sop) { o[0] = 0xFF; o[1] = 0x91; o[2] = 0x00; o[3] = 0x04; o[4] = (uint8_t)((unsigned)P.layer >> 8); o[5] = (uint8_t)P.layer; }
    {
        bool any = false;
        (void)t2_header_checks(P, cbs + P.cb0, lane, any);
        const uint64_t total = t2_fast_build(P, cbs + P.cb0, any, bits, lane);
        if (total) {
            bool f;
            (void)t2_fast_emit(bits, total, v, o + (sop ? 6 : 0), lane, f);
        } else {
            T2Sink w{o + (sop ? 6 : 0), 0, 0, 0, v != 0};
            (void)t2_header_wave(w, 1, P, cbs + P.cb0, fld, lane);
        }
    }
    if (lane == 0 && eph) { uint8_t *e = o + (sop ? 6 : 0) + sizes[p].hlen[v]; e[0] = 0xFF; e[1] = 0x92; }
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
#define T2_BODY_SLICES 8
// The bodies in block order behind the header (t2.go:279-287).  Wavefront (p, s) copies the code-blocks j = s mod gridDim.y of
// packet p (frame calls: as many slices as the plan's largest packet has blocks, at most 32 -- 8 -> 32 on the C2 frame: 25 -> 19 us); every wavefront sizes all of them (64 at a time, a scan over the lanes) to know where its own go.
__global__ __launch_bounds__(64) void t2_body_kernel(const j2k_t2_dev_packet *__restrict__ packets, long npackets, const j2k_t2_dev_cb *__restrict__ cbs,
                                                     const uint8_t *__restrict__ data, const T2Size *__restrict__ sizes, const uint64_t *__restrict__ offs,
                                                     const uint8_t *__restrict__ var, int fixed, uint8_t *__restrict__ out, uint64_t cap,
                                                     const uint64_t *__restrict__ result, const int32_t *__restrict__ ptile, uint64_t extra,
                                                     const BlockJob *__restrict__ slot_jobs, const uint32_t *__restrict__ maglens, int ht) {
    const long p = blockIdx.x;
    const int slice = blockIdx.y, lane = threadIdx.x;
    if (result[2] != 0 || offs[npackets] + extra > cap) return;
    const j2k_t2_dev_packet P = packets[p];
    uint8_t *o = out + offs[p] + (ptile ? 14ull * (uint64_t)(ptile[p] + 1) : 0ull);
    uint64_t pos = (uint64_t)fixed + sizes[p].hlen[var[p]];
    for (int64_t i0 = 0; i0 < P.ncb; i0 += 64) {
        const int64_t i = i0 + lane;
        j2k_t2_dev_cb cb{};
        if (i < P.ncb) cb = cbs[P.cb0 + i];
        const uint32_t mylen = (i < P.ncb && t2_contributes(cb, P.layer)) ? cb.data_len : 0u;
        uint64_t incl = mylen;
        for (int d = 1; d < 64; d <<= 1) { const uint64_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        const uint64_t myoff = pos + incl - mylen;
        const int cnt = (int)(P.ncb - i0 < 64 ? P.ncb - i0 : 64);
        for (int j = slice; j < cnt; j += (int)gridDim.y) {
            const uint32_t len = __shfl(mylen, j);
            if (!len) continue;
            const uint64_t dof = __shfl(myoff, j), sof = __shfl(cb.data_off, j);
            uint8_t *d = o + dof;
            if (slot_jobs) {                                         // frame encoder: block P.cb0 + i0 + j of the plan, from its coding slot (no dense stream in between)
                const int64_t jb = P.cb0 + i0 + j;
                gather_job(slot_jobs[jb], data, d, len, ht != 0, maglens ? maglens[jb] : 0u, lane);
                continue;
            }
            const uint8_t *s = data + sof;
            const uint32_t words = len >> 2;
            for (uint32_t k = lane; k < words; k += 64) *reinterpret_cast<u32_unaligned *>(d + 4 * k) = *reinterpret_cast<const u32_unaligned *>(s + 4 * k);
            if (lane < (int)(len & 3)) d[4 * words + lane] = s[4 * words + lane];
        }
        pos += __shfl(incl, 63);
    }
}

// The block coder's outputs as packet tables: code-block j of a plan (its job order is component, resolution, band, block row,
// block column: encoder.go:616-673) with the bytes lens[j] at offs[j] of the compacted stream.  IncludedInLayers 0, Passes = the
// 3 * numBPS - 2 coding passes EncodeFast5 ran (t1_fast5.go:66-70; HT: one), ZeroBitPlanes = max(mb - numBPS, 0).
__global__ __launch_bounds__(256) void t2_fill_cbs_kernel(long n, const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens,
                                                          const uint8_t *__restrict__ numbps, int mb, int ht, j2k_t2_dev_cb *__restrict__ cbs,
                                                          uint64_t *__restrict__ reset) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j == 0 && reset) { reset[0] = 0; reset[1] = 0; reset[2] = 0; }   // (the packet coder's result words, for the launches behind this one)
    if (j >= n) return;
    const int nb = numbps[j];
    j2k_t2_dev_cb cb{};
    cb.included_in_layers = ((ht & 2) && lens[j] == 0) ? 1 : 0;     // closed-loop mode (bit 1): a block without data is in no layer, and says so
    cb.zero_bit_planes = mb > nb ? mb - nb : 0;
    cb.num_passes = nb == 0 ? 0 : ((ht & 1) ? 1 : 3 * nb - 2);
    cb.data_len = lens[j];
    cb.data_off = offs ? offs[j] : 0;                              // (no stream: the bytes are gathered from the coding slots)
    cbs[j] = cb;
}

hipError_t launch_t2_fill_cbs(hipStream_t s, long n, const uint64_t *offs, const uint32_t *lens, const uint8_t *numbps, int mb, int ht, j2k_t2_dev_cb *cbs, uint64_t *reset) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(t2_fill_cbs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, offs, lens, numbps, mb, ht, cbs, reset);
    return hipGetLastError();
}

static unsigned t2_body_slices() {                                  // J2K_T2_BODY_SLICES (tuning): wavefronts that share a packet's bodies
    static const unsigned n = [] { const char *e = tuning_env("J2K_T2_BODY_SLICES"); const long v = e ? atol(e) : 0; return (unsigned)(v >= 1 && v <= 64 ? v : T2_BODY_SLICES); }();
    return n;
}
size_t t2_dev_workspace(long npackets) { return (size_t)npackets * sizeof(T2Size) + (size_t)npackets + 64; }

// ws: t2_dev_workspace(npackets) bytes; result: 3 x uint64 {total, flag out, fault}, zeroed by the caller on this stream
hipError_t launch_t2_encode_packets(hipStream_t s, const j2k_t2_dev_packet *packets, long npackets, const j2k_t2_dev_cb *cbs, uint64_t ncbs, const uint8_t *data,
                                    int sop, int eph, int delay_in, uint8_t *out, uint64_t cap, uint64_t *offs, void *ws, uint64_t *result) {
    T2Size *sizes = reinterpret_cast<T2Size *>(ws);
    uint8_t *var = reinterpret_cast<uint8_t *>(sizes + npackets);
    const int fixed = (sop ? 6 : 0) + (eph ? 2 : 0);
    if (npackets > 0) hipLaunchKernelGGL(t2_size_kernel, dim3((unsigned)npackets), dim3(64), 0, s, packets, npackets, cbs, ncbs, sizes, result);
    hipLaunchKernelGGL(t2_scan_kernel, dim3(1), dim3(256), 0, s, sizes, npackets, fixed, delay_in, offs, var, result, T2Heads{});
    if (npackets > 0) {
        hipLaunchKernelGGL(t2_header_kernel, dim3((unsigned)npackets), dim3(64), 0, s, packets, npackets, cbs, sizes, offs, var, sop, eph, out, cap, result,
                           (const int32_t *)nullptr, (uint64_t)0);
        hipLaunchKernelGGL(t2_body_kernel, dim3((unsigned)npackets, t2_body_slices()), dim3(64), 0, s, packets, npackets, cbs, data, sizes, offs, var, fixed, out, cap, result,
                           (const int32_t *)nullptr, (uint64_t)0, (const BlockJob *)nullptr, (const uint32_t *)nullptr, 0);
    }
    return hipGetLastError();
}

// SOT | SOD of every tile-part in front of its packets (tcd CreateTileHeader: FF90 000A Isot Psot TPsot = 0 TNsot = 1, FF93), where each
// starts, the total -- and the capacity verdict (nothing is written by anyone when the caller's buffer is too small).  Runs at the end of
// t2_scan_kernel (a launch of its own cost 4.6 us of a frame's 130)
__device__ void t2_tile_heads(const uint64_t *__restrict__ offs, long npackets, const T2Heads &Hd, const uint64_t *__restrict__ result, int t0, int tstep) {
    const int ntiles = Hd.ntiles;
    const uint64_t total = offs[npackets] + 14ull * (uint64_t)ntiles;
    if (t0 == 0) {
        Hd.tile_offs[ntiles] = total;
        if (Hd.status && result[2] != 0) atomicMin(Hd.status, J2K_ERR_GO_PANIC);       // (a tree width of 0: no plan makes one)
        else if (Hd.status && total > Hd.cap) atomicMin(Hd.status, J2K_ERR_CAPACITY);
    }
    for (int t = t0; t < ntiles; t += tstep) {
        const uint64_t o0 = offs[Hd.tile_packet0[t]], o1 = offs[Hd.tile_packet0[t + 1]];
        Hd.tile_offs[t] = o0 + 14ull * (uint64_t)t;
        if (result[2] != 0 || total > Hd.cap) continue;
        uint8_t *dst = Hd.out + o0 + 14ull * (uint64_t)t;
        const uint32_t idx = (uint32_t)(Hd.tile_first + t) & 0xFFFFu, psot = (uint32_t)(14 + (o1 - o0));
        const uint8_t hdr[14] = {0xFF, 0x90, 0x00, 0x0A, (uint8_t)(idx >> 8), (uint8_t)idx, (uint8_t)(psot >> 24), (uint8_t)(psot >> 16),
                                 (uint8_t)(psot >> 8), (uint8_t)psot, 0x00, 0x01, 0xFF, 0x93};
#pragma unroll
        for (int k = 0; k < 14; k++) dst[k] = hdr[k];
    }
}
// The packets of a frame written where they end up: tile-parts in `out` (SOT | SOD | packets per tile), no packet stream and no dense
// block stream in between.  data + cbs[].data_off: the blocks' bytes (the compacted stream of j2k_plan_encode_stream), or, with
// slot_jobs, data = the plan's coding slots and block j is gathered from slot_jobs[j] (HT: MagSgn | MEL zeros made here | VLC, maglens).
hipError_t launch_t2_encode_tile_parts(hipStream_t s, const j2k_t2_dev_packet *packets, long npackets, const j2k_t2_dev_cb *cbs, uint64_t ncbs, const uint8_t *data,
                                       int sop, int eph, uint8_t *out, uint64_t cap, uint64_t *offs, void *ws, uint64_t *result, const int32_t *ptile,
                                       const int *tile_packet0, int ntiles, int tile_first, uint64_t *tile_offs, int *status, const BlockJob *slot_jobs,
                                       const uint32_t *maglens, int ht, int body_slices) {
    if (npackets <= 0 || ntiles <= 0) return hipSuccess;
    T2Size *sizes = reinterpret_cast<T2Size *>(ws);
    uint8_t *var = reinterpret_cast<uint8_t *>(sizes + npackets);
    const int fixed = (sop ? 6 : 0) + (eph ? 2 : 0);
    const uint64_t extra = 14ull * (uint64_t)ntiles;
    hipLaunchKernelGGL(t2_size_kernel, dim3((unsigned)npackets), dim3(64), 0, s, packets, npackets, cbs, ncbs, sizes, result);
    hipLaunchKernelGGL(t2_scan_kernel, dim3(1), dim3(256), 0, s, sizes, npackets, fixed, 0, offs, var, result, T2Heads{tile_packet0, ntiles, tile_first, out, cap, tile_offs, status});
    hipLaunchKernelGGL(t2_header_kernel, dim3((unsigned)npackets), dim3(64), 0, s, packets, npackets, cbs, sizes, offs, var, sop, eph, out, cap, result, ptile, extra);
    hipLaunchKernelGGL(t2_body_kernel, dim3((unsigned)npackets, tuning_env("J2K_T2_BODY_SLICES") || body_slices < 1 ? t2_body_slices() : (unsigned)std::min(body_slices, 64)), dim3(64), 0, s, packets, npackets, cbs, data, sizes, offs, var, fixed, out, cap, result,
                       ptile, extra, slot_jobs, maglens, ht);
    return hipGetLastError();
}

}  // namespace j2k
