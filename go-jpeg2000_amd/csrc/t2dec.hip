// t2dec.hip -- Tier-2 packet DECODING on device buffers and the rest of the decode body:
//   t2_decode_kernel      PacketDecoder.DecodePacket (internal/tcd/t2.go:463-652) for chains of packets, one decoder object (one
//                         wavefront) per chain -- the reference's decoder as it is written, or, with the J2K_T2_* flags of the
//                         closed-loop mode (include/j2kgfx.h), one that can read what the encoder wrote
//   t2_tile_chains_kernel the tile-parts of a frame (SOT ... SOD, codestream.Parser.ReadTilePartHeader, parser.go:894-983) as chains
//   t2_marks_kernel, t2_seed_kernel   SOP + EPH streams: every packet's start guessed from the markers, one chain per PACKET; the tile
//                         launch of t2_decode_kernel verifies the guesses against the serial rule before it would run the tile's chain
//   t2_blocks_kernel / t2_finish_kernel   the decoded code-block fields as the block decoder's inputs (offsets, lengths, bit-plane counts)
//   place_blocks_kernel   decoded blocks -> their windows of the coefficient planes (the step decoder.decodeTile leaves out,
//                         decoder.go:375-411)
// What is serial and what is not.  A packet header is a prefix code: the place of every field depends on every field before it,
// and the place of the NEXT packet on the lengths this header carries -- one chain per decoder object, i.e. per tile in the
// closed-loop mode (a new PacketDecoder per tile, as the encoder side takes a new PacketEncoder) and per run in the reference's
// mode (its header reader runs over the whole buffer on its own).  Where the stream carries SOP and EPH markers the second dependence
// can be guessed away and checked afterwards ("the packets of a tile side by side", below).  A chain is parsed by one wavefront with every lane holding the
// same state (wave-uniform: the instruction stream of one lane, and all 64 there for the refills of the LDS window the bytes are
// read through); the chains of a frame run side by side.  Nothing is copied: a block's body stays where it is and the block
// decoder reads it from there.
#include "j2k_internal.h"

namespace j2k {

#define T2D_RAW 4096                    // bytes of the buffer staged in LDS at a time (a tile's first packets -- a few blocks each -- lie within it)
#define T2D_CHUNK 256                   // raw bytes un-stuffed at a time into the bit buffer (a header of a C2 tile: 10 ... 250 bytes)

// bio.ByteStuffingReader (bio.go:105-155) in two stages.  (1) By the whole wavefront: T2D_CHUNK raw bytes -> their bits end to end in
// an LDS bit buffer.  A byte holds 8 bits, or 7 behind a 0xFF byte (bio.go:127-131): the widths depend on the RAW bytes only, so four
// bytes per lane, one scan over the lanes for the bit offsets, OR-deposits.  (2) The chain -- every lane the same state -- reads
// fields from a 64-bit register window over that buffer: a field is a shift, a unary value a count of leading zeros; one LDS word per
// 32 bits consumed.  The reader's position in the reference's terms (rpos, cnt, buf, sawFF) is recovered from the byte -> bit map
// whenever it is asked for (a packet's end), so the two stages give exactly the reference's state at every field boundary.
// every lane of the parsing wavefront holds the same reader state: a value read from LDS is taken through readfirstlane so that the compiler
// KNOWS it is uniform and keeps the state in scalar registers, with scalar branches (left in vector registers the field loop was ~150
// vector instructions a code-block under exec-mask control flow)
__device__ __forceinline__ uint32_t t2_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct T2Rd {
    const uint8_t *data;                // the decoder's buffer is data[0, end)
    uint64_t end;
    uint64_t *raw;                      // LDS: T2D_RAW / 8 words, data[...] from ADDRESS rwbase (8-byte aligned; ~0: nothing loaded)
    uint64_t rwbase;
    uint32_t *bits;                     // LDS: T2D_CHUNK * 8 / 32 + 4 words, most significant bit first
    uint16_t *sbit;                     // LDS: T2D_CHUNK + 1: first bit of every raw byte of the chunk; [nbytes] = the chunk's bit count
    uint64_t p0;                        // the chunk starts at data[p0] ...
    uint32_t nbytes, nbits;             // ... holds this many raw bytes / bits (nbytes < T2D_CHUNK only at the end of the buffer)
    uint32_t bitpos;                    // the reader's position in the chunk
    uint64_t w;                         // the next `wav` bits of the chunk, from bit 63 down
    uint32_t wav, wnext;                // wnext: index of the next buffer word to append
    bool have_chunk, eof;
    bool stale;                         // the register window does not match bitpos (the fast block path moved it): fetch() starts it again
#ifdef J2K_T2D_STATS
    uint64_t t_reload = 0, t_build = 0; uint32_t n_reload = 0, n_build = 0;
#endif
    uint64_t pend_pos; uint32_t pend_first; bool pend_sff;      // seat()
    int lane;

    __device__ __forceinline__ void raw_ensure(uint64_t a0, uint64_t a1) {       // addresses [a0, a1) (+ one byte before) staged
        if (rwbase != ~0ull && a0 >= rwbase + 1 && a1 <= rwbase + T2D_RAW) return;
#ifdef J2K_T2D_STATS
        const uint64_t ts_ = wall_clock64(); n_reload++;
#endif
        __syncthreads();
        rwbase = (a0 - 1) & ~7ull;                                               // (the byte before a0 decides a0's width)
        const uint64_t lo = (uint64_t)(uintptr_t)data & ~7ull, lim = ((uint64_t)(uintptr_t)data + end + 7) & ~7ull;   // words that overlap data[0, end) only
        for (int k = lane; k < T2D_RAW / 8; k += 64) {
            const uint64_t a = rwbase + 8ull * (uint64_t)k;
            raw[k] = (a >= lo && a < lim) ? *reinterpret_cast<const uint64_t *>(data + (int64_t)(a - (uint64_t)(uintptr_t)data)) : 0ull;   // (from `data`: a global load, not a flat one)
        }
        __syncthreads();
#ifdef J2K_T2D_STATS
        t_reload += wall_clock64() - ts_;
#endif
    }
    __device__ __forceinline__ uint32_t raw_at(uint64_t addr) const { return reinterpret_cast<const uint8_t *>(raw)[addr - rwbase]; }
    __device__ __forceinline__ uint32_t byte_at(uint64_t pos) {    // pos < end (the marker tests, t2.go:470-486)
        const uint64_t a = (uint64_t)(uintptr_t)data + pos;
        raw_ensure(a, a + 1);
        return t2_uni(raw_at(a));
    }
    // the chunk that starts at data[pos]: its first byte holds `first` bits (0: by the rule, from the byte before it or `sff` when pos == 0
    // or the caller says so with force_sff >= 0)
    __device__ __forceinline__ void build(uint64_t pos, uint32_t first, int force_sff) {
        const uint64_t a0 = (uint64_t)(uintptr_t)data + pos;
        const uint32_t nb = (uint32_t)(end - pos < T2D_CHUNK ? end - pos : T2D_CHUNK);
#ifdef J2K_T2D_STATS
        const uint64_t tb_ = wall_clock64(); n_build++;
#endif
        raw_ensure(a0, a0 + (nb ? nb : 1));
        for (int k = lane; k < T2D_CHUNK * 8 / 32 + 4; k += 64) bits[k] = 0;
        __syncthreads();
        uint32_t wd[4], by[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i = 4u * (uint32_t)lane + (uint32_t)j;
            by[j] = i < nb ? raw_at(a0 + i) : 0u;
            uint32_t prev_ff;
            if (i == 0) prev_ff = force_sff >= 0 ? (uint32_t)force_sff : (pos > 0 ? (uint32_t)(raw_at(a0 - 1) == 0xFFu) : 0u);
            else prev_ff = (uint32_t)(raw_at(a0 + i - 1) == 0xFFu);
            wd[j] = i < nb ? (prev_ff ? 7u : 8u) : 0u;
            if (i == 0 && first) wd[j] = nb ? first : 0u;
            tot += wd[j];
        }
        uint32_t incl = tot;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        uint32_t off = incl - tot;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i = 4u * (uint32_t)lane + (uint32_t)j;
            if (i <= nb) sbit[i] = (uint16_t)off;
            if (wd[j]) {
                const uint32_t v = by[j] & (0xFFu >> (8u - wd[j])), wi = off >> 5, sh = off & 31u;
                if (sh + wd[j] <= 32) atomicOr(&bits[wi], v << (32 - sh - wd[j]));
                else { const uint32_t lo_ = sh + wd[j] - 32; atomicOr(&bits[wi], v >> lo_); atomicOr(&bits[wi + 1], v << (32 - lo_)); }
            }
            off += wd[j];
        }
        nbits = t2_uni((uint32_t)__shfl(incl, 63));
        if (lane == 63 && nb == T2D_CHUNK) sbit[T2D_CHUNK] = (uint16_t)nbits;    // (lanes cover i = 0 .. 255; the closing entry)
        __syncthreads();
        p0 = pos; nbytes = nb; bitpos = 0; w = 0; wav = 0; wnext = 0; have_chunk = true; stale = false;
#ifdef J2K_T2D_STATS
        t_build += wall_clock64() - tb_;
#endif
    }
    // the raw byte the reader is in, or would take next: bytes [0, j) of the chunk are used up entirely
    __device__ __forceinline__ uint32_t byte_index() const {
        // sbit[j] <= bitpos < sbit[j + 1]; bitpos == nbits -> nbytes
        if (bitpos >= nbits) return nbytes;
        uint32_t lo = bitpos >> 3, hi = bitpos / 7u + 3u;                        // a byte holds 7 or 8 bits (the chunk's first one: 1 ... 8)
        if (hi > nbytes) hi = nbytes;
        uint32_t j = lo;
        for (uint32_t i = lo; i < hi; i++) if (t2_uni(sbit[i]) <= bitpos) j = i;    // (a handful of entries; uniform)
        return j;
    }
    // where the reader stands before its first chunk is made: at data[pos], `first` bits of that byte left (0: an untouched byte whose
    // width the flag decides -- the reference's sawFF)
    __device__ __forceinline__ void seat(uint64_t pos, uint32_t first, bool sff) { pend_pos = pos; pend_first = first; pend_sff = sff; have_chunk = false; }
    __device__ __forceinline__ void fetch() {                       // keep more than 32 bits in the window while the chunk has them
        if (stale) {
            const uint32_t idx = bitpos >> 5, sh = bitpos & 31u;
            w = idx * 32u < nbits ? (uint64_t)(t2_uni(bits[idx]) << sh) << 32 : 0ull;
            wav = 32u - sh; wnext = idx + 1;
            stale = false;
        }
        while (wav <= 32 && wnext * 32u < nbits) { w |= (uint64_t)t2_uni(bits[wnext]) << (32u - wav); wav += 32; wnext++; }
    }
    // THE place where chunks are made (one call site in the kernel: the chunk builder is inlined once): the first chunk after seat(),
    // and a new one from the reader's position whenever fewer than 128 bits are left in a chunk that is not the buffer's last (128: what the
    // whole-code-block path below asks for -- with 64 the last blocks of every chunk went through the stepwise reader: 181 -> 168 us per tile chain)
    __device__ __forceinline__ void ensure() {
        uint64_t bp = 0; uint32_t bf = 0; int bs = -1; bool go = false;
        if (!have_chunk) {
            if (pend_pos < end) { bp = pend_pos; bf = pend_first; bs = pend_first ? -1 : (pend_sff ? 1 : 0); go = true; }
        } else if (nbits - bitpos < 128 && p0 + nbytes < end) {
            const uint32_t j = byte_index();
            bp = p0 + j; bf = j < nbytes ? t2_uni(sbit[j + 1]) - bitpos : 0u; go = true;      // (the bits of byte j still unread; a fresh byte: by the rule)
        }
        if (go) build(bp, bf, bs);
    }
    __device__ __forceinline__ bool more_data() const { return have_chunk ? p0 + nbytes < end : pend_pos < end; }
    __device__ __forceinline__ uint32_t get(uint32_t n) {           // ReadBits(n), 0 <= n <= 32, after ensure(); 0 at the end of the buffer (eof set)
        if (n == 0) return 0;
        if (!have_chunk || nbits - bitpos < n) { eof = true; return 0; }
        fetch();
        const uint32_t v = (uint32_t)(w >> (64u - n));
        w <<= n; wav -= n; bitpos += n;
        return v;
    }
    // decodeTagTreeValue (t2.go:574-590), resumable: zeros are added to v up to the first one (true) or to the end of the chunk (false:
    // ensure() and call again -- a value can be longer than any chunk)
    __device__ __forceinline__ bool unary_some(uint32_t &v) {
        for (;;) {
            if (!have_chunk || bitpos >= nbits) return false;
            fetch();
            const uint32_t left = nbits - bitpos, m = wav < left ? wav : left;      // real bits in the window
            const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
            if (lz >= m) { v += m; w = m < 64 ? w << m : 0; wav -= m; bitpos += m; continue; }
            v += lz;
            w <<= lz + 1; wav -= lz + 1; bitpos += lz + 1;
            return true;
        }
    }
    // the reference reader's state: bytes taken so far, bits left in the last one, that byte, whether it was 0xFF
    __device__ __forceinline__ void state(uint64_t &rpos, uint32_t &cnt, uint32_t &buf, bool &sff) {
        if (!have_chunk) {                                          // nothing read since seat()
            rpos = pend_first ? pend_pos + 1 : pend_pos; cnt = pend_first; sff = pend_sff;
            buf = sff ? 0xFFu : 0u;
            if (pend_first && pend_pos < end) buf = byte_at(pend_pos);
            return;
        }
        uint32_t j = byte_index();                                  // byte j is the one in progress (cnt > 0) or the next one (cnt == 0)
        const bool partial = j < nbytes && t2_uni(sbit[j]) < bitpos;
        const uint32_t taken = partial ? j + 1 : j;
        rpos = p0 + taken;
        cnt = partial ? t2_uni(sbit[j + 1]) - bitpos : 0u;
        if (taken > 0 || p0 > 0) {
            const uint64_t a = (uint64_t)(uintptr_t)data + rpos - 1;
            raw_ensure(a, a + 1);
            buf = t2_uni(raw_at(a));
        } else buf = 0;
        sff = buf == 0xFFu && rpos > 0;
    }
};

// one decoder object's packets: [packet0, packet0 + npackets) on data[start, end); the state in and out as j2k_t2_dec_state
struct T2Chain {
    uint64_t start, end;                // the decoder's buffer within d_data (positions in the state are relative to `start`)
    int64_t packet0, npackets;
    j2k_t2_dec_state st;
    int32_t status, done;               // out: J2K_OK or the first failing packet's status; packets decoded before it
    int32_t skip, pad_;                 // != 0: the chain was found malformed before it started (status set): nothing to do
#ifdef J2K_T2D_STATS
    uint64_t t_total, t_reload, t_build; uint32_t n_reload, n_build, n_blocks, n_packets;   // dev: wall_clock64 ticks (100 MHz) per part
#endif
};

__global__ __launch_bounds__(64) void t2_decode_kernel(T2Chain *__restrict__ chains, const j2k_t2_dev_packet *__restrict__ packets, long npackets_all,
                                                       j2k_t2_dev_cb *__restrict__ cbs, uint64_t ncbs, const uint8_t *__restrict__ data, int sop, int eph, int clean,
                                                       uint64_t *__restrict__ body_base, int *__restrict__ frame_status,
                                                       const T2Chain *__restrict__ pchains, const uint64_t *__restrict__ seeds, const uint32_t *__restrict__ tile_par,
                                                       const int *__restrict__ tile_packet0) {
    __shared__ uint64_t raw[T2D_RAW / 8];
    __shared__ uint32_t bits[T2D_CHUNK * 8 / 32 + 4];
    __shared__ uint16_t sbit[T2D_CHUNK + 2];
    const int lane = threadIdx.x;
    T2Chain &Cn = chains[blockIdx.x];
#ifdef J2K_T2D_STATS
    const uint64_t t_start_ = wall_clock64(); uint32_t nblk_ = 0; uint64_t t_fast_ = 0;
#endif
    if (Cn.skip) {
        if (Cn.status) {                                            // a tile-part found malformed: its packets have no bodies (not a stale entry of the frame before)
            for (int64_t k = lane; k < Cn.npackets && Cn.packet0 >= 0 && Cn.packet0 + k < npackets_all; k += 64) body_base[Cn.packet0 + k] = ~0ull;
            if (lane == 0 && frame_status) atomicMin(frame_status, Cn.status);
        }
        return;
    }
    if (pchains && tile_par[blockIdx.x]) {
        // The tile launch behind a packet-parallel one (see "the packets of a tile side by side" below): every packet of this tile was
        // decoded on its own from a guessed state.  Kept if each packet ENDED in the state (position, carried flag) the next one was
        // started from -- by induction the run this chain would make; otherwise the chain runs (and writes every field again).
        const int p0 = tile_packet0[blockIdx.x], npk = tile_packet0[blockIdx.x + 1] - p0;
        bool bad = false;
        for (int p = lane; p < npk; p += 64) {
            const T2Chain &Q = pchains[p0 + p];
            if (Q.status != J2K_OK || Q.done != 1) bad = true;
            else if (p + 1 < npk && (Q.st.pos << 1 | (uint64_t)(Q.st.saw_ff != 0)) != seeds[p0 + p + 1]) bad = true;
        }
        if (__ballot(bad) == 0) {
            if (lane == 0) {
                Cn.status = J2K_OK; Cn.done = npk; Cn.st = pchains[p0 + npk - 1].st;
                if (frame_status) atomicAdd(&frame_status[1], 1);   // (a count for j2k_plan_frame_parallel_tiles)
            }
            return;
        }
    }
#ifdef J2K_T2D_STATS
    const uint64_t t_a_ = wall_clock64(); uint64_t t_hdr_ = 0, t_pre_ = 0, t_post_ = 0;
#endif
    const uint64_t base = Cn.start;
    T2Rd r;
    r.data = data + base; r.end = Cn.end - base; r.raw = raw; r.rwbase = ~0ull; r.bits = bits; r.sbit = sbit; r.lane = lane; r.eof = false;
    r.p0 = 0; r.nbytes = r.nbits = r.bitpos = r.wav = r.wnext = 0; r.w = 0; r.stale = false;
    // the decoder object's header reader as it was left (a zeroed state: NewPacketDecoder): cnt bits of byte rpos - 1 still unread
    {
        const uint32_t cnt = Cn.st.cnt > 8 ? 8u : Cn.st.cnt;
        if (cnt && Cn.st.rpos >= 1) r.seat(Cn.st.rpos - 1, cnt, Cn.st.saw_ff != 0);
        else r.seat(Cn.st.rpos, 0, Cn.st.saw_ff != 0);
    }
    bool carried_ff = Cn.st.saw_ff != 0;                            // closed-loop mode: the last header byte of the previous packet was 0xFF
    uint64_t pos = Cn.st.pos;
    const uint64_t end = r.end;
    int status = J2K_OK;
    int64_t done = 0;
    if (Cn.packet0 < 0 || Cn.npackets < 0 || Cn.packet0 + Cn.npackets > npackets_all) status = J2K_ERR_INVALID_ARG;
    for (int64_t k = 0; status == J2K_OK && k < Cn.npackets; k++) {
#ifdef J2K_T2D_STATS
        const uint64_t t_p0_ = wall_clock64();
#endif
        const int64_t pk = Cn.packet0 + k;
        const j2k_t2_dev_packet P = packets[pk];
        if (P.ncb < 0 || P.cb0 < 0 || (uint64_t)P.cb0 + (uint64_t)P.ncb > ncbs) { status = J2K_ERR_INVALID_ARG; break; }
        const int layer = P.layer;
        const uint32_t lenbits = (P.flags & J2K_T2_WIDE_LEN) ? 5u : 3u;
        if (P.flags & J2K_T2_FRESH) { carried_ff = false; if (!(P.flags & J2K_T2_SEATED)) r.seat(0, 0, false); }   // NewPacketDecoder
        if (sop && pos + 6 <= end && r.byte_at(pos) == 0xFFu && r.byte_at(pos + 1) == 0x91u) pos += 6;     // t2.go:470-474
        if (P.flags & J2K_T2_SEATED) r.seat(pos, 0, carried_ff);                                // closed-loop mode: the header starts at Position()
        uint64_t body = 0;                                                                       // bytes of the bodies this packet carries
        j2k_t2_dev_cb *pc = cbs + P.cb0;
        // The body loop (t2.go:489-499) takes bytes for every block with IncludedInLayers == layer and data -- also one this
        // header did not touch but whose fields say so from before (a table the caller filled, or an earlier packet of the same
        // layer).  `clean`: layer 0 is decoded only and the table holds nothing from before -- 1: the caller zeroed it; 2 (frame calls): it
        // may hold anything, this kernel writes EVERY field of every block of the packets it decodes (also of an empty packet) and flags
        // the packets it did not reach (body_base = ~0), which is all t2_finish_kernel looks at.  No such block, nothing is read.
        const bool old_matters = !clean || layer != 0;
        // decodePacketHeader (t2.go:506-571) as ONE loop over its reading steps -- presence bit, then per code-block inclusion,
        // zero bit planes, pass count + length -- so that the chunk builder above has a single call site
#ifdef J2K_T2D_STATS
        const uint64_t t_b_ = wall_clock64(); t_pre_ += t_b_ - t_p0_;
#endif
        enum { S_PRESENT, S_INCL, S_IMSB, S_REST, S_DONE };
        int step = S_PRESENT;
        int64_t i = 0;
        uint32_t present = 0, uv = 0;
        int incl_layers = 0;
        j2k_t2_dev_cb old{};
        const bool fast_ok = layer == 0 && !old_matters && P.incl_tree_w != 0 && P.imsb_tree_w != 0;
        while (step != S_DONE) {
            r.ensure();
            if (step == S_INCL && fast_ok && r.have_chunk && r.nbits - r.bitpos >= 128) {
                // Whole code-blocks straight from a 128-bit register window (four LDS words, one round trip each): inclusion value, zero
                // bit planes, pass count, length -- a tight loop over the blocks for as long as the chunk holds 128 bits more.  Anything
                // that might not fit (long unary values) is left to the stepwise reader below: nothing is committed before a block is complete.
#ifdef J2K_T2D_STATS
                const uint64_t tf_ = wall_clock64(); const int64_t i0_ = i;
#endif
                uint32_t bp = r.bitpos;
                const uint32_t lim = r.nbits - 128u;
                const uint32_t *bits_ = r.bits;
                bool bail = false;
                while (bp <= lim && i < P.ncb) {
                    // (the loop-carried state through readfirstlane as well: the compiler cannot see that it is uniform across the trips)
                    bp = t2_uni(bp);
                    i = (int64_t)t2_uni((uint32_t)i);                                            // (i < ncb < 2^31)
                    body = (uint64_t)t2_uni((uint32_t)(body >> 32)) << 32 | t2_uni((uint32_t)body);
                    const uint32_t idx = bp >> 5, sh = bp & 31u;
                    uint64_t hi = (uint64_t)t2_uni(bits_[idx]) << 32 | t2_uni(bits_[idx + 1]), lo = (uint64_t)t2_uni(bits_[idx + 2]) << 32 | t2_uni(bits_[idx + 3]);
                    hi = (hi << sh) | ((lo >> 1) >> (63u - sh)); lo <<= sh;                      // >= 97 bits from bit 63 of hi down
                    uint32_t used = 0;
                    auto take = [&](uint32_t n) -> uint32_t {                                    // 1 <= n <= 32
                        const uint32_t v = (uint32_t)(hi >> (64u - n));
                        hi = hi << n | lo >> (64u - n); lo <<= n; used += n;
                        return v;
                    };
                    const uint32_t z1 = hi ? (uint32_t)__builtin_clzll(hi) : 64u;
                    if (z1 > 30) { bail = true; break; }
                    (void)take(z1 + 1);
                    if (z1 != 0) {                                                               // not in this layer: only IncludedInLayers is written
                        if (lane == 0) *reinterpret_cast<int4 *>(&pc[i]) = int4{(int)z1, 0, 0, 0};   // (every field: the table need not be zero before)
                        bp += used; i++;
                        continue;
                    }
                    const uint32_t z2 = hi ? (uint32_t)__builtin_clzll(hi) : 64u;
                    if (z2 > 40) { bail = true; break; }                                         // (zero bit planes: 31 - numBPS in the frame calls)
                    (void)take(z2 + 1);
                    int np;                                                                      // t2.go:592-631
                    if (take(1) == 0) np = 1;
                    else if (take(1) == 0) np = 2;
                    else {
                        uint32_t v = take(2);
                        if (v < 3) np = (int)v + 3;
                        else {
                            v = take(5);
                            if (v < 31) np = (int)v + 6;
                            else np = (int)take(7) + 37;
                        }
                    }
                    const uint32_t nb = take(lenbits);
                    const uint32_t length = nb ? take(nb) : 0u;                                  // (nb <= 31; used <= 42 + 16 + 5 + 31 = 94 of the >= 97 bits)
                    if (lane == 0) {
                        *reinterpret_cast<int4 *>(&pc[i]) = int4{0, (int)z2, np, (int)length};
                        pc[i].data_off = body;
                    }
                    body += length;
                    bp += used; i++;
                }
#ifdef J2K_T2D_STATS
                t_fast_ += wall_clock64() - tf_; nblk_ += (uint32_t)(i - i0_);
#endif
                if (bp != r.bitpos) { r.bitpos = bp; r.stale = true; }
                if (i == P.ncb) { step = S_DONE; continue; }
                if (!bail) continue;                                                             // the chunk is nearly used up: ensure() makes the next
            }
            if (step == S_PRESENT) {
                present = r.get(1);
                if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                step = (present && P.ncb > 0) ? S_INCL : S_DONE;
                if (step == S_INCL && old_matters) old = pc[0];
                uv = 0;
            } else if (step == S_INCL) {                                                         // t2.go:516-540
                bool inc;
                if (layer == 0) {
                    if (P.incl_tree_w == 0) { status = J2K_ERR_GO_PANIC; break; }
                    if (!r.unary_some(uv)) { if (!r.more_data()) { status = J2K_ERR_INVALID_ARG; break; } continue; }
                    inc = uv == 0;
                    incl_layers = (int)uv;
                    if (lane == 0) { if (clean) *reinterpret_cast<int4 *>(&pc[i]) = int4{incl_layers, 0, 0, 0}; else pc[i].included_in_layers = incl_layers; }
                } else {
                    inc = r.get(1) == 1;
                    if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                    incl_layers = inc ? layer : old.included_in_layers;
                    if (inc && lane == 0) pc[i].included_in_layers = layer;
                }
                uv = 0;
                if (!inc) {
                    if (incl_layers == layer && old.data_len > 0) { if (lane == 0) pc[i].data_off = body; body += old.data_len; }
                    i++;
                    if (i == P.ncb) step = S_DONE;
                    else if (old_matters) old = pc[i];
                } else step = incl_layers == layer ? S_IMSB : S_REST;
            } else if (step == S_IMSB) {                                                         // t2.go:544-551
                if (P.imsb_tree_w == 0) { status = J2K_ERR_GO_PANIC; break; }
                if (!r.unary_some(uv)) { if (!r.more_data()) { status = J2K_ERR_INVALID_ARG; break; } continue; }
                if (lane == 0) pc[i].zero_bit_planes = (int)uv;
                uv = 0;
                step = S_REST;
            } else {                                                                             // t2.go:553-568, 592-648: at most 16 + 5 + 31 bits
                int np;
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
                const uint32_t nb = r.get(lenbits);
                const uint32_t length = r.get(nb);
                if (r.eof) { status = J2K_ERR_INVALID_ARG; break; }
                if (lane == 0) { pc[i].num_passes = np; pc[i].data_len = length; pc[i].data_off = body; }   // (offset within the packet's bodies; t2_bodies_kernel adds where they start)
                body += length;
                i++;
                if (i == P.ncb) step = S_DONE;
                else { step = S_INCL; if (old_matters) old = pc[i]; }
            }
        }
#ifdef J2K_T2D_STATS
        const uint64_t t_c_ = wall_clock64(); t_hdr_ += t_c_ - t_b_;
#endif
        if (status != J2K_OK) break;
        if (!present && clean == 2)                                                              // an empty packet: its blocks hold nothing
            for (int64_t q = lane; q < P.ncb; q += 64) pc[q] = j2k_t2_dev_cb{};
        if (!present && old_matters) {
            for (int64_t q = 0; q < P.ncb; q++) {
                const j2k_t2_dev_cb o = pc[q];
                if (o.included_in_layers == layer && o.data_len > 0) { if (lane == 0) pc[q].data_off = body; body += o.data_len; }
            }
        }
        if (P.flags & J2K_T2_SEATED) {                                                           // ... and Position() moves past the header
            uint32_t cnt_, buf_;
            r.state(pos, cnt_, buf_, carried_ff);
        }
        if (eph && pos + 2 <= end && r.byte_at(pos) == 0xFFu && r.byte_at(pos + 1) == 0x92u) pos += 2;     // t2.go:481-486
        if (pos + body > end) { status = J2K_ERR_INVALID_ARG; break; }                           // "unexpected end of packet data"
        if (lane == 0) body_base[pk] = base + pos;
        pos += body;
        done = k + 1;
#ifdef J2K_T2D_STATS
        t_post_ += wall_clock64() - t_c_;
#endif
    }
#ifdef J2K_T2D_STATS
    const uint64_t t_e_ = wall_clock64();
#endif
    uint64_t st_rpos; uint32_t st_cnt, st_buf; bool st_ff;
    r.state(st_rpos, st_cnt, st_buf, st_ff);                       // (by every lane: it may restage the window)
    if (lane == 0) {
        for (int64_t k = done; k < Cn.npackets && Cn.packet0 + k < npackets_all && Cn.packet0 + k >= 0; k++) body_base[Cn.packet0 + k] = ~0ull;
        Cn.st.pos = pos; Cn.st.rpos = st_rpos; Cn.st.buf = (uint8_t)st_buf; Cn.st.cnt = (uint8_t)st_cnt; Cn.st.saw_ff = st_ff ? 1 : 0;
        Cn.status = status; Cn.done = (int32_t)done;
#ifdef J2K_T2D_STATS
        Cn.t_total = wall_clock64() - t_start_; Cn.t_reload = r.t_reload; Cn.t_build = r.t_build; Cn.n_reload = r.n_reload; Cn.n_build = r.n_build; Cn.n_packets = (uint32_t)done;
        if (blockIdx.x == 0 || blockIdx.x == 39 || blockIdx.x == 5) printf("chain %d of %d: total %llu ticks, reload %llu (%u), build %llu (%u) incl. reloads, packets %u, fast loop %llu ticks for %u blocks; prologue %llu, per packet: before header %llu, header %llu, after %llu; loop end at %llu\n", (int)blockIdx.x, (int)gridDim.x, (unsigned long long)Cn.t_total,
                                                        (unsigned long long)Cn.t_reload, Cn.n_reload, (unsigned long long)Cn.t_build, Cn.n_build, Cn.n_packets, (unsigned long long)t_fast_, nblk_, (unsigned long long)(t_a_ - t_start_), (unsigned long long)t_pre_, (unsigned long long)t_hdr_, (unsigned long long)t_post_, (unsigned long long)(t_e_ - t_start_));
#endif
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
    if (at >= len || len - at < 14) return J2K_ERR_INVALID_ARG;      // (`at` is the caller's: no sum of it may wrap)
    uint32_t b[14];                                                 // SOT and the marker behind it in one round trip (normally SOD)
#pragma unroll
    for (int k = 0; k < 14; k++) b[k] = cs[at + k];
    if (b[0] != 0xFF || b[1] != 0x90) return J2K_ERR_INVALID_ARG;
    const uint32_t lsot = b[2] << 8 | b[3], isot = b[4] << 8 | b[5];
    const uint32_t psot = b[6] << 24 | b[7] << 16 | b[8] << 8 | b[9];
    if (lsot != 10 || isot != ((uint32_t)want_index & 0xFFFFu)) return J2K_ERR_INVALID_ARG;
    const uint64_t stop = psot ? at + psot : len;
    if (stop > len || stop < at + 14) return J2K_ERR_INVALID_ARG;
    uint64_t p = at + 12;
    if (b[12] == 0xFF && b[13] == 0x93) p += 2;
    else for (;;) {                                                 // parser.go:180-190: segments by length until SOD
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

// ---- the packets of a tile side by side (SOP + EPH streams) ------------------------------------------------------------------
// A tile's packets are one chain: packet p + 1 starts where the lengths in packet p's header say.  With SOP and EPH markers in the
// stream that place can be GUESSED without reading a header: neither FF91 nor FF92 can occur inside a header (a byte behind 0xFF keeps
// its top bit clear, bio.go:127-131) or inside an MQ-coded body (behind 0xFF the coder emits at most 0x8F, mqc.go byteout).  So:
//   t2_marks_kernel   every FF91 / FF92 of the tile-parts, found by all CUs (an unordered list per tile)
//   t2_seed_kernel    per tile: the list sorted; if it reads SOP EPH SOP EPH ... with one pair per packet of the plan, packet p's guess is
//                     (start = its SOP, carried flag = the byte before packet p - 1's EPH is 0xFF) -- one chain of ONE packet each
//   t2_decode_kernel  the same decoder as ever, a wavefront per packet
//   t2_decode_kernel  once more, a wavefront per TILE: first it checks that every packet was decoded and that the state packet p ENDED in
//                     (position, flag) is the state packet p + 1 was started from.  By induction that is the serial decoder's run, field for field.  If anything is off -- a marker pair
//                     inside an HT body (its bytes are not marker-free), a stream without the markers, a malformed header -- the
//                     wavefront runs the tile's chain as before, writing every field again: the guess decides speed, never the result.
#define T2P_MAXM 1024                   // markers of one tile that the sort takes (2 per packet)
__global__ __launch_bounds__(256) void t2_marks_kernel(const T2Chain *__restrict__ chains, const int *__restrict__ tile_packet0, const uint8_t *__restrict__ cs,
                                                       uint64_t *__restrict__ marks, uint32_t *__restrict__ cnt) {
    const int t = blockIdx.y;
    if (chains[t].skip) return;
    const uint64_t a0 = chains[t].start, a1 = chains[t].end;
    const uint32_t cap = 2u * (uint32_t)(tile_packet0[t + 1] - tile_packet0[t]);
    uint64_t *m = marks + 2 * (size_t)tile_packet0[t];
    auto found = [&](uint64_t q, uint32_t nx) {
        const uint32_t k = atomicAdd(&cnt[t], 1u);
        if (k < cap) m[k] = q << 1 | (uint64_t)(nx == 0x92u);
    };
    const uint64_t base = (uint64_t)(uintptr_t)cs;
    // positions [b0, b1): the 16-byte pieces (by ADDRESS) that lie inside the tile-part; [a0, b0) and [b1, a1) byte by byte
    uint64_t b0 = ((base + a0 + 15) & ~15ull) - base, b1 = ((base + a1) & ~15ull) - base;
    if (b1 < b0) b0 = b1 = a0;
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const uint64_t q = threadIdx.x < 16 ? a0 + threadIdx.x : (b1 > a0 ? b1 : a0) + (threadIdx.x - 16);
        const bool in = (threadIdx.x < 16 ? q < b0 : q >= b1) && q + 1 < a1;
        const uint32_t c0 = cs[in ? q : a0], c1 = cs[in ? q + 1 : a0];              // (a1 - a0 >= 1: the chain was checked)
        if (in && c0 == 0xFFu && (c1 == 0x91u || c1 == 0x92u)) found(q, c1);
    }
    const uint64_t nvec = (b1 - b0) >> 4;
    // 16 KiB per workgroup and trip: four pieces per thread and the byte behind each, every load unconditional (clamped) and in flight
    // before the first is looked at
    for (uint64_t c0 = (uint64_t)blockIdx.x * 1024; c0 < nvec; c0 += (uint64_t)gridDim.x * 1024) {
        uint4 x[4]; uint32_t nx4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t v = c0 + (uint64_t)(u * 256 + (int)threadIdx.x), vc = v < nvec ? v : nvec - 1;
            const uint64_t p = b0 + 16 * vc;
            x[u] = *reinterpret_cast<const uint4 *>(cs + (int64_t)p);
            nx4[u] = cs[p + 16 < a1 ? p + 16 : p];                                   // (the byte behind the piece; none behind the buffer's last)
        }
        // any byte 0xFF with 0x91 / 0x92 behind it?  Four bytes a step: z has a zero byte exactly there (the usual zero-byte test is exact
        // for "is there one").  Looking at every 0xFF byte instead (one piece in sixteen has one) made this kernel compute-bound.
        uint32_t hit = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t d[5] = {x[u].x, x[u].y, x[u].z, x[u].w, nx4[u]};
            uint32_t any = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e = __builtin_amdgcn_alignbyte(d[j + 1], d[j], 1), nf = ~d[j];
                const uint32_t z1 = nf | (e ^ 0x91919191u), z2 = nf | (e ^ 0x92929292u);
                any |= ((z1 - 0x01010101u) & ~z1 & 0x80808080u) | ((z2 - 0x01010101u) & ~z2 & 0x80808080u);
            }
            if (any && c0 + (uint64_t)(u * 256 + (int)threadIdx.x) < nvec) hit |= 1u << u;
        }
        if (__ballot(hit != 0) == 0) continue;                      // (most wavefronts: a tile holds a few dozen markers)
        // a wavefront with a marker (one in three on a 4K frame): the same words again with the EXACT zero-byte flags (bit 7 of every
        // zero byte and of no other), a marker per set bit -- all in registers (reading the piece again byte by byte cost a round trip each)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!(hit >> u & 1)) continue;
            const uint32_t d[5] = {x[u].x, x[u].y, x[u].z, x[u].w, nx4[u]};
            const uint64_t p = b0 + 16 * (c0 + (uint64_t)(u * 256 + (int)threadIdx.x));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e = __builtin_amdgcn_alignbyte(d[j + 1], d[j], 1), nf = ~d[j];
                const uint32_t z1 = nf | (e ^ 0x91919191u), z2 = nf | (e ^ 0x92929292u);
                const uint32_t f1 = ~(((z1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z1 | 0x7F7F7F7Fu), f2 = ~(((z2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z2 | 0x7F7F7F7Fu);
                uint32_t f = f1 | f2 >> 1;                          // bit 7: SOP at that byte, bit 6: EPH
                while (f) {
                    const int bit = __builtin_ctz(f);
                    f &= f - 1;
                    const uint64_t q = p + (uint64_t)(4 * j + (bit >> 3));
                    if (q + 1 < a1) found(q, (bit & 7) == 7 ? 0x91u : 0x92u);
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void t2_seed_kernel(const T2Chain *__restrict__ chains, const int *__restrict__ tile_packet0, const uint8_t *__restrict__ cs,
                                                      const uint64_t *__restrict__ marks, uint32_t *__restrict__ cnt, T2Chain *__restrict__ pchains,
                                                      uint64_t *__restrict__ seeds, uint32_t *__restrict__ tile_par) {
    __shared__ uint64_t in[T2P_MAXM], srt[T2P_MAXM];
    __shared__ int bad;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int p0 = tile_packet0[t], npk = tile_packet0[t + 1] - p0;
    const uint64_t a0 = chains[t].start, a1 = chains[t].end;
    const uint32_t n = cnt[t];
    bool ok = !chains[t].skip && npk > 0 && n == 2u * (uint32_t)npk && n <= T2P_MAXM;
    if (tid == 0) bad = 0;
    if (ok) {
        for (uint32_t j = tid; j < n; j += 256) in[j] = marks[2 * (size_t)p0 + j];
        __syncthreads();
        for (uint32_t j = tid; j < n; j += 256) {                   // positions are distinct: an entry's rank is its place
            const uint64_t e = in[j];
            uint32_t rk = 0;
            for (uint32_t q = 0; q < n; q++) rk += in[q] < e;
            srt[rk] = e;
        }
        __syncthreads();
        for (uint32_t j = tid; j < n; j += 256) {
            const uint64_t e = srt[j];
            if ((e & 1) != (j & 1)) bad = 1;                        // SOP EPH SOP EPH ...
            if ((j & 1) && (e >> 1) < (srt[j - 1] >> 1) + 7) bad = 1;   // ... with a header of a byte or more between them
        }
        if (tid == 0 && (srt[0] >> 1) != a0) bad = 1;
    }
    __syncthreads();
    ok = ok && !bad;
    if (tid == 0) cnt[t] = 0;                                       // (for the next frame: the workspace starts zeroed and every count is taken here)
    for (int p = tid; p < npk; p += 256) {
        T2Chain Q{};
        Q.start = a0; Q.end = a1; Q.packet0 = p0 + p; Q.npackets = 1; Q.skip = !ok;
        if (ok) {
            const uint64_t s = (srt[2 * p] >> 1) - a0;
            const uint32_t ff = p > 0 && cs[(srt[2 * p - 1] >> 1) - 1] == 0xFFu;
            Q.st.pos = s; Q.st.saw_ff = (uint8_t)ff;
            seeds[p0 + p] = s << 1 | ff;
        }
        pchains[p0 + p] = Q;
    }
    if (tid == 0) tile_par[t] = ok;
}
// Code-block fields -> what j2k_plan_decode_blocks takes.  A block the packets did not include (or a chain that failed before it)
// has no data: length 0, no bit planes -- tcd.DecodeCodeBlock leaves its coefficients alone (tcd.go:394-396).  Bit planes: the MQ
// coder's pass count is 3 * numBPS - 2 (t1_fast5.go:66-70); an HT block carries one pass and the decoder does not use the count,
// so it is mb - ZeroBitPlanes there (what j2k_plan_t2_fill_cbs wrote).
__device__ __forceinline__ void t2_block_out(long j, const j2k_t2_dev_cb &cb, bool decoded, int ht, int mb, uint64_t total, uint64_t *__restrict__ offs,
                                             uint32_t *__restrict__ lens, uint8_t *__restrict__ numbps, int *__restrict__ status) {
    // (a body outside the buffer can only come from a chain that failed half way -- the frame's status says so; nothing is read there)
    const bool has = decoded && cb.included_in_layers == 0 && cb.data_len > 0 && cb.num_passes > 0 && cb.data_off <= total && cb.data_len <= total - cb.data_off;
    int nb = 0;
    if (has) nb = ht ? (mb > cb.zero_bit_planes ? mb - cb.zero_bit_planes : 0) : (cb.num_passes + 2) / 3;
    // more bit planes than an int32 coefficient has: no encoder of this library writes that (numBPS <= 31) -- a foreign stream; the block is
    // dropped and the frame's status says why
    const bool bad = nb > 31;
    if (bad && status) atomicMin(status, J2K_ERR_INVALID_ARG);
    offs[j] = has && !bad ? cb.data_off : 0;
    lens[j] = has && !bad ? cb.data_len : 0u;
    numbps[j] = (uint8_t)(bad ? 0 : nb);
}
__global__ __launch_bounds__(256) void t2_blocks_kernel(long n, const j2k_t2_dev_cb *__restrict__ cbs, int ht, int mb, uint64_t total, uint64_t *__restrict__ offs,
                                                        uint32_t *__restrict__ lens, uint8_t *__restrict__ numbps, int *__restrict__ status) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    t2_block_out(j, cbs[j], true, ht, mb, total, offs, lens, numbps, status);
}
// t2_bodies_kernel and t2_blocks_kernel in one pass for a plan's own packets (every block of the plan is in exactly one of them): one
// workgroup per packet; a packet its chain did not reach (body_base = ~0) has no data
__global__ __launch_bounds__(256) void t2_finish_kernel(const j2k_t2_dev_packet *__restrict__ packets, j2k_t2_dev_cb *__restrict__ cbs, uint64_t ncbs,
                                                        const uint64_t *__restrict__ body_base, int ht, int mb, uint64_t total, uint64_t *__restrict__ offs,
                                                        uint32_t *__restrict__ lens, uint8_t *__restrict__ numbps, int *__restrict__ status) {
    const j2k_t2_dev_packet P = packets[blockIdx.x];
    const uint64_t b = body_base[blockIdx.x];
    if (P.ncb < 0 || P.cb0 < 0 || (uint64_t)P.cb0 + (uint64_t)P.ncb > ncbs) return;
    for (int64_t i = threadIdx.x; i < P.ncb; i += 256) {
        j2k_t2_dev_cb cb = cbs[P.cb0 + i];
        if (b != ~0ull && cb.included_in_layers == P.layer && cb.data_len > 0) { cb.data_off += b; cbs[P.cb0 + i].data_off = cb.data_off; }
        t2_block_out((long)(P.cb0 + i), cb, b != ~0ull, ht, mb, total, offs, lens, numbps, status);
    }
}

// decoded block j (dense w x h at D.out_off) -> its window of the coefficient planes (S.src_off, row stride S.stride); one
// workgroup per block and 64 rows.  ystep = 4: only the rows y % 4 == 0 -- all the reference's HT decoder ever writes; the
// frame decoder keeps both buffers for itself, zeroed once, so the other rows are zero on both sides already
__global__ __launch_bounds__(256) void place_blocks_kernel(const BlockJob *__restrict__ src_jobs, const BlockJob *__restrict__ dec_jobs,
                                                           const int32_t *__restrict__ decoded, int32_t *__restrict__ coeff, int ystep) {
    const BlockJob S = src_jobs[blockIdx.x];
    const int64_t doff = dec_jobs[blockIdx.x].out_off;
    const int w = S.w, h = S.h;
    const int y0 = blockIdx.y * 64, y1 = min(h, y0 + 64);
    if (y0 >= h) return;
    const int nrows = (y1 - y0 + ystep - 1) / ystep;                // rows y0, y0 + ystep, ... (y0 is a multiple of 64)
    const int32_t *src = decoded + doff;
    int32_t *dst = coeff + S.src_off;
    if (!(w & 3) && !(doff & 3) && !(S.src_off & 3) && !(S.stride & 3)) {
        const int wq = w >> 2;
        for (int i = threadIdx.x; i < nrows * wq; i += 256) {
            const int y = y0 + (i / wq) * ystep, x = (i - (i / wq) * wq) << 2;
            *reinterpret_cast<int4 *>(dst + (int64_t)y * S.stride + x) = *reinterpret_cast<const int4 *>(src + (int64_t)y * w + x);
        }
    } else {
        for (int i = threadIdx.x; i < nrows * w; i += 256) {
            const int y = y0 + (i / w) * ystep, x = i - (i / w) * w;
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
                       clean, body_base, frame_status, (const T2Chain *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const int *)nullptr);
    hipLaunchKernelGGL(t2_bodies_kernel, dim3((unsigned)npackets), dim3(256), 0, s, packets, cbs, ncbs, body_base);
    return hipGetLastError();
}
// a frame's tile chains (launch_t2_tile_chains), each tile's packets side by side where its markers allow it; ws: t2_par_workspace() bytes
size_t t2_par_workspace(long npackets, int ntiles) {
    return (size_t)npackets * (sizeof(T2Chain) + 16 + 8) + (size_t)ntiles * 8 + 64;
}
hipError_t launch_t2_decode_tiles(hipStream_t s, void *chains, int ntiles, const int *tile_packet0, const j2k_t2_dev_packet *packets, long npackets,
                                  j2k_t2_dev_cb *cbs, uint64_t ncbs, const uint8_t *data, uint64_t len, int sop, int eph, uint64_t *body_base, int *frame_status, void *ws,
                                  int ht, int mb, uint64_t *offs, uint32_t *lens, uint8_t *numbps) {
    if (ntiles <= 0 || npackets <= 0) return hipSuccess;
    T2Chain *tc = reinterpret_cast<T2Chain *>(chains);
    if (sop && eph && ws) {
        T2Chain *pch = reinterpret_cast<T2Chain *>(ws);
        uint64_t *marks = reinterpret_cast<uint64_t *>(pch + npackets), *seeds = marks + 2 * npackets;
        uint32_t *cnt = reinterpret_cast<uint32_t *>(seeds + npackets), *tile_par = cnt + ntiles;
        // pieces of 16 bytes, four per thread and trip: workgroups of 16 KiB of the average tile-part
        // (`len` is the caller's buffer, often a bound well above the bytes in use: at most ~2048 workgroups -- one that finds nothing to
        // do still waits for the tile's chain record)
        const unsigned seg = (unsigned)std::min<uint64_t>(std::max(1, 2048 / ntiles), std::max<uint64_t>(1, len / (uint64_t)ntiles / 16384 + 1));
        hipLaunchKernelGGL(t2_marks_kernel, dim3(seg, (unsigned)ntiles), dim3(256), 0, s, tc, tile_packet0, data, marks, cnt);
        hipLaunchKernelGGL(t2_seed_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, tc, tile_packet0, data, marks, cnt, pch, seeds, tile_par);
        hipLaunchKernelGGL(t2_decode_kernel, dim3((unsigned)npackets), dim3(64), 0, s, pch, packets, npackets, cbs, ncbs, data, sop, eph, 2, body_base, (int *)nullptr,
                           (const T2Chain *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const int *)nullptr);
        hipLaunchKernelGGL(t2_decode_kernel, dim3((unsigned)ntiles), dim3(64), 0, s, tc, packets, npackets, cbs, ncbs, data, sop, eph, 2, body_base, frame_status,
                           (const T2Chain *)pch, (const uint64_t *)seeds, (const uint32_t *)tile_par, tile_packet0);
    } else
        hipLaunchKernelGGL(t2_decode_kernel, dim3((unsigned)ntiles), dim3(64), 0, s, tc, packets, npackets, cbs, ncbs, data, sop, eph, 2, body_base, frame_status,
                           (const T2Chain *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const int *)nullptr);
    hipLaunchKernelGGL(t2_finish_kernel, dim3((unsigned)npackets), dim3(256), 0, s, packets, cbs, ncbs, body_base, ht, mb, len, offs, lens, numbps, frame_status);
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
hipError_t launch_t2_blocks(hipStream_t s, long n, const j2k_t2_dev_cb *cbs, int ht, int mb, uint64_t total, uint64_t *offs, uint32_t *lens, uint8_t *numbps, int *status) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(t2_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, cbs, ht, mb, total, offs, lens, numbps, status);
    return hipGetLastError();
}
hipError_t launch_place_blocks(hipStream_t s, const BlockJob *src_jobs, const BlockJob *dec_jobs, int njobs, int max_h, const int32_t *decoded, int32_t *coeff, int ystep) {
    if (njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(place_blocks_kernel, dim3((unsigned)njobs, (unsigned)((max_h + 63) / 64)), dim3(256), 0, s, src_jobs, dec_jobs, decoded, coeff, ystep == 4 ? 4 : 1);
    return hipGetLastError();
}

}  // namespace j2k
