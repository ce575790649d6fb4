// ht.hip -- the reference's "HT" block coder (internal/entropy/ht.go), bug for bug.
//
//   HTEncoder.SetData + Encode  ht.go:935-1045 (+ encodeCleanup :1048-1196, encodeVLCQuad :1199-1226,
//                               encodeUVLC :1229-1263, vlcWrite/Flush :1266-1300, magSgnWrite/Flush :1303-1341)
//   HTDecoder.Decode            ht.go:93-150 (+ initMEL :153-195, initVLC :276-314, rev* :317-396,
//                               initMagSgn :399-429, frwd* :432-519, decodeCleanup :583-713, UVLC :716-864)
//
// What the reference coder actually does (SURVEY.md 8a quirk list) and therefore what is
// reproduced: only row y of every 4-row stripe is coded; quad = 4 horizontally adjacent
// samples; the first quad's context is always 0, the second's is rho>>2; the encoder ORs the
// UNMASKED 7-bit table index into the VLC stream while advancing by the entry's length; MEL is
// never fed (max(64,2wh)/4 zero bytes); output = MagSgn | MEL zeros | VLC (write order) | SCUP.
//
// Encoder shape: one code-block per wavefront.  The wave scans the whole block for max|x| (nil /
// numbps decision); every quad pair's code words and MagSgn fields are formed by one lane, their
// bit positions are prefix sums of the lengths, and they are OR-deposited into two LDS bit strings
// (ht_form); the bytes are then cut from the strings 256 at a time with the 0xFF stuffing rules
// (ht_emit).  Blocks with more than HT_FAST_MAX_SAMPLES coded samples take a bit-serial path on
// lane 0.  Decoder shape: three kernels, see "parallel decoder" below.  Go semantics kept: uint32
// wraparound, shifts >= 32 give 0.
#include <cstdlib>

#include "ht_tables.h"
#include "j2k_internal.h"

namespace j2k {

__device__ __constant__ uint16_t c_vlc_tbl0[1024] = J2K_HT_VLC_TBL0_INIT;
__device__ __constant__ uint16_t c_vlc_tbl1[1024] = J2K_HT_VLC_TBL1_INIT;
// encoder view of the same tables: [initial][context 0..3][rho 0..15] -> (cwd << 4) | len of the FIRST
// cwd in [0,128) whose entry has that rho and a non-zero length (ht.go:1211-1222); 0x0001 = fallback (0,1)
__device__ uint16_t g_vlc_enc[2 * 4 * 16];

__global__ void ht_build_enc_table() {
    const int t = threadIdx.x;  // 128 threads
    if (t >= 128) return;
    const int initial = t >> 6, context = (t >> 4) & 3, rho = t & 15;
    const uint16_t *tbl = initial ? c_vlc_tbl0 : c_vlc_tbl1;
    uint16_t r = 0x0001;
    for (uint32_t cwd = 0; cwd < 128; cwd++) {
        const uint16_t e = tbl[(context << 7) | cwd];
        if (((e >> 4) & 0xF) == rho && (e & 0xF) > 0) { r = (uint16_t)((cwd << 4) | (e & 0xF)); break; }
    }
    g_vlc_enc[t] = r;
}

// decoder view for the rows after the first (tbl1): the next 14 stream bits -> BOTH quads of a quad pair, valid whenever
// the first quad's "length" nibble is <= 7 (then the second quad's 7 index bits lie inside the 14):
//   rho1 | rho2 << 4 | (len1 + len2) << 8 | uOff2 << 13 | uOff1 << 14 ; bit 15 = first length > 7, take the two-step path
// (ht.go:600-640: first quad context 0, second quad context rho1 >> 2)
__device__ uint16_t g_vlc_pair1[16384];

__global__ void ht_build_pair_table() {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 16384) return;
    const uint32_t e1 = c_vlc_tbl1[idx & 0x7F];
    const uint32_t len1 = e1 & 0xF, rho1 = (e1 >> 4) & 0xF, uo1 = (e1 >> 3) & 1;
    uint32_t r = 0x8000;
    if (len1 <= 7) {
        const uint32_t e2 = c_vlc_tbl1[((rho1 >> 2) << 7) | ((idx >> len1) & 0x7F)];
        const uint32_t len2 = e2 & 0xF, rho2 = (e2 >> 4) & 0xF, uo2 = (e2 >> 3) & 1;
        r = rho1 | rho2 << 4 | (len1 + len2) << 8 | uo2 << 13 | uo1 << 14;
    }
    g_vlc_pair1[idx] = (uint16_t)r;
}

__device__ __forceinline__ uint32_t shl32(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
__device__ __forceinline__ uint64_t shl64(uint64_t x, uint32_t n) { return n >= 64 ? 0ull : x << n; }
__device__ __forceinline__ uint64_t shr64(uint64_t x, uint32_t n) { return n >= 64 ? 0ull : x >> n; }
__device__ __forceinline__ uint32_t uabs(int v) { return v < 0 ? 0u - (uint32_t)v : (uint32_t)v; }

// Inclusive prefix sum over the 64 lanes on the DPP cross-lane network (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU instructions and no LDS round trip -- the
// __shfl_up ladder it replaces is six dependent ds_bpermute_b32 (~100 cycles each), and these scans sit on the one
// wavefront's critical path in the encoder's and decoder's bit-position bookkeeping.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// ---------------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------------
struct BitWriter {
    uint8_t *p;       // destination
    long pos, cap;    // bytes written / capacity (Go: len(data))
    uint64_t tmp;
    int bits;
    uint32_t last;
    int fault;
};

__device__ __forceinline__ void vlc_write(BitWriter &v, uint32_t val, uint32_t nbits) {  // ht.go:1266-1286
    v.tmp |= shl64((uint64_t)val, (uint32_t)v.bits);
    v.bits += (int)nbits;
    while (v.bits >= 8) {
        uint32_t b = (uint32_t)(v.tmp & 0xFF);
        if (v.last > 0x8F && (b & 0x7F) == 0x7F) b &= 0x7F;
        if (v.pos >= v.cap) { v.fault = 1; v.bits = 0; return; }
        v.p[v.pos++] = (uint8_t)b;
        v.last = b;
        v.tmp >>= 8;
        v.bits -= 8;
    }
}

__device__ __forceinline__ void ms_write(BitWriter &m, uint32_t val, uint32_t nbits) {  // ht.go:1303-1327
    m.tmp |= shl64((uint64_t)val, (uint32_t)m.bits);
    m.bits += (int)nbits;
    while (m.bits >= 8) {
        uint32_t b = (uint32_t)(m.tmp & 0xFF);
        if (m.pos >= m.cap) { m.fault = 1; m.bits = 0; return; }
        if (m.last == 0xFF) { b &= 0x7F; m.p[m.pos++] = (uint8_t)b; m.tmp >>= 7; m.bits -= 7; }
        else { m.p[m.pos++] = (uint8_t)b; m.tmp >>= 8; m.bits -= 8; }
        m.last = b;
    }
}

__device__ __forceinline__ void flush_bits(BitWriter &w) {  // vlcFlush / magSgnFlush: ht.go:1289-1300, 1330-1341
    while (w.bits > 0) {
        if (w.pos >= w.cap) { w.fault = 1; return; }
        w.p[w.pos++] = (uint8_t)(w.tmp & 0xFF);
        w.tmp >>= 8;
        w.bits -= 8;
        if (w.bits < 0) w.bits = 0;
    }
}

__device__ __forceinline__ void uvlc_one(BitWriter &v, uint32_t u) {  // ht.go:1242-1249
    if (u <= 1) vlc_write(v, 1, 1);
    else if (u <= 2) vlc_write(v, 2, 2);
    else { vlc_write(v, 0, 3); vlc_write(v, u - 3, 5); }
}

// ---- wave-parallel encoder (fast path) --------------------------------------------------------
// Every quad pair of every coded row is an ITEM; its VLC bits (two table code words + UVLC) and its
// MagSgn bits depend only on its own 8 samples (the first quad's context is always 0, the second's is
// rho>>2), so all items are formed in parallel, one per lane.  An exclusive prefix sum of the bit
// lengths gives every item its position; the bits are OR-ed into LDS bit strings (OR, because the
// reference ORs the UNMASKED table index into its buffer and the stray high bits overlap the next
// code word -- ht.go:1266-1268).  Byte stuffing is then applied while the bytes are emitted:
//   VLC   : a byte keeps its position, only bit 7 may be cleared depending on the previous FINAL byte;
//   MagSgn: after a 0xFF byte the next byte takes 7 bits, which shifts everything behind it, so
//           emission runs in 64-byte chunks that stop at the first 0xFF of the chunk.
#define HT_FAST_MAX_SAMPLES 1024                       /* coded samples (rows y%4==0) per block      */
#define HT_MS_WORDS (HT_FAST_MAX_SAMPLES * 31 / 32 + 4) /* worst case 31 bits per coded sample       */
#define HT_VLC_WORDS (46 * (HT_FAST_MAX_SAMPLES / 8) / 32 + 8) /* <= 15+15+16 bits per item (the "length" nibble reaches 15) */

__device__ __forceinline__ void or_bits(uint32_t *buf, uint32_t bitpos, uint64_t val) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint32_t lo = (uint32_t)(val << sh);
    const uint64_t hi = sh ? (val >> (32 - sh)) : (val >> 32);
    if (lo) atomicOr(&buf[wd], lo);
    if ((uint32_t)hi) atomicOr(&buf[wd + 1], (uint32_t)hi);
    if ((uint32_t)(hi >> 32)) atomicOr(&buf[wd + 2], (uint32_t)(hi >> 32));
}
// n zero bytes at p (any alignment), one wavefront: bytes up to the next 16-byte boundary, 16-byte stores, tail bytes.
// (The MEL segment is max(64, 2wh)/4 zero bytes -- 2 KB per 64x64 block; as 64 one-byte lanes per store it was 32 store
// instructions per block and a measurable part of the encoder.)
__device__ __forceinline__ void zero_bytes(uint8_t *p, size_t n, int lane) {
    const size_t head = min(n, (size_t)((16 - ((uintptr_t)p & 15)) & 15));
    if ((size_t)lane < head) p[lane] = 0;
    uint8_t *q = p + head;
    const size_t nv = (n - head) >> 4;
    for (size_t i = lane; i < nv; i += 64) reinterpret_cast<uint4 *>(q)[i] = make_uint4(0, 0, 0, 0);
    const size_t done = head + (nv << 4);
    if (done + lane < n) p[done + lane] = 0;
}

__device__ __forceinline__ uint32_t ms_bits32(const uint32_t *buf, uint32_t bitpos) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint64_t two = (uint64_t)buf[wd] | ((uint64_t)buf[wd + 1] << 32);
    return (uint32_t)(two >> sh);
}
__device__ __forceinline__ uint32_t get_bits8(const uint32_t *buf, uint32_t bitpos) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint64_t two = (uint64_t)buf[wd] | ((uint64_t)buf[wd + 1] << 32);
    return (uint32_t)(two >> sh) & 0xFF;
}

// Part 1: every quad pair's code words and MagSgn fields deposited into the two LDS bit strings.  Returns false on the
// reference's panic domain (MinInt32).  TM / TV = bits in the MagSgn / VLC strings.
// LROWS: the coded rows (y % 4 == 0) were staged in LDS by the caller's max scan (row r at lrows + r * w): no second trip to memory.
template <bool LROWS>
__device__ __forceinline__ bool ht_form(const BlockJob &J, const int32_t *__restrict__ src, int lane, uint32_t *vbuf, uint32_t *mbuf,
                        uint32_t &TMout, uint32_t &TVout, const int32_t *lrows, const uint16_t *enc) {
    const int w = J.w, h = J.h, stride = LROWS ? J.w : J.stride;
    const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, R = (h + 3) / 4, N = R * P;
    for (int i = lane; i < HT_VLC_WORDS; i += 64) vbuf[i] = 0;
    for (int i = lane; i < HT_MS_WORDS; i += 64) mbuf[i] = 0;
    __syncthreads();
    uint32_t vbase = 0, mbase = 0;   // running bit totals
    int bad = 0;
    const bool vec_ok = ((J.stride & 3) == 0) && ((J.src_off & 3) == 0);
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int it = i0 + lane;
        uint64_t vv = 0; uint32_t vl = 0, ml = 0;
        uint32_t mval[8], mlen[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { mval[i] = 0; mlen[i] = 0; }
        if (it < N) {
            const int r = it / P, pi = it - r * P;
            const int initial = (r == 0);
            const int32_t *row = LROWS ? lrows + r * w : src + (size_t)(4 * r) * stride;
            const int xb = pi * 8;
            int v[8];
            if ((LROWS || vec_ok) && xb + 8 <= w) {
                const int4 a = *reinterpret_cast<const int4 *>(row + xb), b = *reinterpret_cast<const int4 *>(row + xb + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int t = row[xb + i < w ? xb + i : w - 1];        // clamped index: unconditional load
                    v[i] = (xb + i < w) ? t : 0;
                }
            }
            uint32_t rho = 0, rho2 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (v[i] != 0) rho |= 1u << i;
                if (v[4 + i] != 0) rho2 |= 1u << i;
            }
            const uint32_t e1 = enc[(initial << 6) | rho];
            const uint32_t e2 = enc[(initial << 6) | ((rho >> 2) << 4) | rho2];
            vv = (uint64_t)(e1 >> 4);
            vl = e1 & 0xF;
            vv |= (uint64_t)(e2 >> 4) << vl;
            vl += e2 & 0xF;
            if (rho | rho2) {
                uint32_t u1 = 1, u2 = 1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (xb + i < w && uabs(v[i]) >= shl32(1, u1)) u1++;
                    if (xb + 4 + i < w && uabs(v[4 + i]) >= shl32(1, u2)) u2++;
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (!(q ? rho2 : rho)) continue;
                    const uint32_t u = q ? u2 : u1;
                    if (u <= 1) { vv |= (uint64_t)1 << vl; vl += 1; }
                    else if (u <= 2) { vv |= (uint64_t)2 << vl; vl += 2; }
                    else { vv |= (uint64_t)(u - 3) << (vl + 3); vl += 8; }   // (0,3) then (u-3,5)
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t rr = (i < 4) ? rho : rho2;
                if (!((rr >> (i & 3)) & 1)) continue;
                const uint32_t mag = uabs(v[i]);
                if (mag >= 0x80000000u) { bad = 1; continue; }
                const uint32_t emb = 32 - __clz(mag);
                mval[i] = (mag & (shl32(1, emb - 1) - 1)) | ((v[i] < 0 ? 1u : 0u) << (emb - 1));
                mlen[i] = emb;
                ml += emb;
            }
        }
        // exclusive prefix sums over the 64 items of this round
        const uint32_t vs = wave_incl_scan(vl), ms = wave_incl_scan(ml);
        const uint32_t vtot = wave_last(vs), mtot = wave_last(ms);
        uint32_t vpos = vbase + vs - vl, mpos = mbase + ms - ml;
        if (it < N) {
            or_bits(vbuf, vpos, vv);
            if (ml <= 64) {
                // the item's fields concatenated in registers, one deposit: an eighth of the LDS atomics, and blocks of small
                // magnitudes (1-bit fields, 32 to a word) no longer serialise on one word
                uint64_t acc = 0;
                uint32_t sh = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    acc |= shl64((uint64_t)mval[i], sh);       // mval < 2^mlen; an absent field is 0 bits of 0
                    sh += mlen[i];
                }
                if (acc) or_bits(mbuf, mpos, acc);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (mlen[i]) { or_bits(mbuf, mpos, (uint64_t)mval[i]); mpos += mlen[i]; }
            }
        }
        vbase += vtot; mbase += mtot;
    }
    bad = __any(bad);
    __syncthreads();
    if (bad) return false;
    TMout = mbase; TVout = vbase;
    return true;
}

// Where the 0xFF bytes of the MagSgn segment fall.  The stuffing rule (after a 0xFF byte the next byte takes 7 bits,
// ht.go:1303-1327) makes byte boundaries depend on every 0xFF before them, but a 0xFF byte is eight consecutive one bits
// of the unstuffed string starting at a byte boundary, and such places are rare: all lanes look for 8-runs of ones in
// their words, and only those candidates are visited in order (a wave-uniform loop, a dozen scalar instructions each):
// a candidate at p is a 0xFF byte iff p is a byte boundary of the current alignment; the byte after it starts at p + 8
// and is 7 bits wide (never 0xFF), the next 8-bit byte starts at p + 15.  flist[i] = start bit of the i-th 0xFF byte.
#define HT_MS_FF_CAP 2176                      /* > 31 * HT_FAST_MAX_SAMPLES / 15 */
__device__ __forceinline__ int ht_ms_find_ff(const uint32_t *mbuf, uint32_t TM, int lane, uint16_t *flist, uint32_t &myF) {
    int nF = 0;
    uint32_t base = 0;                         // byte boundaries: the positions >= base congruent to base mod 8
    myF = 0;                                   // lane i also keeps flist[i] (i < 64) in a register
    const uint32_t nwords = (TM + 31) >> 5;
    for (uint32_t w0 = 0; w0 < nwords; w0 += 64) {
        const uint32_t wi = w0 + lane;
        uint64_t x = 0;
        if (wi < nwords) x = (uint64_t)mbuf[wi] | ((uint64_t)mbuf[wi + 1] << 32);   // (the string is followed by zero words)
        uint64_t y = x & (x >> 1);
        y &= y >> 2;
        y &= y >> 4;                           // bit p: bits p .. p+7 of x are all ones
        const uint32_t m = (uint32_t)y;        // runs that start inside my word
        // One vector step per 0xFF byte: every lane masks its candidates down to those on the current byte grid at or behind
        // `base`; the lowest one in the wave is the next 0xFF byte and moves the grid.  (A scalar walk over the candidate
        // words cost ~200 cycles per word, branches mostly; blocks of small magnitudes have dozens of them and few real 0xFF.)
        const uint32_t p0 = 32u * wi;
        for (;;) {
            const uint32_t lo = base > p0 ? base - p0 : 0u;
            const uint32_t cand = lo >= 32 ? 0u : (m & (0xFFFFFFFFu << lo) & (0x01010101u << (base & 7)));   // p0 is a multiple of 8
            const unsigned long long who = __ballot(cand != 0);
            if (!who) break;
            const int L = __ffsll((long long)who) - 1;
            const uint32_t cL = (uint32_t)__builtin_amdgcn_readlane((int)cand, L);
            const uint32_t p = 32u * (w0 + (uint32_t)L) + (uint32_t)__ffs((int)cL) - 1;
            if (lane == 0) flist[nF] = (uint16_t)p;
            if (lane == nF) myF = p;
            nF++;
            base = p + 15;
        }
    }
    return nF;
}

// Part 2: the bytes.  WRITE = false only counts them (the fused encode + compact kernel needs the length of a block
// before it knows where the block goes).  flist: HT_MS_FF_CAP halfwords of LDS.
template <bool WRITE>
__device__ bool ht_emit(const BlockJob &J, uint8_t *__restrict__ out, int lane, const uint32_t *vbuf, const uint32_t *mbuf,
                        uint32_t mbase, uint32_t vbase, long &magLenOut, long &vlcLenOut, uint16_t *flist) {
    const size_t nsamp = (size_t)J.w * J.h;
    const size_t maxSize = nsamp * 2 < 64 ? 64 : nsamp * 2;
    const long msCap = (long)(maxSize / 2), vlcCap = (long)(maxSize / 2);
    const size_t melLen = maxSize / 4;
    // ---- MagSgn emission with 0xFF stuffing (ht.go:1303-1341), all bytes in parallel ----
    // With the 0xFF bytes known (ht_ms_find_ff; byte index of the i-th: kF(i) = (flist[i] + i) / 8), byte k starts at bit
    // 8k - #{i : kF(i) <= k - 2} and is 7 bits wide iff byte k - 1 is one of them.  The write loop of the reference emits a
    // byte while at least 8 bits are left (also for a 7-bit byte), the flush one more byte if any bit is left, unmasked.
    const uint32_t TM = mbase;
#ifdef J2K_ENC_STAMP
    const long long e0 = wall_clock64();
#endif
    uint32_t myF;
    const int nF = ht_ms_find_ff(mbuf, TM, lane, flist, myF);
    __syncthreads();
#ifdef J2K_ENC_STAMP
    const long long e05 = wall_clock64();
#endif
    auto kF = [&](int i) { return (long)(((uint32_t)flist[i] + (uint32_t)i) >> 3); };
    long K = 0;                                    // bytes out of the write loop
    uint32_t posK = 0;                             // bits consumed by them
    if (TM >= 8) {
        K = (long)((TM - 8 + (uint32_t)nF) >> 3) + 1;          // right when byte K - 1 lies behind the last 0xFF's 7-bit successor
        posK = (uint32_t)(8 * K) - (uint32_t)nF;
        if (nF) {
            const uint32_t Fl = flist[nF - 1];
            const long kl = kF(nF - 1);
            if (K - 1 < kl + 2) {
                if (Fl + 16 <= TM) { K = kl + 2; posK = Fl + 15; }   // the 7-bit byte is the last one
                else { K = kl + 1; posK = Fl + 8; }                   // the 0xFF byte is
            }
        }
    }
    const bool tail = TM > posK;                   // magSgnFlush: the remaining < 8 bits, no stuffing rule
    const long magLen = K + (tail ? 1 : 0);
    if (magLen > msCap) return false;
    if (WRITE) {
        // up to 64 0xFF bytes (practically always): their byte indices live in registers, counted with ballots and fetched
        // with v_readlane; more than that: the same from the LDS list
        const bool inreg = nF <= 64;
        const long myK = (lane < nF) ? (long)((myF + (uint32_t)lane) >> 3) : (long)0x7FFFFFFF;
        int fi = 0;                                // 0xFF bytes that shift every byte of the chunk: kF <= kb - 2
        for (long kb = 0; kb < K; kb += 256) {
            int fe;                                // .. that shift some of them: kF <= kb + 254
            if (inreg) {
                fi = __popcll(__ballot(myK <= kb - 2));
                fe = __popcll(__ballot(myK <= kb + 254));
            } else {
                while (fi < nF && kF(fi) <= kb - 2) fi++;
                fe = fi;
                while (fe < nF && kF(fe) <= kb + 254) fe++;
            }
            const long k0 = kb + 4 * lane;
            uint32_t dw = 0;
            if (fe == fi) {
                if (k0 < K) dw = ms_bits32(mbuf, (uint32_t)(8 * k0) - (uint32_t)fi);
            } else {
                uint32_t n[4] = {(uint32_t)fi, (uint32_t)fi, (uint32_t)fi, (uint32_t)fi}, narrow = 0;
                for (int i = fi; i < fe; i++) {
                    const long kk = inreg ? (long)__builtin_amdgcn_readlane((int)myK, i) : kF(i);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (kk <= k0 + j - 2) n[j]++;
                        if (kk == k0 + j - 1) narrow |= 1u << j;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (k0 + j >= K) break;
                    const uint32_t bj = get_bits8(mbuf, (uint32_t)(8 * (k0 + j)) - n[j]) & (((narrow >> j) & 1) ? 0x7Fu : 0xFFu);
                    dw |= bj << (8 * j);
                }
            }
            if (k0 >= K) continue;
            uint8_t *q = out + k0;
            if (k0 + 4 <= K) __builtin_memcpy(q, &dw, 4);               // unaligned 4-byte store
            else {
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (k0 + j < K) q[j] = (uint8_t)(dw >> (8 * j));
            }
        }
        if (tail && lane == 0) out[K] = (uint8_t)get_bits8(mbuf, posK);
    }
#ifdef J2K_ENC_STAMP
    const long long e1 = wall_clock64();
#endif
    // ---- VLC bytes: position-preserving stuffing (ht.go:1271-1300) ----
    const uint32_t TV = vbase;
    const long nfull = TV >> 3, vlcLen = (TV + 7) >> 3;
    if (vlcLen > vlcCap) return false;
    uint8_t *vout = out + magLen + melLen;
    for (long i0 = 4 * lane; WRITE && i0 < vlcLen; i0 += 256) {               // four bytes per lane
        uint32_t dw = ms_bits32(vbuf, (uint32_t)(8 * i0));
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const long i = i0 + jj;
            uint32_t b = (dw >> (8 * jj)) & 0xFF;
            if (i < nfull && (b & 0x7F) == 0x7F && i > 0) {
                // final value of the previous byte: walk back over the run of bytes whose low 7 bits are all ones
                long j = i - 1;
                while (j > 0 && (get_bits8(vbuf, (uint32_t)(8 * j)) & 0x7F) == 0x7F) j--;
                uint32_t prev = (j == 0 && (get_bits8(vbuf, 0) & 0x7F) == 0x7F) ? get_bits8(vbuf, 0)      // byte 0: lastByte starts at 0, never masked
                                                                                 : get_bits8(vbuf, (uint32_t)(8 * j));
                for (long k = j + 1; k < i; k++) {
                    uint32_t rb = get_bits8(vbuf, (uint32_t)(8 * k));
                    if (prev > 0x8F && (rb & 0x7F) == 0x7F) rb &= 0x7F;
                    prev = rb;
                }
                if (prev > 0x8F) dw &= ~(0x80u << (8 * jj));
            }
        }
        uint8_t *q = vout + i0;
        if (i0 + 4 <= vlcLen) __builtin_memcpy(q, &dw, 4);
        else {
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
                if (i0 + jj < vlcLen) q[jj] = (uint8_t)(dw >> (8 * jj));
        }
    }
    magLenOut = magLen; vlcLenOut = vlcLen;
#ifdef J2K_ENC_STAMP
    if (((blockIdx.x % 200) == 7) && lane == 0) printf("   wg %d find_ff %lld (nF %d) ms emit %lld vlc emit %lld\n", (int)blockIdx.x, e05 - e0, nF, e1 - e05, wall_clock64() - e1);
#endif
    return true;
}


template <bool LROWS>
__device__ __forceinline__ bool ht_encode_fast(const BlockJob &J, const int32_t *__restrict__ src, uint8_t *__restrict__ out, int lane,
                               uint32_t *vbuf, uint32_t *mbuf, long &magLenOut, long &vlcLenOut, const int32_t *lrows, const uint16_t *enc,
                               uint16_t *flist) {
    uint32_t TM, TV;
#ifdef J2K_ENC_STAMP
    const long long f0 = wall_clock64(), c0 = clock64();
#endif
    if (!ht_form<LROWS>(J, src, lane, vbuf, mbuf, TM, TV, lrows, enc)) return false;
#ifdef J2K_ENC_STAMP
    if (0) printf("   wg %d form %lld wall ticks, %lld shader clocks\n", (int)blockIdx.x, wall_clock64() - f0, clock64() - c0);
#endif
    return ht_emit<true>(J, out, lane, vbuf, mbuf, TM, TV, magLenOut, vlcLenOut, flist);
}

// utab / alias_ids (j2k_plan_encode_stream only): the block coder addresses every band's blocks from the TOP-LEFT of the plane
// (encoder.go:763-795), so the three bands of a resolution read the same windows, and this coder ignores the band
// (ht.go:942): jobs with the same window are byte-identical.  The plan lists one entry per distinct window (utab: the job
// itself + where its list of job ids -- its own first -- sits in alias_ids); the kernel codes each distinct window once and
// reports its length / bit-plane count / MagSgn length for every job of the list, one lane per job.
//
// A wavefront's life here is a chain of memory round trips, not work (1851 wavefronts per 4K frame on 1024 SIMDs), so the
// kernel is organised to make as few as possible: ONE for the table entry, then the alias ids, the code-word table and all
// sixteen 16-byte loads of a 64x64 block's max scan go out together; the coded rows (y % 4 == 0) are parked in LDS by the
// scan, so forming the code words does not go back to memory.  (Before: job id -> job -> two rounds of scan loads -> two
// rounds of row loads, each followed by table look-ups in global memory -> the alias chain, link by link: 33 us.)
__global__ __launch_bounds__(64) void ht_encode_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                       const int32_t *__restrict__ coef, uint8_t *__restrict__ slots,
                                                       uint32_t *__restrict__ lens, uint8_t *__restrict__ numbps,
                                                       int *__restrict__ fault, uint32_t *__restrict__ maglens,
                                                       const HtUJob *__restrict__ utab, const int *__restrict__ alias_ids) {
    // maglens != NULL (j2k_plan_encode_stream): also report where the MagSgn bytes end, and do NOT write the MEL
    // segment's zero bytes into the slot -- the gather puts zeros straight into the stream instead of copying them
    __shared__ uint32_t s_vbuf[HT_VLC_WORDS];
    __shared__ uint32_t s_mbuf[HT_MS_WORDS];
    __shared__ __align__(16) int32_t s_rows[HT_FAST_MAX_SAMPLES];
    __shared__ uint32_t s_enc[64];
    __shared__ uint16_t s_ff[HT_MS_FF_CAP];
    if ((int)blockIdx.x >= njobs) return;          // njobs = entries of utab when given
    const int lane = threadIdx.x;
#ifdef J2K_ENC_STAMP
    const long long st0 = wall_clock64();
#endif
    BlockJob J;
    int jid = (int)blockIdx.x, nalias = 1, aoff = 0;
    if (utab) {
        const HtUJob U = utab[blockIdx.x];
        J = U.J; jid = U.jid; nalias = U.nalias; aoff = U.alias_off;
    } else {
        J = jobs[jid];
    }
    int myalias = jid;
    if (utab && lane < nalias) myalias = alias_ids[aoff + lane];
    s_enc[lane] = reinterpret_cast<const uint32_t *>(g_vlc_enc)[lane];
    const int w = J.w, h = J.h, stride = J.stride;
    const int32_t *src = coef + J.src_off;
    uint8_t *out = slots + J.out_off;
    auto publish = [&](uint32_t len, uint32_t nb, bool has_mag, uint32_t mag) {
        for (int k = lane; k < nalias; k += 64) {
            const int j = k < 64 ? myalias : alias_ids[aoff + k];
            lens[j] = len;
            numbps[j] = (uint8_t)nb;
            if (maglens && has_mag) maglens[j] = mag;
        }
    };
    const bool fast = (size_t)((h + 3) / 4) * (size_t)w <= HT_FAST_MAX_SAMPLES;

    // ---- max |x| over the WHOLE block: nil decision (ht.go:947-960) and numbps ----
    int maxMag = 0;  // Go compares int32: -MinInt32 stays negative and never wins
    const bool quads = (w & 3) == 0 && (stride & 3) == 0 && (J.src_off & 3) == 0;
    const bool lrows = quads && fast;              // the scan parks the coded rows in s_rows
    if (quads) {
        // (row, quad) kept incrementally: a division by the block width per load was a third of this loop's instructions
        const int wq = w >> 2, nq = wq * h;
        const int dy = 64 / wq, dx = 64 - dy * wq;
        int y = lane / wq, xq = lane - y * wq;
        // sixteen loads in flight per step (clamped, unconditional addresses): a whole 64x64 block in one round trip
        for (int e0 = lane; e0 < nq; e0 += 64 * 16) {
            int4 qv[16];
            int ys[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const bool ok = e0 + 64 * u < nq;
                const int4 t = *reinterpret_cast<const int4 *>(ok ? src + (size_t)y * stride + 4 * xq : src);
                qv[u] = ok ? t : make_int4(0, 0, 0, 0);
                ys[u] = ok ? (y << 8 | xq) : -1;
                y += dy; xq += dx;
                if (xq >= wq) { xq -= wq; y++; }
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int4 q = qv[u];
                if (lrows && ys[u] >= 0 && ((ys[u] >> 8) & 3) == 0)
                    *reinterpret_cast<int4 *>(&s_rows[(ys[u] >> 10) * w + 4 * (ys[u] & 0xFF)]) = q;
                const int a0 = q.x < 0 ? (int)(0u - (uint32_t)q.x) : q.x, a1 = q.y < 0 ? (int)(0u - (uint32_t)q.y) : q.y;
                const int a2 = q.z < 0 ? (int)(0u - (uint32_t)q.z) : q.z, a3 = q.w < 0 ? (int)(0u - (uint32_t)q.w) : q.w;
                maxMag = max(max(maxMag, max(a0, a1)), max(a2, a3));
            }
        }
    } else {
        for (int y = 0; y < h; y++)
            for (int x = lane; x < w; x += 64) {
                int v = src[(size_t)y * stride + x];
                if (v < 0) v = (int)(0u - (uint32_t)v);
                maxMag = max(maxMag, v);
            }
    }
    for (int o = 32; o > 0; o >>= 1) maxMag = max(maxMag, __shfl_xor(maxMag, o));
#ifdef J2K_ENC_STAMP
    const long long st1 = wall_clock64();
#endif
    if (maxMag == 0) {
        publish(0, 0, false, 0);
        return;
    }
    if (fast) {
        long mLen = 0, vLen = 0;
        const size_t nsamp_ = (size_t)w * h;
        const size_t maxSize_ = nsamp_ * 2 < 64 ? 64 : nsamp_ * 2;
        const size_t melLen_ = maxSize_ / 4;
        const uint16_t *enc = reinterpret_cast<const uint16_t *>(s_enc);
        const bool ok = lrows ? ht_encode_fast<true>(J, src, out, lane, s_vbuf, s_mbuf, mLen, vLen, s_rows, enc, s_ff)
                              : ht_encode_fast<false>(J, src, out, lane, s_vbuf, s_mbuf, mLen, vLen, nullptr, enc, s_ff);
        if (!ok) {
            if (lane == 0) atomicMax(fault, 1);
            publish(0, 0, false, 0);
            return;
        }
        if (!maglens) zero_bytes(out + mLen, melLen_, lane);
        const size_t scup = melLen_ + (size_t)vLen + 2;
        const size_t total = (size_t)mLen + scup;
        if (lane == 0) {
            out[total - 2] = (uint8_t)(scup >> 8);
            out[total - 1] = (uint8_t)(scup & 0xFF);
        }
        publish((uint32_t)total, 32 - __clz((uint32_t)maxMag), true, (uint32_t)mLen);
#ifdef J2K_ENC_STAMP
        const long long st2 = wall_clock64();
        if (((blockIdx.x % 200) == 7 || blockIdx.x >= njobs - 3) && lane == 0)
            printf("enc wg %d: start %lld end %lld scan %lld  form+emit %lld (x10 ns)  mLen %ld vLen %ld\n", (int)blockIdx.x, st0 % 100000, st2 % 100000, st1 - st0, st2 - st1, mLen, vLen);
#endif
        return;
    }
    // ---- generic path for large blocks: bit-serial packing on lane 0 ----
    const size_t nsamp = (size_t)w * h;
    const size_t maxSize = nsamp * 2 < 64 ? 64 : nsamp * 2;   // ht.go:969-972
    const size_t msCap = maxSize / 2, melLen = maxSize / 4, vlcCap = maxSize / 2;
    uint8_t *vlcScratch = out + msCap + melLen;               // VLC bytes in write order, moved down at the end

    long magLen = 0, vlcLen = 0;
    int bad = 0;
    if (lane == 0) {
        BitWriter vlc{vlcScratch, 0, (long)vlcCap, 0, 0, 0, 0};
        BitWriter ms{out, 0, (long)msCap, 0, 0, 0, 0};
        const int quadCols = (w + 3) / 4;
        for (int y = 0; y < h && !vlc.fault && !ms.fault; y += 4) {   // only row y of each stripe (ht.go:1054)
            const int initial = (y == 0);
            const int32_t *row = src + (size_t)y * stride;
            for (int qx = 0; qx < quadCols && !vlc.fault && !ms.fault; qx += 2) {
                int v[8];
                uint32_t rho = 0, rho2 = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int x = qx * 4 + i;
                    v[i] = (x < w) ? row[x] : 0;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (qx * 4 + i < w && v[i] != 0) rho |= 1u << i;
                    if ((qx + 1) * 4 + i < w && v[4 + i] != 0) rho2 |= 1u << i;
                }
                // first quad: context is always 0 (sigma1 holds 4-bit rho, >>4 == 0: ht.go:1079-1085)
                uint16_t e1 = g_vlc_enc[(initial << 6) | rho];
                vlc_write(vlc, e1 >> 4, e1 & 0xF);
                // second quad: context = rho >> 2 (ht.go:1094)
                uint16_t e2 = g_vlc_enc[(initial << 6) | ((rho >> 2) << 4) | rho2];
                vlc_write(vlc, e2 >> 4, e2 & 0xF);
                if (rho | rho2) {                                   // ht.go:1105-1142
                    uint32_t u1 = 1, u2 = 1;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (qx * 4 + i < w && uabs(v[i]) >= shl32(1, u1)) u1++;
                        if ((qx + 1) * 4 + i < w && uabs(v[4 + i]) >= shl32(1, u2)) u2++;
                    }
                    if (rho && rho2) { uvlc_one(vlc, u1); uvlc_one(vlc, u2); }
                    else uvlc_one(vlc, rho ? u1 : u2);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {                       // ht.go:1145-1193
                    const uint32_t r = (i < 4) ? rho : rho2;
                    if (!((r >> (i & 3)) & 1)) continue;
                    const uint32_t mag = uabs(v[i]);
                    if (mag >= 0x80000000u) { ms.fault = 1; break; }  // Go: emb loop never terminates
                    const uint32_t emb = 32 - __clz(mag);           // bit length
                    ms_write(ms, mag & (shl32(1, emb - 1) - 1), emb - 1);
                    ms_write(ms, v[i] < 0 ? 1u : 0u, 1);
                }
            }
        }
        if (!vlc.fault) flush_bits(vlc);
        if (!ms.fault) flush_bits(ms);
        magLen = ms.pos; vlcLen = vlc.pos;
        bad = vlc.fault | ms.fault;
    }
    magLen = __shfl(magLen, 0); vlcLen = __shfl(vlcLen, 0); bad = __shfl(bad, 0);
    if (bad) {
        if (lane == 0) atomicMax(fault, 1);
        publish(0, 0, false, 0);
        return;
    }
    // ---- assemble: MagSgn | MEL zeros | VLC | SCUP (ht.go:1017-1042) ----
    zero_bytes(out + magLen, melLen, lane);
    __syncthreads();  // single-wave block: orders the zero fill before the move below
    const size_t D = (size_t)magLen + melLen;                  // <= msCap + melLen: the move goes downward
    for (long base = 0; base < vlcLen; base += 64) {
        const long i = base + lane;
        uint8_t b = 0;
        if (i < vlcLen) b = vlcScratch[i];
        __syncthreads();
        if (i < vlcLen) out[D + i] = b;
        __syncthreads();
    }
    {
        const size_t scup = melLen + (size_t)vlcLen + 2;
        const size_t total = (size_t)magLen + scup;
        if (lane == 0) {
            out[total - 2] = (uint8_t)(scup >> 8);
            out[total - 1] = (uint8_t)(scup & 0xFF);
        }
        publish((uint32_t)total, 32 - __clz((uint32_t)maxMag), true, (uint32_t)magLen);
    }
}

// ================================================================================================
// Encode + compact in one kernel (blocks on the parallel path only): the dense stream the reference builds with
// `tileData = append(tileData, encoded...)` (encoder.go:684) without the slot -> stream copy.
// A block's place is the sum of the lengths before it in job order.  Each workgroup learns its length BEFORE writing a
// byte (ht_emit<false> counts), publishes it, and looks back over its predecessors' status words ("decoupled look-back":
// a word holds either a block's own length or its inclusive prefix).  Only status words travel between workgroups --
// relaxed agent-scope atomics, no fences -- and workgroups are dispatched in index order, so every predecessor a
// workgroup waits for is already running: no deadlock; the wait is bounded anyway and reports a fault if it ever hits
// the bound.  A launch is tagged with an epoch so the status array needs no clearing between launches.
#define HT_ST_AGG 1ull
#define HT_ST_PREFIX 2ull
__device__ __forceinline__ unsigned long long ht_status(unsigned long long flag, uint32_t epoch, unsigned long long value) {
    return flag << 62 | (unsigned long long)(epoch & 0x3FFFFF) << 40 | (value & 0xFFFFFFFFFFull);
}

__global__ __launch_bounds__(64) void ht_encode_stream_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                              const int32_t *__restrict__ coef, uint8_t *__restrict__ stream,
                                                              unsigned long long *__restrict__ offs, uint32_t *__restrict__ lens,
                                                              uint8_t *__restrict__ numbps, unsigned long long *__restrict__ status,
                                                              uint32_t epoch, int *__restrict__ fault) {
    __shared__ uint32_t s_vbuf[HT_VLC_WORDS];
    __shared__ uint32_t s_mbuf[HT_MS_WORDS];
    __shared__ uint16_t s_ff[HT_MS_FF_CAP];
    const int jid = blockIdx.x;
    if (jid >= njobs) return;
    const int lane = threadIdx.x;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h, stride = J.stride;
    const int32_t *src = coef + J.src_off;
    int maxMag = 0;  // Go compares int32: -MinInt32 stays negative and never wins (ht.go:947-960)
    if ((w & 3) == 0 && (stride & 3) == 0 && (J.src_off & 3) == 0) {
        // (row, quad) kept incrementally: a division by the block width per load was a third of this loop's instructions
        const int wq = w >> 2, nq = wq * h;
        const int dy = 64 / wq, dx = 64 - dy * wq;
        int y = lane / wq, xq = lane - y * wq;
        for (int e = lane; e < nq; e += 64) {
            const int4 q = *reinterpret_cast<const int4 *>(src + (size_t)y * stride + 4 * xq);
            const int a0 = q.x < 0 ? (int)(0u - (uint32_t)q.x) : q.x, a1 = q.y < 0 ? (int)(0u - (uint32_t)q.y) : q.y;
            const int a2 = q.z < 0 ? (int)(0u - (uint32_t)q.z) : q.z, a3 = q.w < 0 ? (int)(0u - (uint32_t)q.w) : q.w;
            maxMag = max(max(maxMag, max(a0, a1)), max(a2, a3));
            y += dy; xq += dx;
            if (xq >= wq) { xq -= wq; y++; }
        }
    } else {
        for (int y = 0; y < h; y++)
            for (int x = lane; x < w; x += 64) {
                int v = src[(size_t)y * stride + x];
                if (v < 0) v = (int)(0u - (uint32_t)v);
                maxMag = max(maxMag, v);
            }
    }
    for (int o = 32; o > 0; o >>= 1) maxMag = max(maxMag, __shfl_xor(maxMag, o));
    const size_t nsamp = (size_t)w * h;
    const size_t maxSize = nsamp * 2 < 64 ? 64 : nsamp * 2;
    const size_t melLen = maxSize / 4;
    uint32_t TM = 0, TV = 0;
    long mLen = 0, vLen = 0;
    unsigned long long total = 0;
    if (maxMag != 0) {
        bool ok = ht_form<false>(J, src, lane, s_vbuf, s_mbuf, TM, TV, nullptr, g_vlc_enc);
        if (ok) ok = ht_emit<false>(J, nullptr, lane, s_vbuf, s_mbuf, TM, TV, mLen, vLen, s_ff);
        if (ok) total = (unsigned long long)mLen + melLen + (unsigned long long)vLen + 2;
        else if (lane == 0) atomicMax(fault, 1);                      // the reference panics on this input
    }
    // ---- decoupled look-back over the predecessors ----
    if (lane == 0) __hip_atomic_store(&status[jid], ht_status(HT_ST_AGG, epoch, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long excl = 0;
    int base = jid - 1;
    uint32_t spins = 0;
    while (base >= 0) {
        const int idx = base - lane;
        const unsigned long long v = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : ht_status(HT_ST_PREFIX, epoch, 0);
        const unsigned long long flag = v >> 62;
        const bool valid = flag != 0 && (uint32_t)((v >> 40) & 0x3FFFFF) == (epoch & 0x3FFFFF);
        const unsigned long long inv = __ballot(!valid), pre = __ballot(valid && flag == HT_ST_PREFIX);
        const int first_inv = inv ? __ffsll((long long)inv) - 1 : 64, first_pre = pre ? __ffsll((long long)pre) - 1 : 64;
        if (first_inv < first_pre) {                                  // a predecessor this side of the nearest prefix has not published yet
            if (++spins > (1u << 22)) { if (lane == 0) atomicMax(fault, 3); break; }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        const int upto = min(first_pre, 63);
        unsigned long long part = lane <= upto ? (v & 0xFFFFFFFFFFull) : 0ull;
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        excl += part;
        if (first_pre < 64) break;
        base -= 64;
    }
    if (lane == 0) {
        __hip_atomic_store(&status[jid], ht_status(HT_ST_PREFIX, epoch, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        offs[jid] = excl;
        if (jid == njobs - 1) offs[njobs] = excl + total;
        lens[jid] = (uint32_t)total;
        numbps[jid] = total ? (uint8_t)(32 - __clz((uint32_t)maxMag)) : 0;
    }
    if (!total) return;
    // ---- the bytes, straight into their final place ----
    uint8_t *out = stream + excl;
    ht_emit<true>(J, out, lane, s_vbuf, s_mbuf, TM, TV, mLen, vLen, s_ff);
    zero_bytes(out + mLen, melLen, lane);
    if (lane == 0) {
        const size_t scup = melLen + (size_t)vLen + 2;
        out[total - 2] = (uint8_t)(scup >> 8);
        out[total - 1] = (uint8_t)(scup & 0xFF);
    }
}

// ---------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------
struct RevStream { const uint8_t *data; long len, pos, size; uint64_t tmp; uint32_t bits; int unstuff; };
struct FwdStream { const uint8_t *data; long len, pos, size; uint64_t tmp; uint32_t bits; int unstuff; uint32_t x; };

__device__ void rev_read(RevStream &v) {                    // ht.go:317-378
    if (v.bits > 32) return;
    uint32_t val = 0;
    if (v.size > 3) {
        const long p = v.pos - 3;
        if (p >= 0 && p + 3 < v.len)
            val = (uint32_t)v.data[p] | (uint32_t)v.data[p + 1] << 8 | (uint32_t)v.data[p + 2] << 16 | (uint32_t)v.data[p + 3] << 24;
        v.pos -= 4; v.size -= 4;
    } else if (v.size > 0) {
        int i = 24;
        while (v.size > 0) {
            if (v.pos >= 0 && v.pos < v.len) { val |= (uint32_t)v.data[v.pos] << i; v.pos--; }
            v.size--; i -= 8;
        }
    }
    uint32_t tmp = val >> 24, bits = 8;
    if (v.unstuff && ((val >> 24) & 0x7F) == 0x7F) bits = 7;
    int unstuff = (val >> 24) > 0x8F;
    tmp |= ((val >> 16) & 0xFF) << bits;
    bits += (unstuff && ((val >> 16) & 0x7F) == 0x7F) ? 7 : 8;
    unstuff = ((val >> 16) & 0xFF) > 0x8F;
    tmp |= ((val >> 8) & 0xFF) << bits;
    bits += (unstuff && ((val >> 8) & 0x7F) == 0x7F) ? 7 : 8;
    unstuff = ((val >> 8) & 0xFF) > 0x8F;
    tmp |= (val & 0xFF) << bits;
    bits += (unstuff && (val & 0x7F) == 0x7F) ? 7 : 8;
    v.unstuff = (val & 0xFF) > 0x8F;
    v.tmp |= shl64((uint64_t)tmp, v.bits);
    v.bits += bits;
}
__device__ __forceinline__ uint32_t rev_fetch(RevStream &v) {  // ht.go:381-389
    if (v.bits < 32) { rev_read(v); if (v.bits < 32) rev_read(v); }
    return (uint32_t)v.tmp;
}
__device__ __forceinline__ void rev_advance(RevStream &v, uint32_t n) { v.tmp = shr64(v.tmp, n); v.bits -= n; }

__device__ void fwd_read(FwdStream &f) {                    // ht.go:432-501
    if (f.bits > 32) return;
    uint32_t val = 0;
    if (f.size > 3) {
        if (f.pos + 3 < f.len)
            val = (uint32_t)f.data[f.pos] | (uint32_t)f.data[f.pos + 1] << 8 | (uint32_t)f.data[f.pos + 2] << 16 | (uint32_t)f.data[f.pos + 3] << 24;
        f.pos += 4; f.size -= 4;
    } else if (f.size > 0) {
        if (f.x != 0) val = 0xFFFFFFFFu;
        int i = 0;
        while (f.size > 0) {
            if (f.pos < f.len) {
                const uint32_t b = f.data[f.pos];
                val = (val & ~((uint32_t)0xFF << i)) | (b << i);
                f.pos++;
            }
            f.size--; i += 8;
        }
    } else if (f.x != 0) {
        val = 0xFFFFFFFFu;
    }
    uint32_t bits = f.unstuff ? 7 : 8;
    uint32_t t = val & 0xFF;
    int unstuff = (val & 0xFF) == 0xFF;
    t |= ((val >> 8) & 0xFF) << bits;
    bits += unstuff ? 7 : 8;
    unstuff = ((val >> 8) & 0xFF) == 0xFF;
    t |= ((val >> 16) & 0xFF) << bits;
    bits += unstuff ? 7 : 8;
    unstuff = ((val >> 16) & 0xFF) == 0xFF;
    t |= ((val >> 24) & 0xFF) << bits;
    bits += unstuff ? 7 : 8;
    f.unstuff = ((val >> 24) & 0xFF) == 0xFF;
    f.tmp |= shl64((uint64_t)t, f.bits);
    f.bits += bits;
}
__device__ __forceinline__ uint32_t fwd_fetch(FwdStream &f) {  // ht.go:504-512
    if (f.bits < 32) { fwd_read(f); if (f.bits < 32) fwd_read(f); }
    return (uint32_t)f.tmp;
}
__device__ __forceinline__ void fwd_advance(FwdStream &f, uint32_t n) { f.tmp = shr64(f.tmp, n); f.bits -= n; }

__device__ __forceinline__ uint32_t uvlc_entry(uint32_t idx) {  // ht.go:718-727: prefix len | suffix len<<2 | base<<5
    // {3|5<<2|5<<5, 1|1<<5, 2|2<<5, 1|1<<5, 3|1<<2|3<<5, 1|1<<5, 2|2<<5, 1|1<<5} = {183,33,66,33,103,33,66,33}, one byte each
    return (uint32_t)(0x21422167214221B7ull >> (8 * (idx & 7))) & 0xFF;
}

__device__ uint32_t decode_uvlc(uint32_t vlc, uint32_t mode, uint32_t (&u)[2], int initial) {  // ht.go:716-864
    uint32_t consumed = 0;
    u[0] = 1; u[1] = 1;
    if (mode == 0) return 0;
    if (mode <= 2) {
        const uint32_t t = uvlc_entry(vlc);
        const uint32_t pl = t & 3; vlc >>= pl; consumed += pl;
        const uint32_t sl = (t >> 2) & 7; consumed += sl;
        const uint32_t val = (t >> 5) + (vlc & (shl32(1, sl) - 1));
        if (mode == 1) u[0] = val + 1; else u[1] = val + 1;
    } else if (mode == 3) {
        const uint32_t t1 = uvlc_entry(vlc);
        const uint32_t pl1 = t1 & 3; vlc >>= pl1; consumed += pl1;
        if (initial && pl1 > 2) {                               // ht.go:756-764
            u[1] = (vlc & 1) + 2; consumed++; vlc >>= 1;
            const uint32_t sl = (t1 >> 2) & 7; consumed += sl;
            u[0] = (t1 >> 5) + (vlc & (shl32(1, sl) - 1)) + 1;
        } else {
            const uint32_t t2 = uvlc_entry(vlc);
            const uint32_t pl2 = t2 & 3; vlc >>= pl2; consumed += pl2;
            const uint32_t sl1 = (t1 >> 2) & 7; consumed += sl1;
            u[0] = (t1 >> 5) + (vlc & (shl32(1, sl1) - 1)) + 1;
            vlc >>= sl1;
            const uint32_t sl2 = (t2 >> 2) & 7; consumed += sl2;
            u[1] = (t2 >> 5) + (vlc & (shl32(1, sl2) - 1)) + 1;
        }
    }
    return consumed;
}

// the same for the rows after the first, branch-free: a disabled u is the all-zero table entry.  Note the reference's
// pairing (ht.go:736-751): mode = uOff1 << 1 | uOff2, mode 1 decodes u[0] and mode 2 decodes u[1] -- so u[0] is
// present iff uOff2 and u[1] iff uOff1.
__device__ __forceinline__ uint32_t decode_uvlc_later(uint32_t vlc, uint32_t uOff1, uint32_t uOff2, uint32_t &u0, uint32_t &u1) {
    const uint32_t t1 = uOff2 ? uvlc_entry(vlc) : 0u;
    const uint32_t pl1 = t1 & 3; vlc >>= pl1;
    const uint32_t t2 = uOff1 ? uvlc_entry(vlc) : 0u;
    const uint32_t pl2 = t2 & 3; vlc >>= pl2;
    const uint32_t sl1 = (t1 >> 2) & 7, sl2 = (t2 >> 2) & 7;
    u0 = (t1 >> 5) + (vlc & ((1u << sl1) - 1)) + 1;
    vlc >>= sl1;
    u1 = (t2 >> 5) + (vlc & ((1u << sl2) - 1)) + 1;
    return pl1 + pl2 + sl1 + sl2;
}

__device__ bool init_mel_ok(const uint8_t *data, long len, long lcup, long scup) {  // ht.go:153-195
    long pos = lcup - scup, size = scup - 1;
    int unstuff = 0;
    long num = 4 - (pos & 3);
    if (num > 4) num = 4;
    for (long i = 0; i < num && size > 0; i++) {
        if (unstuff && pos < len && data[pos] > 0x8F) return false;
        uint32_t b;
        if (size > 0 && pos < len) { b = data[pos]; pos++; size--; } else b = 0xFF;
        if (size == 1) b |= 0x0F;
        unstuff = (b == 0xFF);
    }
    return true;
}

// ---- parallel decoder (fast path), three kernels ------------------------------------------------
// The VLC stream is inherently sequential (each code word's length comes out of the table lookup of the
// previous one).  Run as "one block per wavefront" that walk saturates the CU's single scalar unit
// (measured: 36 us for one block, 290 us for 7005).  So:
//   ht_vlcprep_kernel: one block per WAVEFRONT.  Validates SCUP / the MEL start and undoes the reverse reader's
//                      byte stuffing in parallel: the VLC bytes become a linear bit string in a global scratch.
//   ht_walk_kernel   : one block per LANE.  64 bit strings staged in LDS, 64 blocks walked at once; one record
//                      (rho, rho2, u-VLC mode, the 16 stream bits at the u-VLC) per quad pair.
//   ht_decode_kernel : one block per WAVEFRONT.  u values from the records, parallel unstuffing of the MagSgn bytes into an
//                      LDS bit string (a byte is 7 bits wide iff its predecessor is 0xFF; prefix sum of
//                      widths; OR-deposit because the reference ORs a full byte at a 7-bit advance,
//                      ht.go:467-500; past the segment everything reads as ones), then every lane extracts
//                      the samples of its pairs: a pair's bit position is the prefix sum of
//                      popcount(rho)*(u+1) over the pairs before it (ht.go:661-710).
// Blocks with more than HT_FAST_MAX_SAMPLES coded samples / HT_WALK_MAX_PAIRS pairs, or where a decoded u
// exceeds 32 (the reference's uint32 bit counter then wraps, ht.go:515-519), take the bit-serial path.
#define HT_WALK_MAX_PAIRS 128
#define HT_DEC_MWORDS (33 * HT_FAST_MAX_SAMPLES / 32 + 8)
#define HT_WALK_REC 132                        /* words per block: 128 pair records + flags (+pad) */
#define HT_PAIR_SERIAL 0xFFFFFFFFu              /* record[0]: decode this block with the serial path */
#define HT_PAIR_ZERO 0xFFFFFFFEu                /* record[0]: invalid stream -> output stays zero     */

// Single-wavefront workgroups: DS instructions of one wave execute in order, so LDS written by one lane is
// visible to the others without an s_barrier; only the compiler must not reorder.  Unlike __syncthreads()
// this does NOT wait for outstanding global stores (vmcnt), which lets the zero fill drain in the background.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t get_bits32(const uint32_t *buf, uint32_t bitpos) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint64_t two = (uint64_t)buf[wd] | ((uint64_t)buf[wd + 1] << 32);
    return (uint32_t)(two >> sh);
}

// reads for blocks with u > 32: positions clamped into the staged string (its tail is all ones) and cut to zero at L
__device__ __forceinline__ uint32_t get_bits32_cut(const uint32_t *buf, uint32_t bitpos, uint32_t L) {
    if (bitpos >= L) return 0;
    const uint32_t p = bitpos < (HT_DEC_MWORDS - 2) * 32u ? bitpos : (HT_DEC_MWORDS - 2) * 32u + (bitpos & 31);
    uint32_t v = get_bits32(buf, p);
    if (L - bitpos < 32) v &= (1u << (L - bitpos)) - 1;
    return v;
}

#define HT_VBITS_WORDS 192                     /* unstuffed VLC bit string per block: 46 bits x 128 pairs + slack */
#define HT_VROW 193                            /* LDS row stride in the walk kernel (odd: no bank conflicts)        */

// ---- kernel 1: one block per wavefront -- validate the stream, unstuff the VLC bytes into a linear bit string ----
// initVLC + revRead (ht.go:276-378): the high nibble of byte lcup-2 (3 or 4 bits), then bytes lcup-3, lcup-4, ...
// each 8 bits wide, or 7 when the previous byte is > 0x8F and this one's low 7 bits are all ones (the byte itself
// is OR-ed in whole at the shorter advance); after scup-2 bytes, zeros.  Byte widths depend only on the neighbour
// byte, so all bytes are placed in parallel: prefix sum of widths, OR-deposit into LDS, coalesced copy to global.
__global__ __launch_bounds__(256) void ht_vlcprep_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                         const uint8_t *__restrict__ stream, const uint64_t *__restrict__ offs,
                                                         const uint32_t *__restrict__ lens, uint32_t *__restrict__ vbits,
                                                         uint32_t *__restrict__ pairs) {
    __shared__ uint32_t vb4[4][HT_VBITS_WORDS + 8];
    uint32_t *vb = vb4[threadIdx.x >> 6];
    const int jid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (jid >= njobs) return;
    const int lane = threadIdx.x & 63;
    const int w = jobs[jid].w, h = jobs[jid].h;
    const long len = (long)lens[jid];
    const uint8_t *data = stream + offs[jid];
    uint32_t *rec = pairs + (size_t)jid * HT_WALK_REC;
    const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, R = (h + 3) / 4, N = R * P;
    const bool fast = (size_t)R * (size_t)w <= HT_FAST_MAX_SAMPLES && N <= HT_WALK_MAX_PAIRS;
    uint32_t tag = 0;
    long scup = 0;
    if (!fast) tag = HT_PAIR_SERIAL;
    else if (len < 2) tag = HT_PAIR_ZERO;                            // ht.go:94-100
    else {
        scup = (long)data[len - 1] + ((long)(data[len - 2] & 0x0F) << 8);
        if (scup < 2 || scup > len) tag = HT_PAIR_ZERO;              // ht.go:104-111
        else if (!init_mel_ok(data, len, len, scup)) tag = HT_PAIR_ZERO;   // ht.go:117-122
    }
    if (tag) {                                                       // wave-uniform
        if (lane == 0) { rec[0] = tag; rec[HT_WALK_MAX_PAIRS] = 0; }
        return;
    }
    for (int i = lane; i < HT_VBITS_WORDS + 8; i += 64) vb[i] = 0;
    wave_sync();
    const long lcup = len;
    const uint32_t b0 = data[lcup - 2];
    const uint32_t t0 = b0 >> 4;
    uint32_t off = 4 - ((t0 & 7) >> 2);
    if (lane == 0) atomicOr(&vb[0], t0);
    const long size = scup - 2;
    const long maxbytes = (long)(46 * N + 7) / 8 + 8;
    const long nb = size < maxbytes ? size : maxbytes;
    // Four stream bytes per lane per step: reverse-reader bytes k = 256c + 4l + 1 .. + 4 are the four bytes ENDING at
    // data[lcup - 3 - 256c - 4l]; one (unaligned) 4-byte load, byte-swapped so that byte j is the j-th in reading order.
    // A byte's width (7 or 8) needs only its predecessor; one prefix sum of the per-lane totals, one OR-deposit of the
    // four bytes merged at their offsets (a 7-bit advance lets the next byte overlap the top bit: OR, as revRead does).
    uint32_t carry = b0 | 0x0F;                                      // "previous byte" of k = 1
    for (long k0 = 1; k0 <= nb; k0 += 256) {
        const long k = k0 + 4 * lane;                                // first of this lane's four bytes
        const long e = lcup - 2 - k;                                 // its address; the other three sit below it
        uint32_t dw = 0;
        if (k + 3 <= nb) {
            __builtin_memcpy(&dw, data + e - 3, 4);
            dw = __builtin_bswap32(dw);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (k + j <= nb) dw |= (uint32_t)data[e - j] << (8 * j);
        }
        uint32_t prev = __shfl_up(dw >> 24, 1);
        if (lane == 0) prev = carry;
        carry = __shfl(dw >> 24, 63);                                // only read again when all 256 bytes of this step exist
        uint32_t wsum = 0;
        uint64_t merged = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t bj = (dw >> (8 * j)) & 0xFF;
            const uint32_t wj = (k + j <= nb) ? ((prev > 0x8F && (bj & 0x7F) == 0x7F) ? 7u : 8u) : 0u;
            merged |= (uint64_t)bj << wsum;
            wsum += wj;
            prev = bj;
        }
        const uint32_t ws = wave_incl_scan(wsum);
        if (merged) or_bits(vb, off + ws - wsum, merged);
        off += wave_last(ws);
    }
    wave_sync();
    uint32_t *dst = vbits + (size_t)jid * HT_VBITS_WORDS;
    for (int i = lane; i < HT_VBITS_WORDS; i += 64) dst[i] = vb[i];
    if (lane == 0) { rec[0] = 0; rec[HT_WALK_MAX_PAIRS] = (uint32_t)scup; }   // not a tag; SCUP travels in the flag word
}

#ifndef HT_WALK_BLOCKS
#define HT_WALK_BLOCKS 64                      /* blocks (= walking lanes) per workgroup: 64 or 32 */
#endif
struct HtWalkShared {
    uint32_t vb[HT_WALK_BLOCKS][HT_VROW];
    uint16_t pair1[16384];
    uint16_t tbl0[512], tbl1[512];
};

// bits the u-VLC of a quad pair consumes, rows after the first (see decode_uvlc_later)
__device__ __forceinline__ uint32_t uvlc_used_later(uint32_t vlc, uint32_t uOff1, uint32_t uOff2) {
    const uint32_t t1 = uOff2 ? uvlc_entry(vlc) : 0u;
    const uint32_t pl1 = t1 & 3;
    const uint32_t t2 = uOff1 ? uvlc_entry(vlc >> pl1) : 0u;
    return pl1 + (t2 & 3) + ((t1 >> 2) & 7) + ((t2 >> 2) & 7);
}

// ---- kernel 2: one block per LANE -- the sequential VLC walk (ht.go:589-658) on the linear bit strings ----
// The walk is a dependent chain (a code word's length comes out of the lookup of the previous one) and a block is at
// most 128 quad pairs long, so the kernel time is 128 x the latency of one pair: everything that is not needed to find
// the NEXT pair's bit position is left to ht_decode_kernel.  Per pair: fetch 64 bits at the current position
// (three LDS words + two v_alignbit), ONE LDS lookup that resolves both quads (g_vlc_pair1), the u-VLC's LENGTH, one
// record store:  rho | rho2 << 4 | (uOff1 << 1 | uOff2) << 13 | (the 16 stream bits at the u-VLC) << 16  (bits 8..12: scratch).
// The first row (other table, other u-VLC rule) and code words with a length nibble above 7 use two lookups.
// A pair consumes at most 46 bits and HT_VBITS_WORDS covers 128 of them plus the read-ahead, so no clamping.
__global__ __launch_bounds__(256) void ht_walk_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                     const uint32_t *__restrict__ vbits, uint32_t *__restrict__ pairs) {
    extern __shared__ __align__(16) unsigned char walk_smem[];
    HtWalkShared &S = *reinterpret_cast<HtWalkShared *>(walk_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int jid0 = blockIdx.x * HT_WALK_BLOCKS;
    const int jid = jid0 + lane;
    const bool have = lane < HT_WALK_BLOCKS && jid < njobs;
#ifdef J2K_WALK_STAMP
    const long long st0 = __builtin_amdgcn_s_memtime();
#endif
    // Staging by all four wavefronts, every load issued before the first use (a lone wavefront doing this in
    // dependent batches spent 40% of the kernel here): 32 KiB pair table = 8 x 16 B per thread, 64 blocks x 192 words
    // = 12 x 16 B per thread, the two 1 KiB tables.  Afterwards only wavefront 0 walks.
    const int nblk = min(HT_WALK_BLOCKS, njobs - jid0);
    // the walking wavefront's own inputs ride along with the staging loads
    uint32_t *rec = pairs + (size_t)jid * HT_WALK_REC;
    uint32_t tag = HT_PAIR_ZERO;
    int w = 0, h = 0;
    if (tid < 64 && have) { tag = rec[0]; w = jobs[jid].w; h = jobs[jid].h; }
    {
        constexpr int NV = HT_WALK_BLOCKS * 48 / 256;          // 16-byte pieces of bit string per thread
        uint4 t[8], v[NV];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = reinterpret_cast<const uint4 *>(g_vlc_pair1)[j * 256 + tid];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int q = j * 256 + tid;                      // 16-byte piece q: block q / 48, words 4 * (q % 48) ..
            const int bsel = min(q / 48, nblk - 1);
            v[j] = reinterpret_cast<const uint4 *>(vbits + (size_t)(jid0 + bsel) * HT_VBITS_WORDS)[q % 48];
        }
        if (tid < 64) reinterpret_cast<uint4 *>(S.tbl0)[tid] = reinterpret_cast<const uint4 *>(c_vlc_tbl0)[tid];   // contexts 0..3 = 1 KiB
        else if (tid < 128) reinterpret_cast<uint4 *>(S.tbl1)[tid - 64] = reinterpret_cast<const uint4 *>(c_vlc_tbl1)[tid - 64];
#pragma unroll
        for (int j = 0; j < 8; j++) reinterpret_cast<uint4 *>(S.pair1)[j * 256 + tid] = t[j];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int q = j * 256 + tid;
            uint32_t *d = &S.vb[q / 48][4 * (q % 48)];        // rows have an odd word stride: four 4-byte stores
            d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
        }
    }
    __syncthreads();
#ifdef J2K_WALK_STAMP
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif
    if (tid >= 64) return;
    const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, R = (h + 3) / 4;
    // the walking wavefront's trip count: the largest pair count among its blocks -- reduced while all 64 lanes are still
    // active (a butterfly over a wave with holes leaves different partial maxima in different lanes: taken after the early
    // returns below it cut some blocks' walks short, found by tools/fuzz_gpu.py on a frame of mixed block sizes)
    const bool walks = have && tag != HT_PAIR_SERIAL && tag != HT_PAIR_ZERO;
    int nsteps = walks ? R * P : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, o));
    nsteps = min(__builtin_amdgcn_readfirstlane(nsteps), HT_WALK_MAX_PAIRS);
    if (!walks) return;
    const uint32_t *row = &S.vb[lane][0];
    uint32_t pos = 0;
#define HT_FETCH(w0, w1)                                                          \
    uint32_t w0, w1;                                                               \
    {                                                                              \
        const uint32_t *p_ = row + (pos >> 5);                                     \
        const uint32_t d0_ = p_[0], d1_ = p_[1], d2_ = p_[2];                      \
        w0 = __builtin_amdgcn_alignbit(d1_, d0_, pos & 31);                        \
        w1 = __builtin_amdgcn_alignbit(d2_, d1_, pos & 31);                        \
    }
    // ---- first row: tbl0, general u-VLC ----
    for (int pi = 0; pi < P; pi++) {
        HT_FETCH(w0, w1)
        const uint32_t qinf = S.tbl0[w0 & 0x7F];                    // first quad: context is always 0
        const uint32_t rho = (qinf >> 4) & 0xF, uOff1 = (qinf >> 3) & 1, len1 = qinf & 0xF;
        const uint32_t qinf2 = S.tbl0[((rho >> 2) << 7) | ((w0 >> len1) & 0x7F)];
        const uint32_t rho2 = (qinf2 >> 4) & 0xF, uOff2 = (qinf2 >> 3) & 1, len = len1 + (qinf2 & 0xF);
        const uint32_t vw = __builtin_amdgcn_alignbit(w1, w0, len);  // len <= 30
        const uint32_t mode = (uOff1 << 1) | uOff2;
        uint32_t u[2];
        const uint32_t used = decode_uvlc(vw, mode, u, 1);
        pos += len + used;
        rec[pi] = rho | rho2 << 4 | mode << 13 | vw << 16;           // (the later rows' layout; bits 8..12 are not read)
    }
    // ---- later rows ----
    // A lone wavefront retires about one instruction per 8-9 cycles whatever their dependencies (measured here and on the MQ
    // coder), so what a step costs is its instruction COUNT plus the one LDS round trip of the table look-up: the loop is
    // kept to the instructions the next position needs.  The u-VLC's length comes from two nibble tables in 32-bit
    // constants (prefix lengths 3,1,2,1,3,1,2,1 and totals 8,1,2,1,4,1,2,1 for the three prefix bits), the record is the
    // table entry as it is (rho | rho2 << 4 | len << 8 | uOff2 << 13 | uOff1 << 14) with the 16 stream bits at the u-VLC on
    // top, and the trip count is the wavefront's maximum (scalar loop control): a lane whose block has fewer pairs walks on
    // through its zero-padded row and writes records nobody reads (every block owns HT_WALK_MAX_PAIRS record words).
    // (Tried: the stream words prefetched a step ahead into registers -- the selects cost what the round trip saves.)
    {
        for (int it = P; it < nsteps; it++) {
            HT_FETCH(w0, w1)
            uint32_t e = S.pair1[w0 & 0x3FFF];
            if (e & 0x8000) {                                            // rare: length nibble > 7
                const uint32_t qinf = S.tbl1[w0 & 0x7F];
                const uint32_t rho = (qinf >> 4) & 0xF, len1 = qinf & 0xF;
                const uint32_t qinf2 = S.tbl1[((rho >> 2) << 7) | ((w0 >> len1) & 0x7F)];
                e = rho | (qinf2 & 0xF0) | (len1 + (qinf2 & 0xF)) << 8 | ((qinf2 >> 3) & 1) << 13 | ((qinf >> 3) & 1) << 14;
            }
            const uint32_t len = (e >> 8) & 0x1F;
            const uint32_t vw = __builtin_amdgcn_alignbit(w1, w0, len);
            // u-VLC length (see decode_uvlc_later): the first prefix is read iff uOff2 (bit 13), the second iff uOff1 (bit 14)
            const uint32_t m1 = (uint32_t)((int32_t)(e << 18) >> 31), m2 = (uint32_t)((int32_t)(e << 17) >> 31);
            const uint32_t pl1 = (0x12131213u >> ((vw & 7) << 2)) & 3 & m1;
            const uint32_t t1 = (0x12141218u >> ((vw & 7) << 2)) & 0xF & m1;
            const uint32_t t2 = (0x12141218u >> (((vw >> pl1) & 7) << 2)) & 0xF & m2;
            pos += len + t1 + t2;
            rec[it] = (e & 0x7FFF) | vw << 16;
        }
    }
#undef HT_FETCH
#ifdef J2K_WALK_STAMP
    const long long st2 = __builtin_amdgcn_s_memtime();
    if ((blockIdx.x == 0 || blockIdx.x == 50) && lane == 0)
        printf("walk wg %d: staging %lld cyc, loop %lld cyc (%d pairs, %d bits)\n", (int)blockIdx.x, st1 - st0, st2 - st1, R * P, (int)pos);
#endif
}

#define HT_MAX_FF 64
// Decoded blocks (115 MB per 4K frame, 3/4 of it zeros) are a pure output stream: nothing in the library reads them back,
// so they are stored non-temporally.  Plain stores leave them dirty in the 256 MB Infinity Cache, and whatever kernel
// runs next pays their write-back on top of its own traffic (measured with tools/probe/l0_fwd_dev.hip: the level-0
// forward transform takes 31 us after a 300 MB plain-store copy, 27 us after the same copy with nt stores).
#ifndef J2K_DECODED_NT
#define J2K_DECODED_NT 1
#endif
__device__ __forceinline__ void st_decoded(int32_t *p, int a, int b, int c, int d) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const v4i_ v = {a, b, c, d};
    if (J2K_DECODED_NT) __builtin_nontemporal_store(v, reinterpret_cast<v4i_ *>(p));
    else *reinterpret_cast<v4i_ *>(p) = v;
}

#ifdef J2K_DEC_STAMP
static __device__ long long g_dbg_unstuff_dummy;
#define g_dbg_unstuff S.dbg_t
#endif
struct HtDecShared {
#ifdef J2K_DEC_STAMP
    long long dbg_t;
#endif
    uint32_t mbuf[HT_DEC_MWORDS];
    uint32_t pair[HT_WALK_MAX_PAIRS];
    uint32_t ffpos[HT_MAX_FF];      // indices of 0xFF bytes in the MagSgn segment (only needed when a u exceeds 32)
    uint32_t nff;
};

// MagSgn unstuffing + extraction for one block (one wavefront); records come from ht_walk_kernel
// bit offset the reference's forward reader has LOADED after j 4-byte reads (ht.go:399-501): byte k of the
// segment starts at 8k - #{0xFF bytes among b[0..k-2]}; past the segment the reader feeds 0xFF bytes, 7 bits each
// (8 for the first one unless the last real byte is 0xFF).
__device__ uint32_t ht_loaded_bits(const HtDecShared &S, uint32_t nff, long segLen, uint32_t j) {
    const long k = 4 * (long)j;
    const long kk = k < segLen ? k : segLen;
    uint32_t f = 0, last_ff = 0;
    for (uint32_t i = 0; i < nff; i++) {
        if ((long)S.ffpos[i] + 2 <= kk) f++;
        if ((long)S.ffpos[i] == segLen - 1) last_ff = 1;
    }
    uint32_t b = (uint32_t)(8 * kk) - f;
    if (k > segLen) b += (last_ff ? 7u : 8u) + 7u * (uint32_t)(k - segLen - 1);
    return b;
}

// returns false when the block needs the bit-serial decoder (never on encoder output)
// mpos2[t] = bit position of pair lane + 64 t in the MagSgn bit string, total_bits = the bits all pairs consume (both from the
// caller, which has the records in registers): the LDS string is prepared only as far as it will be read.
// STRIDED (the closed-loop frame decoder): row y of the block lies at out + y * os_ -- its window of a coefficient plane -- instead of out + y * w
template <bool STRIDED>
__device__ bool ht_extract_fast(HtDecShared &S, const uint8_t *__restrict__ data, long len, long scup, int w, int h,
                                int32_t *__restrict__ out, int lane, bool bigu, const uint32_t (&mpos2)[2], uint32_t total_bits,
                                const uint32_t (&raw0)[4], int os_) {
    const size_t os64 = STRIDED ? (size_t)os_ : (size_t)64, osw = STRIDED ? (size_t)os_ : (size_t)w;
    const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, R = (h + 3) / 4, N = R * P;
    const long lcup = len;
    {
        const long segLen0 = lcup - scup;
        const long cap0 = (long)HT_DEC_MWORDS * 4 - 16;
        const uint32_t zw = bigu ? (uint32_t)HT_DEC_MWORDS
                                 : min((uint32_t)HT_DEC_MWORDS, (uint32_t)(((segLen0 < cap0 ? segLen0 : cap0) * 8) >> 5) + 3u);
        for (uint32_t i = lane; i < zw; i += 64) S.mbuf[i] = 0;       // deposits reach 8 x (bytes staged) bits
    }
    if (lane == 0) S.nff = 0;
    wave_sync();
    constexpr int PF = 4;   // prefetch 4 x 256 bytes (one aligned dword per lane each) before processing
    {
        const long segLen = lcup - scup;
        const long maxbytes = (long)HT_DEC_MWORDS * 4 - 16;
        const long nb = segLen < maxbytes ? segLen : maxbytes;
        const long d = (long)((uintptr_t)data & 3);                // segment byte k lives at aligned byte d + k
        // (pointer arithmetic, not an integer round trip: a pointer made from an integer is a flat pointer)
        const uint32_t *wsrc = reinterpret_cast<const uint32_t *>(data - d);
        const long ndw = (d + nb + 3) >> 2;                          // aligned dwords covering the segment
        uint32_t off = 0, carry = 0;
        for (long j0 = 0; j0 < ndw; j0 += 64 * PF) {
            uint32_t raw[PF];
#pragma unroll
            for (int c = 0; c < PF; c++) {
                const long j = j0 + 64 * c + lane;
                if (j0 == 0) raw[c] = raw0[c];                      // the first 1 KB was fetched by the caller, ahead of its zero fill
                else raw[c] = wsrc[j < ndw ? j : ndw - 1];          // clamped: unconditional, issued back to back
            }
#pragma unroll
            for (int c = 0; c < PF; c++) {
                if (j0 + 64 * c >= ndw) break;
                const long j = j0 + 64 * c + lane;
                const uint32_t dw = raw[c];
                uint32_t prev3 = __shfl_up(dw >> 24, 1);            // last byte of the previous dword
                if (lane == 0) prev3 = carry;
                carry = __shfl(dw >> 24, 63);
                const long k0 = 4 * j - d;                           // segment index of this dword's byte 0
                uint32_t wd[4], tot = 0;
                bool plain = (j < ndw) && k0 >= 1 && k0 + 4 <= nb && prev3 != 0xFF;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const long k = k0 + t;
                    const uint32_t pb = t ? ((dw >> (8 * (t - 1))) & 0xFF) : prev3;
                    const bool valid = (j < ndw) && k >= 0 && k < nb;
                    wd[t] = valid ? ((k > 0 && pb == 0xFF) ? 7u : 8u) : 0u;
                    if (t && pb == 0xFF) plain = false;
                    tot += wd[t];
                    if (bigu && valid && ((dw >> (8 * t)) & 0xFF) == 0xFF) {      // rare path: remember where the 0xFF bytes are
                        const uint32_t idx = atomicAdd(&S.nff, 1u);
                        if (idx < HT_MAX_FF) S.ffpos[idx] = (uint32_t)k;
                    }
                }
                const uint32_t ws = wave_incl_scan(tot);
                uint32_t pos = off + ws - tot;
                if (plain) {
                    if (dw) or_bits(S.mbuf, pos, (uint64_t)dw);      // four 8-bit bytes: one deposit
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const uint32_t bt = (dw >> (8 * t)) & 0xFF;
                        if (wd[t] && bt) or_bits(S.mbuf, pos, (uint64_t)bt);
                        pos += wd[t];
                    }
                }
                off += wave_last(ws);
            }
        }
        wave_sync();
        if (bigu && (nb != segLen || S.nff > HT_MAX_FF)) return false;
        if (nb == segLen) {   // everything past the segment reads as ones (ht.go:407, 447-449, 462-464)
            const uint32_t wd0 = off >> 5;
            const uint32_t wend = bigu ? (uint32_t)HT_DEC_MWORDS : min((uint32_t)HT_DEC_MWORDS, (total_bits >> 5) + 3u);   // .. as far as it is read
            for (uint32_t i = wd0 + lane; i < wend; i += 64)
                S.mbuf[i] = (i == wd0) ? (S.mbuf[i] | (0xFFFFFFFFu << (off & 31))) : 0xFFFFFFFFu;
        }
    }
    wave_sync();
#ifdef J2K_DEC_STAMP
    g_dbg_unstuff = wall_clock64();
#endif
#ifdef J2K_DEC_STOP_B
    if (total_bits != 0x7fffffff) return true;      // DEV experiment: .. + the MagSgn unstuffing
#endif
    // ---- u > 32: the reference's uint32 bit counter wraps when it advances by more bits than it has loaded
    //      (ht.go:515-519); from then on the reader never refills and everything reads as zero.  Equivalent:
    //      the bit string is cut to zeros at L = "bits loaded when the first such advance happens".  Find L. ----
    uint32_t Lcut = 0xFFFFFFFFu;
    if (bigu) {
        const long segLen = lcup - scup;
        const uint32_t nff = S.nff;
        uint32_t best_key = 0xFFFFFFFFu, best_L = 0xFFFFFFFFu, mb = 0;
        for (int i0 = 0; i0 < N; i0 += 64) {
            const int it = i0 + lane;
            uint32_t nbits = 0, rho = 0, rho2 = 0, u0 = 1, u1 = 1;
            if (it < N) {
                const uint32_t info = S.pair[it];
                const int pi = it % P, xb = pi * 8;
                rho = info & 0xF; rho2 = (info >> 4) & 0xF; u0 = (info >> 8) & 0x3F; u1 = (info >> 14) & 0x3F;
                const uint32_t m1 = (xb + 4 <= w) ? 0xFu : ((xb < w) ? ((1u << (w - xb)) - 1) : 0u);
                const uint32_t m2 = (xb + 8 <= w) ? 0xFu : ((xb + 4 < w) ? ((1u << (w - xb - 4)) - 1) : 0u);
                rho &= m1; rho2 &= m2;
                nbits = __popc(rho) * (u0 + 1) + __popc(rho2) * (u1 + 1);
            }
            const uint32_t ns = wave_incl_scan(nbits);
            uint32_t mpos = mb + ns - nbits;
            mb += wave_last(ns);
            if (it < N) {
                for (int i = 0; i < 8; i++) {
                    const uint32_t rr = (i < 4) ? rho : rho2, emb = (i < 4) ? u0 : u1;
                    if (!((rr >> (i & 3)) & 1)) continue;
                    if (emb > 32 && best_key == 0xFFFFFFFFu) {
                        uint32_t j = (mpos + 32) / 32;
                        if (j < 2) j = 2;                              // two reads happen during initMagSgn
                        uint32_t G = ht_loaded_bits(S, nff, segLen, j);
                        while (G < mpos + 32) G = ht_loaded_bits(S, nff, segLen, ++j);
                        if (G - mpos < emb) { best_key = (uint32_t)(it * 8 + i); best_L = G; }
                    }
                    mpos += emb + 1;
                }
            }
        }
        uint32_t k = best_key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) k = min(k, (uint32_t)__shfl_xor(k, o));
        const unsigned long long who = __ballot(best_key == k && k != 0xFFFFFFFFu);
        if (who) Lcut = __shfl(best_L, __ffsll((long long)who) - 1);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (64 * t >= N) break;
        const int it = 64 * t + lane;
        uint32_t rho = 0, rho2 = 0, u0 = 1, u1 = 1;
        int r = 0, pi = 0;
        if (it < N) {
            const uint32_t info = S.pair[it];
            r = it / P; pi = it - r * P;
            rho = info & 0xF; rho2 = (info >> 4) & 0xF; u0 = (info >> 8) & 0x3F; u1 = (info >> 14) & 0x3F;
            const int xb = pi * 8;
            // samples beyond the block width are skipped even when their rho bit is set (ht.go:661, 689)
            const uint32_t m1 = (xb + 4 <= w) ? 0xFu : ((xb < w) ? ((1u << (w - xb)) - 1) : 0u);
            const uint32_t m2 = (xb + 8 <= w) ? 0xFu : ((xb + 4 < w) ? ((1u << (w - xb - 4)) - 1) : 0u);
            rho &= m1; rho2 &= m2;
        }
        uint32_t mpos = mpos2[t];
        int vals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (it < N && !bigu) {
            // every u is at most 31 here (bigu starts at 32): one 32-bit window holds a sample's magnitude bits and the sign
            // behind them.  Branch-free: an absent sample reads the window at the current position and advances by nothing.
            const uint32_t half0 = 1u << (u0 - 1), half1 = 1u << (u1 - 1);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t rr = (i < 4) ? rho : rho2, emb = (i < 4) ? u0 : u1;
                const bool on = (rr >> (i & 3)) & 1;
                const uint32_t wd = mpos >> 5;
                const uint32_t win = __builtin_amdgcn_alignbit(S.mbuf[wd + 1], S.mbuf[wd], mpos & 31);
                const uint32_t mag = __builtin_amdgcn_ubfe(win, 0, emb) + ((i < 4) ? half0 : half1);
                const uint32_t sg = 0u - ((win >> emb) & 1);
                vals[i] = on ? (int32_t)((mag ^ sg) - sg) : 0;
                mpos += on ? emb + 1 : 0u;
            }
        } else if (it < N) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t rr = (i < 4) ? rho : rho2, emb = (i < 4) ? u0 : u1;
                if (!((rr >> (i & 3)) & 1)) continue;
                const uint32_t magVal = get_bits32_cut(S.mbuf, mpos, Lcut);
                const uint32_t sign = get_bits32_cut(S.mbuf, mpos + emb, Lcut) & 1;
                const uint32_t mag = (magVal & (shl32(1, emb) - 1)) + shl32(1, emb - 1);
                mpos += emb + 1;
                vals[i] = sign ? (int32_t)(0u - mag) : (int32_t)mag;
            }
        }
        // the pair's 8 columns of the coded row are written exactly once (zeros included): the zero fill
        // of the kernel skips coded rows, so no store ever has to be ordered behind another
        if (w == 64 && (((uintptr_t)out) & 15) == 0 && (!STRIDED || (os_ & 3) == 0)) {
            // 64-wide blocks: lane = (row of this round, pair) holds 32 bytes of a 256-byte row, and storing them as they lie
            // makes every store instruction write 16-byte pieces 32 bytes apart -- half-written 128-byte lines, which the
            // memory system turns into partial writes (measured: the kernel's 102 MB left at 3.2 TB/s against 5.4 TB/s for
            // the zero rows alone).  One cross-lane move later each instruction writes four whole rows.
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int drow = half * 4 + (lane >> 4), q = lane & 15;     // the row of this round / 16-byte piece I store
                const int src = drow * 8 + (q >> 1);                        // the lane that decoded it
                int o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int a = __shfl(vals[i], src), b = __shfl(vals[4 + i], src);
                    o[i] = (q & 1) ? b : a;
                }
                const int rg = 8 * t + drow;                                // coded row index in the block
#ifdef J2K_DEC_NO_CODED_STORES
                if (o[0] == 0x7fffffff)            // DEV experiment: everything but the coded rows' stores
#endif
                if (rg < R) st_decoded(out + (size_t)(4 * rg) * os64 + 4 * q, o[0], o[1], o[2], o[3]);
            }
        } else if (it < N) {
            const int xb = pi * 8;
            int32_t *orow = out + (size_t)(4 * r) * osw + xb;
            if (xb + 8 <= w && (((uintptr_t)orow) & 15) == 0) {
                st_decoded(orow, vals[0], vals[1], vals[2], vals[3]);
                st_decoded(orow + 4, vals[4], vals[5], vals[6], vals[7]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (xb + i < w) orow[i] = vals[i];
            }
        }
    }
    return true;
}

// four independent wavefronts per workgroup, one block each (no workgroup barrier anywhere on the fast path)
// STRIDED: J.out_off / J.stride are the block's window of a coefficient plane (closed-loop plans: the windows partition the plane) and only the
// coded rows are ever written -- the caller's planes were zeroed once (j2k_plan_decode_frame_pixels); no dense block, no placement copy
template <bool STRIDED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void ht_decode_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                        const uint8_t *__restrict__ stream, const uint64_t *__restrict__ offs,
                                                        const uint32_t *__restrict__ lens, int32_t *__restrict__ decoded,
                                                        const uint32_t *__restrict__ pairs, int coded_rows_only_) {
    const int coded_rows_only = STRIDED ? 1 : coded_rows_only_;
    __shared__ HtDecShared S4[4];
    HtDecShared &S = S4[threadIdx.x >> 6];
    const int jid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (jid >= njobs) return;
    const int lane = threadIdx.x & 63;
#ifdef J2K_DEC_STAMP
    const long long d0 = wall_clock64();
#endif
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h;
    const int os_ = STRIDED ? J.stride : w;                       // row stride of the output
    int32_t *out = decoded + J.out_off;
    const size_t n = (size_t)w * h;
    // records first (their latency hides behind the zero fill)
    const uint32_t *rec = pairs + (size_t)jid * HT_WALK_REC;
    const uint32_t r0 = rec[lane], r1 = rec[64 + lane];
    const uint64_t my_off = offs[jid];                         // (issued with the records: one round trip for all of them)
    const uint32_t my_len = lens[jid], my_scup = rec[HT_WALK_MAX_PAIRS];
    const uint32_t tag = __shfl(r0, 0);
    const bool fast = (tag != HT_PAIR_ZERO) && (tag != HT_PAIR_SERIAL);
    // fresh NewHTDecoder: zeroed data.  On the fast path the coded rows (y % 4 == 0) are written in full by the
    // extraction and the other rows are zeroed LAST (vmcnt retires in order: a load issued behind these
    // stores would wait for all of them).  Other paths zero everything up front.
    // (row index kept incrementally: a 64-bit i / wq per iteration costs ~150 VALU instructions and made this
    //  loop the most expensive part of the kernel)
    // coded_rows_only (j2k_plan_set_decode_coded_rows_only): the rows the reference's decoder never writes (y % 4 != 0,
    // SURVEY fact 3 / row a16) are not written here either -- the caller owns a buffer it zeroed once, as a POOLED
    // HTDecoder's data slice is (ht.go:1393-1429: GetHTDecoder hands back a decoder whose slice is not cleared on the
    // normal path) -- so three quarters of this kernel's store stream disappear.  The coded rows are always written in
    // full, zeros included.
    auto zero_fill = [&](bool skip_coded) {
        if (coded_rows_only) {
            if (skip_coded) return;                              // fast path: nothing but the coded rows, which the extraction writes
            // other paths: zero the coded rows, which the serial decoder (or nobody: invalid stream) then writes
            const uint32_t uw = (uint32_t)w, nrow = (uint32_t)(h + 3) >> 2;
            for (uint32_t i = lane; i < uw * nrow; i += 64) out[(size_t)(i / uw) * 4 * (STRIDED ? (uint32_t)os_ : uw) + i % uw] = 0;
            return;
        }
        if (w == 64 && (J.out_off & 3) == 0) {
            // 16 quads per row: the 64 lanes cover four rows per step, lanes 0..15 always the coded one
            if (skip_coded && lane < 16) return;
            int32_t *q = out + 4 * (size_t)lane;
            for (int i = 0; i < h / 4; i++) st_decoded(q + 256 * (size_t)i, 0, 0, 0, 0);
            const int rem = h & 3;                               // last, partial stripe
            if (lane < 16 * rem) st_decoded(q + 256 * (size_t)(h / 4), 0, 0, 0, 0);
        } else if ((w & 3) == 0 && (J.out_off & 3) == 0) {
            const uint32_t wq = (uint32_t)w >> 2, nq = (uint32_t)(n >> 2);
            const uint32_t dy = 64u / wq, dx = 64u % wq;
            uint32_t y = (uint32_t)lane / wq, x = (uint32_t)lane % wq;
            for (uint32_t i = lane; i < nq; i += 64) {
                if (!skip_coded || (y & 3)) st_decoded(out + 4 * (size_t)i, 0, 0, 0, 0);
                y += dy; x += dx;
                if (x >= wq) { x -= wq; y++; }
            }
        } else {
            const uint32_t uw = (uint32_t)w;
            const uint32_t dy = 64u / uw, dx = 64u % uw;
            uint32_t y = (uint32_t)lane / uw, x = (uint32_t)lane % uw;
            for (uint32_t i = lane; i < (uint32_t)n; i += 64) {
                if (!skip_coded || (y & 3)) out[i] = 0;
                y += dy; x += dx;
                if (x >= uw) { x -= uw; y++; }
            }
        }
    };
    if (!fast) zero_fill(false);
    // Fast path: the first 1 KB of the MagSgn segment is requested NOW and the zero fill of the rows that carry no coded
    // sample goes out right behind it -- 12 of a block's 16 KB, fire and forget, so the memory system writes while the
    // wavefront unstuffs and extracts.  (Stores first and loads behind them would make the loads wait for every store:
    // vmcnt retires in order.  Stores last -- the round-1 order -- left HBM idle for the first half of the kernel and
    // then had all 7005 wavefronts store at once: the kernel ran at half the write bandwidth.)
    uint32_t raw0[4] = {0, 0, 0, 0};
#ifdef J2K_DEC_STAMP
    const long long d_a = wall_clock64();
#endif
    if (fast) {
        const uint8_t *fdata = stream + my_off;
        const long segLen = (long)my_len - (long)my_scup;
        const long maxbytes = (long)HT_DEC_MWORDS * 4 - 16;
        const long nb = segLen < maxbytes ? segLen : maxbytes;
        const long d = (long)((uintptr_t)fdata & 3);
        const uint32_t *wsrc = reinterpret_cast<const uint32_t *>(fdata - d);
        const long ndw = (d + nb + 3) >> 2;
        if (ndw > 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const long j = 64 * c + lane;
                raw0[c] = wsrc[j < ndw ? j : ndw - 1];
            }
        }
        zero_fill(true);
#ifdef J2K_DEC_ONLY_ZERO
        return;                                    // DEV experiment: the store stream alone
#endif
    }
#ifdef J2K_DEC_STAMP
    const long long d_b = wall_clock64();
#endif
    // finish the walk's records: u0/u1 from the 16 stream bits kept at the u-VLC (ht.go:716-864) -> rho | rho2 << 4 | u0 << 8 | u1 << 14,
    // and every pair's position in the MagSgn bit string: the prefix sum of popcount(rho) * (u + 1) (ht.go:661-710)
    bool bigu = false;
    uint32_t mpos2[2] = {0, 0}, total_bits = 0;
    {
        const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, N = ((h + 3) / 4) * P;
        uint32_t rr[2] = {r0, r1};
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int it = lane + 64 * t;
            const uint32_t r = rr[t], mode = (r >> 13) & 3, vw = r >> 16;
            uint32_t u0, u1;
            decode_uvlc_later(vw, mode >> 1, mode & 1, u0, u1);
            if (it < P) {                                         // first row: the other u-VLC rule (a few lanes of round 0)
                uint32_t u[2];
                decode_uvlc(vw, mode, u, 1);
                u0 = u[0]; u1 = u[1];
            }
            if (it < N && (u0 > 31 || u1 > 31)) bigu = true;   // 32 itself only because the fast extraction reads a sample's u + 1 bits from one 32-bit window
            S.pair[it] = (r & 0xFF) | (u0 & 0x3F) << 8 | (u1 & 0x3F) << 14;
            uint32_t nbits = 0;
            if (it < N) {
                const int pi = it % P, xb = pi * 8;
                // samples beyond the block width are skipped even when their rho bit is set (ht.go:661, 689)
                const uint32_t m1 = (xb + 4 <= w) ? 0xFu : ((xb < w) ? ((1u << (w - xb)) - 1) : 0u);
                const uint32_t m2 = (xb + 8 <= w) ? 0xFu : ((xb + 4 < w) ? ((1u << (w - xb - 4)) - 1) : 0u);
                nbits = __popc(r & 0xF & m1) * ((u0 & 0x3F) + 1) + __popc((r >> 4) & 0xF & m2) * ((u1 & 0x3F) + 1);
            }
            const uint32_t ns = wave_incl_scan(nbits);
            mpos2[t] = total_bits + ns - nbits;
            total_bits += wave_last(ns);
        }
        bigu = __any(bigu);
    }
    wave_sync();
#ifdef J2K_DEC_STAMP
    const long long d_c = wall_clock64();
#endif
#ifdef J2K_DEC_STOP_A
    if (total_bits != 0x7fffffff) return;           // DEV experiment: loads, zero rows, u-VLC + positions only
#endif
    if (tag == HT_PAIR_ZERO) return;
    if (tag != HT_PAIR_SERIAL) {
        const uint8_t *fdata = stream + my_off;
        const long flen = (long)my_len;
        const long fscup = (long)my_scup;                              // validated by ht_vlcprep_kernel
        if (ht_extract_fast<STRIDED>(S, fdata, flen, fscup, w, h, out, lane, bigu, mpos2, total_bits, raw0, os_)) {
#ifdef J2K_DEC_STAMP
            if ((jid % 700) == 13 && lane == 0) printf("dec job %d: start %lld | job+records %lld | prefetch+zero issue %lld | uvlc+pos %lld | unstuff %lld | extract %lld | end %lld (x10 ns)\n", jid, d0 % 100000, d_a - d0, d_b - d_a, d_c - d_b, g_dbg_unstuff - d_c, wall_clock64() - g_dbg_unstuff, wall_clock64() % 100000);
#endif
            return;
        }
        zero_fill(false);       // exotic input (more than HT_MAX_FF 0xFF bytes together with u > 32): serial decoder
    }
    // serial path: the zero fill must be complete before lane 0 overwrites samples (single wave: wait, no barrier)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    const uint8_t *data = stream + offs[jid];
    const long len = (long)lens[jid];
    if (len < 2) return;                                        // ht.go:94-100
    const long scup = (long)data[len - 1] + ((long)(data[len - 2] & 0x0F) << 8);
    if (scup < 2 || scup > len) return;                         // ht.go:104-111
    const long lcup = len;
    if (!init_mel_ok(data, len, lcup, scup)) return;            // ht.go:117-122

    RevStream vlc{data, len, lcup - 2, scup - 2, 0, 0, 0};      // initVLC ht.go:276-314
    if (vlc.pos >= 0 && vlc.pos < len) {
        const uint32_t b = data[vlc.pos];
        vlc.pos--;
        vlc.tmp = (uint64_t)(b >> 4);
        vlc.bits = 4 - (uint32_t)((vlc.tmp & 7) >> 2);
        vlc.unstuff = (b | 0x0F) > 0x8F;
    }
    {
        long num = 1 + (vlc.pos & 3);
        if (num > vlc.size) num = vlc.size;
        for (long i = 0; i < num; i++) {
            uint32_t b = 0;
            if (vlc.pos >= 0 && vlc.pos < len) { b = data[vlc.pos]; vlc.pos--; }
            const uint32_t dBits = (vlc.unstuff && (b & 0x7F) == 0x7F) ? 7 : 8;
            vlc.tmp |= shl64((uint64_t)b, vlc.bits);
            vlc.bits += dBits;
            vlc.unstuff = b > 0x8F;
        }
        vlc.size -= num;
        rev_read(vlc);
    }
    FwdStream ms{data, len, 0, lcup - scup, 0, 0, 0, 0xFF};     // initMagSgn ht.go:399-429
    for (int i = 0; i < 4; i++) {
        uint32_t b;
        if (ms.size > 0 && ms.pos < len) { b = data[ms.pos]; ms.pos++; ms.size--; } else b = 0xFF;
        const uint32_t dBits = ms.unstuff ? 7 : 8;
        ms.tmp |= shl64((uint64_t)b, ms.bits);
        ms.bits += dBits;
        ms.unstuff = (b == 0xFF);
    }
    fwd_read(ms);

    const int quadCols = (w + 3) / 4;
    for (int y = 0; y < h; y += 4) {                            // ht.go:589-711
        const int initial = (y == 0);
        const uint16_t *tbl = initial ? c_vlc_tbl0 : c_vlc_tbl1;
        for (int qx = 0; qx < quadCols; qx += 2) {
            uint32_t vlcVal = rev_fetch(vlc);
            // first quad context is always 0: sigma1 holds 4-bit rho (>>4 == 0), lineState is never written
            const uint32_t qinf = tbl[vlcVal & 0x7F];
            const uint32_t rho = (qinf >> 4) & 0xF, uOff1 = (qinf >> 3) & 1;
            rev_advance(vlc, qinf & 0xF);
            vlcVal = rev_fetch(vlc);
            const uint32_t qinf2 = tbl[((rho >> 2) << 7) | (vlcVal & 0x7F)];
            const uint32_t rho2 = (qinf2 >> 4) & 0xF, uOff2 = (qinf2 >> 3) & 1;
            rev_advance(vlc, qinf2 & 0xF);
            uint32_t u[2] = {1, 1};
            const uint32_t mode = (uOff1 << 1) | uOff2;
            if (mode > 0) {
                vlcVal = rev_fetch(vlc);
                rev_advance(vlc, decode_uvlc(vlcVal, mode, u, initial));
            }
            for (int q = 0; q < 2; q++) {                       // ht.go:661-710
                const uint32_t r = q ? rho2 : rho, emb = u[q];
                const int base = (qx + q) * 4;
                for (int i = 0; i < 4 && base + i < w; i++) {
                    if (!((r >> i) & 1)) continue;
                    const uint32_t magVal = fwd_fetch(ms);
                    const uint32_t mag = (magVal & (shl32(1, emb) - 1)) + shl32(1, emb - 1);
                    fwd_advance(ms, emb);
                    const uint32_t sign = fwd_fetch(ms) & 1;
                    fwd_advance(ms, 1);
                    out[(size_t)y * (STRIDED ? (size_t)os_ : (size_t)w) + base + i] = sign ? (int32_t)(0u - mag) : (int32_t)mag;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
static bool g_tables_ready[16] = {false};

// derived tables live in device globals of this code object: build them once per device, in stream order
static hipError_t ht_tables_ready(hipStream_t s) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (g_tables_ready[dev]) return hipSuccess;
    hipLaunchKernelGGL(ht_build_enc_table, dim3(1), dim3(128), 0, s);
    hipLaunchKernelGGL(ht_build_pair_table, dim3(64), dim3(256), 0, s);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(ht_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(HtWalkShared))) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;   // once per device: other streams may use the tables next
    g_tables_ready[dev] = true;
    return hipSuccess;
}

hipError_t launch_ht_encode(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *slots,
                            uint32_t *lens, uint8_t *numbps, int *fault, uint32_t *maglens, const HtUJob *utab, int nunique,
                            const int *alias_ids) {
    if (njobs <= 0) return hipSuccess;
    hipError_t e;
    if ((e = ht_tables_ready(s)) != hipSuccess) return e;
    const int n = utab ? nunique : njobs;
    hipLaunchKernelGGL(ht_encode_kernel, dim3(n), dim3(64), 0, s, jobs, n, coef, slots, lens, numbps, fault, maglens, utab, alias_ids);
    return hipGetLastError();
}

// every block of the batch must be on the parallel path ((h+3)/4 * w <= HT_FAST_MAX_SAMPLES); status = njobs u64 words
// that are never cleared (epoch-tagged)
hipError_t launch_ht_encode_stream(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *stream,
                                   uint64_t *offs, uint32_t *lens, uint8_t *numbps, uint64_t *status, uint32_t epoch, int *fault) {
    if (njobs <= 0) return hipSuccess;
    hipError_t e;
    if ((e = ht_tables_ready(s)) != hipSuccess) return e;
    hipLaunchKernelGGL(ht_encode_stream_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, coef, stream,
                       reinterpret_cast<unsigned long long *>(offs), lens, numbps, reinterpret_cast<unsigned long long *>(status), epoch, fault);
    return hipGetLastError();
}
int ht_fast_max_samples() { return HT_FAST_MAX_SAMPLES; }

// words of device scratch launch_ht_decode needs for njobs blocks: per-pair records + unstuffed VLC bit strings
size_t ht_decode_scratch_words(int njobs) { return (size_t)njobs * (HT_WALK_REC + HT_VBITS_WORDS); }

// placed_jobs != NULL: the blocks are written straight into their windows of the coefficient planes `decoded` (ht_decode_kernel<true>)
hipError_t launch_ht_decode(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs,
                            const uint32_t *lens, int32_t *decoded, uint32_t *scratch, int coded_rows_only, const BlockJob *placed_jobs) {
    if (njobs <= 0) return hipSuccess;
    hipError_t e;
    if ((e = ht_tables_ready(s)) != hipSuccess) return e;
    uint32_t *pairs = scratch, *vbits = scratch + (size_t)njobs * HT_WALK_REC;
    for (int rep_ = 0; rep_ < dev_reps(32); rep_++)
    hipLaunchKernelGGL(ht_vlcprep_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, jobs, njobs, stream, offs, lens, vbits, pairs);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    for (int rep_ = 0; rep_ < dev_reps(64); rep_++)
    hipLaunchKernelGGL(ht_walk_kernel, dim3((njobs + HT_WALK_BLOCKS - 1) / HT_WALK_BLOCKS), dim3(256), sizeof(HtWalkShared), s, jobs, njobs, vbits, pairs);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    for (int rep_ = 0; rep_ < dev_reps(128); rep_++)
    if (placed_jobs) hipLaunchKernelGGL(ht_decode_kernel<true>, dim3((njobs + 3) / 4), dim3(256), 0, s, placed_jobs, njobs, stream, offs, lens, decoded, pairs, 1);
    else
    hipLaunchKernelGGL(ht_decode_kernel<false>, dim3((njobs + 3) / 4), dim3(256), 0, s, jobs, njobs, stream, offs, lens, decoded, pairs, coded_rows_only);
    return hipGetLastError();
}

}  // namespace j2k
