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
// Kernel shape: one code-block per wavefront.  The wave scans the whole block for max|x|
// (nil / numbps decision) with coalesced row loads; the bit-serial stream packing -- every
// code word's position depends on all previous lengths and on byte stuffing -- runs on lane 0
// straight into the block's output slot; the MEL zero fill and the final VLC move are done by
// all 64 lanes.  Go semantics kept: uint32 wraparound, shifts >= 32 give 0.
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

__device__ __forceinline__ uint32_t shl32(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
__device__ __forceinline__ uint64_t shl64(uint64_t x, uint32_t n) { return n >= 64 ? 0ull : x << n; }
__device__ __forceinline__ uint64_t shr64(uint64_t x, uint32_t n) { return n >= 64 ? 0ull : x >> n; }
__device__ __forceinline__ uint32_t uabs(int v) { return v < 0 ? 0u - (uint32_t)v : (uint32_t)v; }

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
#define HT_FAST_MAX_SAMPLES 2048                       /* coded samples (rows y%4==0) per block      */
#define HT_MS_WORDS (HT_FAST_MAX_SAMPLES * 31 / 32 + 4) /* worst case 31 bits per coded sample       */
#define HT_VLC_WORDS (HT_FAST_MAX_SAMPLES / 8 + 8)      /* <= 30 bits per item of 8 samples          */

__device__ __forceinline__ void or_bits(uint32_t *buf, uint32_t bitpos, uint64_t val) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint32_t lo = (uint32_t)(val << sh);
    const uint64_t hi = sh ? (val >> (32 - sh)) : (val >> 32);
    if (lo) atomicOr(&buf[wd], lo);
    if ((uint32_t)hi) atomicOr(&buf[wd + 1], (uint32_t)hi);
    if ((uint32_t)(hi >> 32)) atomicOr(&buf[wd + 2], (uint32_t)(hi >> 32));
}
__device__ __forceinline__ uint32_t get_bits8(const uint32_t *buf, uint32_t bitpos) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint64_t two = (uint64_t)buf[wd] | ((uint64_t)buf[wd + 1] << 32);
    return (uint32_t)(two >> sh) & 0xFF;
}

__device__ bool ht_encode_fast(const BlockJob &J, const int32_t *__restrict__ src, uint8_t *__restrict__ out, int lane,
                               uint32_t *vbuf, uint32_t *mbuf, long &magLenOut, long &vlcLenOut) {
    const int w = J.w, h = J.h, stride = J.stride;
    const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, R = (h + 3) / 4, N = R * P;
    const size_t nsamp = (size_t)w * h;
    const size_t maxSize = nsamp * 2 < 64 ? 64 : nsamp * 2;
    const long msCap = (long)(maxSize / 2), vlcCap = (long)(maxSize / 2);
    const size_t melLen = maxSize / 4;
    for (int i = lane; i < HT_VLC_WORDS; i += 64) vbuf[i] = 0;
    for (int i = lane; i < HT_MS_WORDS; i += 64) mbuf[i] = 0;
    __syncthreads();
    uint32_t vbase = 0, mbase = 0;   // running bit totals
    int bad = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int it = i0 + lane;
        uint64_t vv = 0; uint32_t vl = 0, ml = 0;
        uint32_t mval[8], mlen[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { mval[i] = 0; mlen[i] = 0; }
        if (it < N) {
            const int r = it / P, pi = it - r * P;
            const int initial = (r == 0);
            const int32_t *row = src + (size_t)(4 * r) * stride;
            const int xb = pi * 8;
            int v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (xb + i < w) ? row[xb + i] : 0;
            uint32_t rho = 0, rho2 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (v[i] != 0) rho |= 1u << i;
                if (v[4 + i] != 0) rho2 |= 1u << i;
            }
            const uint32_t e1 = g_vlc_enc[(initial << 6) | rho];
            const uint32_t e2 = g_vlc_enc[(initial << 6) | ((rho >> 2) << 4) | rho2];
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
        uint32_t vs = vl, ms = ml;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(vs, o), b = __shfl_up(ms, o);
            if (lane >= o) { vs += a; ms += b; }
        }
        const uint32_t vtot = __shfl(vs, 63), mtot = __shfl(ms, 63);
        uint32_t vpos = vbase + vs - vl, mpos = mbase + ms - ml;
        if (it < N) {
            or_bits(vbuf, vpos, vv);
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (mlen[i]) { or_bits(mbuf, mpos, (uint64_t)mval[i]); mpos += mlen[i]; }
        }
        vbase += vtot; mbase += mtot;
    }
    bad = __any(bad);
    __syncthreads();
    if (bad) return false;
    // ---- MagSgn emission with 0xFF stuffing (ht.go:1303-1341) ----
    const uint32_t TM = mbase;
    uint32_t pos = 0, last = 0;
    long outpos = 0;
    while (TM - pos >= 8) {
        if (last == 0xFF) {                         // wave-uniform: one 7-bit byte
            const uint32_t b = get_bits8(mbuf, pos) & 0x7F;
            if (outpos >= msCap) return false;
            if (lane == 0) out[outpos] = (uint8_t)b;
            outpos++; pos += 7; last = b;
            continue;
        }
        const uint32_t p = pos + 8 * lane;
        const bool valid = (p + 8 <= TM);
        const uint32_t b = valid ? get_bits8(mbuf, p) : 0;
        const unsigned long long vmask = __ballot(valid), fmask = __ballot(valid && b == 0xFF);
        const int nvalid = __popcll(vmask);
        const int first = fmask ? __ffsll((long long)fmask) - 1 : 64;
        const int count = min(nvalid, first + 1);
        if (outpos + count > msCap) return false;
        if (lane < count) out[outpos + lane] = (uint8_t)b;
        outpos += count; pos += 8 * count;
        last = (first < nvalid) ? 0xFF : __shfl(b, count - 1);
    }
    if (TM > pos) {                                 // magSgnFlush: the remaining < 8 bits, no stuffing rule
        if (outpos >= msCap) return false;
        if (lane == 0) out[outpos] = (uint8_t)get_bits8(mbuf, pos);
        outpos++;
    }
    const long magLen = outpos;
    // ---- VLC bytes: position-preserving stuffing (ht.go:1271-1300) ----
    const uint32_t TV = vbase;
    const long nfull = TV >> 3, vlcLen = (TV + 7) >> 3;
    if (vlcLen > vlcCap) return false;
    uint8_t *vout = out + magLen + melLen;
    for (long i = lane; i < vlcLen; i += 64) {
        uint32_t b = get_bits8(vbuf, (uint32_t)(8 * i));
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
            if (prev > 0x8F) b &= 0x7F;
        }
        vout[i] = (uint8_t)b;
    }
    magLenOut = magLen; vlcLenOut = vlcLen;
    return true;
}

__global__ __launch_bounds__(64) void ht_encode_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                       const int32_t *__restrict__ coef, uint8_t *__restrict__ slots,
                                                       uint32_t *__restrict__ lens, uint8_t *__restrict__ numbps,
                                                       int *__restrict__ fault) {
    __shared__ uint32_t s_vbuf[HT_VLC_WORDS];
    __shared__ uint32_t s_mbuf[HT_MS_WORDS];
    const int jid = blockIdx.x;
    if (jid >= njobs) return;
    const int lane = threadIdx.x;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h, stride = J.stride;
    const int32_t *src = coef + J.src_off;
    uint8_t *out = slots + J.out_off;

    // ---- max |x| over the WHOLE block: nil decision (ht.go:947-960) and numbps ----
    int maxMag = 0;  // Go compares int32: -MinInt32 stays negative and never wins
    if ((w & 3) == 0 && (stride & 3) == 0 && (J.src_off & 3) == 0) {
        const int wq = w >> 2, nq = wq * h;
        for (int e = lane; e < nq; e += 64) {
            const int y = e / wq, xq = e - y * wq;
            const int4 q = *reinterpret_cast<const int4 *>(src + (size_t)y * stride + 4 * xq);
            const int a0 = q.x < 0 ? (int)(0u - (uint32_t)q.x) : q.x, a1 = q.y < 0 ? (int)(0u - (uint32_t)q.y) : q.y;
            const int a2 = q.z < 0 ? (int)(0u - (uint32_t)q.z) : q.z, a3 = q.w < 0 ? (int)(0u - (uint32_t)q.w) : q.w;
            maxMag = max(max(maxMag, max(a0, a1)), max(a2, a3));
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
    if (maxMag == 0) {
        if (lane == 0) { lens[jid] = 0; numbps[jid] = 0; }
        return;
    }
    if ((size_t)((h + 3) / 4) * (size_t)w <= HT_FAST_MAX_SAMPLES) {
        long mLen = 0, vLen = 0;
        const size_t nsamp_ = (size_t)w * h;
        const size_t maxSize_ = nsamp_ * 2 < 64 ? 64 : nsamp_ * 2;
        const size_t melLen_ = maxSize_ / 4;
        if (!ht_encode_fast(J, src, out, lane, s_vbuf, s_mbuf, mLen, vLen)) {
            if (lane == 0) { atomicMax(fault, 1); lens[jid] = 0; numbps[jid] = 0; }
            return;
        }
        for (size_t i = lane; i < melLen_; i += 64) out[mLen + i] = 0;
        if (lane == 0) {
            const size_t scup = melLen_ + (size_t)vLen + 2;
            const size_t total = (size_t)mLen + scup;
            out[total - 2] = (uint8_t)(scup >> 8);
            out[total - 1] = (uint8_t)(scup & 0xFF);
            lens[jid] = (uint32_t)total;
            numbps[jid] = (uint8_t)(32 - __clz((uint32_t)maxMag));
        }
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
        if (lane == 0) { atomicMax(fault, 1); lens[jid] = 0; numbps[jid] = 0; }
        return;
    }
    // ---- assemble: MagSgn | MEL zeros | VLC | SCUP (ht.go:1017-1042) ----
    for (size_t i = lane; i < melLen; i += 64) out[magLen + i] = 0;
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
    if (lane == 0) {
        const size_t scup = melLen + (size_t)vlcLen + 2;
        const size_t total = (size_t)magLen + scup;
        out[total - 2] = (uint8_t)(scup >> 8);
        out[total - 1] = (uint8_t)(scup & 0xFF);
        lens[jid] = (uint32_t)total;
        numbps[jid] = (uint8_t)(32 - __clz((uint32_t)maxMag));
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
    // {3|5<<2|5<<5, 1|1<<5, 2|2<<5, 1|1<<5, 3|1<<2|3<<5, 1|1<<5, 2|2<<5, 1|1<<5}
    const uint32_t packed[8] = {3 | (5 << 2) | (5 << 5), 1 | (1 << 5), 2 | (2 << 5), 1 | (1 << 5),
                                3 | (1 << 2) | (3 << 5), 1 | (1 << 5), 2 | (2 << 5), 1 | (1 << 5)};
    return packed[idx & 7] & 0xFF;   // the Go table is [8]uint8
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

// ---- wave-parallel decoder (fast path) ---------------------------------------------------------
// 1. both byte streams are UNSTUFFED in parallel into LDS bit strings (a byte's width -- 7 or 8 bits --
//    depends only on its predecessor; prefix sum of widths; OR-deposit, because the reference ORs a full
//    byte at a 7-bit advance: ht.go:344-377, 467-500);
// 2. lane 0 walks the VLC bit string (the only truly sequential part: each code word's length comes out
//    of the table lookup of the previous one) and records (rho, rho2, u0, u1) per quad pair;
// 3. all lanes extract magnitudes and signs: a pair's MagSgn position is the prefix sum of
//    popcount(rho)*(u+1) over the pairs before it.
// Falls back to the bit-serial decoder for blocks with more than HT_FAST_MAX_SAMPLES coded samples or when
// a decoded u exceeds 32 (the reference's uint32 bit counter then wraps, ht.go:515-519).
#define HT_DEC_VWORDS (46 * (HT_FAST_MAX_SAMPLES / 8) / 32 + 8)
#define HT_DEC_MWORDS (33 * HT_FAST_MAX_SAMPLES / 32 + 8)

__device__ __forceinline__ uint32_t get_bits32(const uint32_t *buf, uint32_t bitpos) {
    const uint32_t wd = bitpos >> 5, sh = bitpos & 31;
    const uint64_t two = (uint64_t)buf[wd] | ((uint64_t)buf[wd + 1] << 32);
    return (uint32_t)(two >> sh);
}

struct HtDecShared {
    uint32_t vbuf[HT_DEC_VWORDS];
    uint32_t mbuf[HT_DEC_MWORDS];
    uint32_t pair[HT_FAST_MAX_SAMPLES / 8];
    uint16_t tbl0[512], tbl1[512];
};

// returns false if the block must be decoded by the serial path instead
__device__ bool ht_decode_fast(HtDecShared &S, const uint8_t *__restrict__ data, long len, long scup, int w, int h,
                               int32_t *__restrict__ out, int lane) {
    const int quadCols = (w + 3) / 4, P = (quadCols + 1) / 2, R = (h + 3) / 4, N = R * P;
    const long lcup = len;
    for (int i = lane; i < HT_DEC_VWORDS; i += 64) S.vbuf[i] = 0;
    for (int i = lane; i < HT_DEC_MWORDS; i += 64) S.mbuf[i] = 0;
    for (int i = lane; i < 512; i += 64) { S.tbl0[i] = c_vlc_tbl0[i]; S.tbl1[i] = c_vlc_tbl1[i]; }
    __syncthreads();
    // ---- VLC (reverse) bit string: initVLC + revRead (ht.go:276-378) ----
    {
        const uint32_t b0 = data[lcup - 2];
        const uint32_t t0 = b0 >> 4;
        uint32_t off = 4 - ((t0 & 7) >> 2);
        if (lane == 0) atomicOr(&S.vbuf[0], t0);
        const long size = scup - 2;
        const long maxbytes = (long)(46 * N + 7) / 8 + 8;
        const long nb = size < maxbytes ? size : maxbytes;
        for (long k0 = 1; k0 <= nb; k0 += 64) {
            const long k = k0 + lane;
            uint32_t b = 0, width = 0;
            if (k <= nb) {
                b = data[lcup - 2 - k];
                const uint32_t prev = (k == 1) ? (b0 | 0x0F) : data[lcup - 1 - k];
                width = (prev > 0x8F && (b & 0x7F) == 0x7F) ? 7 : 8;
            }
            uint32_t ws = width;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(ws, o); if (lane >= o) ws += a; }
            if (k <= nb && b) or_bits(S.vbuf, off + ws - width, (uint64_t)b);
            off += __shfl(ws, 63);
        }
    }
    // ---- MagSgn (forward) bit string: initMagSgn + frwdRead (ht.go:399-501); exhausted -> all ones ----
    {
        const long segLen = lcup - scup;
        const long maxbytes = (long)HT_DEC_MWORDS * 4 - 16;
        const long nb = segLen < maxbytes ? segLen : maxbytes;
        uint32_t off = 0;
        for (long k0 = 0; k0 < nb; k0 += 64) {
            const long k = k0 + lane;
            uint32_t b = 0, width = 0;
            if (k < nb) {
                b = data[k];
                width = (k > 0 && data[k - 1] == 0xFF) ? 7 : 8;
            }
            uint32_t ws = width;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(ws, o); if (lane >= o) ws += a; }
            if (k < nb && b) or_bits(S.mbuf, off + ws - width, (uint64_t)b);
            off += __shfl(ws, 63);
        }
        __syncthreads();
        if (nb == segLen) {   // everything past the segment reads as ones
            const uint32_t wd0 = off >> 5;
            for (uint32_t i = wd0 + lane; i < HT_DEC_MWORDS; i += 64)
                S.mbuf[i] = (i == wd0) ? (S.mbuf[i] | (0xFFFFFFFFu << (off & 31))) : 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    // ---- sequential VLC walk on lane 0 (ht.go:589-658) ----
    int too_big = 0;
    if (lane == 0) {
        uint32_t cv = 0;
        for (int r = 0; r < R; r++) {
            const int initial = (r == 0);
            const uint16_t *tbl = initial ? S.tbl0 : S.tbl1;
            for (int pi = 0; pi < P; pi++) {
                uint32_t vlcVal = get_bits32(S.vbuf, cv);
                const uint32_t qinf = tbl[vlcVal & 0x7F];
                const uint32_t rho = (qinf >> 4) & 0xF, uOff1 = (qinf >> 3) & 1;
                cv += qinf & 0xF;
                vlcVal = get_bits32(S.vbuf, cv);
                const uint32_t qinf2 = tbl[((rho >> 2) << 7) | (vlcVal & 0x7F)];
                const uint32_t rho2 = (qinf2 >> 4) & 0xF, uOff2 = (qinf2 >> 3) & 1;
                cv += qinf2 & 0xF;
                uint32_t u[2] = {1, 1};
                const uint32_t mode = (uOff1 << 1) | uOff2;
                if (mode > 0) cv += decode_uvlc(get_bits32(S.vbuf, cv), mode, u, initial);
                if (u[0] > 32 || u[1] > 32) too_big = 1;
                S.pair[r * P + pi] = rho | rho2 << 4 | (u[0] & 0x3F) << 8 | (u[1] & 0x3F) << 16;
            }
        }
    }
    too_big = __shfl(too_big, 0);
    __syncthreads();
    if (too_big) return false;
    // ---- parallel MagSgn extraction (ht.go:661-710) ----
    uint32_t mbase = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int it = i0 + lane;
        uint32_t info = 0, nbits = 0, rho = 0, rho2 = 0, u0 = 1, u1 = 1;
        int r = 0, pi = 0;
        if (it < N) {
            info = S.pair[it];
            r = it / P; pi = it - r * P;
            rho = info & 0xF; rho2 = (info >> 4) & 0xF; u0 = (info >> 8) & 0x3F; u1 = (info >> 16) & 0x3F;
            const int xb = pi * 8;
            // samples beyond the block width are skipped even when their rho bit is set (ht.go:661, 689)
            const uint32_t m1 = (xb + 4 <= w) ? 0xFu : ((xb < w) ? ((1u << (w - xb)) - 1) : 0u);
            const uint32_t m2 = (xb + 8 <= w) ? 0xFu : ((xb + 4 < w) ? ((1u << (w - xb - 4)) - 1) : 0u);
            rho &= m1; rho2 &= m2;
            nbits = __popc(rho) * (u0 + 1) + __popc(rho2) * (u1 + 1);
        }
        uint32_t ns = nbits;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(ns, o); if (lane >= o) ns += a; }
        uint32_t mpos = mbase + ns - nbits;
        mbase += __shfl(ns, 63);
        if (it < N) {
            int32_t *orow = out + (size_t)(4 * r) * w + pi * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t rr = (i < 4) ? rho : rho2, emb = (i < 4) ? u0 : u1;
                if (!((rr >> (i & 3)) & 1)) continue;
                const uint32_t magVal = get_bits32(S.mbuf, mpos);
                const uint32_t mag = (magVal & (shl32(1, emb) - 1)) + shl32(1, emb - 1);
                mpos += emb;
                const uint32_t sign = get_bits32(S.mbuf, mpos) & 1;
                mpos += 1;
                orow[i] = sign ? (int32_t)(0u - mag) : (int32_t)mag;
            }
        }
    }
    return true;
}

__global__ __launch_bounds__(64) void ht_decode_kernel(const BlockJob *__restrict__ jobs, int njobs,
                                                       const uint8_t *__restrict__ stream, const uint64_t *__restrict__ offs,
                                                       const uint32_t *__restrict__ lens, int32_t *__restrict__ decoded) {
    __shared__ HtDecShared S;
    const int jid = blockIdx.x;
    if (jid >= njobs) return;
    const int lane = threadIdx.x;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h;
    int32_t *out = decoded + J.out_off;
    const size_t n = (size_t)w * h;
    for (size_t i = lane; i < n; i += 64) out[i] = 0;           // fresh NewHTDecoder: zeroed data
    __syncthreads();
    if ((size_t)((h + 3) / 4) * (size_t)w <= HT_FAST_MAX_SAMPLES) {
        const uint8_t *fdata = stream + offs[jid];
        const long flen = (long)lens[jid];
        if (flen < 2) return;                                   // ht.go:94-100
        const long fscup = (long)fdata[flen - 1] + ((long)(fdata[flen - 2] & 0x0F) << 8);
        if (fscup < 2 || fscup > flen) return;                  // ht.go:104-111
        if (!init_mel_ok(fdata, flen, flen, fscup)) return;     // ht.go:117-122
        if (ht_decode_fast(S, fdata, flen, fscup, w, h, out, lane)) return;
        __syncthreads();
    }
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
                    out[(size_t)y * w + base + i] = sign ? (int32_t)(0u - mag) : (int32_t)mag;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
static bool g_enc_table_ready[16] = {false};

hipError_t launch_ht_encode(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *slots,
                            uint32_t *lens, uint8_t *numbps, int *fault) {
    if (njobs <= 0) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16 && !g_enc_table_ready[dev]) {
        hipLaunchKernelGGL(ht_build_enc_table, dim3(1), dim3(128), 0, s);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        g_enc_table_ready[dev] = true;
    }
    hipLaunchKernelGGL(ht_encode_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, coef, slots, lens, numbps, fault);
    return hipGetLastError();
}

hipError_t launch_ht_decode(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs,
                            const uint32_t *lens, int32_t *decoded) {
    if (njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(ht_decode_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, stream, offs, lens, decoded);
    return hipGetLastError();
}

}  // namespace j2k
