// t1.hip -- the reference's MQ-based Tier-1 block coder (what public jpeg2000.Encode runs).
//
//   T1.SetData + T1.Encode == EncodeFast5   internal/entropy/t1.go:292-304, t1_fast5.go:10-899
//   mqByteOutLocal                          internal/entropy/t1_fast.go:11-34
//   T1.Decode + decode*Pass, MQDecoder      internal/entropy/t1.go:1261-1410, mqc.go:352-497
//   MQ state table / contexts / LUT rules   mqc.go:21-166, t1_luts.go:32-231
//
// Reference behaviour reproduced (it is NOT ISO 15444-1 EBCOT): every bit-plane, including the
// first, runs SigProp (raster order) -> MagRef (raster order) -> Cleanup (4-row stripes, column
// by column, run-length mode); all contexts start at state 0 except UNI = 92; the sign-context
// LUT maps (hc==1, vc not in {0,1}) to SC1 and hc==2 to SC3; output = MQ bytes without the
// leading sentinel, trailing 0xFF dropped, nil for an all-zero block.
//
// Kernel shape (round 1): one code-block per wavefront.  The wave loads the block, strips the
// signs, finds the bit-plane count and builds the LUTs cooperatively; the context-adaptive
// state machine itself is sequential by construction (every decision's context depends on the
// significance state left by the previous decision, and the MQ interval on every previous
// symbol) and runs on lane 0 with flags + magnitudes + tables in LDS (or in a global
// workspace for blocks too large for LDS).
#include "j2k_internal.h"

namespace j2k {

enum { T1Sig = 1, T1Visit = 2, T1Refine = 4, T1SignNeg = 8, T1SigN = 16, T1SigS = 32, T1SigE = 64, T1SigW = 128 };  // t1.go:74-91
enum { CtxSC0 = 9, CtxMag0 = 14, CtxMag1 = 15, CtxMag2 = 16, CtxRL = 17, CtxUni = 18, NumContexts = 19 };         // mqc.go:135-166

// ISO/IEC 15444-1 Table C.2 (Qe, NMPS, NLPS, SWITCH); expanded to the reference's 94-entry
// MPS-interleaved form (mqc.go:21-116) by rule: state 2i+m, nmps = 2*NMPS+m, nlps = 2*NLPS + (SWITCH ? 1-m : m).
#define J2K_ISO_QE {0x5601, 0x3401, 0x1801, 0x0AC1, 0x0521, 0x0221, 0x5601, 0x5401, 0x4801, 0x3801, 0x3001, 0x2401, 0x1C01, 0x1601, 0x5601, 0x5401, \
    0x5101, 0x4801, 0x3801, 0x3401, 0x3001, 0x2801, 0x2401, 0x2201, 0x1C01, 0x1801, 0x1601, 0x1401, 0x1201, 0x1101, 0x0AC1, 0x09C1, \
    0x08A1, 0x0521, 0x0441, 0x02A1, 0x0221, 0x0141, 0x0111, 0x0085, 0x0049, 0x0025, 0x0015, 0x0009, 0x0005, 0x0001, 0x5601}
#define J2K_ISO_NMPS {1, 2, 3, 4, 5, 38, 7, 8, 9, 10, 11, 12, 13, 29, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, \
    25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 45, 46}
#define J2K_ISO_NLPS {1, 6, 9, 12, 29, 33, 6, 14, 14, 14, 17, 18, 20, 21, 14, 14, 15, 16, 17, 18, 19, 19, 20, 21, \
    22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 46}
#define J2K_ISO_SWITCH {1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, \
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
__device__ __constant__ uint16_t c_iso_qe[47] = J2K_ISO_QE;
__device__ __constant__ uint8_t c_iso_nmps[47] = J2K_ISO_NMPS;
__device__ __constant__ uint8_t c_iso_nlps[47] = J2K_ISO_NLPS;
__device__ __constant__ uint8_t c_iso_switch[47] = J2K_ISO_SWITCH;
// the same 94 entries (qe | nmps << 16 | nlps << 24) as a constant table in memory, for the decoders whose LDS budget has
// no room for it (t1_decode64_kernel): read only when a context changes state
struct Mq94 {
    uint32_t v[94];
    constexpr Mq94() : v{} {
        constexpr uint16_t qe[47] = J2K_ISO_QE;
        constexpr uint8_t nmps[47] = J2K_ISO_NMPS, nlps[47] = J2K_ISO_NLPS, sw[47] = J2K_ISO_SWITCH;
        for (int s = 0; s < 94; s++) {
            const int i = s >> 1, m = s & 1;
            v[s] = (uint32_t)qe[i] | (uint32_t)(2 * nmps[i] + m) << 16 | (uint32_t)(2 * nlps[i] + (sw[i] ? 1 - m : m)) << 24;
        }
    }
};
__device__ __constant__ Mq94 c_mq94 = Mq94();

// LDS-resident tables shared by encoder and decoder
struct T1Tables {
    uint32_t mq[96];      // qe | nmps << 16 | nlps << 24
    uint8_t zc[256];      // lutZCCtx for this block's band (t1_luts.go:34-110)
    uint8_t sc[256];      // lutSignCtx | lutSignPred << 3 (t1_luts.go:153-230)
    uint8_t ctx[32];      // MQ context states (encoder)
    uint32_t ent[32];     // decoder: the table entry of each context's current state
    uint8_t mrctx[128];   // decoder: per-column context lists of the wave-level passes (two of 64 bytes)
};

__device__ void build_tables(T1Tables &T, int band, int lane) {
    for (int s = lane; s < 94; s += 64) {
        const int i = s >> 1, m = s & 1;
        const uint32_t nm = 2 * c_iso_nmps[i] + m;
        const uint32_t nl = 2 * c_iso_nlps[i] + (c_iso_switch[i] ? 1 - m : m);
        T.mq[s] = (uint32_t)c_iso_qe[i] | nm << 16 | nl << 24;
    }
    for (int p = lane; p < 256; p += 64) {
        const int w = p & 1, e = (p >> 1) & 1, n = (p >> 2) & 1, s = (p >> 3) & 1;
        const int d = ((p >> 4) & 1) + ((p >> 5) & 1) + ((p >> 6) & 1) + ((p >> 7) & 1);
        int hh = w + e, v = n + s, ctx;
        if (band == 1) { const int t = hh; hh = v; v = t; }   // HL: swap h and v
        if (band == 3) {
            const int hv = hh + v;
            if (hv >= 3) ctx = 8;
            else if (hv == 2) ctx = d >= 2 ? 7 : (d >= 1 ? 6 : 5);
            else if (hv == 1) ctx = d >= 2 ? 4 : 3;
            else ctx = d >= 2 ? 2 : (d >= 1 ? 1 : 0);
        } else {
            if (hh == 2) ctx = 8;
            else if (hh == 1) ctx = v >= 1 ? 7 : (d >= 1 ? 6 : 5);
            else if (v == 2) ctx = 4;
            else if (v == 1) ctx = d >= 1 ? 3 : 2;
            else ctx = d >= 2 ? 1 : 0;
        }
        T.zc[p] = (uint8_t)ctx;
        // sign LUT: index bits 0=W_sig 1=W_chi 2=E_sig 3=E_chi 4=N_sig 5=N_chi 6=S_sig 7=S_chi
        int hc = 0, vc = 0, pred = 0, sctx = 0;
        if (p & 1) hc += (p & 2) ? -1 : 1;
        if (p & 4) hc += (p & 8) ? -1 : 1;
        if (p & 16) vc += (p & 32) ? -1 : 1;
        if (p & 64) vc += (p & 128) ? -1 : 1;
        if (hc < 0) { pred = 1; hc = -hc; }
        if (hc == 0 && vc < 0) { pred = 1; vc = -vc; }
        if (hc == 1) sctx = vc == 1 ? 4 : (vc == 0 ? 2 : 1);
        else if (hc == 0) sctx = vc == 1 ? 1 : 0;
        else if (hc == 2) sctx = 3;
        T.sc[p] = (uint8_t)(sctx | pred << 3);
    }
    if (lane < NumContexts) T.ctx[lane] = (lane == CtxUni) ? 92 : 0;
}
// after build_tables + a barrier: the decoder's per-context entries (contexts start at state 0, UNI at 92)
__device__ void init_dec_contexts(T1Tables &T, int lane) {
    if (lane < NumContexts) T.ent[lane] = T.mq[lane == CtxUni ? 92 : 0];
}

// ---- MQ encoder with the "current byte" (buf[bp]) held in a register ---------------------
struct MqEnc {
    uint32_t A, C, CT;
    uint32_t cur;      // value of buf[bp]
    long bp;           // index of the byte in `cur`; byte k>=1 goes to out[k-1]
    uint8_t *out;
    long cap;
    int overflow;
};

__device__ __forceinline__ void mq_advance(MqEnc &e, uint32_t nv) {
    if (e.bp >= 1) {
        if (e.bp - 1 < e.cap) e.out[e.bp - 1] = (uint8_t)e.cur; else e.overflow = 1;
    }
    e.bp++;
    e.cur = nv & 0xFF;
}

__device__ __forceinline__ void mq_byte_out(MqEnc &e) {   // t1_fast.go:11-34
    if (e.cur == 0xFF) { mq_advance(e, e.C >> 20); e.C &= 0xFFFFF; e.CT = 7; return; }
    if ((e.C & 0x8000000) == 0) { mq_advance(e, e.C >> 19); e.C &= 0x7FFFF; e.CT = 8; return; }
    e.cur = (e.cur + 1) & 0xFF;
    if (e.cur == 0xFF) { e.C &= 0x7FFFFFF; mq_advance(e, e.C >> 20); e.C &= 0xFFFFF; e.CT = 7; return; }
    mq_advance(e, e.C >> 19); e.C &= 0x7FFFF; e.CT = 8;
}

__device__ __forceinline__ void mq_encode(MqEnc &e, T1Tables &T, int ctx, int d) {   // mqc.go:224-255
    const uint32_t st = T.ctx[ctx];
    const uint32_t ent = T.mq[st];
    const uint32_t qe = ent & 0xFFFF;
    e.A -= qe;
    if ((uint32_t)d == (st & 1)) {
        if (e.A & 0x8000) { e.C += qe; return; }
        if (e.A < qe) e.A = qe; else e.C += qe;
        T.ctx[ctx] = (uint8_t)((ent >> 16) & 0xFF);
    } else {
        if (e.A < qe) e.C += qe; else e.A = qe;
        T.ctx[ctx] = (uint8_t)(ent >> 24);
    }
    do {   // renorm (mqc.go:258-267)
        e.A <<= 1; e.C <<= 1; e.CT--;
        if (e.CT == 0) mq_byte_out(e);
    } while ((e.A & 0x8000) == 0);
}

__device__ __forceinline__ int zc_packed(const uint8_t *f, int stride) {   // t1_fast5.go:118-125
    return (f[-1] & T1Sig) | ((f[1] & T1Sig) << 1) | ((f[-stride] & T1Sig) << 2) | ((f[stride] & T1Sig) << 3) |
           ((f[-stride - 1] & T1Sig) << 4) | ((f[-stride + 1] & T1Sig) << 5) | ((f[stride - 1] & T1Sig) << 6) |
           ((f[stride + 1] & T1Sig) << 7);
}
__device__ __forceinline__ int sc_index(uint32_t fW, uint32_t fE, uint32_t fN, uint32_t fS) {   // t1_fast5.go:171-181
    return (fW & T1Sig) | (((fW & T1SignNeg) >> 3) << 1) | ((fE & T1Sig) << 2) | (((fE & T1SignNeg) >> 3) << 3) |
           ((fN & T1Sig) << 4) | (((fN & T1SignNeg) >> 3) << 5) | ((fS & T1Sig) << 6) | (((fS & T1SignNeg) >> 3) << 7);
}
__device__ __forceinline__ void set_significant(uint8_t *f, int x, int y, int w, int h, int stride) {   // t1_fast5.go:233-245
    *f |= T1Sig;
    if (y > 0) f[-stride] |= T1SigS;
    if (y < h - 1) f[stride] |= T1SigN;
    if (x > 0) f[-1] |= T1SigE;
    if (x < w - 1) f[1] |= T1SigW;
}
// Decoder only: the directional bits are not read there (contexts come from the neighbours' T1Sig), so bit 16 means
// "one of my 8 neighbours is significant", set when the neighbour turns significant.  The scans of T1.Decode then test
// one flag byte per sample instead of nine (any_sig8 below); the flag array has a border, so no bounds checks.
enum { T1HasNb = 16 };
__device__ __forceinline__ void set_significant_dec(uint8_t *f, int stride) {
    *f |= T1Sig;
    f[-stride - 1] |= T1HasNb; f[-stride] |= T1HasNb; f[-stride + 1] |= T1HasNb;
    f[-1] |= T1HasNb; f[1] |= T1HasNb;
    f[stride - 1] |= T1HasNb; f[stride] |= T1HasNb; f[stride + 1] |= T1HasNb;
}
__device__ __forceinline__ bool any_sig8(const uint8_t *f, int stride) {
    return ((f[-1] | f[1] | f[-stride] | f[stride] | f[-stride - 1] | f[-stride + 1] | f[stride - 1] | f[stride + 1]) & T1Sig) != 0;
}

// Decoder flag layout: rows of T1D_STRIDE(w) bytes (a multiple of 4), sample (x, y) at (y + 1) * stride + T1D_XO + x: rows
// start on a 4-byte boundary so the SigProp / MagRef scans test four samples with one load; border samples on every side
// (x = -1 at offset 3; x = w in the row's padding or, for w % 4 == 0, at offset 0 of the next row, which no row uses).
#define T1D_XO 4
#define T1D_STRIDE(w) (((w) + T1D_XO + 3) & ~3)
size_t t1_flag_bytes(int w, int h) { return ((size_t)T1D_STRIDE(w) * (h + 2) + 4 + 15) & ~size_t(15); }
size_t t1_work_bytes(int w, int h) {
    const size_t flags = ((size_t)(w + 2) * (h + 2) + 15) & ~size_t(15);
    return flags + (size_t)w * h * 4;
}

// bytes of the reference's mqBuf for a block of n samples (t1_fast5.go:47-56: 2n + 1024, never fewer than 16384) = the slot
// size j2k_block_bound gives the block; running past it is the Go index panic
__host__ __device__ inline size_t t1_mqbuf_bytes(size_t n) { return n * 2 + 1024 < 16384 ? (size_t)16384 : n * 2 + 1024; }
#define T1_LDS_LIMIT (60 * 1024)

// LDSW: the flag / magnitude workspace is in LDS (every block fits) or in `work`; a template parameter so that the
// accesses compile to ds_* / global_* instead of flat_* (a pointer chosen at run time is a flat pointer).
#define T1B_OVERFLOW 0xFFFFFFFFu
template <bool LDSW>
__global__ __launch_bounds__(64) void t1_encode_kernel(const BlockJob *__restrict__ jobs, int njobs, const int32_t *__restrict__ coef,
                                                       uint8_t *__restrict__ slots, uint32_t *__restrict__ lens,
                                                       uint8_t *__restrict__ numbps, uint8_t *__restrict__ work, size_t work_per_job,
                                                       int lds_work_bytes, int *__restrict__ fault, int skip_small,
                                                       const uint32_t *__restrict__ marked) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int jid = blockIdx.x;                  // (vector loop control as in t1_decode64_kernel is slower here: 87 -> 118 ms)
    if (jid >= njobs) return;
    const int lane = threadIdx.x;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h, stride = w + 2;
    // skip_small = 64 / 256: t1_encode64_kernel (and t1_encode_big_kernel) take these -- unless the two-kernel form of t1_big.inc marked
    // the block as one whose symbols did not fit its list
    if (skip_small && w <= skip_small && h <= skip_small && !(marked && marked[jid] == T1B_OVERFLOW)) return;
    const size_t n = (size_t)w * h;
    T1Tables &T = *reinterpret_cast<T1Tables *>(smem);
    const size_t flag_bytes = ((size_t)(w + 2) * (h + 2) + 15) & ~size_t(15);
    uint8_t *wk = LDSW ? smem + ((sizeof(T1Tables) + 15) & ~size_t(15)) : work + (size_t)jid * work_per_job;
    uint8_t *flags = wk;
    int32_t *data = reinterpret_cast<int32_t *>(wk + flag_bytes);

    build_tables(T, J.band, lane);
    for (size_t i = lane; i < (size_t)(w + 2) * (h + 2); i += 64) flags[i] = 0;
    __syncthreads();
    // SetData (t1.go:292-304) + max magnitude (t1_fast5.go:13-28)
    const int32_t *src = coef + J.src_off;
    int maxVal = 0;
    for (int y = 0; y < h; y++)
        for (int x = lane; x < w; x += 64) {
            int v = src[(size_t)y * J.stride + x];
            if (v < 0) {
                v = (int)(0u - (uint32_t)v);
                flags[(size_t)(y + 1) * stride + x + 1] = T1SignNeg;
            }
            data[(size_t)y * w + x] = v;
            maxVal = max(maxVal, v);
        }
    for (int o = 32; o > 0; o >>= 1) maxVal = max(maxVal, __shfl_xor(maxVal, o));
    __syncthreads();
    if (maxVal == 0) {
        if (lane == 0) { lens[jid] = 0; numbps[jid] = 0; }
        return;
    }
    if (lane != 0) return;
    const int numBPS = 32 - __clz((uint32_t)maxVal);
    uint8_t *out = slots + J.out_off;
    MqEnc e{0x8000, 0, 12, 0, 0, out, (long)t1_mqbuf_bytes(n), 0};

    for (int bp = numBPS - 1; bp >= 0; bp--) {
        const int32_t bit = (int32_t)(1u << bp);
        // ---- significance propagation (t1_fast5.go:72-249) ----
        for (int y = 0; y < h; y++) {
            uint8_t *frow = flags + (size_t)(y + 1) * stride + 1;
            const int32_t *drow = data + (size_t)y * w;
            for (int x = 0; x < w; x++) {
                uint8_t *f = frow + x;
                const uint32_t fv = *f;
                if (fv & T1Sig) continue;
                uint32_t fW = 0, fE = 0, fN = 0, fS = 0, fNW, fNE, fSW, fSE;
                if ((fv & (T1SigN | T1SigS | T1SigE | T1SigW)) == 0) {
                    fNW = f[-stride - 1]; fNE = f[-stride + 1]; fSW = f[stride - 1]; fSE = f[stride + 1];
                    if (((fNW | fNE | fSW | fSE) & T1Sig) == 0) continue;
                } else {
                    fW = f[-1]; fE = f[1]; fN = f[-stride]; fS = f[stride];
                    fNW = f[-stride - 1]; fNE = f[-stride + 1]; fSW = f[stride - 1]; fSE = f[stride + 1];
                }
                const int sig = (drow[x] >> bp) & 1;
                const int packed = (fW & T1Sig) | ((fE & T1Sig) << 1) | ((fN & T1Sig) << 2) | ((fS & T1Sig) << 3) |
                                   ((fNW & T1Sig) << 4) | ((fNE & T1Sig) << 5) | ((fSW & T1Sig) << 6) | ((fSE & T1Sig) << 7);
                mq_encode(e, T, T.zc[packed], sig);
                if (sig) {
                    const uint32_t sc = T.sc[sc_index(fW, fE, fN, fS)];
                    mq_encode(e, T, CtxSC0 + (sc & 7), ((fv & T1SignNeg) ? 1 : 0) ^ (sc >> 3));
                    set_significant(f, x, y, w, h, stride);
                }
                *f |= T1Visit;
            }
        }
        // ---- magnitude refinement (t1_fast5.go:252-335) ----
        for (int y = 0; y < h; y++) {
            uint8_t *frow = flags + (size_t)(y + 1) * stride + 1;
            const int32_t *drow = data + (size_t)y * w;
            for (int x = 0; x < w; x++) {
                uint8_t *f = frow + x;
                const uint32_t fv = *f;
                if ((fv & T1Sig) == 0 || (fv & T1Visit) != 0) continue;
                int ctx;
                if ((fv & T1Refine) == 0) ctx = any_sig8(f, stride) ? CtxMag1 : CtxMag0;
                else ctx = CtxMag2;
                mq_encode(e, T, ctx, (drow[x] & bit) ? 1 : 0);
                *f = (uint8_t)(fv | T1Refine);
            }
        }
        // ---- cleanup (t1_fast5.go:338-876) ----
        for (int y = 0; y < h; y += 4) {
            for (int x = 0; x < w; x++) {
                bool canRL = (y + 4 <= h);
                if (canRL) {
                    for (int yy = 0; yy < 4; yy++) {
                        const uint8_t *f = flags + (size_t)(y + yy + 1) * stride + x + 1;
                        if ((*f & (T1Sig | T1Visit)) || any_sig8(f, stride)) { canRL = false; break; }
                    }
                }
                if (canRL) {
                    int firstSig = -1;
                    for (int i = 0; i < 4; i++)
                        if (data[(size_t)(y + i) * w + x] & bit) { firstSig = i; break; }
                    mq_encode(e, T, CtxRL, firstSig >= 0 ? 1 : 0);
                    if (firstSig < 0) continue;
                    mq_encode(e, T, CtxUni, (firstSig >> 1) & 1);
                    mq_encode(e, T, CtxUni, firstSig & 1);
                    {
                        const int yy = y + firstSig;
                        uint8_t *f = flags + (size_t)(yy + 1) * stride + x + 1;
                        const uint32_t sc = T.sc[sc_index(f[-1], f[1], f[-stride], f[stride])];
                        mq_encode(e, T, CtxSC0 + (sc & 7), ((*f & T1SignNeg) ? 1 : 0) ^ (sc >> 3));
                        set_significant(f, x, yy, w, h, stride);
                    }
                    for (int i = firstSig + 1; i < 4; i++) {
                        const int yy = y + i;
                        uint8_t *f = flags + (size_t)(yy + 1) * stride + x + 1;
                        const int sig = (data[(size_t)yy * w + x] & bit) ? 1 : 0;
                        const uint32_t fW = f[-1], fE = f[1], fN = f[-stride], fS = f[stride];
                        mq_encode(e, T, T.zc[zc_packed(f, stride)], sig);
                        if (sig) {
                            const uint32_t sc = T.sc[sc_index(fW, fE, fN, fS)];
                            mq_encode(e, T, CtxSC0 + (sc & 7), ((*f & T1SignNeg) ? 1 : 0) ^ (sc >> 3));
                            set_significant(f, x, yy, w, h, stride);
                        }
                    }
                    continue;
                }
                const int yEnd = min(y + 4, h);
                for (int yy = y; yy < yEnd; yy++) {
                    uint8_t *f = flags + (size_t)(yy + 1) * stride + x + 1;
                    const uint32_t fv = *f;
                    if (fv & T1Visit) { *f = (uint8_t)(fv & ~T1Visit); continue; }
                    if (fv & T1Sig) continue;
                    const int sig = (data[(size_t)yy * w + x] & bit) ? 1 : 0;
                    const uint32_t fW = f[-1], fE = f[1], fN = f[-stride], fS = f[stride];
                    mq_encode(e, T, T.zc[zc_packed(f, stride)], sig);
                    if (sig) {
                        const uint32_t sc = T.sc[sc_index(fW, fE, fN, fS)];
                        mq_encode(e, T, CtxSC0 + (sc & 7), ((fv & T1SignNeg) ? 1 : 0) ^ (sc >> 3));
                        set_significant(f, x, yy, w, h, stride);
                    }
                }
            }
        }
    }
    // ---- flush (t1_fast5.go:878-898) ----
    const uint32_t tempC = e.C + e.A;
    e.C |= 0xFFFF;
    if (e.C >= tempC) e.C -= 0x8000;
    e.C <<= e.CT; mq_byte_out(e);
    e.C <<= e.CT; mq_byte_out(e);
    // bytes buf[1..bp]; the last one is still in `cur`
    long end = e.bp + 1;                       // endPos
    if (e.cur == 0xFF) end--;                  // drop a trailing 0xFF
    else if (e.bp >= 1) { if (e.bp - 1 < e.cap) out[e.bp - 1] = (uint8_t)e.cur; else e.overflow = 1; }
    if (e.overflow) atomicMax(fault, 2);
    lens[jid] = end > 1 ? (uint32_t)(end - 1) : 0;
    numbps[jid] = (uint8_t)numBPS;
}

// ================================================================================================
// Encoder for blocks up to 64 x 64: wave-parallel context formation, MQ coding from a symbol list
// ================================================================================================
// The only truly sequential part of EncodeFast5 is the MQ coder (every symbol moves A, C and one context's state).
// Everything that decides WHICH (context, decision) pairs are coded, and in which order, is a function of
// significance bit masks:
//   * one wavefront per block; lane y keeps row y's masks in registers: S (significant), NEG (sign), REF (refined);
//     the bit planes of the magnitudes are 64-bit row masks in LDS (built with ballots);
//   * SigProp membership (raster order) is the recurrence  new[x] = G[x] | (Pr[x] & new[x-1])  along a row, with
//     Pr = ~S & B (not yet significant, bit set) and G = Pr & (significant neighbour among N-row-updated, S-row-old,
//     E-old, W-old): a carry chain, solved for a whole row with ONE 64-bit add  (cin = (Pr + G) ^ Pr ^ G).  Rows
//     depend on the row above, so the 64 lanes iterate the row update until nothing changes (<= h steps);
//   * MagRef members are the samples significant before this plane; Cleanup members are the samples neither
//     significant nor visited, and every member whose bit is set becomes significant -- so the significance state a
//     neighbour had "at visit time" is just a choice between two masks, by coding order;
//   * symbols are then produced with lanes = columns (row masks broadcast with v_readlane), compacted in coding
//     order with popcounts / a wave scan into an LDS list, and lane 0 runs the MQ coder over the list.
// Output is byte-identical to the serial kernel above (which stays for larger blocks).
#define T1F_SYM_CAP 3072    /* symbol list; drained by the MQ coder whenever the next row / stripe might not fit */
#define T1F_SYM_STEP 640    /* most symbols one emission step adds: a cleanup stripe = 10 per column x 64 columns */
struct T1Fast {
    T1Tables T;
    uint8_t zc2[256];          // ZC context by this kernel's index: NW | N<<1 | NE<<2 | SW<<3 | S<<4 | SE<<5 | W<<6 | E<<7
    uint32_t ctxent[32];       // per context: the MQ table entry of its current state
    alignas(16) uint8_t sym[T1F_SYM_CAP];  // ctx | decision << 5
};
#define T1F_NULL_SYM 19u     /* context 19 (the first unused one) is a no-op for the lane-parallel MQ kernel (Qe = 0): pads a chunk to 16 symbols */
#define T1F_NCTX 20          /* entries per lane in that kernel's LDS: the 19 contexts and the no-op */
#define T1F_SKIPPED 0xFFFFFFFFu   /* nsyms[] mark: block over the symbol-plane budget, left to the one-kernel path */

__device__ __forceinline__ uint64_t rl64(uint64_t v, int r) {      // row mask of lane r, broadcast (r wave-uniform)
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, r), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), r);
    return (uint64_t)hi << 32 | lo;
}
__device__ __forceinline__ uint32_t bit_at(uint64_t m, int x) { return (uint32_t)(m >> x) & 1u; }
// bits x-1, x, x+1 of m as bits 0, 1, 2 (columns outside 0..63 read as 0)
__device__ __forceinline__ uint32_t win3(uint64_t m, int x) { return (uint32_t)((x == 0) ? (m << 1) : (m >> (x - 1))) & 7u; }
__device__ __forceinline__ uint64_t spread3(uint64_t m) { return m | (m << 1) | (m >> 1); }

// MQ coder over a symbol list (mqc.go:224-267), arranged for the serial lane: the per-context state is kept as the
// state's TABLE ENTRY (qe | nmps << 16 | nlps << 24; the MPS is the parity of nmps), so the common case -- MPS coded,
// interval still >= 0x8000 -- costs one LDS read and a handful of ALU ops; the table is consulted only when a context
// changes state; renormalisation shifts by count-leading-zeros instead of bit by bit (byte-out when CT reaches 0).
__device__ __forceinline__ void mq_run(MqEnc &e, const uint32_t *mqtab, uint32_t *ctxent, const uint8_t *sym, uint32_t n) {
#ifdef J2K_T1_NOMQ
    e.C += n; return;     // timing experiment: context formation only
#endif
    // Four symbols per trip, their contexts' table entries requested together with the NEXT four symbols: the chain of a lone
    // lane is latency, and an LDS round trip per symbol (entry after symbol after entry ...) was most of it -- 150 ns per symbol
    // for the 650 k symbols of a 256 x 256 block of noise.  An entry fetched early is stale only if an earlier symbol of the
    // same group changed the same context's state (a renormalisation): patched where it happens.
    uint32_t A = e.A, C = e.C, CT = e.CT;
    uint32_t four = n ? *reinterpret_cast<const uint32_t *>(sym) : 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 4) {
        const uint32_t cur = four;
        if (i0 + 4 < n) four = *reinterpret_cast<const uint32_t *>(sym + i0 + 4);
        const uint32_t m = min(n - i0, 4u);
        uint32_t cx[4], ent[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { cx[k] = (cur >> (8 * k)) & 31; ent[k] = ctxent[cx[k]]; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k >= m) break;
            const uint32_t ctx = cx[k], d = (cur >> (8 * k + 5)) & 1;
            const uint32_t en = ent[k];
            const uint32_t qe = en & 0xFFFF;
            const uint32_t A1 = A - qe;
            const bool isM = d == ((en >> 16) & 1);
            if (isM && (A1 & 0x8000)) { A = A1; C += qe; continue; }      // the common case; everything below is select-only:
            // a divergent branch costs exec-mask bookkeeping on the CU's one scalar unit (measured: as many SALU as VALU
            // instructions per symbol when every `if` of mqc.go:224-255 is a branch)
            const bool lt = A1 < qe;
            C += (isM != lt) ? qe : 0u;                                   // MPS: C += qe unless A < qe; LPS: only if A < qe
            A = (isM == lt) ? qe : A1;
            const uint32_t ne = mqtab[isM ? ((en >> 16) & 0xFF) : (en >> 24)];
            ctxent[ctx] = ne;
#pragma unroll
            for (int j = k + 1; j < 4; j++) if (cx[j] == ctx) ent[j] = ne;   // the entries fetched ahead for the same context
            uint32_t shift = (uint32_t)__clz(A) - 16;                     // A < 0x8000 here: 1..15 doublings
            A <<= shift;
            if (shift < CT) { C <<= shift; CT -= shift; continue; }
            do {                                                          // a byte leaves the register
                const uint32_t sft = min(shift, CT);
                C <<= sft; CT -= sft; shift -= sft;
                if (CT == 0) { e.C = C; mq_byte_out(e); C = e.C; CT = e.CT; }
            } while (shift);
        }
    }
    e.A = A; e.C = C; e.CT = CT;
}

// SPLIT = false: contexts + MQ coder in this kernel (lane 0 codes each chunk of symbols).  only_skipped != null: just the
//                 blocks the split path marked T1F_SKIPPED.
// SPLIT = true : the symbols go to `gsym` (job j at j * sym_stride, chunks padded to 16 with no-op symbols, count in
//                nsyms[j]) for t1_mq_lanes_kernel; blocks with more than `plane_budget` bit planes are marked T1F_SKIPPED.
template <bool SPLIT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void t1_encode64_kernel(const BlockJob *__restrict__ jobs, int njobs, const int32_t *__restrict__ coef,
                                                         uint8_t *__restrict__ slots, uint32_t *__restrict__ lens,
                                                         uint8_t *__restrict__ numbps, int *__restrict__ fault,
                                                         uint8_t *__restrict__ gsym, size_t sym_stride, uint32_t *__restrict__ nsyms,
                                                         int plane_budget, const uint32_t *__restrict__ only_skipped) {
    __shared__ T1Fast F;
    const int jid = blockIdx.x;
    if (jid >= njobs) return;
    if (!SPLIT && only_skipped && only_skipped[jid] != T1F_SKIPPED) return;
    const int lane = threadIdx.x;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h;
    if (w > 64 || h > 64) {                       // the serial kernel takes these
        if (SPLIT && lane == 0) nsyms[jid] = 0;
        return;
    }
    const size_t n = (size_t)w * h;
    build_tables(F.T, J.band, lane);
    __syncthreads();
    for (int p = lane; p < 256; p += 64) {
        const int ref = ((p >> 6) & 1) | ((p >> 7) & 1) << 1 | ((p >> 1) & 1) << 2 | ((p >> 4) & 1) << 3 |
                        (p & 1) << 4 | ((p >> 2) & 1) << 5 | ((p >> 3) & 1) << 6 | ((p >> 5) & 1) << 7;
        F.zc2[p] = F.T.zc[ref];
    }
    if (lane < NumContexts) F.ctxent[lane] = F.T.mq[lane == CtxUni ? 92 : 0];
    // ---- SetData (t1.go:292-304) + bit-plane count (t1_fast5.go:13-28): lanes = columns ----
    const int32_t *src = coef + J.src_off;
    const bool colok = lane < w;
    uint32_t maxVal = 0;
    for (int y0 = 0; y0 < h; y0 += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[(size_t)min(y0 + k, h - 1) * J.stride + (colok ? lane : 0)];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (y0 + k < h && colok) maxVal = max(maxVal, (uint32_t)(v[k] < 0 ? 0u - (uint32_t)v[k] : (uint32_t)v[k]));
    }
    // the reference compares int32 magnitudes: |MinInt32| stays negative and never wins the max (t1_fast5.go:16-22)
    {
        int m = (int)maxVal < 0 ? 0 : (int)maxVal;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        maxVal = (uint32_t)m;
    }
    if (maxVal == 0) {
        if (lane == 0) { lens[jid] = 0; numbps[jid] = 0; if (SPLIT) nsyms[jid] = 0; }
        return;
    }
    const int numBPS = 32 - __clz(maxVal);
    if (SPLIT && numBPS > plane_budget) {
        if (lane == 0) nsyms[jid] = T1F_SKIPPED;
        return;
    }
    uint8_t *gdst = SPLIT ? gsym + (size_t)jid * sym_stride : nullptr;
    uint32_t gtotal = 0;
    bool govf = false;
    // row masks of one bit of the samples (bit 31 = sign): loads with lanes = columns, a ballot per row, kept by lane = row.
    // Re-read from L2 for every plane (16 KB per block) rather than parked in LDS: LDS is what limits how many of the
    // serial MQ chains a SIMD can interleave.
    auto row_masks = [&](int bit) -> uint64_t {
        uint64_t mine = 0;
        for (int y0 = 0; y0 < h; y0 += 8) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[(size_t)min(y0 + k, h - 1) * J.stride + (colok ? lane : 0)];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t t = bit == 31 ? (uint32_t)v[k] : (v[k] < 0 ? 0u - (uint32_t)v[k] : (uint32_t)v[k]);
                const uint64_t b = __ballot(colok && ((t >> bit) & 1));
                if (lane == y0 + k) mine = b;
            }
        }
        return lane < h ? mine : 0ull;
    };
    const uint64_t NEG = row_masks(31);             // lanes = rows from here on for the register masks

    const uint64_t wmask = (w == 64) ? ~0ull : ((1ull << w) - 1);
    const uint64_t rowok = (lane < h) ? wmask : 0ull;
    const uint64_t lt = (1ull << lane) - 1;         // columns before this lane's column
    uint64_t S = 0, REF = 0;
    uint8_t *out = slots + J.out_off;
    MqEnc e{0x8000, 0, 12, 0, 0, out, (long)t1_mqbuf_bytes(n), 0};
    const int x = lane;
    uint32_t nsym = 0;
#define T1F_DRAIN()                                           \
    do {                                                      \
        if (SPLIT) {                                          \
            const uint32_t npad = (nsym + 15u) & ~15u;        \
            if (nsym + lane < npad) F.sym[nsym + lane] = (uint8_t)T1F_NULL_SYM; \
            __syncthreads();                                  \
            if ((size_t)gtotal + npad > sym_stride) govf = true; \
            else {                                            \
                for (uint32_t i = (uint32_t)lane * 16; i < npad; i += 1024) \
                    *reinterpret_cast<uint4 *>(gdst + gtotal + i) = *reinterpret_cast<const uint4 *>(F.sym + i); \
                gtotal += npad;                               \
            }                                                 \
        } else {                                              \
            __syncthreads();                                  \
            if (lane == 0) mq_run(e, F.T.mq, F.ctxent, F.sym, nsym); \
        }                                                     \
        nsym = 0;                                             \
        __syncthreads();                                      \
    } while (0)
#define T1F_ROOM() do { if (nsym + T1F_SYM_STEP > T1F_SYM_CAP) T1F_DRAIN(); } while (0)

    for (int bp = numBPS - 1; bp >= 0; bp--) {
        const uint64_t B = row_masks(bp);
        // =========== significance propagation (t1_fast5.go:72-249) ===========
        uint64_t newsig = 0, vis = 0;
        if (__any(S != 0)) {
            uint64_t Dn = __shfl_down(S, 1);
            if (lane == 63) Dn = 0;
            const uint64_t stat = spread3(Dn) | (S << 1) | (S >> 1);      // S-row old, E/W old
            const uint64_t P = ~S & rowok, Pr = P & B;
            uint64_t base = stat;
            for (int it = 0; it <= h; it++) {
                uint64_t Up = __shfl_up(S | newsig, 1);
                if (lane == 0) Up = 0;
                base = stat | spread3(Up);
                const uint64_t G = Pr & base;
                const uint64_t cin = (Pr + G) ^ Pr ^ G;
                const uint64_t ns = G | (Pr & cin);
                const bool changed = ns != newsig;
                newsig = ns;
                if (!__any(changed)) break;
            }
            vis = P & (base | (newsig << 1));
        }
        const uint64_t Snew = S | newsig;
        {
            uint64_t rows = __ballot(vis != 0);
            while (rows) {
                const int r = __ffsll((long long)rows) - 1;
                rows &= rows - 1;
                T1F_ROOM();
                const uint64_t Vis = rl64(vis, r), New = rl64(newsig, r), Bm = rl64(B, r);
                const uint64_t Uf = r > 0 ? rl64(Snew, r - 1) : 0ull, On = rl64(Snew, r), Oo = rl64(S, r);
                const uint64_t Dd = r < 63 ? rl64(S, r + 1) : 0ull;
                const uint64_t Nu = r > 0 ? rl64(NEG, r - 1) : 0ull, No = rl64(NEG, r), Nd = r < 63 ? rl64(NEG, r + 1) : 0ull;
                if (bit_at(Vis, x)) {
                    const uint32_t n3 = win3(Uf, x), s3 = win3(Dd, x);
                    const uint32_t Wb = win3(On, x) & 1, Eb = (win3(Oo, x) >> 2) & 1;
                    const uint32_t d = bit_at(Bm, x);
                    const uint32_t pos = nsym + __popcll(Vis & lt) + __popcll(New & lt);
                    F.sym[pos] = (uint8_t)(F.zc2[n3 | s3 << 3 | Wb << 6 | Eb << 7] | d << 5);
                    if (d) {
                        const uint32_t g3 = win3(No, x);
                        const uint32_t sci = Wb | (g3 & 1) << 1 | Eb << 2 | ((g3 >> 2) & 1) << 3 | ((n3 >> 1) & 1) << 4 |
                                             bit_at(Nu, x) << 5 | ((s3 >> 1) & 1) << 6 | bit_at(Nd, x) << 7;
                        const uint32_t sc = F.T.sc[sci];
                        F.sym[pos + 1] = (uint8_t)((CtxSC0 + (sc & 7)) | ((((g3 >> 1) & 1) ^ (sc >> 3)) & 1) << 5);
                    }
                }
                nsym += __popcll(Vis) + __popcll(New);
            }
        }
        // =========== magnitude refinement (t1_fast5.go:252-335): members = significant before this plane ===========
        {
            // "a significant neighbour" of every sample of a row, for all rows at once (lanes = rows) instead of per row on the scalar
            // unit: the per-row part of this loop was four more broadcasts and ten 64-bit scalar operations, for every row of
            // every plane -- the largest single item of this kernel's instruction count
            uint64_t Up = __shfl_up(Snew, 1), Dn2 = __shfl_down(Snew, 1);
            if (lane == 0) Up = 0;
            if (lane == 63) Dn2 = 0;
            const uint64_t ANY = spread3(Up) | spread3(Dn2) | (Snew << 1) | (Snew >> 1);
            uint64_t rows = __ballot(S != 0);
            while (rows) {
                const int r = __ffsll((long long)rows) - 1;
                rows &= rows - 1;
                T1F_ROOM();
                const uint64_t M = rl64(S, r), Rf = rl64(REF, r), Bm = rl64(B, r), any8 = rl64(ANY, r);
                if (bit_at(M, x)) {
                    const uint32_t ctx = bit_at(Rf, x) ? CtxMag2 : (bit_at(any8, x) ? CtxMag1 : CtxMag0);
                    F.sym[nsym + __popcll(M & lt)] = (uint8_t)(ctx | bit_at(Bm, x) << 5);
                }
                nsym += __popcll(M);
            }
        }
        REF |= S;
        // =========== cleanup (t1_fast5.go:338-876) ===========
        const uint64_t mem = ~Snew & ~vis & rowok;          // coded here; every member whose bit is set becomes significant
        const uint64_t Sc = Snew | (mem & B);
        for (int r0 = 0; r0 < h; r0 += 4) {
            uint64_t Mm[4], Bm[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { Mm[i] = r0 + i < 64 ? rl64(mem, r0 + i) : 0ull; Bm[i] = r0 + i < 64 ? rl64(B, r0 + i) : 0ull; }
            if ((Mm[0] | Mm[1] | Mm[2] | Mm[3]) == 0) continue;
            T1F_ROOM();
            // rows r0-1 .. r0+4 as j = 0..5
            uint32_t L[6], R[6], Cc[6], Ca[6], Ng[6], NgL[6], NgR[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int r = r0 - 1 + j;
                const bool in = r >= 0 && r < 64;
                const uint64_t a = in ? rl64(Snew, r) : 0ull, c = in ? rl64(Sc, r) : 0ull, g = in ? rl64(NEG, r) : 0ull;
                // column x-1 was coded before this column (rows of this stripe and above: updated; next stripe: not yet);
                // column x+1 after it (only the previous stripe's row is updated)
                L[j] = win3(j == 5 ? a : c, x) & 1;
                R[j] = (win3(j == 0 ? c : a, x) >> 2) & 1;
                Cc[j] = bit_at(c, x);
                Ca[j] = bit_at(a, x);
                const uint32_t g3 = win3(g, x);
                Ng[j] = (g3 >> 1) & 1; NgL[j] = g3 & 1; NgR[j] = (g3 >> 2) & 1;
            }
            uint32_t m[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { m[i] = bit_at(Mm[i], x); b[i] = bit_at(Bm[i], x); }
            uint8_t sy[10];
            uint32_t cnt = 0;
            const bool full = r0 + 4 <= h;
            uint32_t anyn = Cc[0] | Ca[5];
#pragma unroll
            for (int j = 0; j < 6; j++) anyn |= L[j] | R[j];
            const bool canRL = full && (m[0] & m[1] & m[2] & m[3]) && !anyn;
            int first = 0;                                   // first row of this column coded sample by sample
            if (canRL) {
                const int fs = b[0] ? 0 : (b[1] ? 1 : (b[2] ? 2 : (b[3] ? 3 : -1)));
                sy[cnt++] = (uint8_t)(CtxRL | (fs >= 0 ? 1u : 0u) << 5);
                if (fs >= 0) {
                    sy[cnt++] = (uint8_t)(CtxUni | ((fs >> 1) & 1) << 5);
                    sy[cnt++] = (uint8_t)(CtxUni | (fs & 1) << 5);
                }
                first = fs >= 0 ? fs : 4;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i < first || !m[i]) continue;
                const int j = i + 1;                          // index of this row in the 6-row window
                const bool runhead = canRL && i == first;     // its significance was coded by the run-length symbols
                if (!runhead) {
                    const uint32_t idx = L[j - 1] | Cc[j - 1] << 1 | R[j - 1] << 2 | L[j + 1] << 3 | Ca[j + 1] << 4 | R[j + 1] << 5 |
                                         L[j] << 6 | R[j] << 7;
                    sy[cnt++] = (uint8_t)(F.zc2[idx] | b[i] << 5);
                }
                if (b[i]) {
                    const uint32_t sci = L[j] | NgL[j] << 1 | R[j] << 2 | NgR[j] << 3 | Cc[j - 1] << 4 | Ng[j - 1] << 5 |
                                         Ca[j + 1] << 6 | Ng[j + 1] << 7;
                    const uint32_t sc = F.T.sc[sci];
                    sy[cnt++] = (uint8_t)((CtxSC0 + (sc & 7)) | ((Ng[j] ^ (sc >> 3)) & 1) << 5);
                }
            }
            // coding order inside a stripe = column order: exclusive scan of the per-column symbol counts
            uint32_t incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const uint32_t base = nsym + incl - cnt;
#pragma unroll
            for (int k = 0; k < 10; k++)
                if ((uint32_t)k < cnt) F.sym[base + k] = sy[k];
            nsym += __shfl(incl, 63);
        }
        S = Sc;
    }
    T1F_DRAIN();
#undef T1F_DRAIN
#undef T1F_ROOM
    if (lane != 0) return;
    if (SPLIT) {
        if (govf) atomicMax(fault, 2);
        nsyms[jid] = govf ? 0u : gtotal;
        if (govf) lens[jid] = 0;
        numbps[jid] = (uint8_t)numBPS;
        return;
    }
    // ---- flush (t1_fast5.go:878-898) ----
    const uint32_t tempC = e.C + e.A;
    e.C |= 0xFFFF;
    if (e.C >= tempC) e.C -= 0x8000;
    e.C <<= e.CT; mq_byte_out(e);
    e.C <<= e.CT; mq_byte_out(e);
    long end = e.bp + 1;                       // endPos
    if (e.cur == 0xFF) end--;                  // drop a trailing 0xFF
    else if (e.bp >= 1) { if (e.bp - 1 < e.cap) out[e.bp - 1] = (uint8_t)e.cur; else e.overflow = 1; }
    if (e.overflow) atomicMax(fault, 2);
    lens[jid] = end > 1 ? (uint32_t)(end - 1) : 0;
    numbps[jid] = (uint8_t)numBPS;
}

#include "t1_big.inc"

// The lanes kernels run at the latency of one chain on a few hundred wavefronts; with frames in flight they share SIMDs with the
// device-filling kernels of other frames (context formation, the plane kernels, the transforms), eight wavefronts to a SIMD, and
// a chain that gets every ninth issue slot takes nine times as long.  Raised wave priority lets the chain issue whenever it can;
// the throughput wavefronts fill the slots its dependent instructions leave.
#ifndef J2K_T1_LANES_PRIO
#define J2K_T1_LANES_PRIO 3
#endif
#define T1_LANES_PRIO() __builtin_amdgcn_s_setprio(J2K_T1_LANES_PRIO)
// Wavefronts per workgroup of the lanes kernels.  Every wavefront works alone (its own 64 blocks, its own LDS, no barrier);
// putting four of them into one workgroup only makes the dispatcher place them on the four SIMDs of one compute unit, one each,
// instead of wherever a single-wavefront workgroup happens to land -- with twenty frames' lanes kernels in flight the busiest SIMD
// decides how long a kernel takes.
#ifndef T1_LANES_WPW
#define T1_LANES_WPW 4
#endif
__device__ __forceinline__ void t1_wave_sync() {       // LDS written by this wavefront is visible to all its lanes (DS ops of a wave run in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- MQ coder, K blocks per wavefront in lock step (mqc.go:224-267, flush t1_fast5.go:878-898) ----
// t1_encode64_kernel<true> leaves every block's (context, decision) list in memory; here LANE l of workgroup g codes the
// list of block g*K + l.  One serial chain per block is bound by instruction issue (≈45 wave instructions per symbol,
// most of them on the CU's one scalar unit); K chains per wavefront share every instruction, so the issue cost per
// symbol falls by K and the kernel runs at the latency of one chain.  All lane-varying state is in registers (A, C, CT,
// the pending byte) or in a lane-interleaved LDS array (the table entry of each context's current state).  Lanes whose
// list has ended are fed the no-op symbol (context 31: Qe = 0, never renormalises).
__global__ __launch_bounds__(64 * T1_LANES_WPW) void t1_mq_lanes_kernel(const BlockJob *__restrict__ jobs, int njobs, int K, const uint8_t *__restrict__ gsym,
                                                         size_t sym_stride, const uint32_t *__restrict__ nsyms, uint8_t *__restrict__ slots,
                                                         uint32_t *__restrict__ lens, int *__restrict__ fault, const uint32_t *__restrict__ perm) {
    __shared__ uint32_t mqtab_w[T1_LANES_WPW][96];
    __shared__ uint32_t ce_w[T1_LANES_WPW][T1F_NCTX * 64];
    T1_LANES_PRIO();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    uint32_t *const mqtab = mqtab_w[wv], *const ce = ce_w[wv];
    // perm: lane order by symbol count (t1_order_kernel), so that the chains of a wavefront end together
    const long slot = ((long)blockIdx.x * T1_LANES_WPW + wv) * K + lane;
    const uint32_t pj = (lane < K && slot < njobs) ? (perm ? perm[slot] : (uint32_t)slot) : 0xFFFFFFFFu;
    const long jid = (long)pj;
    const bool live = pj != 0xFFFFFFFFu;
    for (int s = lane; s < 94; s += 64) {
        const int i = s >> 1, m = s & 1;
        const uint32_t nm = 2 * c_iso_nmps[i] + m;
        const uint32_t nl = 2 * c_iso_nlps[i] + (c_iso_switch[i] ? 1 - m : m);
        mqtab[s] = (uint32_t)c_iso_qe[i] | nm << 16 | nl << 24;
    }
    t1_wave_sync();
    for (int c = 0; c < T1F_NCTX; c++) ce[c * 64 + lane] = c < NumContexts ? mqtab[c == CtxUni ? 92 : 0] : 0u;
    uint32_t n = live ? nsyms[jid] : 0u;
    if (n == T1F_SKIPPED) n = 0;
    uint32_t nmax = n;
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o));
    if (nmax == 0) return;
    const BlockJob J = jobs[live ? jid : 0];
    const uint8_t *src = gsym + (size_t)(live ? jid : 0) * sym_stride;
    uint8_t *const out = slots + J.out_off;
    const uint32_t cap = (uint32_t)t1_mqbuf_bytes((size_t)J.w * J.h);
    uint32_t A = 0x8000, C = 0, CT = 12;
    uint32_t curb = 0, bp = 0, ovf = 0;            // pending byte buf[bp]; byte k >= 1 goes to out[k - 1] (t1_fast.go:11-34)
    // One byte leaves C (mqc.go:270-299 as t1_fast.go restates it), select-only: the pending byte takes the carry unless it
    // is 0xFF; a pending 0xFF makes the next byte a 7-bit one.
    auto byte_out = [&](bool on) {
        const bool carry = curb != 0xFF && (C & 0x8000000u);
        const uint32_t emit = curb + (carry ? 1u : 0u);
        const uint32_t Cm = carry ? (C & 0x7FFFFFFu) : C;
        const bool ff = emit == 0xFF;
        const bool room = bp - 1u < cap;               // bp == 0: the byte before the stream, never stored (wraps to "no room")
        if (on && room) out[bp - 1] = (uint8_t)emit;
        ovf |= (on && bp >= 1 && !room) ? 1u : 0u;
        const uint32_t sh = ff ? 20u : 19u;
        if (on) {
            bp++;
            curb = (Cm >> sh) & 0xFF;
            C = Cm & ((1u << sh) - 1u);
            CT = ff ? 7u : 8u;
        }
    };
    const uint4 null16 = make_uint4(T1F_NULL_SYM * 0x01010101u, T1F_NULL_SYM * 0x01010101u, T1F_NULL_SYM * 0x01010101u, T1F_NULL_SYM * 0x01010101u);
    const uint32_t nclamp = n ? n - 16u : 0u;
    auto fetch = [&](uint32_t i0) -> uint4 {      // unconditional load from a clamped offset, then select (n is a multiple of 16)
        uint4 v;
        __builtin_memcpy(&v, src + min(i0, nclamp), 16);
        const bool in = i0 < n;
        return make_uint4(in ? v.x : null16.x, in ? v.y : null16.y, in ? v.z : null16.z, in ? v.w : null16.w);
    };
    uint4 cur = fetch(0);
    uint32_t idx = (cur.x & 31u) * 64 + lane;
    uint32_t ent = ce[idx];
    for (uint32_t i0 = 0; i0 < nmax; i0 += 16) {
        const uint4 nxt = fetch(i0 + 16);
        const uint32_t words[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t sy = words[q >> 2] >> ((q & 3) * 8);
            const uint32_t syn = words[(q + 1) >> 2] >> (((q + 1) & 3) * 8);
            const uint32_t d = (sy >> 5) & 1;
            // the next symbol's entry is read BEFORE this symbol's update is written and patched below if it is the same
            // context: the LDS round trip of the state update stays off the chain of the following symbol
            const uint32_t idxn = (syn & 31u) * 64 + lane;
            const uint32_t entn = ce[idxn];
            // both successor entries, read while the interval arithmetic runs (the pick waits for isM and the renormalisation)
            const uint32_t candM = mqtab[(ent >> 16) & 0xFF], candL = mqtab[ent >> 24];
            const uint32_t qe = ent & 0xFFFF;
            const uint32_t A1 = A - qe;
            const bool isM = d == ((ent >> 16) & 1);
            const bool lt = A1 < qe;
            C += (isM != lt) ? qe : 0u;                 // MPS: C += qe unless A < qe; LPS: only if A < qe
            A = (isM == lt) ? qe : A1;
            uint32_t shift = (uint32_t)__builtin_clz(A) - 16;   // A != 0; 0 when the interval is still >= 0x8000
            // no wave-level branch around the renormalisation (with K lanes some lane nearly always renormalises, and a
            // taken branch costs more than the dozen instructions it would skip): shift = 0 lanes pass through unchanged
            {
                const uint32_t ne = isM ? candM : candL;
                const bool upd = shift != 0;            // a context changes state exactly when it renormalises
                const uint32_t nent = upd ? ne : ent;
                ce[idx] = nent;
                ent = idxn == idx ? nent : entn;
                A <<= shift;
                uint32_t s1 = min(shift, CT);
                C <<= s1; CT -= s1; shift -= s1;
                if (__any(CT == 0)) {                   // a byte leaves the register (at most three times per symbol)
                    bool on = CT == 0;
                    byte_out(on);
                    s1 = on ? min(shift, CT) : 0u;
                    C <<= s1; CT -= s1; shift -= s1;
                    while (__any(CT == 0)) {
                        on = CT == 0;
                        byte_out(on);
                        s1 = on ? min(shift, CT) : 0u;
                        C <<= s1; CT -= s1; shift -= s1;
                    }
                }
            }
            idx = idxn;
        }
        cur = nxt;
    }
    if (n == 0) return;
    // ---- flush (t1_fast5.go:878-898) ----
    const uint32_t tempC = C + A;
    C |= 0xFFFF;
    if (C >= tempC) C -= 0x8000;
    C <<= CT; byte_out(true);
    C <<= CT; byte_out(true);
    uint32_t end = bp + 1;
    if (curb == 0xFF) end--;
    else if (bp >= 1) { if (bp - 1 < cap) out[bp - 1] = (uint8_t)curb; else ovf = 1; }
    if (ovf) atomicMax(fault, 2);
    lens[jid] = end > 1 ? (uint32_t)(end - 1) : 0;
}

// ---- MQ decoder (mqc.go:352-497) --------------------------------------------------------------
struct MqDec { uint32_t C, A, CT; long bp, len; const uint8_t *data; };

__device__ __forceinline__ void mq_byte_in(MqDec &d) {   // mqc.go:402-439
    if (d.bp < 0) d.bp = 0;
    if (d.bp >= d.len) { d.C += 0xFF00; d.CT = 8; return; }
    const uint32_t next = (d.bp + 1 < d.len) ? d.data[d.bp + 1] : 0xFF;
    if (d.data[d.bp] == 0xFF) {
        if (next > 0x8F) { d.C += 0xFF00; d.CT = 8; }
        else { d.bp++; d.C += next << 9; d.CT = 7; }
    } else { d.bp++; d.C += next << 8; d.CT = 8; }
}
__device__ __forceinline__ void mq_renorm_dec(MqDec &d) {   // mqc.go:488-497
    do {
        if (d.CT == 0) mq_byte_in(d);
        d.A <<= 1; d.C <<= 1; d.CT--;
    } while ((d.A & 0x8000) == 0);
}
// mqc.go:443-485.  The per-context state is the state's TABLE ENTRY (mq word: qe | nmps << 16 | nlps << 24; the MPS
// is the parity of nmps) kept in ent[], so a decision costs one LDS read unless the context changes state.
__device__ __forceinline__ int mq_decode(MqDec &d, uint32_t *ent, const uint32_t *mq, int ctx) {
    const uint32_t e = ent[ctx];
    const uint32_t qe = e & 0xFFFF;
    const int mps = (e >> 16) & 1;
    int dec;
    d.A -= qe;
    if ((d.C >> 16) < qe) {
        uint32_t nst;
        if (d.A < qe) { dec = mps; nst = (e >> 16) & 0xFF; }
        else { dec = 1 - mps; nst = e >> 24; }
        ent[ctx] = mq[nst];
        d.A = qe;
        mq_renorm_dec(d);
        return dec;
    }
    d.C -= qe << 16;
    if ((d.A & 0x8000) == 0) {
        uint32_t nst;
        if (d.A < qe) { dec = 1 - mps; nst = e >> 24; }
        else { dec = mps; nst = (e >> 16) & 0xFF; }
        ent[ctx] = mq[nst];
        mq_renorm_dec(d);
        return dec;
    }
    return mps;
}
__device__ __forceinline__ int mq_decode(MqDec &d, T1Tables &T, int ctx) { return mq_decode(d, T.ent, T.mq, ctx); }

// One block, one lane: T.Decode's three passes per bit plane (t1.go:1291-1410) on byte flags (`flags` has a 1-sample
// border, `stride` = w + 2).  ent = this lane's context entries, mq / zc / sc = the shared tables (zc for this block's band).
struct T1DecLane {
    MqDec d;
    uint32_t *ent;
    const uint32_t *mq;
    const uint8_t *zcsc;     // zc[i] in the low nibble, sc[i] (sign context | prediction << 3) in the high one
    uint8_t *flags;
    int32_t *data;
    int w, h, stride;
};
__device__ __forceinline__ void dec_sign(T1DecLane &L, uint8_t *f) {   // t1.go:1322-1328
    const uint32_t sc = (uint32_t)L.zcsc[sc_index(f[-1], f[1], f[-L.stride], f[L.stride])] >> 4;
    if (mq_decode(L.d, L.ent, L.mq, CtxSC0 + (sc & 7)) ^ (int)(sc >> 3)) *f |= T1SignNeg;
}
// MagRef by the whole wavefront (t1.go:1331-1347).  Its contexts depend only on flags that the pass does not change for
// OTHER samples, so it is not a serial pass: per row (64 columns at a time) all lanes find the members and their contexts,
// lane 0 runs the MQ decoder over that short list, and all lanes apply the decisions.  L.d is lane 0's decoder.
__device__ __forceinline__ void t1_dec_magref_wave(T1DecLane &L, int32_t bit, uint8_t *mrctx, int lane) {
    const int w = L.w, h = L.h, stride = L.stride;
    const uint64_t lt_lane = (1ull << lane) - 1;
    for (int y = 0; y < h; y++)
        for (int x0 = 0; x0 < w; x0 += 64) {
            const int x = x0 + lane;
            uint8_t *const f = L.flags + (size_t)(y + 1) * stride + T1D_XO + x;
            const uint32_t fv = x < w ? *f : 0u;
            const bool member = (fv & T1Sig) && !(fv & T1Visit);
            const uint64_t mask = __ballot(member);
            if (mask == 0) continue;
            const int pos = __popcll(mask & lt_lane);
            if (member) mrctx[pos] = (uint8_t)((fv & T1Refine) ? CtxMag2 : ((fv & T1HasNb) ? CtxMag1 : CtxMag0));
            __syncthreads();
            uint64_t bits = 0;
            if (lane == 0) {
                const int nm = __popcll(mask);
                for (int i = 0; i < nm; i++) bits |= (uint64_t)mq_decode(L.d, L.ent, L.mq, mrctx[i]) << i;
            }
            bits = (uint64_t)__shfl((int)(bits >> 32), 0) << 32 | (uint32_t)__shfl((int)bits, 0);
            if (member) {
                if ((bits >> pos) & 1) atomicOr(&L.data[(size_t)y * w + x], bit);
                *f = (uint8_t)(fv | T1Refine);
            }
            __syncthreads();
        }
}
// SigProp (t1.go:1295-1319).  Per row (64 columns at a time) all lanes find the candidates (not significant, a significant
// neighbour) with a ballot and compute, for every sample that is not significant yet, its ZC and sign context INDICES
// from the flags.  Inside the row those can only be changed by the left-hand neighbour turning significant, which lane 0
// knows: it walks the candidates touching no flag at all -- index from the list, patch the W bits, decode, keep the new
// significances / signs / visits as three 64-bit masks -- and adds the right-hand neighbour of a sample that turns
// significant to the candidates.  All lanes then write the row's flags, the neighbour bits of the rows above and below and
// the magnitudes from the masks.
__device__ __forceinline__ void t1_dec_sigprop_wave(T1DecLane &L, int32_t bit, uint8_t *zcl, uint8_t *scl, int lane) {
    const int w = L.w, h = L.h, stride = L.stride;
    for (int y = 0; y < h; y++)
        for (int x0 = 0; x0 < w; x0 += 64) {
            uint8_t *const row = L.flags + (size_t)(y + 1) * stride + T1D_XO + x0;
            uint8_t *const f = row + lane;
            const bool inb = x0 + lane < w;
            const uint32_t fv = inb ? *f : (uint32_t)T1Sig;
            const uint64_t sigmask = __ballot((fv & T1Sig) != 0);
            uint64_t m = __ballot((fv & (T1Sig | T1HasNb)) == T1HasNb);
            if (m == 0) continue;
            if (!(fv & T1Sig)) {
                zcl[lane] = (uint8_t)zc_packed(f, stride);
                scl[lane] = (uint8_t)sc_index(f[-1], f[1], f[-stride], f[stride]);
            }
            __syncthreads();
            uint64_t newsig = 0, newneg = 0, vis = 0;
            if (lane == 0) {
                while (m) {
                    const int xi = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const uint64_t bx = 1ull << xi;
                    const uint32_t wnew = xi > 0 ? (uint32_t)(newsig >> (xi - 1)) & 1u : 0u;   // (column x0 - 1: already in the flags)
                    vis |= bx;
                    if (mq_decode(L.d, L.ent, L.mq, L.zcsc[zcl[xi] | wnew] & 15)) {
                        uint32_t sci = scl[xi];
                        if (wnew) sci |= 1u | ((uint32_t)(newneg >> (xi - 1)) & 1u) << 1;
                        const uint32_t sc = (uint32_t)L.zcsc[sci] >> 4;
                        const uint64_t neg = (uint64_t)(mq_decode(L.d, L.ent, L.mq, CtxSC0 + (sc & 7)) ^ (int)(sc >> 3));
                        newsig |= bx;
                        newneg |= neg << xi;
                        if (xi < 63 && x0 + xi + 1 < w && !((sigmask >> (xi + 1)) & 1)) m |= bx << 1;
                    }
                }
            }
            newsig = (uint64_t)__shfl((int)(newsig >> 32), 0) << 32 | (uint32_t)__shfl((int)newsig, 0);
            newneg = (uint64_t)__shfl((int)(newneg >> 32), 0) << 32 | (uint32_t)__shfl((int)newneg, 0);
            vis = (uint64_t)__shfl((int)(vis >> 32), 0) << 32 | (uint32_t)__shfl((int)vis, 0);
            if (inb) {
                const uint32_t me = (uint32_t)(newsig >> lane) & 1u;
                const uint64_t side = (newsig << 1) | (newsig >> 1);
                uint32_t nf = fv | (((vis >> lane) & 1) ? T1Visit : 0u) | (me ? T1Sig : 0u) | (((newneg >> lane) & 1) ? T1SignNeg : 0u) |
                              (((side >> lane) & 1) ? T1HasNb : 0u);
                if (nf != fv) *f = (uint8_t)nf;
                if (((side | newsig) >> lane) & 1) { f[-stride] |= T1HasNb; f[stride] |= T1HasNb; }
                if (me) L.data[(size_t)y * w + x0 + lane] = bit;
            }
            if (lane == 0) {                         // the columns next to the chunk (border, or the neighbouring chunk)
                if (newsig & 1) { row[-1] |= T1HasNb; row[-1 - stride] |= T1HasNb; row[-1 + stride] |= T1HasNb; }
                if (newsig >> 63) { row[64] |= T1HasNb; row[64 - stride] |= T1HasNb; row[64 + stride] |= T1HasNb; }
            }
            __syncthreads();
        }
}
// Cleanup (t1.go:1350-1410) with the classification done by all lanes: per 4-row stripe (64 columns at a time) every lane
// looks at its column -- which rows are members (neither significant nor visited), is the column eligible for the
// run-length symbol -- clears the visit flags and leaves one byte per column; lane 0 then walks only the columns that
// have a member.  A column's run-length eligibility can be lost to a sample of the column before it that turns
// significant in this very stripe: that one case is re-checked from the flags.
__device__ __forceinline__ void t1_dec_cleanup_wave(T1DecLane &L, int32_t bit, uint8_t *colinfo, int lane) {
    const int w = L.w, h = L.h, stride = L.stride;
    int last_sig = -2;
    for (int y = 0; y < h; y += 4)
        for (int x0 = 0; x0 < w; x0 += 64) {
            const int x = x0 + lane;
            uint32_t m4 = 0, any = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (x < w && y + i < h) {
                    uint8_t *f = L.flags + (size_t)(y + i + 1) * stride + T1D_XO + x;
                    const uint32_t fv = *f;
                    any |= fv;
                    if (!(fv & (T1Sig | T1Visit))) m4 |= 1u << i;
                    if (fv & T1Visit) *f = (uint8_t)(fv & ~T1Visit);
                }
            }
            const bool clean = x < w && y + 4 <= h && !(any & (T1Sig | T1Visit | T1HasNb));
            colinfo[lane] = (uint8_t)(m4 | (clean ? 16u : 0u));
            uint64_t work = __ballot(m4 != 0);
            if (work == 0) continue;
            __syncthreads();
            if (lane == 0) {
                while (work) {
                    const int xi = __ffsll((long long)work) - 1, xx = x0 + xi;
                    work &= work - 1;
                    const uint32_t info = colinfo[xi];
                    const uint32_t mem = info & 15u;
                    bool canRL = (info & 16u) != 0;
                    if (canRL && last_sig == xx - 1)
                        for (int i = 0; i < 4; i++)
                            if (L.flags[(size_t)(y + i + 1) * stride + T1D_XO + xx] & T1HasNb) canRL = false;
                    int first = 0;
                    if (canRL) {
                        if (mq_decode(L.d, L.ent, L.mq, CtxRL) == 0) continue;
                        int pos = mq_decode(L.d, L.ent, L.mq, CtxUni) << 1;
                        pos |= mq_decode(L.d, L.ent, L.mq, CtxUni);
                        uint8_t *f = L.flags + (size_t)(y + pos + 1) * stride + T1D_XO + xx;
                        L.data[(size_t)(y + pos) * w + xx] = bit;
                        dec_sign(L, f);
                        set_significant_dec(f, stride);
                        last_sig = xx;
                        first = pos + 1;
                    }
                    for (int i = first; i < 4; i++) {
                        if (!((mem >> i) & 1)) continue;
                        uint8_t *f = L.flags + (size_t)(y + i + 1) * stride + T1D_XO + xx;
                        if (mq_decode(L.d, L.ent, L.mq, L.zcsc[zc_packed(f, stride)] & 15)) {
                            L.data[(size_t)(y + i) * w + xx] = bit;
                            dec_sign(L, f);
                            set_significant_dec(f, stride);
                            last_sig = xx;
                        }
                    }
                }
            }
            __syncthreads();
        }
}
// One block, one wavefront; all lanes call this.  Only the MQ decisions and what depends on them run on lane 0.
__device__ __forceinline__ void t1_decode_block_wave(T1DecLane &L, int numBPS, uint8_t *mrctx, int lane) {
    for (int bp = numBPS - 1; bp >= 0; bp--) {
        const int32_t bit = bp < 32 ? (int32_t)(1u << bp) : 0;
        t1_dec_sigprop_wave(L, bit, mrctx, mrctx + 64, lane);
        t1_dec_magref_wave(L, bit, mrctx, lane);
        t1_dec_cleanup_wave(L, bit, mrctx, lane);
    }
}
__device__ __forceinline__ void mq_dec_init(MqDec &d, const uint8_t *data, long len) {   // NewMQDecoder mqc.go:370-399
    d = MqDec{0, 0x8000, 0, -1, len, data};
    if (d.len == 0) d.C = 0xFFu << 16; else { d.bp = 0; d.C = (uint32_t)d.data[0] << 16; }
    mq_byte_in(d);
    d.C <<= 7; d.CT -= 7; d.A = 0x8000;
}

template <bool LDSW>
__global__ __launch_bounds__(64) void t1_decode_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ stream,
                                                       const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens,
                                                       const uint8_t *__restrict__ numbps, int32_t *__restrict__ decoded,
                                                       uint8_t *__restrict__ work, size_t work_per_job, int lds_work_bytes, int skip_small) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int vzero;                                   // see t1_decode64_kernel: keeps the serial chain's loop control off the scalar unit
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const int jid = (int)blockIdx.x + vzero;
    if (jid >= njobs) return;
    const int lane = threadIdx.x;
    const BlockJob J = jobs[jid];
    if (skip_small && J.w <= skip_small && J.h <= skip_small) return;   // skip_small = 64 / 256: t1_decode64_kernel (and t1_decode_big_kernel) take these
    const int w = J.w, h = J.h, stride = T1D_STRIDE(w);
    const size_t n = (size_t)w * h;
    T1Tables &T = *reinterpret_cast<T1Tables *>(smem);
    // Only the flags live in the workspace (LDS when they fit): the magnitudes are built in the output buffer itself
    // (first significance = plain store, refinement = fire-and-forget atomic OR), so LDS does not limit how many of
    // these serial chains a SIMD interleaves.
    uint8_t *flags = LDSW ? smem + ((sizeof(T1Tables) + 15) & ~size_t(15)) : work + (size_t)jid * work_per_job;
    int32_t *out = decoded + J.out_off;

    build_tables(T, J.band, lane);
    for (size_t i = lane; i < (size_t)stride * (h + 2) + 4; i += 64) flags[i] = 0;
    for (size_t i = lane; i < n; i += 64) out[i] = 0;
    __syncthreads();
    init_dec_contexts(T, lane);
    for (int p = lane; p < 256; p += 64) T.zc[p] = (uint8_t)(T.zc[p] | T.sc[p] << 4);     // the decoders' packed look-up (T1DecLane::zcsc)
    __syncthreads();
    {
        T1DecLane L;
        if (lane == 0) mq_dec_init(L.d, stream + offs[jid], (long)lens[jid]);
        L.ent = T.ent; L.mq = T.mq; L.zcsc = T.zc; L.flags = flags; L.data = out; L.w = w; L.h = h; L.stride = stride;
        t1_decode_block_wave(L, __shfl((int)numbps[jid], 0), T.mrctx, lane);
    }
    __syncthreads();                                                      // includes the wait for the stores and atomics above
    for (size_t i = lane; i < n; i += 64) {                               // t1.go:1281-1289
        if (!(flags[(i / w + 1) * stride + T1D_XO + (i % w)] & T1SignNeg)) continue;
        const int v = __hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2, not this CU's L1
        out[i] = (int32_t)(0u - (uint32_t)v);
    }
}

// The same for blocks up to 64x64 with the flags in LDS BY CONSTRUCTION: a workspace pointer chosen at run time (LDS or
// `work`) is a flat pointer, and flat loads / stores cost the serial chain far more than ds_* ones (C3: 84 -> 58 ms).
// (Tried and removed: K blocks per wavefront on K lanes of this same code -- the loops are the same for every block, so
// lanes stay position-synchronous and share the scans -- K = 2: 62.7 ms, 4: 96.9, 12: 138: two blocks rarely take a
// decision at the same sample of the same pass, so the decoder's instructions are not shared, only serialised.)
// LDS per block decides how many chains are resident.  The 7005 blocks of a 4K frame are 27.4 per CU on 256 CUs; round 2 fitted
// exactly 28 per CU (5632 bytes = 11 granules of 512, 157.7 of 160 KB) -- and ran at 28 ms per frame on four boxes and 58-63 ms on
// two others with the SAME code object, LDS size, register counts and grid (rocprofv3 traces of six runs):
// every other kernel of those runs took the same time on both kinds of box, only this one doubled.  At 27.4 of 28 slots there is
// no slack: a CU that holds one block fewer (LDS allocations of finished blocks free in a different order and fragment, a box
// with fewer enabled CUs) pushes the overflow into a second round of 28 ms chains.  Now 10 granules: 32 blocks per CU = the
// wave limit, 8192 slots for 7005 blocks.  What went: the 94-entry MQ table (read from constant memory: it is consulted only
// when a context changes state; the current entry of every context stays in LDS), and zc / sc share one byte per index.
#define T1D64_FLAGS ((66 * T1D_STRIDE(64) + 4 + 15) & ~15)
struct T1Dec64Shared {
    uint32_t ent[20];
    uint8_t zcsc[256];                           // lutZCCtx in the low nibble, lutSignCtx | lutSignPred << 3 in the high one
    alignas(16) uint8_t flags[T1D64_FLAGS];      // doubles as the scratch the tables are built in
    uint8_t mrctx[128];                          // per-column context lists of the wave-level passes (two of 64 bytes)
};
static_assert(sizeof(T1Dec64Shared) <= 5120, "t1_decode64_kernel: LDS per block above 10 granules (32 blocks per CU)");
__global__ __launch_bounds__(64) void t1_decode64_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ stream,
                                                         const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens,
                                                         const uint8_t *__restrict__ numbps, int32_t *__restrict__ decoded, int min_bps) {
    __shared__ T1Dec64Shared S;
    if ((int)blockIdx.x >= njobs) return;
    if ((int)numbps[blockIdx.x] < min_bps) return;                   // the plane-stepped path took this block
    const int lane = threadIdx.x;
    // The block index as a VECTOR value the compiler cannot prove uniform: everything derived from it (block size, loop
    // counters, flag addresses) then lives in VGPRs and runs on the SIMD's own ALU.  As uniform values they go to the
    // scalar unit, which all four SIMDs of a CU share -- with 28 serial chains per CU that unit is the bottleneck
    // (C3: 67.7 ms with scalar loop control, 58 ms with vector).
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const int jid = (int)blockIdx.x + vzero;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h, stride = T1D_STRIDE(w);
    if (w > 64 || h > 64) return;                                    // the general kernel takes these
    const int n = w * h;
    int32_t *out = decoded + J.out_off;
    {
        T1Tables &T = *reinterpret_cast<T1Tables *>(S.flags);
        build_tables(T, J.band, lane);
        __syncthreads();
        for (int p = lane; p < 256; p += 64) S.zcsc[p] = (uint8_t)(T.zc[p] | T.sc[p] << 4);
        if (lane < 20) S.ent[lane] = lane < NumContexts ? T.mq[lane == CtxUni ? 92 : 0] : 0u;
        __syncthreads();
    }
    for (int i = lane; i < (stride * (h + 2) + 4 + 3) / 4; i += 64) reinterpret_cast<uint32_t *>(S.flags)[i] = 0;
    for (int i = lane; i < n; i += 64) out[i] = 0;
    __syncthreads();
    {
        T1DecLane L;                                              // L.d is lane 0's decoder
        if (lane == 0) mq_dec_init(L.d, stream + offs[jid], (long)lens[jid]);
        L.ent = S.ent; L.mq = c_mq94.v; L.zcsc = S.zcsc; L.flags = S.flags; L.data = out; L.w = w; L.h = h; L.stride = stride;
        t1_decode_block_wave(L, __shfl((int)numbps[jid], 0), S.mrctx, lane);
    }
    __syncthreads();                                                      // includes the wait for lane 0's stores and atomics
    for (int i = lane; i < n; i += 64) {                                  // t1.go:1281-1289
        if (!(S.flags[(i / w + 1) * stride + T1D_XO + (i % w)] & T1SignNeg)) continue;
        const int v = __hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2, not this CU's L1
        out[i] = (int32_t)(0u - (uint32_t)v);
    }
}

#include "t1_bigdec.inc"

// ---- plane-stepped decoder for blocks up to 64x64 -----------------------------------------------------------------
// A frame at 12 bits and Quality 75 has 18 bit planes per block and ~78 % of its MQ decisions are magnitude refinements
// (every sample that is already significant takes one per plane).  MagRef's contexts (Mag0 / Mag1 / Mag2) depend only on
// flags the pass does not change for other samples, so the pass is a LIST of contexts known before its first decision --
// a pure MQ chain, like the encoder's -- and chains of different blocks can share instructions, one block per LANE.
// The decoder therefore runs as a sequence of launches, launch k working on every block's k-th plane from ITS OWN top,
// p = numBPS - 1 - k (so the planes in which a block's samples turn significant -- its first three, where nearly all of
// the ZC and sign decisions are -- coincide for all blocks and those launches fill the device):
//   t1_dec_step_kernel(k)          one block per wavefront, its LDS image (flags, context entries, tables) reloaded from
//                                  the workspace: apply the MagRef decisions of plane p + 1, Cleanup(p + 1), SigProp(p),
//                                  write MagRef(p)'s context list (one code per member, coding order), save the image;
//   t1_dec_magref_lanes_kernel(k)  lane l of wavefront g runs the MQ decoder of block 64 g + l over its list, all lanes in
//                                  lock step (select-only step, no divergent branch), and leaves one bit per decision.
// The MagRef chains -- 64 per wavefront, a hundred wavefronts per 4K frame -- cost a per-mille of the device's issue slots and
// run at the latency of one chain, beside the step kernels of other frames in flight; what stays issue-bound is SigProp and
// Cleanup (t1_decode64_kernel's wave-level passes, unchanged).  The launch sequence is fixed (32 + 31 launches: the host does
// not know the blocks' bit-plane counts); a block is finished after launch k = numBPS, later launches find nothing to do.  Blocks with more than 31 planes (bit 0 for p >= 32, t1.go:1291) and blocks larger than
// 64x64 keep the one-launch kernels.
struct T1DecState { uint32_t C, A, CT, nmr; long long bp, len; };
#define T1DS_IMG ((sizeof(T1Dec64Shared) + 15) & ~size_t(15))
#define T1DS_STATE T1DS_IMG
#define T1DS_BITS (T1DS_STATE + 32)
#define T1DS_LIST (T1DS_BITS + 512)
#define T1DS_STRIDE (T1DS_LIST + 4096)
#define T1DS_MAXP 31

__global__ __launch_bounds__(64) void t1_dec_step_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ stream,
                                                         const uint64_t *__restrict__ offs, const uint32_t *__restrict__ lens,
                                                         const uint8_t *__restrict__ numbps, int32_t *__restrict__ decoded,
                                                         uint8_t *__restrict__ ws, int k) {
    __shared__ T1Dec64Shared S;
    if ((int)blockIdx.x >= njobs) return;
    const int nb = (int)numbps[blockIdx.x];
    const int p = nb - 1 - k;                                        // this launch: SigProp of the block's k-th plane from ITS top
    const bool first = k == 0;
    if (nb > T1DS_MAXP || p < -1) return;                            // deep blocks: t1_decode64_kernel; block finished at k = nb
    const int lane = threadIdx.x;
    int vzero;                                                       // see t1_decode64_kernel
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const int jid = (int)blockIdx.x + vzero;
    const BlockJob J = jobs[jid];
    const int w = J.w, h = J.h, stride = T1D_STRIDE(w);
    if (w > 64 || h > 64) return;                                    // the general kernel takes these
    const int n = w * h;
    int32_t *out = decoded + J.out_off;
    uint8_t *const rec = ws + (size_t)blockIdx.x * T1DS_STRIDE;
    T1DecState *const st = reinterpret_cast<T1DecState *>(rec + T1DS_STATE);
    T1DecLane L;                                                     // L.d is lane 0's decoder
    if (first) {
        {
            T1Tables &T = *reinterpret_cast<T1Tables *>(S.flags);
            build_tables(T, J.band, lane);
            __syncthreads();
            for (int q = lane; q < 256; q += 64) S.zcsc[q] = (uint8_t)(T.zc[q] | T.sc[q] << 4);
            if (lane < 20) S.ent[lane] = lane < NumContexts ? T.mq[lane == CtxUni ? 92 : 0] : 0u;
            __syncthreads();
        }
        for (int i = lane; i < (stride * (h + 2) + 4 + 3) / 4; i += 64) reinterpret_cast<uint32_t *>(S.flags)[i] = 0;
        for (int i = lane; i < n; i += 64) out[i] = 0;
        if (lane == 0) mq_dec_init(L.d, stream + offs[jid], (long)lens[jid]);
    } else {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(rec);
        for (int i = lane; i < (int)(sizeof(T1Dec64Shared) / 4); i += 64) reinterpret_cast<uint32_t *>(&S)[i] = src[i];
        if (lane == 0) {
            const T1DecState t = *st;
            L.d = MqDec{t.C, t.A, t.CT, (long)t.bp, (long)t.len, stream + offs[jid]};
        }
    }
    __syncthreads();
    L.ent = S.ent; L.mq = c_mq94.v; L.zcsc = S.zcsc; L.flags = S.flags; L.data = out; L.w = w; L.h = h; L.stride = stride;
    const uint64_t lt_lane = (1ull << lane) - 1;
    if (!first && p + 1 < nb) {
        // MagRef(p + 1), second half (t1.go:1331-1347): the same member ballots as when the list was written -- nothing
        // touched the flags in between -- give every member its position in the list; its decision is that bit.
        const int32_t bit = (int32_t)(1u << (p + 1));
        const uint32_t *bits = reinterpret_cast<const uint32_t *>(rec + T1DS_BITS);
        int off = 0;
        for (int y = 0; y < h; y++) {
            uint8_t *const f = S.flags + (size_t)(y + 1) * stride + T1D_XO + lane;
            const uint32_t fv = lane < w ? *f : 0u;
            const bool member = (fv & T1Sig) && !(fv & T1Visit);
            const uint64_t mask = __ballot(member);
            if (member) {
                const int pos = off + __popcll(mask & lt_lane);
                if ((bits[pos >> 5] >> (pos & 31)) & 1) atomicOr(&out[(size_t)y * w + lane], bit);
                *f = (uint8_t)(fv | T1Refine);
            }
            off += __popcll(mask);
        }
        __syncthreads();
        t1_dec_cleanup_wave(L, bit, S.mrctx, lane);
    }
    if (p < 0) {                                                     // t1.go:1281-1289
        __syncthreads();                                             // includes the wait for the stores and atomics above
        for (int i = lane; i < n; i += 64) {
            if (!(S.flags[(i / w + 1) * stride + T1D_XO + (i % w)] & T1SignNeg)) continue;
            const int v = __hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2, not this CU's L1
            out[i] = (int32_t)(0u - (uint32_t)v);
        }
        return;
    }
    int nmr = 0;
    if (p < nb) {
        t1_dec_sigprop_wave(L, (int32_t)(1u << p), S.mrctx, S.mrctx + 64, lane);
        // MagRef(p), first half: members in coding order (rows, then columns) and their contexts
        uint8_t *list = rec + T1DS_LIST;
        for (int y = 0; y < h; y++) {
            const uint32_t fv = lane < w ? S.flags[(size_t)(y + 1) * stride + T1D_XO + lane] : 0u;
            const bool member = (fv & T1Sig) && !(fv & T1Visit);
            const uint64_t mask = __ballot(member);
            if (member) list[nmr + __popcll(mask & lt_lane)] = (uint8_t)((fv & T1Refine) ? 2 : ((fv & T1HasNb) ? 1 : 0));
            nmr += __popcll(mask);
        }
    }
    __syncthreads();
    {
        uint32_t *dst = reinterpret_cast<uint32_t *>(rec);
        for (int i = lane; i < (int)(sizeof(T1Dec64Shared) / 4); i += 64) dst[i] = reinterpret_cast<const uint32_t *>(&S)[i];
        if (lane == 0) *st = T1DecState{L.d.C, L.d.A, L.d.CT, (uint32_t)nmr, (long long)L.d.bp, (long long)L.d.len};
    }
}

// ---- the MQ decoder of one block per lane (mqc.go:402-497 as a select-only step), shared by the lanes kernels ----
// Per lane: A, C, CT and the byte position as a 32-bit offset from a 16-byte aligned base below the block's first byte.
// Compressed bytes come through a per-lane ring of T1R_BYTES bytes in LDS (lanes T1R_BYTES + 4 apart: they start in different banks),
// filled a whole ring ahead at the start and looked at every T1R_PERIOD steps (16 for a ring of 128 bytes): a step can consume at
// most three bytes (15 shifts), so with more than half a ring ahead after every look at least 17 are ahead at the next.  (The ring
// is most of these kernels' LDS, and LDS is what limits how many lanes workgroups of twenty frames in flight a CU holds.)  The refill is synchronous (16-byte loads
// issued together, one wait): values loaded one period ahead and carried in registers were copied at every loop header of the
// callers' nested loops, which made every step wait for memory.  Lanes without a decision pass act = false (Qe = 0 changes
// nothing); lanes without a block keep A = 0x8000.
#ifndef T1R_BYTES
#define T1R_BYTES 128                    /* ring bytes per lane (looked at every T1R_BYTES / 8 steps).  With single-wavefront workgroups 256 was 4 % faster at 20 frames in flight; with four wavefronts per workgroup (T1_LANES_WPW) 128 is what lets two such workgroups share a CU: 143.5 -> 133 ms per step */
#endif
#define T1R_STRIDE (T1R_BYTES + 4)
#define T1R_PERIOD (T1R_BYTES / 8)       /* steps between looks: at most 3 bytes each */
struct MqLaneDec {
    uint32_t A, C, CT, pos, end, filled, d0, rlane, nstep;
    uintptr_t abase;
    bool live;
    uint8_t *ring;

    // 16 bytes of the stream at offset `off`, bytes at or beyond `end` reading as 0xFF (a piece that starts before `end` lies in
    // a page of the stream).  The decoder past its last byte then needs no case of its own: data[bp] = 0xFF followed by 0xFF
    // is byteIn's "marker" branch (C += 0xFF00, CT = 8, bp unchanged), which is also what bp >= len does, and the byte after
    // the last one reads 0xFF as mqc.go:412-416 has it.
    __device__ __forceinline__ uint4 ld16(uint32_t off) const {
        uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (live && off < end) v = *reinterpret_cast<const uint4 *>(abase + off);
        const uint32_t valid = off < end ? end - off : 0u;
        auto pad = [&](uint32_t wv, uint32_t first) -> uint32_t {        // bytes first .. first + 3 of the piece
            const uint32_t nv = valid > first ? min(valid - first, 4u) : 0u;
            return nv >= 4u ? wv : (wv | (0xFFFFFFFFu << (8u * nv)));
        };
        return make_uint4(pad(v.x, 0), pad(v.y, 4), pad(v.z, 8), pad(v.w, 12));
    }
    __device__ __forceinline__ void st16(uint32_t off, const uint4 v) {
        uint32_t *q = reinterpret_cast<uint32_t *>(ring + rlane + (off & (T1R_BYTES - 1u)));
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
    __device__ __forceinline__ void start(const uint8_t *blk, const T1DecState &st, bool lv, uint8_t *rg, int lane) {
        live = lv; ring = rg; rlane = (uint32_t)lane * T1R_STRIDE; nstep = 0;
        abase = (uintptr_t)blk & ~uintptr_t(15);
        d0 = (uint32_t)((uintptr_t)blk - abase);
        A = lv ? st.A : 0x8000u; C = st.C; CT = st.CT;
        pos = d0 + (uint32_t)(st.bp < 0 ? 0 : st.bp);
        end = d0 + (uint32_t)st.len;
        filled = pos & ~15u;
        uint4 c[T1R_BYTES / 16];
#pragma unroll
        for (int q = 0; q < T1R_BYTES / 16; q++) c[q] = ld16(filled + 16 * q);
#pragma unroll
        for (int q = 0; q < T1R_BYTES / 16; q++) st16(filled + 16 * q, c[q]);
        filled += T1R_BYTES;
    }
    __device__ __forceinline__ void save(T1DecState &st) const {
        st.A = A; st.C = C; st.CT = CT;
        st.bp = (long long)(pos - d0);                                // (never negative after NewMQDecoder's first byteIn)
    }
    // call before every step (wave-uniform): every 32nd looks at the ring
    __device__ __forceinline__ void tick() {
        if ((nstep++ & (T1R_PERIOD - 1u)) == 0) refill();
    }
    __device__ __forceinline__ void refill() {                      // at most 32 steps apart
        const uint32_t ahead = filled - pos;
        if (!__any(ahead <= T1R_BYTES / 2u)) return;
        constexpr int NQ = T1R_BYTES / 32;                            // 16-byte pieces in half a ring
        const int nq = ahead <= T1R_BYTES / 2u ? NQ : (ahead <= 3u * T1R_BYTES / 4u ? NQ / 2 : 0);
        uint4 c[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) c[q] = q < nq ? ld16(filled + 16 * q) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NQ; q++) if (q < nq) st16(filled + 16 * q, c[q]);
        filled += 16 * nq;
    }
    // the two bytes a byte-in of this step would look at (independent of the step's context: callers read them early)
    __device__ __forceinline__ void peek(uint32_t &b0, uint32_t &b1) const {
        b0 = ring[rlane + (pos & (T1R_BYTES - 1u))]; b1 = ring[rlane + ((pos + 1) & (T1R_BYTES - 1u))];
    }
    // one decision with the context's table entry e (qe | nmps << 16 | nlps << 24; MPS = parity of nmps); e_new = its entry after
    __device__ __forceinline__ uint32_t step(uint32_t e, const uint32_t *mqtab, bool act, uint32_t b0, uint32_t b1, uint32_t &e_new) {
        const uint32_t e_nm = mqtab[(e >> 16) & 0xFF], e_nl = mqtab[e >> 24];          // both successors, off the A / C chain
        const uint32_t qe = act ? (e & 0xFFFFu) : 0u;
        A -= qe;
        const bool lpsx = (C >> 16) < qe;
        const bool a_lt = A < qe;
        C -= lpsx ? 0u : qe << 16;
        const bool need = lpsx || !(A & 0x8000u);
        const bool flip = need && (lpsx != a_lt);
        const uint32_t dec = ((e >> 16) & 1u) ^ (flip ? 1u : 0u);
        A = lpsx ? qe : A;
        e_new = need ? (flip ? e_nl : e_nm) : e;
        uint32_t nsh = need ? (uint32_t)__clz((int)A) - 16u : 0u;
        // renormalise (mqc.go:488-497): a byte-in whenever CT is 0 and a shift is due.  The first round is straight-line code on
        // the bytes read at the top of the step; further rounds (a second byte-in in the same step, rare) re-read the ring.
        auto round = [&]() {
            const bool bin = nsh > 0 && CT == 0;
            const bool ff = b0 == 0xFF;
            const bool stay = ff && b1 > 0x8F;                        // C += 0xFF00, CT = 8, bp unchanged (mqc.go:402-439; past the end: ld16)
            const uint32_t addC = stay ? 0xFF00u : b1 << (ff ? 9 : 8);
            C += bin ? addC : 0u;
            CT = bin ? ((ff && !stay) ? 7u : 8u) : CT;
            const bool adv = bin && !stay;
            pos += adv ? 1u : 0u;
            b0 = adv ? b1 : b0;
            const uint32_t sft = min(nsh, CT);
            A <<= sft; C <<= sft; CT -= sft; nsh -= sft;
        };
        round();
        while (__any(nsh > 0)) {
            b1 = ring[rlane + ((pos + 1) & (T1R_BYTES - 1u))];
            round();
        }
        return dec;
    }
};

// The MagRef chains of 64 blocks in lock step (MqLaneDec above).  Per lane: the decoder and the three context entries in
// registers.  Both bytes a byte-in might need, and both successor entries of the context's state, are read at the top of the
// step, so no memory latency sits on the A / C chain.  Lanes whose list has ended decode with Qe = 0, which changes nothing.
// The MagRef chain of one wavefront's 64 blocks (t1.go:1331-1347 as list + chain): lane l decodes the n decisions of its list at
// `rec` (nmax = the wavefront's largest n), 16 decisions to a 16-bit piece of the decision string.  e0 / e1 / e2: the lane's entries of
// the contexts Mag0 / Mag1 / Mag2.  BITLIST: the context list as two bit strings (t1_lanes.inc's plane work: Mag2 members in one,
// Mag1 in the other, list position = bit position); otherwise one byte per member (round 2's step kernels).
#ifndef T1_MAGREF_UNROLL
#define T1_MAGREF_UNROLL 8
#endif
template <bool BITLIST>
__device__ __forceinline__ void t1_magref_chain(MqLaneDec &mq, const uint32_t *mqtab, uint32_t &e0, uint32_t &e1, uint32_t &e2,
                                                uint8_t *rec, uint32_t n, uint32_t nmax, bool live) {
    const uint4 *const list = reinterpret_cast<const uint4 *>(rec + T1DS_LIST);
    const uint16_t *const lhi = reinterpret_cast<const uint16_t *>(rec + T1DS_LIST), *const llo = lhi + 256;      // BITLIST: 4096 bits each
    uint16_t *const obits = reinterpret_cast<uint16_t *>(rec + T1DS_BITS);
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
    if (BITLIST) { cur.x = lhi[0]; cur.y = llo[0]; nxt.x = lhi[nmax > 16 ? 1 : 0]; nxt.y = llo[nmax > 16 ? 1 : 0]; }
    else { cur = list[0]; nxt = list[nmax > 16 ? 1 : 0]; }
    for (uint32_t i0 = 0; i0 < nmax; i0 += 16) {
        uint32_t acc = 0;
        if (T1R_PERIOD <= 16 || (i0 & 16u) == 0) mq.refill();
#pragma unroll T1_MAGREF_UNROLL
        for (int s = 0; s < 16; s++) {
            const uint32_t i = i0 + s;
            bool is2, is1;
            // (the two strings exclude each other below n; past it the pieces hold whatever an earlier plane left, and a lane
            //  whose list has ended must still pick exactly one entry to write back unchanged)
            if (BITLIST) { is2 = (cur.x >> s) & 1u; is1 = !is2 && ((cur.y >> s) & 1u); }
            else {
                const uint32_t wsel = (s >> 2) == 0 ? cur.x : (s >> 2) == 1 ? cur.y : (s >> 2) == 2 ? cur.z : cur.w;
                const uint32_t code = (wsel >> ((s & 3) * 8)) & 3u;
                is2 = code == 2; is1 = code == 1;
            }
            const uint32_t e = is2 ? e2 : (is1 ? e1 : e0);
            uint32_t b0, b1, e_new;
            mq.peek(b0, b1);
            const uint32_t dec = mq.step(e, mqtab, i < n, b0, b1, e_new);
            e0 = (is2 || is1) ? e0 : e_new;
            e1 = is1 ? e_new : e1;
            e2 = is2 ? e_new : e2;
            acc |= dec << s;
        }
        if (live && i0 < n) obits[i0 >> 4] = (uint16_t)acc;
        cur = nxt;
        const uint32_t nx = (i0 >> 4) + 2;
        if (BITLIST) { nxt.x = lhi[nx < 256 ? nx : 255]; nxt.y = llo[nx < 256 ? nx : 255]; }
        else nxt = list[nx < 256 ? nx : 255];
    }
}

template <bool BITLIST>
__global__ __launch_bounds__(64 * T1_LANES_WPW) void t1_dec_magref_lanes_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ stream,
                                                                 const uint64_t *__restrict__ offs, const uint8_t *__restrict__ numbps,
                                                                 uint8_t *__restrict__ ws, const uint32_t *__restrict__ perm, int k) {
    __shared__ uint32_t mqtab_w[T1_LANES_WPW][96];
    __shared__ __attribute__((aligned(16))) uint8_t ring_w[T1_LANES_WPW][64 * T1R_STRIDE];
    T1_LANES_PRIO();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    uint32_t *const mqtab = mqtab_w[wv];
    uint8_t *const ring = ring_w[wv];
    // perm: the lane order of t1_lanes.inc (blocks of similar length share a wavefront); slots without a job read as past the end
    const long slot = ((long)blockIdx.x * T1_LANES_WPW + wv) * 64 + lane;
    const uint32_t pj = perm ? perm[slot] : (uint32_t)slot;
    const long jid = pj == 0xFFFFFFFFu ? (long)njobs : (long)pj;
    for (int q = lane; q < 94; q += 64) mqtab[q] = c_mq94.v[q];
    bool live = jid < njobs;
    const BlockJob J = jobs[live ? jid : 0];
    const int nb = live ? (int)numbps[jid] : 0;
    live = live && J.w <= 64 && J.h <= 64 && nb <= T1DS_MAXP && nb - 1 - k >= 0;
    uint8_t *const rec = ws + (size_t)(live ? jid : 0) * T1DS_STRIDE;
    T1DecState *const stp = reinterpret_cast<T1DecState *>(rec + T1DS_STATE);
    T1DecState st = *stp;
    const uint32_t n = live ? st.nmr : 0u;
    uint32_t nmax = n;
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o));
    if (nmax == 0) return;
    uint32_t *const ent = reinterpret_cast<uint32_t *>(rec + offsetof(T1Dec64Shared, ent));
    uint32_t e0 = ent[CtxMag0], e1 = ent[CtxMag1], e2 = ent[CtxMag2];
    MqLaneDec mq;
    mq.start(stream + offs[live ? jid : 0], st, live, ring, lane);
    t1_wave_sync();
    t1_magref_chain<BITLIST>(mq, mqtab, e0, e1, e2, rec, n, nmax, live);
    if (live) {
        ent[CtxMag0] = e0; ent[CtxMag1] = e1; ent[CtxMag2] = e2;
        mq.save(st);
        *stp = st;
    }
}

#include "t1_lanes.inc"
size_t t1_dec_split_bytes(size_t njobs) { return t1_dec_lanes_bytes(njobs); }

// ---- stand-alone coders: the reference's MQEncoder / MQDecoder / RawEncoder / RawDecoder as batch calls ----
// mqc.go:169-349: NewMQEncoder, n x Encode(ctx, decision), Flush.  One wavefront, the coder on lane 0 (a single chain).
__global__ __launch_bounds__(64) void mq_encode_kernel(const uint8_t *__restrict__ ctxs, const uint8_t *__restrict__ decs, size_t n,
                                                       uint8_t *__restrict__ out, long cap, uint32_t *__restrict__ out_len, int *__restrict__ fault) {
    __shared__ T1Tables T;
    build_tables(T, 0, threadIdx.x);
    __syncthreads();
    if (threadIdx.x != 0) return;
    MqEnc e{0x8000, 0, 12, 0, 0, out, cap, 0};
    for (size_t i = 0; i < n; i++) {
        const int c = ctxs[i];
        if (c >= NumContexts) { atomicMax(fault, 3); *out_len = 0; return; }      // Go: index out of range panic
        mq_encode(e, T, c, decs[i]);                                              // uint8(decision) == mps: a decision > 1 never matches
    }
    const uint32_t tempC = e.C + e.A;                                             // setbits + Flush, mqc.go:313-341
    e.C |= 0xFFFF;
    if (e.C >= tempC) e.C -= 0x8000;
    e.C <<= e.CT; mq_byte_out(e);
    e.C <<= e.CT; mq_byte_out(e);
    long end = e.bp + 1;
    if (e.cur == 0xFF) end--;
    else if (e.bp >= 1) { if (e.bp - 1 < e.cap) out[e.bp - 1] = (uint8_t)e.cur; else e.overflow = 1; }
    if (e.overflow) atomicMax(fault, 2);
    *out_len = end > 1 ? (uint32_t)(end - 1) : 0;
}

// mqc.go:352-497: NewMQDecoder(data), n x Decode(ctx)
__global__ __launch_bounds__(64) void mq_decode_kernel(const uint8_t *__restrict__ data, long len, const uint8_t *__restrict__ ctxs, size_t n,
                                                       uint8_t *__restrict__ decs, int *__restrict__ fault) {
    __shared__ T1Tables T;
    build_tables(T, 0, threadIdx.x);
    __syncthreads();
    init_dec_contexts(T, threadIdx.x);
    __syncthreads();
    if (threadIdx.x != 0) return;
    MqDec d{0, 0x8000, 0, -1, len, data};
    if (d.len == 0) d.C = 0xFFu << 16; else { d.bp = 0; d.C = (uint32_t)d.data[0] << 16; }
    mq_byte_in(d);
    d.C <<= 7; d.CT -= 7; d.A = 0x8000;
    for (size_t i = 0; i < n; i++) {
        const int c = ctxs[i];
        if (c >= NumContexts) { atomicMax(fault, 3); return; }
        decs[i] = (uint8_t)mq_decode(d, T, c);
    }
}

// mqc.go:560-600: RawEncoder.EncodeBit x n + Flush (bit stuffing: after an 0xFF byte the next byte carries 7 bits)
__global__ void raw_encode_kernel(const uint8_t *__restrict__ bits, size_t n, uint8_t *__restrict__ out, size_t cap,
                                  uint32_t *__restrict__ out_len, int *__restrict__ fault) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t c = 0;
    int ct = 8;
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        ct--;
        c += (uint32_t)(bits[i] & 1) << ct;
        if (ct == 0) {
            if (pos < cap) out[pos] = (uint8_t)c; else atomicMax(fault, 2);
            pos++;
            ct = ((uint8_t)c == 0xFF) ? 7 : 8;
            c = 0;
        }
    }
    if (ct < 8) { if (pos < cap) out[pos] = (uint8_t)c; else atomicMax(fault, 2); pos++; }
    *out_len = (uint32_t)pos;
}

// mqc.go:516-557: RawDecoder.DecodeBit x n
__global__ void raw_decode_kernel(const uint8_t *__restrict__ data, size_t len, size_t n, uint8_t *__restrict__ bits) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    size_t pos = 0;
    uint32_t c = 0;
    int ct = 0;
    for (size_t i = 0; i < n; i++) {
        if (ct == 0) {
            if (c == 0xFF) {
                if (pos < len && data[pos] > 0x8F) { c = 0xFF; ct = 8; }
                else if (pos < len) { c = data[pos++]; ct = 7; }
                else { c = 0xFF; ct = 8; }
            } else {
                if (pos < len) { c = data[pos++]; ct = 8; }
                else { c = 0xFF; ct = 8; }
            }
        }
        ct--;
        bits[i] = (uint8_t)((c >> ct) & 1);
    }
}

hipError_t launch_mq_encode(hipStream_t s, const uint8_t *ctxs, const uint8_t *decs, size_t n, uint8_t *out, size_t cap, uint32_t *out_len, int *fault) {
    hipLaunchKernelGGL(mq_encode_kernel, dim3(1), dim3(64), 0, s, ctxs, decs, n, out, (long)cap, out_len, fault);
    return hipGetLastError();
}
hipError_t launch_mq_decode(hipStream_t s, const uint8_t *data, size_t len, const uint8_t *ctxs, size_t n, uint8_t *decs, int *fault) {
    hipLaunchKernelGGL(mq_decode_kernel, dim3(1), dim3(64), 0, s, data, (long)len, ctxs, n, decs, fault);
    return hipGetLastError();
}
hipError_t launch_raw_encode(hipStream_t s, const uint8_t *bits, size_t n, uint8_t *out, size_t cap, uint32_t *out_len, int *fault) {
    hipLaunchKernelGGL(raw_encode_kernel, dim3(1), dim3(64), 0, s, bits, n, out, cap, out_len, fault);
    return hipGetLastError();
}
hipError_t launch_raw_decode(hipStream_t s, const uint8_t *data, size_t len, size_t n, uint8_t *bits) {
    hipLaunchKernelGGL(raw_decode_kernel, dim3(1), dim3(64), 0, s, data, len, n, bits);
    return hipGetLastError();
}

// gfx950 has 160 KB of LDS per workgroup; above 64 KB of dynamic LDS the kernel attribute has to be raised first.  The
// reference's default 256 x 256 code-block (encoder.go:606-607) has 67 KB of decoder flags: with them in LDS instead of
// global memory every flag access of the lane-0 chain is a ds_read instead of a memory round trip.
#define T1_LDS_BIG_LIMIT (150 * 1024)
static int lds_for(size_t work_per_job, size_t limit = T1_LDS_LIMIT) {
    const size_t tab = (sizeof(T1Tables) + 15) & ~size_t(15);
    size_t wb = work_per_job;
    if (tab + wb > limit) wb = 0;              // largest block does not fit: LDS holds the tables only
    return (int)wb;
}

// symbol workspace of the split path: bytes per job for blocks up to 64x64 with `planes` bit planes.  A block emits at
// most one decision per sample and plane, one sign per sample, and two extra symbols per run-length hit (a 4-sample
// column can be hit once): (planes + 1.5) * 4096, plus the padding of each drained chunk.
size_t t1_sym_stride(int planes) { return ((size_t)(planes + 2) * 4096 + 1024 + 15) & ~size_t(15); }

// max_dim: largest block width or height among the jobs (<= 64: every block takes the wave-parallel kernels).
// sym != null: context formation and MQ coding as two kernels through the symbol workspace (sym_stride bytes per job,
// nsyms = njobs words); blocks with more bit planes than the stride allows fall back to the one-kernel path.
hipError_t launch_t1_encode(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *slots,
                            uint32_t *lens, uint8_t *numbps, uint8_t *work, size_t work_per_job, int *fault, int max_dim,
                            uint8_t *sym, size_t sym_stride, uint32_t *nsyms, int lanes, uint8_t *bigsym, const uint64_t *bigsym_off, uint32_t *bignsyms) {
    if (njobs <= 0) return hipSuccess;
    static int serial_only = -1;   // J2K_T1_SERIAL=1: A/B against the serial kernel
    if (serial_only < 0) serial_only = 0;          // (round 1's A/B switch J2K_T1_SERIAL: gone, the serial kernel only takes what the others leave)
    if (!serial_only) {
        if (sym && nsyms) {
            int planes = (int)((sym_stride - 1024) / 4096) - 2;
            if (planes > 31) planes = 31;
            hipLaunchKernelGGL(t1_encode64_kernel<true>, dim3(njobs), dim3(64), 0, s, jobs, njobs, coef, slots, lens, numbps, fault,
                               sym, sym_stride, nsyms, planes, (const uint32_t *)nullptr);
            // blocks per wavefront: the kernel runs at the latency of one chain whatever K is, so K decides how much of the
            // device a frame's chains occupy while they run.  Measured on a 4K 12-bit frame (7005 blocks): one frame alone
            // 15.8 ms at K = 4, 18.0 at K = 32 (each wavefront waits for its slowest lane); three or more frames in flight
            // 34.8 ms per frame at K = 4, 28.3 at K = 32 -- the chains of one frame then run beside the other frames' decode
            // kernels, which are issue-bound.  lanes > 0: as given (J2K_T1_LANES); 0: latency (blocks / 2048); < 0: throughput
            // (blocks / 256, at most 32) -- the caller passes < 0 while several contexts code with the MQ coder.
            // With the lanes ordered by symbol count (t1_order_kernel; `lane_order`, J2K_T1_ENC_ORDER=0 turns it off) a wavefront's
            // chains end together, and the throughput setting fills all 64 lanes.
            static int lane_order = -1;
            if (lane_order < 0) lane_order = 1;            // (J2K_T1_ENC_ORDER: measured in round 3, ordered lanes kept)
            int K = lanes > 0 ? lanes : (lanes < 0 ? std::min(lane_order ? 64 : 32, (njobs + 255) / 256) : (njobs + 2047) / 2048);
            K = std::min(64, std::max(1, K));
            uint32_t *perm = lane_order ? nsyms + njobs : nullptr;       // njobs + 64 words behind nsyms (t1_workspace in j2k_stages.cpp)
            if (perm) hipLaunchKernelGGL(t1_order_kernel, dim3(1), dim3(1024), 0, s, jobs, njobs, (const uint8_t *)nullptr, (const uint32_t *)nsyms, 1, perm,
                                         (uint32_t *)nullptr, njobs);
            hipLaunchKernelGGL(t1_mq_lanes_kernel, dim3(((njobs + K - 1) / K + T1_LANES_WPW - 1) / T1_LANES_WPW), dim3(64 * T1_LANES_WPW), 0, s, jobs, njobs, K, sym, sym_stride, nsyms,
                               slots, lens, fault, (const uint32_t *)perm);
            if (planes < 31)
                hipLaunchKernelGGL(t1_encode64_kernel<false>, dim3(njobs), dim3(64), 0, s, jobs, njobs, coef, slots, lens, numbps, fault,
                                   (uint8_t *)nullptr, (size_t)0, (uint32_t *)nullptr, 0, (const uint32_t *)nsyms);
        } else {
            hipLaunchKernelGGL(t1_encode64_kernel<false>, dim3(njobs), dim3(64), 0, s, jobs, njobs, coef, slots, lens, numbps, fault,
                               (uint8_t *)nullptr, (size_t)0, (uint32_t *)nullptr, 0, (const uint32_t *)nullptr);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || max_dim <= 64) return e;
        // blocks above 64 x 64, up to 256 x 256 (the reference's default size): wave-parallel context formation (t1_big.inc)
        const uint32_t *marked = nullptr;      // the two-kernel form's overflow marks: those blocks take the serial kernel below (its workspace is global memory: no LDS to wait for)
        static int big_on = -1;        // J2K_T1_BIG=0: A/B against the serial kernel
        if (big_on < 0) { const char *en = tuning_env("J2K_T1_BIG"); big_on = en ? atoi(en) : 1; }
        if (big_on) {
            static bool raised = false;
            if (!raised) {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(t1_encode_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(T1Big));
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(t1_encode_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(T1Big));
                if (e != hipSuccess) return e;
                raised = true;
            }
            int rot = 1;               // J2K_T1_BIG_ROT=0: the chain stays on wave 0 (A/B)
            { const char *en = tuning_env("J2K_T1_BIG_ROT"); if (en) rot = atoi(en); }
            int split = 1;             // J2K_T1_BIG_SPLIT=0: context formation and MQ chain in one kernel (round 3's form; A/B)
            { const char *en = tuning_env("J2K_T1_BIG_SPLIT"); if (en) split = atoi(en); }
            if (split && bigsym && bigsym_off && bignsyms) {
                // two kernels through symbol lists in global memory, so that a block holds its LDS and four waves for the milliseconds of
                // context formation only and its chain -- one wave, no LDS -- is resident beside thousands of others; a block whose
                // symbols do not fit its list (16 per sample) is marked and takes the serial kernel afterwards
                hipLaunchKernelGGL(t1_encode_big_kernel<true>, dim3(njobs), dim3(256), sizeof(T1Big), s, jobs, njobs, coef, slots, lens, numbps, fault, rot, bigsym, bigsym_off, bignsyms);
                hipLaunchKernelGGL(t1_mq_big_kernel, dim3(njobs), dim3(rot ? 256 : 64), 0, s, jobs, njobs, slots, lens, fault, rot, (const uint8_t *)bigsym, bigsym_off, (const uint32_t *)bignsyms);
                marked = bignsyms;
            } else {
                hipLaunchKernelGGL(t1_encode_big_kernel<false>, dim3(njobs), dim3(256), sizeof(T1Big), s, jobs, njobs, coef, slots, lens, numbps, fault, rot, (uint8_t *)nullptr, (const uint64_t *)nullptr, (uint32_t *)nullptr);
            }
            e = hipGetLastError();
            if (e != hipSuccess || (max_dim <= 256 && !marked)) return e;
        }
        const int skip = big_on ? 256 : 64;
        const int wb = lds_for(work_per_job);
        const size_t lds = ((sizeof(T1Tables) + 15) & ~size_t(15)) + (size_t)wb;
        if (wb) hipLaunchKernelGGL(t1_encode_kernel<true>, dim3(njobs), dim3(64), lds, s, jobs, njobs, coef, slots, lens, numbps, work,
                                   work_per_job, wb, fault, skip, marked);
        else hipLaunchKernelGGL(t1_encode_kernel<false>, dim3(njobs), dim3(64), lds, s, jobs, njobs, coef, slots, lens, numbps, work,
                                work_per_job, wb, fault, skip, marked);
        return hipGetLastError();
    }
    const int wb = lds_for(work_per_job);
    const size_t lds = ((sizeof(T1Tables) + 15) & ~size_t(15)) + (size_t)wb;
    if (wb) hipLaunchKernelGGL(t1_encode_kernel<true>, dim3(njobs), dim3(64), lds, s, jobs, njobs, coef, slots, lens, numbps, work,
                               work_per_job, wb, fault, 0, (const uint32_t *)nullptr);
    else hipLaunchKernelGGL(t1_encode_kernel<false>, dim3(njobs), dim3(64), lds, s, jobs, njobs, coef, slots, lens, numbps, work,
                            work_per_job, wb, fault, 0, (const uint32_t *)nullptr);
    return hipGetLastError();
}

// general_only: every block on the general kernel (A/B knob); otherwise blocks up to 64x64 take t1_decode64_kernel
hipError_t launch_t1_decode(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs,
                            const uint32_t *lens, const uint8_t *numbps, int32_t *decoded, uint8_t *work, size_t work_per_job,
                            int max_dim, int general_only, uint8_t *split_ws, int sig_lanes, int throughput) {
    if (njobs <= 0) return hipSuccess;
    if (!general_only) {
        if (split_ws) {
            // plane-stepped path for blocks of at most 31 planes; the one-launch kernel keeps the deeper ones
            // sig_lanes (J2K_T1_DEC_LANES): 2 = the whole plane-stepped decode of a group of 64 blocks in ONE launch (t1_lanes.inc,
            // PERSIST); 1 = its passes as separate launches per plane (significance lanes, plane work, MagRef lanes); 0 = round 2's
            // step kernels (one block per wavefront) + MagRef lanes
            uint64_t *masks = reinterpret_cast<uint64_t *>(split_ws + t1_dec_lanes_mask_offset((size_t)njobs));
            uint64_t *planes = reinterpret_cast<uint64_t *>(split_ws + t1_dec_lanes_planes_offset((size_t)njobs));
            const int ngroups = (njobs + 63) / 64;
            const int nwg = (ngroups + T1_LANES_WPW - 1) / T1_LANES_WPW;
            uint32_t *perm = reinterpret_cast<uint32_t *>(split_ws + t1_dec_lanes_perm_offset((size_t)njobs)), *slot_of = perm + (size_t)ngroups * 64;
            if (sig_lanes) hipLaunchKernelGGL(t1_order_kernel, dim3(1), dim3(1024), 0, s, jobs, njobs, numbps, lens, 0, perm, slot_of, ngroups * 64);
            if (sig_lanes >= 2) {
                hipLaunchKernelGGL(t1_dec_sig_lanes_kernel<true>, dim3(nwg), dim3(64 * T1_LANES_WPW), 0, s, jobs, njobs, stream, offs, lens, numbps,
                                   split_ws, masks, (const uint32_t *)perm, planes, 0);
            } else
            for (int k = 0; k <= T1DS_MAXP; k++) {
                if (sig_lanes) {
                    hipLaunchKernelGGL(t1_dec_sig_lanes_kernel<false>, dim3(nwg), dim3(64 * T1_LANES_WPW), 0, s, jobs, njobs, stream, offs, lens, numbps,
                                       split_ws, masks, (const uint32_t *)perm, planes, k);
                    hipLaunchKernelGGL(t1_dec_plane_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, numbps, decoded, split_ws, masks, slot_of, planes, k);
                } else {
                    hipLaunchKernelGGL(t1_dec_step_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, stream, offs, lens, numbps, decoded, split_ws, k);
                }
                if (k < T1DS_MAXP) {
                    if (sig_lanes) { if (k > 0) hipLaunchKernelGGL(t1_dec_magref_lanes_kernel<true>, dim3(nwg), dim3(64 * T1_LANES_WPW), 0, s, jobs, njobs, stream, offs, numbps, split_ws, (const uint32_t *)perm, k); }
                    else hipLaunchKernelGGL(t1_dec_magref_lanes_kernel<false>, dim3(nwg), dim3(64 * T1_LANES_WPW), 0, s, jobs, njobs, stream, offs, numbps, split_ws, (const uint32_t *)nullptr, k);
                }
            }
        }
        if (split_ws && sig_lanes)
            hipLaunchKernelGGL(t1_dec_assemble_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, numbps, decoded,
                               reinterpret_cast<const uint64_t *>(split_ws + t1_dec_lanes_mask_offset((size_t)njobs)),
                               reinterpret_cast<const uint32_t *>(split_ws + t1_dec_lanes_perm_offset((size_t)njobs)) + (size_t)((njobs + 63) / 64) * 64,
                               reinterpret_cast<const uint64_t *>(split_ws + t1_dec_lanes_planes_offset((size_t)njobs)));
        hipLaunchKernelGGL(t1_decode64_kernel, dim3(njobs), dim3(64), 0, s, jobs, njobs, stream, offs, lens, numbps, decoded,
                           split_ws ? T1DS_MAXP + 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || max_dim <= 64) return e;
    }
    // blocks above 64 x 64, up to 256 x 256 (the reference's default size): the wave-uniform decoder (t1_bigdec.inc)
    int big_dec = 1;                   // J2K_T1_BIG_DEC=0: A/B against the general kernel (read per call: only frames with such blocks get here)
    { const char *en = tuning_env("J2K_T1_BIG_DEC"); if (en) big_dec = atoi(en); }
    const bool use_big = big_dec && !general_only;
    if (use_big) {
        int rot = 1;                   // J2K_T1_BIG_ROT=0: one wave per block, wherever the dispatcher puts it (A/B)
        { const char *en = tuning_env("J2K_T1_BIG_ROT"); if (en) rot = atoi(en); }
        // two sizes of block state (t1_bigdec.inc): each launch takes the blocks of its size
        // throughput (several MQ contexts: frames in flight): two launches, so that the small blocks hold 10 KB of LDS instead of 41 --
        // one after the other on this stream, beside the other streams' launches; one context alone: one launch with the large state
        // for every block (the blocks of a frame decode side by side: latency)
        int classes = throughput;
        { const char *en = tuning_env("J2K_T1_BIG_DEC_CLASSES"); if (en) classes = atoi(en); }
        hipLaunchKernelGGL((t1_decode_big_kernel<4, 258>), dim3(njobs), dim3(rot ? 256 : 64), sizeof(T1BigDec<4, 258>), s, jobs, njobs, stream, offs, lens, numbps, decoded, rot, classes ? 0 : 1);
        if (classes) hipLaunchKernelGGL((t1_decode_big_kernel<2, 130>), dim3(njobs), dim3(rot ? 256 : 64), sizeof(T1BigDec<2, 130>), s, jobs, njobs, stream, offs, lens, numbps, decoded, rot, 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || max_dim <= 256) return e;
    }
    const int skip = general_only ? 0 : (use_big ? 256 : 64);
    const int wb = lds_for(work_per_job, T1_LDS_BIG_LIMIT);     // work_per_job = flag bytes of the largest block
    const size_t lds = ((sizeof(T1Tables) + 15) & ~size_t(15)) + (size_t)wb;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(t1_decode_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, T1_LDS_BIG_LIMIT);
            if (e != hipSuccess) return e;
            raised = true;
        }
    }
    if (wb) hipLaunchKernelGGL(t1_decode_kernel<true>, dim3(njobs), dim3(64), lds, s, jobs, njobs, stream, offs, lens, numbps, decoded,
                               work, work_per_job, wb, skip);
    else hipLaunchKernelGGL(t1_decode_kernel<false>, dim3(njobs), dim3(64), lds, s, jobs, njobs, stream, offs, lens, numbps, decoded,
                            work, work_per_job, wb, skip);
    return hipGetLastError();
}

}  // namespace j2k
