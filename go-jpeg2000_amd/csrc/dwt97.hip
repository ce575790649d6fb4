// dwt97.hip -- 9-7 irreversible lifting DWT (float64, like the reference) for gfx950, fused with
// DC shift + ICT + the encoder's quantisation on the way in and the decoder's rounding + inverse
// ICT + DC shift on the way out.
//
// Replaces (reference, mrjoshuak/go-jpeg2000):
//   dwt.Forward97/Inverse97                 internal/dwt/dwt.go:161-262 (constants :150-157)
//   dwt.Forward2D97/Inverse2D97             dwt.go:432-473
//   one level of Decompose/ReconstructMultiLevel97   dwt.go:551-573
//   encoder.preprocess lossy branch         encoder.go:227-244 (ICT + round half away), 259-276 (v/step +- 0.5)
//   tcd ApplyForwardDWT / ApplyInverseDWT   internal/tcd/tcd.go:520-532, 428-435
//   decoder.decodeTiles lossy tail          decoder.go:326-339 (InverseICT, int32(v+0.5)), 344-348
//
// Same streaming structure as dwt53.hip: a wavefront owns a column strip x a band of row pairs;
// horizontal neighbours by DPP wave shifts (4 per row: one per lifting step); the four vertical
// lifting steps are a software pipeline over row pairs held in registers (look-ahead of two pairs,
// so a band reads 7 halo rows).  Every sample is read once and written once per level.
//
// Bit-exactness vs Go/amd64: compiled with -ffp-contract=off (no FMA), the reference's literal
// constants, the same operand association; the symmetric-extension edge terms are computed as
// c*(x+x), which is bitwise equal to the reference's (2*c)*x (scaling by two is exact).
#include "j2k_internal.h"
#include <hip/hip_ext.h>

namespace j2k {

#define A97 (-1.586134342059924)
#define B97 (-0.052980118572961)
#define G97 (0.882911075530934)
#define D97 (0.443506852043971)
#define K97 (1.230174104914001)
#define K97I (0.812893066115961)

__device__ __forceinline__ double dshift(double v, int ctrl_shr) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl_shr) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dleft(double v) { return dshift(v, 1); }    // lane i <- lane i-1
__device__ __forceinline__ double dright(double v) { return dshift(v, 0); }   // lane i <- lane i+1

// Go int32(float64) on amd64: the compiler lowers it to CVTTSD2SL -- truncation toward zero, and the x86 "integer indefinite"
// 0x80000000 for NaN and every value whose truncation does not fit.  v_cvt_i32_f64 saturates: the same at the negative end,
// INT_MAX at the positive one, 0 for NaN -- one comparison puts those right.  (Round 2 had a 64-bit convert + truncation
// here; the oracle's plain C casts compile to CVTTSD2SI r32 like Go's, and the new out-of-range test showed the difference.)
__device__ __forceinline__ int go_int32(double v) { return v < 2147483648.0 ? (int)v : (int)0x80000000; }
__device__ __forceinline__ int round_half_away(double v) { return v >= 0 ? go_int32(v + 0.5) : go_int32(v - 0.5); }

enum { SRC_I32 = 0, SRC_F64 = 1 };
enum { Q_NONE_ = 0, Q_ENCODER_ = 1, Q_TCD_ = 2 };
enum { DST_F64_SCRATCH = 0, DST_F64_FRAME = 1, DST_I32_FRAME = 2 };

template <int CPL, int NC> struct Row97 { double lo[NC][CPL / 2]; double hi[NC][CPL / 2]; };

// ---- horizontal forward: x[CPL] (columns c..) -> lo/hi (scaled) ----------------------------------
template <int CPL>
__device__ __forceinline__ void hfwd97(const double (&x)[CPL], int c, int w, double (&lo)[CPL / 2], double (&hi)[CPL / 2]) {
    constexpr int H = CPL / 2;
    if (w < 2) {
#pragma unroll
        for (int j = 0; j < H; j++) { lo[j] = x[2 * j]; hi[j] = 0.0; }
        return;
    }
    double d1[H], s1[H], d2[H];
    const double e_r = dright(x[0]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        const double en = (ce + 2 < w) ? ((j + 1 < H) ? x[2 * j + 2] : e_r) : x[2 * j];
        d1[j] = x[2 * j + 1] + A97 * (x[2 * j] + en);
    }
    const double d1_l = dleft(d1[H - 1]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        double dp = (j > 0) ? d1[j - 1] : d1_l;
        if (ce + 1 >= w) d1[j] = dp;           // no odd partner: mirrors d[n-2]
        if (ce == 0) dp = d1[j];
        s1[j] = x[2 * j] + B97 * (dp + d1[j]);
    }
    const double s1_r = dright(s1[0]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        const double sn = (ce + 2 < w) ? ((j + 1 < H) ? s1[j + 1] : s1_r) : s1[j];
        d2[j] = d1[j] + G97 * (s1[j] + sn);
    }
    const double d2_l = dleft(d2[H - 1]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        double dp = (j > 0) ? d2[j - 1] : d2_l;
        if (ce + 1 >= w) d2[j] = dp;
        if (ce == 0) dp = d2[j];
        lo[j] = (s1[j] + D97 * (dp + d2[j])) * K97I;
        hi[j] = d2[j] * K97;
    }
}

// ---- horizontal inverse: lo/hi -> x[CPL] -----------------------------------------------------------
template <int CPL>
__device__ __forceinline__ void hinv97(const double (&lo)[CPL / 2], const double (&hi)[CPL / 2], int c, int w, double (&x)[CPL]) {
    constexpr int H = CPL / 2;
    if (w < 2) {
#pragma unroll
        for (int j = 0; j < H; j++) { x[2 * j] = lo[j]; x[2 * j + 1] = 0.0; }
        return;
    }
    double d2[H], s1[H], d1[H], e[H];
#pragma unroll
    for (int j = 0; j < H; j++) d2[j] = hi[j] * K97I;
    const double d2_l = dleft(d2[H - 1]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        double dp = (j > 0) ? d2[j - 1] : d2_l;
        if (ce + 1 >= w) d2[j] = dp;
        if (ce == 0) dp = d2[j];
        s1[j] = lo[j] * K97 - D97 * (dp + d2[j]);
    }
    const double s1_r = dright(s1[0]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        const double sn = (ce + 2 < w) ? ((j + 1 < H) ? s1[j + 1] : s1_r) : s1[j];
        d1[j] = d2[j] - G97 * (s1[j] + sn);
    }
    const double d1_l = dleft(d1[H - 1]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        double dp = (j > 0) ? d1[j - 1] : d1_l;
        if (ce + 1 >= w) d1[j] = dp;
        if (ce == 0) dp = d1[j];
        e[j] = s1[j] - B97 * (dp + d1[j]);
    }
    const double e_r = dright(e[0]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        const double en = (ce + 2 < w) ? ((j + 1 < H) ? e[j + 1] : e_r) : e[j];
        x[2 * j] = e[j];
        x[2 * j + 1] = d1[j] - A97 * (e[j] + en);
    }
}

// ================================================================================
// forward
// ================================================================================
template <int CPL, int NC>
__device__ __forceinline__ void fwd97_load_row(const void *__restrict__ src, int src_f64, const DwtPlane &P, int r, int c,
                                               int dc_shift, int mct, Row97<CPL, NC> &R) {
    double x[NC][CPL];
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int64_t base = P.src_off[k] + (int64_t)r * P.src_stride;
        if (src_f64) {
            // columns past the row end read a clamped (valid) address: their values never reach an in-range result
            // (every neighbour use in hfwd97 is guarded by the row length), so the loads need no branch
            const double *p = reinterpret_cast<const double *>(src) + base;
#pragma unroll
            for (int i = 0; i < CPL; i++) x[k][i] = p[min(c + i, P.w - 1)];
        } else {
            const int32_t *p = reinterpret_cast<const int32_t *>(src) + base;
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const int v = p[min(c + i, P.w - 1)];
                x[k][i] = (double)(int)((unsigned)v - (unsigned)dc_shift);   // mct.go:96-101, encoder.go:228-233/260-262
            }
        }
    }
    if constexpr (NC == 3) {
        if (mct) {   // mct.go:14-24 then round half away to int32 and back to f64 (encoder.go:235-244, 259-262)
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const double r_ = x[0][i], g_ = x[1][i], b_ = x[2][i];
                const double y = 0.299 * r_ + 0.587 * g_ + 0.114 * b_;
                const double cb = -0.16875 * r_ - 0.33126 * g_ + 0.5 * b_;
                const double cr = 0.5 * r_ - 0.41869 * g_ - 0.08131 * b_;
                x[0][i] = (double)round_half_away(y);
                x[1][i] = (double)round_half_away(cb);
                x[2][i] = (double)round_half_away(cr);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) hfwd97<CPL>(x[k], c, P.w, R.lo[k], R.hi[k]);
}

template <int CPL, int NC>
__device__ __forceinline__ void fwd97_store_row(int32_t *__restrict__ out_i32, double *__restrict__ out_f64, double *__restrict__ nxt,
                                                const DwtPlane &P, int ro, int p0, bool owned, int quant, double step,
                                                const double (&lo)[NC][CPL / 2], const double (&hi)[NC][CPL / 2]) {
    constexpr int H = CPL / 2;
    if (!owned) return;
    const int halfW = (P.w + 1) >> 1;
    const int nL = halfW - p0, nH = (P.w - halfW) - p0;
    const int idxL = ro * P.w + p0, idxH = idxL + halfW;
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int j = 0; j < 2 * H; j++) {
            const bool is_lo = j < H;
            const int jj = is_lo ? j : j - H;
            if (jj >= (is_lo ? nL : nH)) continue;
            const int idx = (is_lo ? idxL : idxH) + jj;
            const double v = is_lo ? lo[k][jj] : hi[k][jj];
            if (idx < P.n_next) nxt[P.nxt_off[k] + idx] = v;
            else if (quant == Q_NONE_) out_f64[P.out_off[k] + idx] = v;
            else if (quant == Q_ENCODER_) out_i32[P.out_off[k] + idx] = v >= 0 ? go_int32(v / step + 0.5) : go_int32(v / step - 0.5);   // encoder.go:270-275
            else out_i32[P.out_off[k] + idx] = round_half_away(v);                                                                        // tcd.go:526-531
        }
}

template <int CPL, int NC>
__global__ __launch_bounds__(256) void dwt97_fwd_kernel(const DwtJob *__restrict__ jobs, int njobs, const DwtPlane *__restrict__ planes,
                                                        const void *__restrict__ src, int src_f64, int32_t *__restrict__ out_i32,
                                                        double *__restrict__ out_f64, double *__restrict__ nxt, int dc_shift,
                                                        int quant, double step, int mct) {
    constexpr int H = CPL / 2;
    constexpr int HL = (H >= 2) ? 1 : 2;   // halo lanes per side
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wave >= njobs) return;
    const int lane = threadIdx.x & 63;
    const DwtJob job = jobs[wave];
    if (job.plane < 0) return;   // padding entry of the XCD-aware job order
    const DwtPlane P = planes[job.plane];
    const int w = P.w, h = P.h;
    const int lane_first = (job.col0 == 0) ? 0 : HL;
    const int c_base = job.col0 - lane_first * CPL;
    const int c = c_base + lane * CPL;
    const bool reach_end = (c_base + 64 * CPL >= w);
    const bool owned = (lane >= lane_first) && (c < w) && (reach_end || lane < 64 - HL);
    const int p0 = c >> 1;
    const int halfH = (h + 1) >> 1;
    const int q0 = job.prow0, q1 = min(job.prow0 + job.nprow, halfH);
    typedef Row97<CPL, NC> Row;

    if (h < 2) {   // columns untouched (no scaling either, dwt.go:162-164)
        if (q0 == 0) {
            Row r0;
            fwd97_load_row<CPL, NC>(src, src_f64, P, 0, c, dc_shift, mct, r0);
            fwd97_store_row<CPL, NC>(out_i32, out_f64, nxt, P, 0, p0, owned, quant, step, r0.lo, r0.hi);
        }
        return;
    }
    const int t_start = max(q0 - 2, 0);
    const int t_last = min(q1, halfH);
    Row e, o, en;                       // rows 2t, 2t+1, 2t+2 after the horizontal pass
    double d1p[2][NC][H], s1p[2][NC][H], d2p[2][NC][H];   // [0]=lo half, [1]=hi half; values of pair t-1 (d2p: pair t-2)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) { d1p[a][k][j] = 0.0; s1p[a][k][j] = 0.0; d2p[a][k][j] = 0.0; }
    fwd97_load_row<CPL, NC>(src, src_f64, P, 2 * t_start, c, dc_shift, mct, e);
    for (int t = t_start; t <= t_last; t++) {
        const bool real = t < halfH;
        const bool o_ex = real && (2 * t + 1 < h), en_ex = real && (2 * t + 2 < h);
        if (real) {   // rows clamped into the plane: both rows' loads are issued back to back; o_ex / en_ex discard what is not there
            fwd97_load_row<CPL, NC>(src, src_f64, P, min(2 * t + 1, h - 1), c, dc_shift, mct, o);
            fwd97_load_row<CPL, NC>(src, src_f64, P, min(2 * t + 2, h - 1), c, dc_shift, mct, en);
        }
        const bool prev_o_ex = (t >= 1) && (2 * (t - 1) + 1 < h);
        double outl[NC][H], outh[NC][H], outl2[NC][H], outh2[NC][H];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {
                    const double ev = a ? e.hi[k][j] : e.lo[k][j];
                    const double ov = a ? o.hi[k][j] : o.lo[k][j];
                    const double env = a ? en.hi[k][j] : en.lo[k][j];
                    double d1t, s1t;
                    if (real) {
                        d1t = o_ex ? ov + A97 * (ev + (en_ex ? env : ev)) : d1p[a][k][j];
                        s1t = ev + B97 * ((t == 0 ? d1t : d1p[a][k][j]) + d1t);
                    } else {
                        d1t = d1p[a][k][j];
                        s1t = s1p[a][k][j];     // mirror: s1[q+1] := s1[q]
                    }
                    if (t >= 1) {               // finalize pair t-1
                        const double d2 = prev_o_ex ? d1p[a][k][j] + G97 * (s1p[a][k][j] + s1t) : d2p[a][k][j];
                        const double s2 = s1p[a][k][j] + D97 * ((t - 1 == 0 ? d2 : d2p[a][k][j]) + d2);
                        if (a == 0) { outl[k][j] = s2 * K97I; outl2[k][j] = d2 * K97; }
                        else { outh[k][j] = s2 * K97I; outh2[k][j] = d2 * K97; }
                        d2p[a][k][j] = d2;
                    }
                    d1p[a][k][j] = d1t;
                    s1p[a][k][j] = s1t;
                }
        if (t >= 1 && t - 1 >= q0) {
            fwd97_store_row<CPL, NC>(out_i32, out_f64, nxt, P, t - 1, p0, owned, quant, step, outl, outh);
            if (prev_o_ex) fwd97_store_row<CPL, NC>(out_i32, out_f64, nxt, P, halfH + t - 1, p0, owned, quant, step, outl2, outh2);
        }
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) { e.lo[k][j] = en.lo[k][j]; e.hi[k][j] = en.hi[k][j]; }
    }
}

// ================================================================================
// inverse
// ================================================================================
template <int CPL, int NC>
__device__ __forceinline__ void inv97_load_row(const void *__restrict__ coef, int coef_f64, const double *__restrict__ prev,
                                               const DwtPlane &P, int ri, int p0, Row97<CPL, NC> &R) {
    constexpr int H = CPL / 2;
    const int halfW = (P.w + 1) >> 1;
    const int nL = halfW - p0, nH = (P.w - halfW) - p0;
    const int idxL = ri * P.w + p0, idxH = idxL + halfW;
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int j = 0; j < 2 * H; j++) {
            const bool is_lo = j < H;
            const int jj = is_lo ? j : j - H;
            // branch-free: an out-of-range element reads the row's first element (valid) and is zeroed by a select; the
            // source (scratch of the coarser level / coefficient plane) is a pointer or value select
            const bool ok = jj < (is_lo ? nL : nH);
            const int idx = ok ? (is_lo ? idxL : idxH) + jj : ri * P.w;
            const bool from_prev = idx < P.n_next;
            double v;
            if (coef_f64) {
                const double *b = from_prev ? prev + P.nxt_off[k] : reinterpret_cast<const double *>(coef) + P.src_off[k];
                v = b[idx];
            } else {
                const double vp = prev[P.nxt_off[k] + (from_prev ? idx : 0)];
                const int vc = reinterpret_cast<const int32_t *>(coef)[P.src_off[k] + idx];                // tcd.go:429-431
                v = from_prev ? vp : (double)vc;
            }
            if (!ok) v = 0.0;
            if (is_lo) R.lo[k][jj] = v; else R.hi[k][jj] = v;
        }
}

template <int CPL, int NC>
__device__ __forceinline__ void inv97_finish_row(void *__restrict__ dst, const DwtPlane &P, int ro, int c, bool owned, int dc_shift,
                                                 int dst_mode, int mct, const double (&lo)[NC][CPL / 2], const double (&hi)[NC][CPL / 2]) {
    double x[NC][CPL];
#pragma unroll
    for (int k = 0; k < NC; k++) hinv97<CPL>(lo[k], hi[k], c, P.w, x[k]);
    if (!owned) return;
    if (dst_mode == DST_I32_FRAME) {
        int v[NC][CPL];
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int i = 0; i < CPL; i++) v[k][i] = go_int32(x[k][i] + 0.5);        // tcd.go:433-435 (negatives round toward +)
        if constexpr (NC == 3) {
            if (mct) {   // decoder.go:326-339 + mct.go:43-53
#pragma unroll
                for (int i = 0; i < CPL; i++) {
                    const double y = (double)v[0][i], cb = (double)v[1][i], cr = (double)v[2][i];
                    const double r_ = y + 1.402 * cr;
                    const double g_ = y - 0.34413 * cb - 0.71414 * cr;
                    const double b_ = y + 1.772 * cb;
                    v[0][i] = go_int32(r_ + 0.5); v[1][i] = go_int32(g_ + 0.5); v[2][i] = go_int32(b_ + 0.5);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            int32_t *p = reinterpret_cast<int32_t *>(dst) + P.out_off[k] + (int64_t)ro * P.out_stride + c;
#pragma unroll
            for (int i = 0; i < CPL; i++)
                if (c + i < P.w) p[i] = (int)((unsigned)v[k][i] + (unsigned)dc_shift);   // mct.go:113-118
        }
    } else {
        const int stride = (dst_mode == DST_F64_FRAME) ? P.out_stride : P.w;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            double *p = reinterpret_cast<double *>(dst) + P.out_off[k] + (int64_t)ro * stride + c;
#pragma unroll
            for (int i = 0; i < CPL; i++)
                if (c + i < P.w) p[i] = x[k][i];
        }
    }
}

template <int CPL, int NC>
__global__ __launch_bounds__(256) void dwt97_inv_kernel(const DwtJob *__restrict__ jobs, int njobs, const DwtPlane *__restrict__ planes,
                                                        const void *__restrict__ coef, int coef_f64, const double *__restrict__ prev,
                                                        void *__restrict__ dst, int dc_shift, int dst_mode, int mct) {
    constexpr int H = CPL / 2;
    constexpr int HL = (H >= 2) ? 1 : 2;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wave >= njobs) return;
    const int lane = threadIdx.x & 63;
    const DwtJob job = jobs[wave];
    if (job.plane < 0) return;   // padding entry of the XCD-aware job order
    const DwtPlane P = planes[job.plane];
    const int w = P.w, h = P.h;
    const int lane_first = (job.col0 == 0) ? 0 : HL;
    const int c_base = job.col0 - lane_first * CPL;
    const int c = c_base + lane * CPL;
    const bool reach_end = (c_base + 64 * CPL >= w);
    const bool owned = (lane >= lane_first) && (c < w) && (reach_end || lane < 64 - HL);
    const int p0 = c >> 1;
    const int halfH = (h + 1) >> 1;
    const int q0 = job.prow0, q1 = min(job.prow0 + job.nprow, halfH);
    typedef Row97<CPL, NC> Row;

    if (h < 2) {
        if (q0 == 0) {
            Row r0;
            inv97_load_row<CPL, NC>(coef, coef_f64, prev, P, 0, p0, r0);
            inv97_finish_row<CPL, NC>(dst, P, 0, c, owned, dc_shift, dst_mode, mct, r0.lo, r0.hi);
        }
        return;
    }
    // software pipeline over input pairs t: s1[t] <- (s2[t], d2[t-1], d2[t]); d1[t-1] <- (d2[t-1], s1[t-1], s1[t]);
    // e[t-1] <- (s1[t-1], d1[t-2], d1[t-1]); o[t-2] <- (d1[t-2], e[t-2], e[t-1]); pair t-2 is complete at step t.
    const int t_start = max(q0 - 2, 0);
    const int t_last = min(q1 + 1, halfH + 1);
    double d2p[2][NC][H], s1p[2][NC][H], d1p[2][NC][H], ep[2][NC][H], e2[2][NC][H];
    // d2p = d2[t-1], s1p = s1[t-1], d1p = d1[t-2], ep = e[t-1] (after update), e2 = e[t-2]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) { d2p[a][k][j] = 0.0; s1p[a][k][j] = 0.0; d1p[a][k][j] = 0.0; ep[a][k][j] = 0.0; e2[a][k][j] = 0.0; }
    for (int t = t_start; t <= t_last; t++) {
        const bool real = t < halfH;
        const bool hi_ex = real && (2 * t + 1 < h);
        Row L, Hh;
        if (real) {   // the high row is clamped into the plane (hi_ex discards it when it does not exist)
            inv97_load_row<CPL, NC>(coef, coef_f64, prev, P, t, p0, L);
            inv97_load_row<CPL, NC>(coef, coef_f64, prev, P, halfH + min(t, h - halfH - 1), p0, Hh);
        }
        const bool p1_real = (t >= 1) && (t - 1 < halfH);            // pair t-1 exists
        const bool p1_hi = p1_real && (2 * (t - 1) + 1 < h);
        const bool p2_real = (t >= 2) && (t - 2 < halfH);            // pair t-2 exists
        const bool p2_hi = p2_real && (2 * (t - 2) + 1 < h);
        double re_lo[NC][H], re_hi[NC][H], ro_lo[NC][H], ro_hi[NC][H];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {
                    double d2t = d2p[a][k][j], s1t = s1p[a][k][j];
                    if (real) {
                        const double s2t = (a ? L.hi[k][j] : L.lo[k][j]) * K97;
                        if (hi_ex) d2t = (a ? Hh.hi[k][j] : Hh.lo[k][j]) * K97I;     // else mirrors d2[t-1]
                        s1t = s2t - D97 * ((t == 0 ? d2t : d2p[a][k][j]) + d2t);
                    }
                    double d1n = d1p[a][k][j], en_ = ep[a][k][j];
                    if (p1_real) {
                        if (p1_hi) d1n = d2p[a][k][j] - G97 * (s1p[a][k][j] + (real ? s1t : s1p[a][k][j]));   // d1[t-1]; else mirrors d1[t-2]
                        en_ = s1p[a][k][j] - B97 * ((t - 1 == 0 ? d1n : d1p[a][k][j]) + d1n);               // e[t-1]
                    }
                    if (p2_real) {
                        const double e_t2 = e2[a][k][j];
                        const double on = d1p[a][k][j] - A97 * (e_t2 + (p1_real ? en_ : e_t2));             // o[t-2]
                        if (a == 0) { re_lo[k][j] = e_t2; ro_lo[k][j] = on; } else { re_hi[k][j] = e_t2; ro_hi[k][j] = on; }
                    }
                    e2[a][k][j] = ep[a][k][j];
                    if (p1_real) { ep[a][k][j] = en_; e2[a][k][j] = en_; }
                    d1p[a][k][j] = d1n;
                    d2p[a][k][j] = d2t;
                    s1p[a][k][j] = s1t;
                }
        if (p2_real && t - 2 >= q0 && t - 2 < q1) {
            inv97_finish_row<CPL, NC>(dst, P, 2 * (t - 2), c, owned, dc_shift, dst_mode, mct, re_lo, re_hi);
            if (p2_hi) inv97_finish_row<CPL, NC>(dst, P, 2 * (t - 2) + 1, c, owned, dc_shift, dst_mode, mct, ro_lo, ro_hi);
        }
    }
}

typedef int v4i __attribute__((ext_vector_type(4)));
#include "dwt97_l0wg.inc"
// waves per SIMD of the inverse workgroup kernel: 4 (128 VGPRs, a dozen of them spilled; two 8-wave workgroups per CU) measured
// 67 us on a 4K frame against 85 us at 3 (136 VGPRs, no spill, one workgroup per CU)
#ifndef J2K_WG97F_WPE
#define J2K_WG97F_WPE 7
#endif
#ifndef J2K_WG97I_WPE
#define J2K_WG97I_WPE 4
#endif
#include "dwt97_l0wg_inv.inc"

// ================================================================================
// launchers
// ================================================================================
template <int NW>
static hipError_t fwd97_wg_go(hipStream_t s, const LevelLaunch &L, const void *src, int32_t *out_i32, double *out_f64, double *nxt,
                              int dc_shift, int quant, double step) {
    const int32_t *p = reinterpret_cast<const int32_t *>(src);
    const double rstep = 1.0 / step;          // RN(1 / step): the reciprocal of the Markstein division (dwt97_l0wg.inc)
#define J2K_WG97(Q) hipExtLaunchKernelGGL((dwt97_fwd_rgb_wg_kernel<NW, Q, J2K_WG97F_WPE, 0>), dim3(L.njobs), dim3(NW * 64), 0, s, L.ev_start, L.ev_stop, 0, \
                                           L.jobs, L.njobs, L.planes, p, (const double *)nullptr, out_i32, out_f64, nxt, dc_shift, step, rstep, 0)
    if (L.pix_stride > 0) {      // packed RGBA8 pixels (j2k_plan_forward_pixels on a lossy plan): eight waves, the encoder's quantiser
        if (NW != 8 || quant != Q_ENCODER_) return hipErrorInvalidValue;
        hipExtLaunchKernelGGL((dwt97_fwd_rgb_wg_kernel<8, Q_ENCODER_, J2K_WG97F_WPE, 3>), dim3(L.njobs), dim3(512), 0, s, L.ev_start, L.ev_stop, 0,
                              L.jobs, L.njobs, L.planes, p, (const double *)nullptr, out_i32, out_f64, nxt, dc_shift, step, rstep, L.pix_stride);
        return hipGetLastError();
    }
    if (quant == Q_ENCODER_) J2K_WG97(Q_ENCODER_);
    else if (quant == Q_TCD_) J2K_WG97(Q_TCD_);
    else J2K_WG97(Q_NONE_);
#undef J2K_WG97
    return hipGetLastError();
}

hipError_t launch_dwt97_fwd(hipStream_t s, const LevelLaunch &L, const void *src, int src_is_f64, int32_t *out_i32, double *out_f64,
                            double *nxt, int dc_shift, int quant, double step, int mct) {
    if (L.njobs <= 0) return hipSuccess;
    if (L.wg_waves > 0) {      // level 0 of an int32 RGB triple with ICT, workgroup form (dwt97_l0wg.inc); geometry checked by the plan
        if (L.ncomp != 3 || src_is_f64 || !mct) return hipErrorInvalidValue;
        if (L.wg_waves == 6) return fwd97_wg_go<6>(s, L, src, out_i32, out_f64, nxt, dc_shift, quant, step);
        if (L.wg_waves == 8) return fwd97_wg_go<8>(s, L, src, out_i32, out_f64, nxt, dc_shift, quant, step);
        if (L.wg_waves == 10) return fwd97_wg_go<10>(s, L, src, out_i32, out_f64, nxt, dc_shift, quant, step);
        if (L.wg_waves == 12) return fwd97_wg_go<12>(s, L, src, out_i32, out_f64, nxt, dc_shift, quant, step);
        if (L.wg_waves == 14) return fwd97_wg_go<14>(s, L, src, out_i32, out_f64, nxt, dc_shift, quant, step);
        if (L.wg_waves == 16) return fwd97_wg_go<16>(s, L, src, out_i32, out_f64, nxt, dc_shift, quant, step);
        return hipErrorInvalidValue;
    }
    if (L.pwaves == 8 && L.pnjobs > 0 && L.ncomp == 1) {   // single planes in workgroup form: a deeper level (float64 in), level 0 of one int32 component, the float64 unit calls
        const double rstep = 1.0 / step;
#define J2K_PWG97(Q, SRC) hipExtLaunchKernelGGL((dwt97_fwd_rgb_wg_kernel<8, Q, 7, SRC>), dim3(L.pnjobs), dim3(512), 0, s, L.ev_start, L.ev_stop, 0, \
                                                 L.pjobs, L.pnjobs, L.planes, reinterpret_cast<const int32_t *>(src), reinterpret_cast<const double *>(src), out_i32, out_f64, nxt, dc_shift, step, rstep, src_is_f64 ? 0 : L.pix_stride)
        if (src_is_f64) {
            if (quant == Q_ENCODER_) J2K_PWG97(Q_ENCODER_, 1);
            else if (quant == Q_TCD_) J2K_PWG97(Q_TCD_, 1);
            else J2K_PWG97(Q_NONE_, 1);
        } else {
            if (quant == Q_ENCODER_) J2K_PWG97(Q_ENCODER_, 2);
            else if (quant == Q_TCD_) J2K_PWG97(Q_TCD_, 2);
            else J2K_PWG97(Q_NONE_, 2);
        }
#undef J2K_PWG97
        return hipGetLastError();
    }
    const int blocks = (L.njobs + 3) / 4;
    if (L.ncomp == 3) {
        hipExtLaunchKernelGGL((dwt97_fwd_kernel<2, 3>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0, L.jobs, L.njobs, L.planes, src, src_is_f64, out_i32, out_f64, nxt, dc_shift, quant, step, mct);
    } else if (L.cpl == 4) {
        hipExtLaunchKernelGGL((dwt97_fwd_kernel<4, 1>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0, L.jobs, L.njobs, L.planes, src, src_is_f64, out_i32, out_f64, nxt, dc_shift, quant, step, mct);
    } else {
        hipExtLaunchKernelGGL((dwt97_fwd_kernel<2, 1>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0, L.jobs, L.njobs, L.planes, src, src_is_f64, out_i32, out_f64, nxt, dc_shift, quant, step, mct);
    }
    return hipGetLastError();
}

hipError_t launch_dwt97_inv(hipStream_t s, const LevelLaunch &L, const void *coef, int coef_is_f64, const double *prev, void *dst,
                            int dc_shift, int final_level, int dst_mode, int mct) {
    (void)final_level;
    if (L.njobs <= 0) return hipSuccess;
    if (L.wg_waves > 0) {      // level 0 of an RGB triple, int32 coefficients -> int32 frame with inverse ICT (dwt97_l0wg_inv.inc)
        if (L.ncomp != 3 || coef_is_f64 || !mct || dst_mode != DST_I32_FRAME) return hipErrorInvalidValue;
#define J2K_WG97I(NW) hipExtLaunchKernelGGL((dwt97_inv_rgb_wg_kernel<NW, J2K_WG97I_WPE, false>), dim3(L.njobs), dim3(NW * 64), 0, s, L.ev_start, L.ev_stop, 0, \
                                             L.jobs, L.njobs, L.planes, reinterpret_cast<const int32_t *>(coef), prev,                       \
                                             reinterpret_cast<int32_t *>(dst), dc_shift, 0)
        if (L.pix_stride > 0) {  // straight to packed RGBA8 pixels (j2k_plan_inverse_pixels on a lossy 8-bit plan): eight waves
            if (L.wg_waves != 8) return hipErrorInvalidValue;
            hipExtLaunchKernelGGL((dwt97_inv_rgb_wg_kernel<8, J2K_WG97I_WPE, true>), dim3(L.njobs), dim3(512), 0, s, L.ev_start, L.ev_stop, 0,
                                  L.jobs, L.njobs, L.planes, reinterpret_cast<const int32_t *>(coef), prev, reinterpret_cast<int32_t *>(dst), dc_shift, L.pix_stride);
            return hipGetLastError();
        }
        if (L.wg_waves == 6) J2K_WG97I(6);
        else if (L.wg_waves == 8) J2K_WG97I(8);
        else if (L.wg_waves == 10) J2K_WG97I(10);
        else if (L.wg_waves == 12) J2K_WG97I(12);
        else return hipErrorInvalidValue;
#undef J2K_WG97I
        return hipGetLastError();
    }
    if (L.pwaves == 8 && L.pnjobs > 0 && L.ncomp == 1) {   // single planes in workgroup form: a deeper level, level 0 of one int32 component, the float64 unit calls
#define J2K_PWG97I(CF, DI) hipLaunchKernelGGL((dwt97_inv_plane_wg_kernel<8, 6, CF, DI>), dim3(L.pnjobs), dim3(512), 0, s, L.pjobs, L.pnjobs, L.planes, \
                                               coef, prev, dst, dc_shift, dst_mode == DST_F64_FRAME ? 1 : 0, dst_mode == DST_I32_FRAME ? L.pix_stride : 0)
        if (dst_mode == DST_I32_FRAME) { if (coef_is_f64) J2K_PWG97I(true, true); else J2K_PWG97I(false, true); }
        else { if (coef_is_f64) J2K_PWG97I(true, false); else J2K_PWG97I(false, false); }
#undef J2K_PWG97I
        return hipGetLastError();
    }
    const int blocks = (L.njobs + 3) / 4;
    if (L.ncomp == 3) {
        hipLaunchKernelGGL((dwt97_inv_kernel<2, 3>), dim3(blocks), dim3(256), 0, s, L.jobs, L.njobs, L.planes, coef, coef_is_f64, prev, dst, dc_shift, dst_mode, mct);
    } else if (L.cpl == 4) {
        hipLaunchKernelGGL((dwt97_inv_kernel<4, 1>), dim3(blocks), dim3(256), 0, s, L.jobs, L.njobs, L.planes, coef, coef_is_f64, prev, dst, dc_shift, dst_mode, mct);
    } else {
        hipLaunchKernelGGL((dwt97_inv_kernel<2, 1>), dim3(blocks), dim3(256), 0, s, L.jobs, L.njobs, L.planes, coef, coef_is_f64, prev, dst, dc_shift, dst_mode, mct);
    }
    return hipGetLastError();
}

}  // namespace j2k
