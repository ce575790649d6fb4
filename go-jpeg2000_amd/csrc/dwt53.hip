// dwt53.hip -- 5-3 reversible lifting DWT for gfx950, fused with DC shift + RCT.
//
// Replaces (reference, mrjoshuak/go-jpeg2000):
//   dwt.Forward53/Inverse53            internal/dwt/dwt.go:73-147
//   dwt.Forward2D53/Inverse2D53        internal/dwt/dwt.go:356-429
//   one level of Decompose/ReconstructMultiLevel53  dwt.go:524-548
//   mct.DCLevelShiftForward/Inverse, mct.ForwardRCT/InverseRCT  internal/mct/mct.go:28-38,56-66,96-118
//
// Design (MI355X-first, not the reference's row-pass + strided column-pass):
//   * The 5-3 filter is local (L[n] <- x[2n-2..2n+2]), so one level is ONE streaming
//     pass: every wavefront owns a column strip x a band of row pairs of one plane.
//   * Each lane holds CPL (2/4/8) consecutive columns of a row in registers
//     (16-byte global loads for CPL>=4); the horizontal lifting neighbours come from
//     the adjacent lane by a DPP wave shift (no LDS, no barrier).
//   * The wave then marches down its band: the vertical lifting is a 2-row sliding
//     window kept in registers, so every sample is read from HBM once and written
//     once -- algorithmic bytes = 2*4*w*h per level.  A band needs three halo rows of its
//     neighbours; the four bands of a workgroup hand them over through LDS (linked bands,
//     see dwt53_fwd_kernel), so only three rows per WORKGROUP are read twice.  Row-contiguous loads and stores only; the reference's
//     4-byte-strided column gather (dwt.go:375-394) does not exist here.
//   * De-interleave happens at the store: L -> column p, H -> column ceil(w/2)+p,
//     low rows -> row q, high rows -> row ceil(h/2)+q.
//   * The reference's multi-level layout (level l+1 re-reads the contiguous prefix
//     data[0 : w'*h'] as a dense matrix, dwt.go:524-531) is kept bit-exact by
//     splitting every store on the LINEAR index: idx < n_next goes to the scratch
//     that feeds the next level, the rest is final and goes to the coefficient plane.
//   * Level 0 optionally loads three component planes, applies DC shift + RCT in
//     registers and lifts the three results (NC=3): no separate elementwise pass.
//
// Go int32 semantics: '>>' is arithmetic, overflow wraps; all sums are done in
// uint32 and shifted as int32.
#include "j2k_internal.h"
#include <hip/hip_ext.h>

namespace j2k {

// lane i <- lane i-1 (wave_shr:1) / lane i <- lane i+1 (wave_shl:1); lane 0 / 63 keep `self`.
#ifndef J2K_NO_DPP_WAVE_SHIFT
__device__ __forceinline__ int from_left(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int from_right(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
#else
__device__ __forceinline__ int from_left(int v) { return __shfl_up(v, 1); }
__device__ __forceinline__ int from_right(int v) { return __shfl_down(v, 1); }
#endif

__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
// (a + b) >> 1 and (a + b + 2) >> 2 with Go wraparound + arithmetic shift
__device__ __forceinline__ int avg1(int a, int b) { return wadd(a, b) >> 1; }
__device__ __forceinline__ int avg2(int a, int b) { return wadd(wadd(a, b), 2) >> 2; }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
// J2K_NT_MODE (set at build time for A/B): bit 0 = non-temporal loads of level-0 source rows,
//                                          bit 1 = non-temporal stores of FINAL coefficients
#ifndef J2K_NT_MODE
#define J2K_NT_MODE 0
#endif
__device__ __forceinline__ int4 ld4(const int32_t *p, bool nt) {
    if ((J2K_NT_MODE & 1) && nt) { const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p)); return make_int4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const int4 *>(p);
}
__device__ __forceinline__ void st4(int32_t *p, int a, int b, int c, int d, bool nt) {
    if ((J2K_NT_MODE & 2) && nt) { v4i v = {a, b, c, d}; __builtin_nontemporal_store(v, reinterpret_cast<v4i *>(p)); return; }
    *reinterpret_cast<int4 *>(p) = make_int4(a, b, c, d);
}

// ---- global access helpers ------------------------------------------------------
template <int CPL, bool VEC>
__device__ __forceinline__ void load_cols(const int32_t *__restrict__ p, int c, int w, int (&x)[CPL], bool nt = false) {
    if constexpr (VEC) {
        // branch-free: lanes past the row end load from column 0 (always valid) and are zeroed by a select, so the
        // loads of consecutive rows can be issued back to back
        const bool in = c < w;
        const int cc = in ? c : 0;
        if constexpr (CPL == 8) {
            int4 a = ld4(p + cc, nt);
            int4 b = ld4(p + cc + 4, nt);
            x[0] = in ? a.x : 0; x[1] = in ? a.y : 0; x[2] = in ? a.z : 0; x[3] = in ? a.w : 0;
            x[4] = in ? b.x : 0; x[5] = in ? b.y : 0; x[6] = in ? b.z : 0; x[7] = in ? b.w : 0;
        } else if constexpr (CPL == 4) {
            int4 a = ld4(p + cc, nt);
            x[0] = in ? a.x : 0; x[1] = in ? a.y : 0; x[2] = in ? a.z : 0; x[3] = in ? a.w : 0;
        } else {
            int2 a = *reinterpret_cast<const int2 *>(p + cc);
            x[0] = in ? a.x : 0; x[1] = in ? a.y : 0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPL; i++) x[i] = (c + i < w) ? p[c + i] : 0;
    }
}

// store n = CPL/2 consecutive values at p[0..n) ; valid = number of in-range elements
template <int H, bool VEC>
__device__ __forceinline__ void store_half(int32_t *__restrict__ p, const int *v, int valid, bool nt = false) {
    if constexpr (VEC) {
        if (valid > 0) {
            if constexpr (H == 4) st4(p, v[0], v[1], v[2], v[3], nt);
            else if constexpr (H == 2) *reinterpret_cast<int2 *>(p) = make_int2(v[0], v[1]);
            else p[0] = v[0];
        }
    } else {
#pragma unroll
        for (int i = 0; i < H; i++)
            if (i < valid) p[i] = v[i];
    }
}

// ---- horizontal forward lifting of one row held across the wave -------------------
// in : x[CPL] = columns c..c+CPL-1 of the row.  out: lo[CPL/2], hi[CPL/2] (pair j = column c+2j).
template <int CPL, bool THIN>
__device__ __forceinline__ void hfwd(const int (&x)[CPL], int c, int w, int (&lo)[CPL / 2], int (&hi)[CPL / 2]) {
    constexpr int H = CPL / 2;
    if (THIN && w < 2) {  // dwt.go:74-76: length < 2 -> untouched
#pragma unroll
        for (int j = 0; j < H; j++) { lo[j] = x[2 * j]; hi[j] = 0; }
        return;
    }
    const int e_right = from_right(x[0]);  // even sample of the lane to the right
    int d[H];
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        const int en = (j + 1 < H) ? x[2 * j + 2] : e_right;
        // predict: d = o - ((e + e_next) >> 1); last odd of an even-length row: d = o - e (dwt.go:89-95)
        const int pred = (ce + 2 < w) ? avg1(x[2 * j], en) : x[2 * j];
        d[j] = wsub(x[2 * j + 1], pred);
    }
    // a row of odd length ends on an even sample with no odd partner: its "d" mirrors d[n-2] (dwt.go:112-114)
    const int d_left_lane = from_left(d[H - 1]);
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int ce = c + 2 * j;
        int dp = (j > 0) ? d[j - 1] : d_left_lane;
        int dc = d[j];
        if (ce + 1 >= w) dc = dp;  // no odd partner
        if (ce == 0) dp = dc;      // dwt.go:99: d[-1] mirrors d[1]
        lo[j] = wadd(x[2 * j], avg2(dp, dc));
        hi[j] = d[j];
    }
}

// ---- horizontal inverse lifting ----------------------------------------------------
// in : lo[H] = L[p0..p0+H), hi[H] = H[p0..p0+H) ; out: x[CPL] = columns 2*p0 ..
template <int CPL>
__device__ __forceinline__ void hinv(const int (&lo)[CPL / 2], const int (&hi)[CPL / 2], int c, int w, int (&x)[CPL]) {
    constexpr int H = CPL / 2;
    if (w < 2) {
#pragma unroll
        for (int j = 0; j < H; j++) { x[2 * j] = lo[j]; x[2 * j + 1] = 0; }
        return;
    }
    const int d_left_lane = from_left(hi[H - 1]);
    int e[H];
#pragma unroll
    for (int j = 0; j < H; j++) {  // undo update (dwt.go:132-138)
        const int ce = c + 2 * j;
        int dp = (j > 0) ? hi[j - 1] : d_left_lane;
        int dc = hi[j];
        if (ce + 1 >= w) dc = dp;
        if (ce == 0) dp = dc;
        e[j] = wsub(lo[j], avg2(dp, dc));
    }
    const int e_right = from_right(e[0]);
#pragma unroll
    for (int j = 0; j < H; j++) {  // undo predict (dwt.go:141-146)
        const int ce = c + 2 * j;
        const int en = (j + 1 < H) ? e[j + 1] : e_right;
        const int pred = (ce + 2 < w) ? avg1(e[j], en) : e[j];
        x[2 * j] = e[j];
        x[2 * j + 1] = wadd(hi[j], pred);
    }
}

// ================================================================================
// forward level kernel
// ================================================================================
template <int CPL, int NC, bool VEC>
struct FwdRow {
    int lo[NC][CPL / 2];
    int hi[NC][CPL / 2];
};
template <int CPL, int NC>
struct RawRow {
    int x[NC][CPL];
};

// Issue the global loads of one source row (NC component planes) and nothing else, so the caller can put the
// loads of several rows in flight before the first use.  Out-of-range lanes/columns read a clamped, valid address:
// their values never reach a stored result (the lifting only looks at in-range neighbours), so they need no mask.
template <int CPL, int NC, bool VEC, bool PIX = false>
__device__ __forceinline__ void fwd_issue_row(const int32_t *__restrict__ src, const DwtPlane &P, int r, int c,
                                              RawRow<CPL, NC> &R, int64_t pix_row0 = -1, int pix_stride = 0) {
    if constexpr (PIX && NC == 3 && CPL == 8 && VEC) {
        {
            // packed RGBA8 source (encoder.go:107-123 fused): eight pixels = two 16-byte loads instead of six
            const int cc = (c < P.w) ? c : 0;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src) + pix_row0 + (int64_t)r * pix_stride + cc;
            const int4 a = ld4(reinterpret_cast<const int32_t *>(p), true), b = ld4(reinterpret_cast<const int32_t *>(p + 4), true);
            const uint32_t px[8] = {(uint32_t)a.x, (uint32_t)a.y, (uint32_t)a.z, (uint32_t)a.w,
                                    (uint32_t)b.x, (uint32_t)b.y, (uint32_t)b.z, (uint32_t)b.w};
#pragma unroll
            for (int i = 0; i < 8; i++) {     // R, G, B are bytes 0, 1, 2 (image.RGBA)
                R.x[0][i] = (int)(px[i] & 0xFF);
                R.x[1][i] = (int)((px[i] >> 8) & 0xFF);
                R.x[2][i] = (int)((px[i] >> 16) & 0xFF);
            }
            return;
        }
    }
    if constexpr (PIX && NC == 1 && CPL == 8 && VEC) {
        {
            // packed Gray16 source (image.Gray16.Pix: two bytes per pixel, high byte first; encoder.go:94-105 fused):
            // eight pixels = one 16-byte load instead of two
            const int cc = (c < P.w) ? c : 0;
            const uint16_t *p = reinterpret_cast<const uint16_t *>(src) + pix_row0 + (int64_t)r * pix_stride + cc;
            const int4 a = ld4(reinterpret_cast<const int32_t *>(p), false);
            const uint32_t dw[4] = {(uint32_t)a.x, (uint32_t)a.y, (uint32_t)a.z, (uint32_t)a.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {     // bytes b0 b1 b2 b3 -> pixels (b0 << 8 | b1), (b2 << 8 | b3)
                R.x[0][2 * i] = (int)(((dw[i] & 0xFF) << 8) | ((dw[i] >> 8) & 0xFF));
                R.x[0][2 * i + 1] = (int)(((dw[i] >> 8) & 0xFF00) | (dw[i] >> 24));
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int32_t *p = src + P.src_off[k] + (int64_t)r * P.src_stride;
        if constexpr (VEC) {
            const int cc = (c < P.w) ? c : 0;
            if constexpr (CPL == 8) {
                const int4 a = ld4(p + cc, NC == 3), b = ld4(p + cc + 4, NC == 3);
                R.x[k][0] = a.x; R.x[k][1] = a.y; R.x[k][2] = a.z; R.x[k][3] = a.w;
                R.x[k][4] = b.x; R.x[k][5] = b.y; R.x[k][6] = b.z; R.x[k][7] = b.w;
            } else if constexpr (CPL == 4) {
                const int4 a = ld4(p + cc, NC == 3);
                R.x[k][0] = a.x; R.x[k][1] = a.y; R.x[k][2] = a.z; R.x[k][3] = a.w;
            } else {
                const int2 a = *reinterpret_cast<const int2 *>(p + cc);
                R.x[k][0] = a.x; R.x[k][1] = a.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPL; i++) R.x[k][i] = p[min(c + i, P.w - 1)];
        }
    }
}

// DC shift (mct.go:96-101) + RCT (mct.go:28-38) + horizontal lifting of one issued row
template <int CPL, int NC, bool VEC>
__device__ __forceinline__ void fwd_finish_row(RawRow<CPL, NC> &X, const DwtPlane &P, int c, int dc_shift, FwdRow<CPL, NC, VEC> &R) {
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int i = 0; i < CPL; i++) X.x[k][i] = wsub(X.x[k][i], dc_shift);
    if constexpr (NC == 3) {
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int r_ = X.x[0][i], g_ = X.x[1][i], b_ = X.x[2][i];
            X.x[0][i] = wadd(wadd(r_, wadd(g_, g_)), b_) >> 2;
            X.x[1][i] = wsub(b_, g_);
            X.x[2][i] = wsub(r_, g_);
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) hfwd<CPL, false>(X.x[k], c, P.w, R.lo[k], R.hi[k]);
}

// thin planes (w < 2 or h < 2): masked loads, the length<2 pass-through rules of dwt.go:74-76
template <int CPL, int NC, bool VEC>
__device__ __forceinline__ void fwd_load_row_thin(const int32_t *__restrict__ src, const DwtPlane &P, int r, int c, int dc_shift,
                                                  FwdRow<CPL, NC, VEC> &R, int64_t pix0, int pix_stride) {
    RawRow<CPL, NC> X;
    if (pix_stride > 0) fwd_issue_row<CPL, NC, VEC, true>(src, P, r, c, X, pix0, pix_stride);
    else fwd_issue_row<CPL, NC, VEC, false>(src, P, r, c, X);
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int i = 0; i < CPL; i++) X.x[k][i] = (c + i < P.w) ? wsub(X.x[k][i], dc_shift) : 0;
    if constexpr (NC == 3) {
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int r_ = X.x[0][i], g_ = X.x[1][i], b_ = X.x[2][i];
            X.x[0][i] = wadd(wadd(r_, wadd(g_, g_)), b_) >> 2;
            X.x[1][i] = wsub(b_, g_);
            X.x[2][i] = wsub(r_, g_);
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) hfwd<CPL, true>(X.x[k], c, P.w, R.lo[k], R.hi[k]);
}

template <int CPL, int NC, bool VEC>
__device__ __forceinline__ void fwd_store_row(int32_t *__restrict__ out, int32_t *__restrict__ nxt, const DwtPlane &P, int ro,
                                              int p0, bool owned, const int (&lo)[NC][CPL / 2], const int (&hi)[NC][CPL / 2]) {
    constexpr int H = CPL / 2;
    if (!owned) return;
    const int halfW = (P.w + 1) >> 1;
    const int nL = halfW - p0;             // valid low columns from p0
    const int nH = (P.w - halfW) - p0;     // valid high columns from p0
    const int idxL = ro * P.w + p0;
    const int idxH = idxL + halfW;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        if constexpr (VEC) {
            int32_t *bl = (idxL < P.n_next) ? nxt + P.nxt_off[k] : out + P.out_off[k];
            int32_t *bh = (idxH < P.n_next) ? nxt + P.nxt_off[k] : out + P.out_off[k];
            store_half<H, true>(bl + idxL, lo[k], nL, idxL >= P.n_next);
            store_half<H, true>(bh + idxH, hi[k], nH, idxH >= P.n_next);
        } else {
#pragma unroll
            for (int j = 0; j < H; j++) {
                if (j < nL) {
                    int32_t *b = (idxL + j < P.n_next) ? nxt + P.nxt_off[k] : out + P.out_off[k];
                    b[idxL + j] = lo[k][j];
                }
                if (j < nH) {
                    int32_t *b = (idxH + j < P.n_next) ? nxt + P.nxt_off[k] : out + P.out_off[k];
                    b[idxH + j] = hi[k][j];
                }
            }
        }
    }
}

// w < 2 or h < 2: one of the two passes is the identity.  Rare (deepest levels of tiny planes); kept simple.
template <int CPL, int NC, bool VEC>
__device__ __forceinline__ void fwd_job_thin(const DwtJob &job, const DwtPlane &P, const int32_t *__restrict__ src,
                                          int32_t *__restrict__ out, int32_t *__restrict__ nxt, int dc_shift, int c, int p0,
                                          bool owned, int64_t pix0, int pix_stride) {
    constexpr int H = CPL / 2;
    typedef FwdRow<CPL, NC, VEC> Row;
    const int h = P.h, halfH = (h + 1) >> 1;
    const int pr_end = min(job.prow0 + job.nprow, halfH);
    Row ye, yo, yn;
    int dvp_lo[NC][H], dvp_hi[NC][H];
    fwd_load_row_thin<CPL, NC, VEC>(src, P, 2 * job.prow0, c, dc_shift, ye, pix0, pix_stride);
    if (h < 2) {
        if (job.prow0 == 0) fwd_store_row<CPL, NC, VEC>(out, nxt, P, 0, p0, owned, ye.lo, ye.hi);
        return;
    }
    if (job.prow0 > 0) {
        Row ym2, ym1;
        fwd_load_row_thin<CPL, NC, VEC>(src, P, 2 * job.prow0 - 2, c, dc_shift, ym2, pix0, pix_stride);
        fwd_load_row_thin<CPL, NC, VEC>(src, P, 2 * job.prow0 - 1, c, dc_shift, ym1, pix0, pix_stride);
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                dvp_lo[k][j] = wsub(ym1.lo[k][j], avg1(ym2.lo[k][j], ye.lo[k][j]));
                dvp_hi[k][j] = wsub(ym1.hi[k][j], avg1(ym2.hi[k][j], ye.hi[k][j]));
            }
    }
    for (int pr = job.prow0; pr < pr_end; pr++) {
        const int r1 = 2 * pr + 1, r2 = 2 * pr + 2;
        const bool has_odd = r1 < h, has_next = r2 < h;
        if (has_odd) fwd_load_row_thin<CPL, NC, VEC>(src, P, r1, c, dc_shift, yo, pix0, pix_stride);
        if (has_next) fwd_load_row_thin<CPL, NC, VEC>(src, P, r2, c, dc_shift, yn, pix0, pix_stride);
        int dv_lo[NC][H], dv_hi[NC][H], sv_lo[NC][H], sv_hi[NC][H];
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                int dl, dh;
                if (has_odd) {
                    dl = wsub(yo.lo[k][j], has_next ? avg1(ye.lo[k][j], yn.lo[k][j]) : ye.lo[k][j]);
                    dh = wsub(yo.hi[k][j], has_next ? avg1(ye.hi[k][j], yn.hi[k][j]) : ye.hi[k][j]);
                } else {
                    dl = dvp_lo[k][j];
                    dh = dvp_hi[k][j];
                }
                sv_lo[k][j] = wadd(ye.lo[k][j], avg2((pr == 0) ? dl : dvp_lo[k][j], dl));
                sv_hi[k][j] = wadd(ye.hi[k][j], avg2((pr == 0) ? dh : dvp_hi[k][j], dh));
                dv_lo[k][j] = dl;
                dv_hi[k][j] = dh;
            }
        fwd_store_row<CPL, NC, VEC>(out, nxt, P, pr, p0, owned, sv_lo, sv_hi);
        if (has_odd) fwd_store_row<CPL, NC, VEC>(out, nxt, P, halfH + pr, p0, owned, dv_lo, dv_hi);
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                dvp_lo[k][j] = dv_lo[k][j]; dvp_hi[k][j] = dv_hi[k][j];
                ye.lo[k][j] = yn.lo[k][j]; ye.hi[k][j] = yn.hi[k][j];
            }
    }
}

#ifdef J2K_FWD_WPE
#define J2K_FWD_ATTR __attribute__((amdgpu_waves_per_eu(J2K_FWD_WPE)))
#else
#define J2K_FWD_ATTR
#endif
// Forward level kernel.  One wavefront = column strip x band of pair-rows (DwtJob); the four wavefronts of a
// workgroup normally hold four vertically adjacent bands of one strip, and then they are LINKED (flags in
// DwtJob::nprow, set by the host): instead of re-reading the three halo rows of its neighbours from memory
//   * a band with J2K_LINK_UP skips the two rows above it; it publishes its first lifted even row and its first
//     vertical d to LDS and leaves the low-pass output row of its first pair-row to the band above,
//   * a band with J2K_LINK_DOWN takes that even row from LDS instead of loading it (last predict step) and, after
//     its last pair-row, emits the first low-pass row of the band below (it owns the d above it).
// One __syncthreads() per wavefront (after its first pair-row) orders publish -> consume; linked bands have >= 2
// pair-rows so the consumer side always comes after the barrier.  Source rows are then read from HBM once plus
// three rows per WORKGROUP (not per wavefront).
// PF: the loads of pair-row q+1 are issued before the vertical lifting and the stores of pair-row q (software
// pipelining, one pair-row deep; costs CPL*NC*2 more live registers)
template <int CPL, int NC, bool VEC, bool PF, bool PIX>
__global__ __launch_bounds__(256) J2K_FWD_ATTR void dwt53_fwd_kernel(const DwtJob *__restrict__ jobs, int njobs,
                                                                     const DwtPlane *__restrict__ planes,
                                                                     const int32_t *__restrict__ src, int32_t *__restrict__ out,
                                                                     int32_t *__restrict__ nxt, int dc_shift, int pix_stride) {
    constexpr int H = CPL / 2;
    constexpr int PUB = 2 * NC * CPL * 64;   // ints one wavefront publishes: {even row, d row} x NC x (lo|hi) x 64 lanes
    __shared__ int sh[4 * PUB];
    const int wv = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    const int lane = threadIdx.x & 63;
    DwtJob job{-1, 0, 0, 0};
    if (wave < njobs) job = jobs[wave];
    if (job.plane < 0) {             // past the end / padding entry of the XCD-aware job order
        __syncthreads();
        return;
    }
    const bool link_up = (job.nprow & J2K_LINK_UP) != 0, link_down = (job.nprow & J2K_LINK_DOWN) != 0;
    job.nprow &= 0xffff;
    const DwtPlane P = planes[job.plane];
    const int w = P.w, h = P.h;
    const int lane_first = (job.col0 == 0) ? 0 : 1;
    const int c_base = job.col0 - lane_first * CPL;
    const int c = c_base + lane * CPL;
    const bool reach_end = (c_base + 64 * CPL >= w);
    const bool owned = (lane >= lane_first) && (c < w) && (reach_end || lane < 63);
    const int p0 = c >> 1;
    // packed-pixel source: pixel index of this tile's origin (src_off[0] = y0 * W + x0 in the planar frame)
    int64_t pix0 = -1;
    if (PIX) {
        const int64_t y0 = P.src_off[0] / P.src_stride;
        pix0 = y0 * pix_stride + (P.src_off[0] - y0 * P.src_stride);
    }
    if (w < 2 || h < 2) {            // never linked
        fwd_job_thin<CPL, NC, VEC>(job, P, src, out, nxt, dc_shift, c, p0, owned, pix0, PIX ? pix_stride : 0);
        __syncthreads();
        return;
    }
    const int halfH = (h + 1) >> 1;
    const int pr_begin = job.prow0;
    const int pr_end = min(job.prow0 + job.nprow, halfH);
    const int hl = h - 1;
    int *pub_mine = sh + wv * PUB + lane;
    const int *pub_below = sh + ((wv + 1) & 3) * PUB + lane;   // only read when link_down (then wv < 3)

    typedef FwdRow<CPL, NC, VEC> Row;
    typedef RawRow<CPL, NC> Raw;
    Row ye, yo, yn;
    int dvp_lo[NC][H], dvp_hi[NC][H];  // vertical d of the previous pair-row
    Raw ra, rb;                        // rows 2q+1, 2q+2 in flight
    {
        // the loads of the prologue are issued before the first use: one (PF) or two memory latencies, not five
        Raw r0, rm2, rm1;
        const int re = 2 * pr_begin;
        fwd_issue_row<CPL, NC, VEC, PIX>(src, P, re, c, r0, pix0, pix_stride);
        if (!link_up) {
            fwd_issue_row<CPL, NC, VEC, PIX>(src, P, max(re - 2, 0), c, rm2, pix0, pix_stride);
            fwd_issue_row<CPL, NC, VEC, PIX>(src, P, max(re - 1, 0), c, rm1, pix0, pix_stride);
        }
        if (PF) {
            fwd_issue_row<CPL, NC, VEC, PIX>(src, P, min(re + 1, hl), c, ra, pix0, pix_stride);
            fwd_issue_row<CPL, NC, VEC, PIX>(src, P, min(re + 2, hl), c, rb, pix0, pix_stride);   // linked bands have >= 2 pair-rows: a real row
        }
        fwd_finish_row<CPL, NC, VEC>(r0, P, c, dc_shift, ye);
        if (!link_up) {
            Row ym2, ym1;
            fwd_finish_row<CPL, NC, VEC>(rm2, P, c, dc_shift, ym2);
            fwd_finish_row<CPL, NC, VEC>(rm1, P, c, dc_shift, ym1);
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {   // unused (pr == 0 mirrors) when the band starts at the top
                    dvp_lo[k][j] = wsub(ym1.lo[k][j], avg1(ym2.lo[k][j], ye.lo[k][j]));
                    dvp_hi[k][j] = wsub(ym1.hi[k][j], avg1(ym2.hi[k][j], ye.hi[k][j]));
                }
        } else {
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) dvp_lo[k][j] = dvp_hi[k][j] = 0;
        }
    }
    for (int pr = pr_begin; pr < pr_end; pr++) {
        const bool first = pr == pr_begin, last = pr + 1 == pr_end;
        const bool has_odd = 2 * pr + 1 < h, has_next = 2 * pr + 2 < h;
        const bool yn_from_lds = link_down && last;
        if (!PF) {
            fwd_issue_row<CPL, NC, VEC, PIX>(src, P, min(2 * pr + 1, hl), c, ra, pix0, pix_stride);
            if (!yn_from_lds) fwd_issue_row<CPL, NC, VEC, PIX>(src, P, min(2 * pr + 2, hl), c, rb, pix0, pix_stride);
        }
        fwd_finish_row<CPL, NC, VEC>(ra, P, c, dc_shift, yo);
        if (yn_from_lds) {
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {
                    yn.lo[k][j] = pub_below[(k * CPL + j) * 64];
                    yn.hi[k][j] = pub_below[(k * CPL + H + j) * 64];
                }
        } else {
            fwd_finish_row<CPL, NC, VEC>(rb, P, c, dc_shift, yn);
        }
        if (PF && !last) {
            fwd_issue_row<CPL, NC, VEC, PIX>(src, P, min(2 * pr + 3, hl), c, ra, pix0, pix_stride);
            if (!(link_down && pr + 2 == pr_end)) fwd_issue_row<CPL, NC, VEC, PIX>(src, P, min(2 * pr + 4, hl), c, rb, pix0, pix_stride);
        }
        int dv_lo[NC][H], dv_hi[NC][H], sv_lo[NC][H], sv_hi[NC][H];
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                // predict: d = o - ((e + e_next) >> 1); no next even row: d = o - e; no odd row (odd height, last even
                // row): the update mirrors d[n-2]
                const int pl = has_next ? avg1(ye.lo[k][j], yn.lo[k][j]) : ye.lo[k][j];
                const int ph = has_next ? avg1(ye.hi[k][j], yn.hi[k][j]) : ye.hi[k][j];
                const int dl = has_odd ? wsub(yo.lo[k][j], pl) : dvp_lo[k][j];
                const int dh = has_odd ? wsub(yo.hi[k][j], ph) : dvp_hi[k][j];
                sv_lo[k][j] = wadd(ye.lo[k][j], avg2((pr == 0) ? dl : dvp_lo[k][j], dl));
                sv_hi[k][j] = wadd(ye.hi[k][j], avg2((pr == 0) ? dh : dvp_hi[k][j], dh));
                dv_lo[k][j] = dl;
                dv_hi[k][j] = dh;
            }
        if (first && link_up) {   // the band above finishes this pair-row's low-pass output
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {
                    pub_mine[(k * CPL + j) * 64] = ye.lo[k][j];
                    pub_mine[(k * CPL + H + j) * 64] = ye.hi[k][j];
                    pub_mine[((NC + k) * CPL + j) * 64] = dv_lo[k][j];
                    pub_mine[((NC + k) * CPL + H + j) * 64] = dv_hi[k][j];
                }
        } else {
            fwd_store_row<CPL, NC, VEC>(out, nxt, P, pr, p0, owned, sv_lo, sv_hi);
        }
        if (has_odd) fwd_store_row<CPL, NC, VEC>(out, nxt, P, halfH + pr, p0, owned, dv_lo, dv_hi);
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                dvp_lo[k][j] = dv_lo[k][j]; dvp_hi[k][j] = dv_hi[k][j];
                ye.lo[k][j] = yn.lo[k][j]; ye.hi[k][j] = yn.hi[k][j];
            }
        if (first) __syncthreads();   // every wavefront of the workgroup arrives exactly once
    }
    if (link_down) {
        // low-pass row of the first pair-row of the band below: s = e + ((d_above + d_own + 2) >> 2), e = `ye` (rotated
        // in from LDS), d_above = this band's last d, d_own = published by the band below
        int sv_lo[NC][H], sv_hi[NC][H];
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                const int dl = pub_below[((NC + k) * CPL + j) * 64], dh = pub_below[((NC + k) * CPL + H + j) * 64];
                sv_lo[k][j] = wadd(ye.lo[k][j], avg2(dvp_lo[k][j], dl));
                sv_hi[k][j] = wadd(ye.hi[k][j], avg2(dvp_hi[k][j], dh));
            }
        fwd_store_row<CPL, NC, VEC>(out, nxt, P, pr_end, p0, owned, sv_lo, sv_hi);
    }
}

// ================================================================================
// inverse level kernel
// ================================================================================
// Input row `ri` of the level matrix, columns [p0,p0+H) of the L part and of the H part,
// each element taken from `prev` (output of the coarser inverse level) when its linear
// index is below n_next, else from the coefficient plane.
template <int CPL, int NC, bool VEC>
__device__ __forceinline__ void inv_load_row(const int32_t *__restrict__ coef, const int32_t *__restrict__ prev,
                                             const DwtPlane &P, int ri, int p0, int c, FwdRow<CPL, NC, VEC> &R) {
    constexpr int H = CPL / 2;
    const int halfW = (P.w + 1) >> 1;
    const int nL = halfW - p0, nH = (P.w - halfW) - p0;
    const int idxL = ri * P.w + p0, idxH = idxL + halfW;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        if constexpr (VEC) {
            const int32_t *bl = (idxL < P.n_next) ? prev + P.nxt_off[k] : coef + P.src_off[k];
            const int32_t *bh = (idxH < P.n_next) ? prev + P.nxt_off[k] : coef + P.src_off[k];
            const bool in = c < P.w;
            const int oL = in ? idxL : 0, oH = in ? idxH : 0;      // clamped: always a valid address, no branch
            if constexpr (H == 4) {
                int4 a = *reinterpret_cast<const int4 *>(bl + oL), b = *reinterpret_cast<const int4 *>(bh + oH);
                R.lo[k][0] = in ? a.x : 0; R.lo[k][1] = in ? a.y : 0; R.lo[k][2] = in ? a.z : 0; R.lo[k][3] = in ? a.w : 0;
                R.hi[k][0] = in ? b.x : 0; R.hi[k][1] = in ? b.y : 0; R.hi[k][2] = in ? b.z : 0; R.hi[k][3] = in ? b.w : 0;
            } else if constexpr (H == 2) {
                int2 a = *reinterpret_cast<const int2 *>(bl + oL), b = *reinterpret_cast<const int2 *>(bh + oH);
                R.lo[k][0] = in ? a.x : 0; R.lo[k][1] = in ? a.y : 0; R.hi[k][0] = in ? b.x : 0; R.hi[k][1] = in ? b.y : 0;
            } else {
                const int a = bl[oL], b = bh[oH];
                R.lo[k][0] = in ? a : 0; R.hi[k][0] = in ? b : 0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < H; j++) {
                int l = 0, hh = 0;
                if (j < nL) {
                    const int32_t *b = (idxL + j < P.n_next) ? prev + P.nxt_off[k] : coef + P.src_off[k];
                    l = b[idxL + j];
                }
                if (j < nH) {
                    const int32_t *b = (idxH + j < P.n_next) ? prev + P.nxt_off[k] : coef + P.src_off[k];
                    hh = b[idxH + j];
                }
                R.lo[k][j] = l; R.hi[k][j] = hh;
            }
        }
    }
}

// horizontal inverse + optional inverse RCT + DC shift + store of one reconstructed row
template <int CPL, int NC, bool VEC, bool PIX = false>
__device__ __forceinline__ void inv_finish_row(int32_t *__restrict__ dst, const DwtPlane &P, int ro, int c, bool owned,
                                               const int (&lo)[NC][CPL / 2], const int (&hi)[NC][CPL / 2], int dc_shift,
                                               bool final_level, int64_t pix0 = -1, int pix_stride = 0) {
    int x[NC][CPL];
#pragma unroll
    for (int k = 0; k < NC; k++) hinv<CPL>(lo[k], hi[k], c, P.w, x[k]);
    if constexpr (NC == 3) {  // mct.go:56-66
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int y_ = x[0][i], u_ = x[1][i], v_ = x[2][i];
            const int g_ = wsub(y_, wadd(u_, v_) >> 2);
            x[0][i] = wadd(v_, g_);
            x[1][i] = g_;
            x[2][i] = wadd(u_, g_);
        }
    }
    if (!owned) return;
    if constexpr (PIX && NC == 3 && CPL == 8 && VEC) {
        {
            // DCLevelShiftInverse + decoder.createImage for 3 components at 8 bit (decoder.go:477-505): clamp to 0..255,
            // alpha 255, eight RGBA pixels = two 16-byte stores instead of six
            uint32_t px[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r_ = min(max(wadd(x[0][i], dc_shift), 0), 255), g_ = min(max(wadd(x[1][i], dc_shift), 0), 255),
                          b_ = min(max(wadd(x[2][i], dc_shift), 0), 255);
                px[i] = (uint32_t)r_ | (uint32_t)g_ << 8 | (uint32_t)b_ << 16 | 0xFF000000u;
            }
            // reconstructed pixels are a pure output stream: non-temporal, so that they do not sit dirty in the Infinity
            // Cache for the next kernel to write back (see st_decoded in ht.hip)
            uint32_t *p = reinterpret_cast<uint32_t *>(dst) + pix0 + (int64_t)ro * pix_stride + c;
            const v4i v0 = {(int)px[0], (int)px[1], (int)px[2], (int)px[3]}, v1 = {(int)px[4], (int)px[5], (int)px[6], (int)px[7]};
            __builtin_nontemporal_store(v0, reinterpret_cast<v4i *>(p));
            __builtin_nontemporal_store(v1, reinterpret_cast<v4i *>(p + 4));
            return;
        }
    }
    if constexpr (PIX && NC == 1 && CPL == 8 && VEC) {
        {
            // DCLevelShiftInverse + decoder.createImage for one component at 16 bit (decoder.go:434-451): clamp to
            // 0..65535, then the reference's rescale v * 65535 / 65535 in int32 -- which WRAPS for v >= 32769 -- and
            // uint16() of that; high byte first, eight pixels = one 16-byte store instead of two
            uint32_t dw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int a0 = min(max(wadd(x[0][2 * i], dc_shift), 0), 65535), b0 = min(max(wadd(x[0][2 * i + 1], dc_shift), 0), 65535);
                const uint32_t a = (uint32_t)((int)((uint32_t)a0 * 65535u) / 65535) & 0xFFFFu;
                const uint32_t b = (uint32_t)((int)((uint32_t)b0 * 65535u) / 65535) & 0xFFFFu;
                dw[i] = (a >> 8) | ((a & 0xFF) << 8) | ((b >> 8) << 16) | ((b & 0xFF) << 24);
            }
            uint16_t *p = reinterpret_cast<uint16_t *>(dst) + pix0 + (int64_t)ro * pix_stride + c;
            *reinterpret_cast<uint4 *>(p) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            return;
        }
    }
    const int stride = final_level ? P.out_stride : P.w;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        int32_t *p = dst + P.out_off[k] + (int64_t)ro * stride + c;
        if constexpr (VEC) {
            if constexpr (CPL == 8) {
                *reinterpret_cast<int4 *>(p) = make_int4(wadd(x[k][0], dc_shift), wadd(x[k][1], dc_shift), wadd(x[k][2], dc_shift), wadd(x[k][3], dc_shift));
                *reinterpret_cast<int4 *>(p + 4) = make_int4(wadd(x[k][4], dc_shift), wadd(x[k][5], dc_shift), wadd(x[k][6], dc_shift), wadd(x[k][7], dc_shift));
            } else if constexpr (CPL == 4) {
                *reinterpret_cast<int4 *>(p) = make_int4(wadd(x[k][0], dc_shift), wadd(x[k][1], dc_shift), wadd(x[k][2], dc_shift), wadd(x[k][3], dc_shift));
            } else {
                *reinterpret_cast<int2 *>(p) = make_int2(wadd(x[k][0], dc_shift), wadd(x[k][1], dc_shift));
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPL; i++)
                if (c + i < P.w) p[i] = wadd(x[k][i], dc_shift);
        }
    }
}

// xe[q] = s[q] - ((d[q-1] + d[q] + 2) >> 2) with the mirror rules of dwt.go:132-138
template <int CPL, int NC, bool VEC>
__device__ __forceinline__ void inv_compute_xe(int q, int nhigh, const FwdRow<CPL, NC, VEC> &S, const FwdRow<CPL, NC, VEC> &Dp,
                                               const FwdRow<CPL, NC, VEC> &Dc, int (&xl)[NC][CPL / 2], int (&xh)[NC][CPL / 2]) {
    constexpr int H = CPL / 2;
    const bool has_d = q < nhigh;
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int j = 0; j < H; j++) {
            int dpl = Dp.lo[k][j], dph = Dp.hi[k][j];
            int dcl = has_d ? Dc.lo[k][j] : dpl, dch = has_d ? Dc.hi[k][j] : dph;
            if (q == 0) { dpl = dcl; dph = dch; }
            xl[k][j] = wsub(S.lo[k][j], avg2(dpl, dcl));
            xh[k][j] = wsub(S.hi[k][j], avg2(dph, dch));
        }
}

// Inverse level kernel.  Same decomposition and the same workgroup links as the forward kernel: a band needs the
// s and d rows of its own pair-rows plus d[q0-1] above and s[q1], d[q1] below.  The two rows below are the FIRST rows
// the band below loads, so a band with J2K_LINK_UP publishes them to LDS and a band with J2K_LINK_DOWN takes them from
// there at its last pair-row; only d[q0-1] is still read twice (one halo row per band instead of three).
template <int CPL, int NC, bool VEC, bool PIX>
__global__ __launch_bounds__(256) void dwt53_inv_kernel(const DwtJob *__restrict__ jobs, int njobs,
                                                        const DwtPlane *__restrict__ planes,
                                                        const int32_t *__restrict__ coef, const int32_t *__restrict__ prev,
                                                        int32_t *__restrict__ dst, int dc_shift, int final_level, int pix_stride) {
    constexpr int H = CPL / 2;
    constexpr int PUB = 2 * NC * CPL * 64;   // ints one wavefront publishes: {s row, d row} x NC x (lo|hi) x 64 lanes
    __shared__ int sh[4 * PUB];
    const int wv = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    const int lane = threadIdx.x & 63;
    DwtJob job{-1, 0, 0, 0};
    if (wave < njobs) job = jobs[wave];
    if (job.plane < 0) {             // past the end / padding entry of the XCD-aware job order
        __syncthreads();
        return;
    }
    const bool link_up = (job.nprow & J2K_LINK_UP) != 0, link_down = (job.nprow & J2K_LINK_DOWN) != 0;
    job.nprow &= 0xffff;
    const DwtPlane P = planes[job.plane];
    const int w = P.w, h = P.h;
    const int lane_first = (job.col0 == 0) ? 0 : 1;
    const int c_base = job.col0 - lane_first * CPL;
    const int c = c_base + lane * CPL;
    const bool reach_end = (c_base + 64 * CPL >= w);
    const bool owned = (lane >= lane_first) && (c < w) && (reach_end || lane < 63);
    const int p0 = c >> 1;
    const int halfH = (h + 1) >> 1;
    const int nhigh = h - halfH;  // rows in the vertical high band
    const int pr_begin = job.prow0;
    const int pr_end = min(job.prow0 + job.nprow, halfH);
    const bool fin = final_level != 0;
    int64_t pix0 = -1;                 // packed RGBA8 destination (final level only): pixel index of the tile origin
    if (PIX) {
        const int64_t y0 = P.out_off[0] / P.out_stride;
        pix0 = y0 * pix_stride + (P.out_off[0] - y0 * P.out_stride);
    }
    int *pub_mine = sh + wv * PUB + lane;
    const int *pub_below = sh + ((wv + 1) & 3) * PUB + lane;   // only read when link_down (then wv < 3)

    typedef FwdRow<CPL, NC, VEC> Row;
    if (h < 2) {  // columns untouched; never linked
        if (pr_begin == 0) {
            Row s0;
            inv_load_row<CPL, NC, VEC>(coef, prev, P, 0, p0, c, s0);
            inv_finish_row<CPL, NC, VEC, PIX>(dst, P, 0, c, owned, s0.lo, s0.hi, dc_shift, fin, pix0, pix_stride);
        }
        __syncthreads();
        return;
    }
    // vertical inverse (columns first, dwt.go:412-421):
    //   xe[q] = s[q] - ((d[q-1] + d[q] + 2) >> 2)   (d[-1] := d[0]; no d[q] (odd h, last) := d[q-1])
    //   xo[q] = d[q] + ((xe[q] + xe[q+1]) >> 1)      (no xe[q+1] := xe[q])
    // Row indices are clamped into the plane so every load is unconditional (issued back to back); a clamped row is
    // only ever combined into values the has_d / has_next selects discard.
    Row s, dcur, dprev;
    int xe_lo[NC][H], xe_hi[NC][H];
    inv_load_row<CPL, NC, VEC>(coef, prev, P, pr_begin, p0, c, s);
    inv_load_row<CPL, NC, VEC>(coef, prev, P, halfH + min(pr_begin, nhigh - 1), p0, c, dcur);
    inv_load_row<CPL, NC, VEC>(coef, prev, P, halfH + max(pr_begin - 1, 0), p0, c, dprev);
    if (link_up) {
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                pub_mine[(k * CPL + j) * 64] = s.lo[k][j];
                pub_mine[(k * CPL + H + j) * 64] = s.hi[k][j];
                pub_mine[((NC + k) * CPL + j) * 64] = dcur.lo[k][j];
                pub_mine[((NC + k) * CPL + H + j) * 64] = dcur.hi[k][j];
            }
    }
    inv_compute_xe<CPL, NC, VEC>(pr_begin, nhigh, s, dprev, dcur, xe_lo, xe_hi);
    for (int q = pr_begin; q < pr_end; q++) {
        const bool first = q == pr_begin, last = q + 1 == pr_end;
        const bool has_d = q < nhigh;           // row 2q+1 exists
        const bool has_next = (q + 1) < halfH;  // row 2q+2 exists
        Row sn, dn;
        int xn_lo[NC][H], xn_hi[NC][H];
        if (last && link_down) {
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {
                    sn.lo[k][j] = pub_below[(k * CPL + j) * 64];
                    sn.hi[k][j] = pub_below[(k * CPL + H + j) * 64];
                    dn.lo[k][j] = pub_below[((NC + k) * CPL + j) * 64];
                    dn.hi[k][j] = pub_below[((NC + k) * CPL + H + j) * 64];
                }
        } else {
            inv_load_row<CPL, NC, VEC>(coef, prev, P, min(q + 1, halfH - 1), p0, c, sn);
            inv_load_row<CPL, NC, VEC>(coef, prev, P, halfH + min(q + 1, nhigh - 1), p0, c, dn);
        }
        inv_compute_xe<CPL, NC, VEC>(q + 1, nhigh, sn, dcur, dn, xn_lo, xn_hi);   // unused when !has_next
        inv_finish_row<CPL, NC, VEC, PIX>(dst, P, 2 * q, c, owned, xe_lo, xe_hi, dc_shift, fin, pix0, pix_stride);
        if (has_d) {
            int xo_lo[NC][H], xo_hi[NC][H];
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int j = 0; j < H; j++) {
                    const int pl = has_next ? avg1(xe_lo[k][j], xn_lo[k][j]) : xe_lo[k][j];
                    const int ph = has_next ? avg1(xe_hi[k][j], xn_hi[k][j]) : xe_hi[k][j];
                    xo_lo[k][j] = wadd(dcur.lo[k][j], pl);
                    xo_hi[k][j] = wadd(dcur.hi[k][j], ph);
                }
            inv_finish_row<CPL, NC, VEC, PIX>(dst, P, 2 * q + 1, c, owned, xo_lo, xo_hi, dc_shift, fin, pix0, pix_stride);
        }
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int j = 0; j < H; j++) {
                xe_lo[k][j] = xn_lo[k][j]; xe_hi[k][j] = xn_hi[k][j];
                dcur.lo[k][j] = dn.lo[k][j]; dcur.hi[k][j] = dn.hi[k][j];
            }
        if (first) __syncthreads();   // every wavefront of the workgroup arrives exactly once
    }
}

// ================================================================================
// fused tail: the small decomposition levels of one plane inside LDS
// ================================================================================
// Once a level's input (w*h <= 16384 samples, w <= 128) fits in LDS, the remaining levels are latency-bound
// launches of a few hundred wavefronts each.  One workgroup per plane runs them all: the level input lives in
// an LDS buffer, each of the 4 wavefronts streams a band of row pairs exactly like the global kernels (one
// column pair per lane, DPP neighbours, register sliding window), final coefficients go straight to global
// memory and the prefix that feeds the next level goes to the other LDS buffer.
#define TAIL_WAVES 16   /* 1024-thread workgroups: 16 wavefronts share one plane's levels */
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0)): in the level loops below every level ends with final coefficients on their way to memory, and a
// barrier that waits for their acknowledgement costs a store round trip (1.5-2 us) per level (measured with phase stamps:
// 4.3 + 2.1 + 1.5 us for three LDS levels whose arithmetic is a few hundred cycles).  Nothing another wave reads goes
// through memory here.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Explicit address spaces for the "LDS or memory, by index" accesses of the level routines: left to the compiler they become
// FLAT loads / stores (a select of two pointers), which tick both memory counters -- the next LDS wait (s_waitcnt lgkmcnt(0))
// then also waits for every flat STORE on its way to memory, i.e. a store round trip per loop iteration.
typedef __attribute__((address_space(3))) int32_t lds_i32_t;
typedef __attribute__((address_space(1))) int32_t glb_i32_t;
__device__ __forceinline__ int ld_lds(const int32_t *p) { return *(const lds_i32_t *)p; }
__device__ __forceinline__ int ld_glb(const int32_t *p) { return *(const glb_i32_t *)p; }
__device__ __forceinline__ void st_lds(int32_t *p, int v) { *(lds_i32_t *)p = v; }
#ifdef J2K_DEEP_NOSTORE_TAIL
__device__ __forceinline__ void st_glb(int32_t *p, int v) { }      // dev timing variant: results are wrong
#else
__device__ __forceinline__ void st_glb(int32_t *p, int v) { *(glb_i32_t *)p = v; }
#endif

// One forward level whose input lives in LDS (w <= 128, any h): cur, nxt: LDS; gout: memory.  A wave takes PER consecutive
// pair-rows at a time (2 columns per lane), requests ALL the rows they need up front -- 2 PER + 3 rows: the pair-row above for
// its d, the even row below -- and only then computes; the halo rows' horizontal pass is recomputed instead of exchanged
// (LDS reads are cheap, a barrier per level is all that is left).  Round 2's version marched row by row with a wait per
// LDS read and a branch per edge rule: 3.6 / 1.9 / 1.4 us for levels of 128 / 64 / 32 columns (phase stamps) against a few
// hundred cycles of arithmetic.
template <int PER>
__device__ __forceinline__ void lds_fwd_level(const int32_t *cur, int32_t *nxt, int32_t *gout, int w, int h, int n_next,
                                              int wave, int lane, int qlo, int qhi) {
    const int halfW = (w + 1) >> 1, halfH = (h + 1) >> 1;
    const int c = 2 * lane;
    const bool owned = c < w;
    constexpr int NROW = 2 * PER + 3;
    auto store = [&](int ro, int lo, int hi) {
        if (!owned) return;
        const int idxL = ro * w + lane, idxH = idxL + halfW;
        if (lane < halfW) { if (idxL < n_next) st_lds(nxt + idxL, lo); else st_glb(gout + idxL, lo); }
        if (lane < w - halfW) { if (idxH < n_next) st_lds(nxt + idxH, hi); else st_glb(gout + idxH, hi); }
    };
    const int c0 = owned ? c : 0, c1 = (c + 1 < w) ? c + 1 : 0;             // clamped: every lane reads a valid address
    for (int qa = qlo + wave * PER; qa < qhi; qa += TAIL_WAVES * PER) {
        int lo[NROW], hi[NROW];
        {
            int x[NROW][2];
#pragma unroll
            for (int i = 0; i < NROW; i++) {
                const int r = min(max(2 * qa - 2 + i, 0), h - 1);
                x[i][0] = ld_lds(cur + r * w + c0);
                x[i][1] = ld_lds(cur + r * w + c1);
            }
#pragma unroll
            for (int i = 0; i < NROW; i++) {
                if (!owned) x[i][0] = 0;
                if (c + 1 >= w) x[i][1] = 0;
                int l1[1], h1[1];
                hfwd<2, true>(x[i], c, w, l1, h1);
                lo[i] = l1[0]; hi[i] = h1[0];
            }
        }
        if (h < 2) { store(0, lo[2], hi[2]); return; }                       // dwt.go:74-76 down the columns: untouched
        // d of the pair-row above (rows 2qa-2, 2qa-1 and 2qa all exist when qa > 0)
        int dpl = wsub(lo[1], avg1(lo[0], lo[2])), dph = wsub(hi[1], avg1(hi[0], hi[2]));
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int q = qa + k;
            if (q >= qhi) break;
            const bool has_odd = 2 * q + 1 < h, has_next = 2 * q + 2 < h;
            const int el = lo[2 + 2 * k], eh = hi[2 + 2 * k];
            int dl = wsub(lo[3 + 2 * k], has_next ? avg1(el, lo[4 + 2 * k]) : el);
            int dh = wsub(hi[3 + 2 * k], has_next ? avg1(eh, hi[4 + 2 * k]) : eh);
            if (!has_odd) { dl = dpl; dh = dph; }                             // odd height, last row: its d mirrors d[n-2] (dwt.go:112-114)
            const int pl = (q == 0) ? dl : dpl, ph = (q == 0) ? dh : dph;   // d[-1] mirrors d[0] (dwt.go:99)
            store(q, wadd(el, avg2(pl, dl)), wadd(eh, avg2(ph, dh)));
            if (has_odd) store(halfH + q, dl, dh);
            dpl = dl; dph = dh;
        }
    }
}
// The same for the shapes that matter (w even, w / 2 a divisor of 64, h >= 2: the levels of power-of-two tiles): lane = one
// column PAIR, 64 / (w/2) bands side by side in a wave (the DPP shifts cross band boundaries, where the mirror rules
// override them), 8-byte LDS reads, no clamps or width tests per element.  About a third of the general routine's
// instructions; the deep workgroup's levels are bound by instruction issue on its one CU (16 waves, 4 per SIMD).
typedef int tl_v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) tl_v2i lds_tl_v2i_t;
template <int PER>
__device__ __forceinline__ void lds_fwd_level_fast(const int32_t *cur, int32_t *nxt, int32_t *gout, int w, int h, int n_next,
                                                   int wave, int lane, int qlo, int qhi) {
    const int HW = w >> 1, halfH = (h + 1) >> 1;
    const int R = 64 / HW;                          // bands per wave
    const int cp = lane & (HW - 1), g = lane / HW;
    const bool first = cp == 0, last = cp == HW - 1;
    constexpr int NROW = 2 * PER + 3;
    auto store = [&](int idx, int v) { if (idx < n_next) st_lds(nxt + idx, v); else st_glb(gout + idx, v); };
    for (int q0 = qlo + wave * R * PER; q0 < qhi; q0 += TAIL_WAVES * R * PER) {
        const int qa = q0 + g * PER;
        // (a band past the range repeats the reads of the range's last pair-row: the caller may hold only the rows the range needs)
        const int qr = min(qa, qhi - 1);
        int lo[NROW], hi[NROW];
        {
            tl_v2i x[NROW];
#pragma unroll
            for (int i = 0; i < NROW; i++) x[i] = *(const lds_tl_v2i_t *)(cur + min(max(2 * qr - 2 + i, 0), min(h - 1, 2 * qhi)) * w + 2 * cp);
#pragma unroll
            for (int i = 0; i < NROW; i++) {
                const int xr = from_right(x[i].x);
                const int d = wsub(x[i].y, last ? x[i].x : avg1(x[i].x, xr));
                int dl = from_left(d);
                if (first) dl = d;
                lo[i] = wadd(x[i].x, avg2(dl, d));
                hi[i] = d;
            }
        }
        int dpl = wsub(lo[1], avg1(lo[0], lo[2])), dph = wsub(hi[1], avg1(hi[0], hi[2]));
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int q = qa + k;
            const bool live = q < qhi, has_odd = 2 * q + 1 < h, has_next = 2 * q + 2 < h;
            const int el = lo[2 + 2 * k], eh = hi[2 + 2 * k];
            int dl = wsub(lo[3 + 2 * k], has_next ? avg1(el, lo[4 + 2 * k]) : el);
            int dh = wsub(hi[3 + 2 * k], has_next ? avg1(eh, hi[4 + 2 * k]) : eh);
            if (!has_odd) { dl = dpl; dh = dph; }
            const int pl = (q == 0) ? dl : dpl, ph = (q == 0) ? dh : dph;
            if (live) {
                store(q * w + cp, wadd(el, avg2(pl, dl)));
                store(q * w + HW + cp, wadd(eh, avg2(ph, dh)));
                if (has_odd) { st_glb(gout + (halfH + q) * w + cp, dl); st_glb(gout + (halfH + q) * w + HW + cp, dh); }   // never in the prefix
            }
            dpl = dl; dph = dh;
        }
    }
}
__device__ __forceinline__ bool tail_level_fast_shape(int w, int h) { return !(w & 1) && w >= 2 && w <= 128 && (64 % (w >> 1)) == 0 && h >= 2; }
// pair-rows [qlo, qhi) of the level (the whole level: 0, ceil(h / 2)); only the rows 2 qlo - 2 .. 2 qhi of `cur` are read
__device__ __forceinline__ void tail_fwd_level(const int32_t *cur, int32_t *nxt, int32_t *gout, int w, int h, int n_next,
                                               int wave, int lane, int qlo, int qhi) {
    const int nq = qhi - qlo;                        // uniform for the workgroup
    if (nq <= 0) return;
    if (tail_level_fast_shape(w, h)) {
        const int per_band = (nq + TAIL_WAVES * (128 / w) - 1) / (TAIL_WAVES * (128 / w));      // pair-rows per band if every band is used once
        if (per_band > 2) lds_fwd_level_fast<4>(cur, nxt, gout, w, h, n_next, wave, lane, qlo, qhi);
        else if (per_band > 1) lds_fwd_level_fast<2>(cur, nxt, gout, w, h, n_next, wave, lane, qlo, qhi);
        else lds_fwd_level_fast<1>(cur, nxt, gout, w, h, n_next, wave, lane, qlo, qhi);
        return;
    }
    if (nq > 2 * TAIL_WAVES) lds_fwd_level<4>(cur, nxt, gout, w, h, n_next, wave, lane, qlo, qhi);
    else if (nq > TAIL_WAVES) lds_fwd_level<2>(cur, nxt, gout, w, h, n_next, wave, lane, qlo, qhi);
    else lds_fwd_level<1>(cur, nxt, gout, w, h, n_next, wave, lane, qlo, qhi);
}
__device__ __forceinline__ void tail_fwd_level(const int32_t *cur, int32_t *nxt, int32_t *gout, int w, int h, int n_next, int wave, int lane) {
    tail_fwd_level(cur, nxt, gout, w, h, n_next, wave, lane, 0, (h + 1) >> 1);
}

__global__ __launch_bounds__(64 * TAIL_WAVES) void dwt53_tail_fwd_kernel(const TailPlane *__restrict__ planes, const int32_t *__restrict__ scr,
                                                             int32_t *__restrict__ coef) {
    extern __shared__ __attribute__((aligned(16))) int32_t tail_lds[];
    const TailPlane P = planes[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int w = P.w, h = P.h;
    const int n0 = w * h;
    int32_t *bufA = tail_lds, *bufB = tail_lds + ((n0 + 3) & ~3);
    const int32_t *src = scr + P.scr_off;
    if ((n0 & 3) == 0 && (P.scr_off & 3) == 0) {
        for (int i = threadIdx.x; i < n0 / 4; i += 64 * TAIL_WAVES) reinterpret_cast<int4 *>(bufA)[i] = reinterpret_cast<const int4 *>(src)[i];
    } else {
        for (int i = threadIdx.x; i < n0; i += 64 * TAIL_WAVES) bufA[i] = src[i];
    }
    __syncthreads();
    int32_t *gout = coef + P.coef_off;
    int32_t *cur = bufA, *nxt = bufB;
    for (int l = 0; l < P.nlev; l++) {
        const int wn = (w + 1) >> 1, hn = (h + 1) >> 1;
        const int n_next = (l == P.nlev - 1) ? 0 : wn * hn;
        tail_fwd_level(cur, nxt, gout, w, h, n_next, wave, lane);
        lds_barrier();
        int32_t *t = cur; cur = nxt; nxt = t;
        w = wn; h = hn;
    }
}

// One inverse level whose output fits LDS.  prev: LDS (the coarser level's result = elements below n_next); gcoef: the level's
// coefficients, LDS (coef_lds) or memory; dst: LDS (dst_lds) or memory -- both flags uniform for the workgroup.  Same shape as
// lds_fwd_level: a wave takes PER pair-rows at a time and requests all their rows first (low-pass rows q .. q+PER, high-pass
// rows q-1 .. q+PER), then runs the vertical steps (dwt.go:132-146 down the columns) and the horizontal inverse of each row.
template <int PER>
__device__ __forceinline__ void lds_inv_level(const int32_t *prev, const int32_t *gcoef, const int32_t *gcoef_hi, int32_t *dst, int w, int h, int n_next,
                                              int wave, int lane, bool coef_lds, bool dst_lds, int qlo, int qhi) {
    const int halfW = (w + 1) >> 1, halfH = (h + 1) >> 1, nhigh = h - halfH;
    const int c = 2 * lane;
    const bool owned = c < w;
    const int lL = (lane < halfW) ? lane : 0, lH = (lane < w - halfW) ? lane : 0;      // clamped: valid addresses for every lane
    auto ld1 = [&](int idx) { return (idx < n_next) ? ld_lds(prev + idx) : (coef_lds ? ld_lds(gcoef + idx) : ld_glb(gcoef + idx)); };
    // the high-pass rows through their own base: a caller that staged the coefficients as two compact runs of rows (dwt53_deep.inc)
    auto ld1h = [&](int idx) { return coef_lds ? ld_lds(gcoef_hi + idx) : ld_glb(gcoef_hi + idx); };
    auto finish = [&](int ro, int lo, int hi) {
        int l1[1] = {lo}, h1[1] = {hi}, x[2];
        hinv<2>(l1, h1, c, w, x);
        if (!owned) return;
        if (dst_lds) { st_lds(dst + ro * w + c, x[0]); if (c + 1 < w) st_lds(dst + ro * w + c + 1, x[1]); }
        else { st_glb(dst + ro * w + c, x[0]); if (c + 1 < w) st_glb(dst + ro * w + c + 1, x[1]); }
    };
    for (int qa = qlo + wave * PER; qa < qhi; qa += TAIL_WAVES * PER) {
        int sl[PER + 1], sh[PER + 1], dl[PER + 2], dh[PER + 2];                  // s_{qa+k}, d_{qa-1+k}
#pragma unroll
        for (int k = 0; k <= PER; k++) {
            const int row = min(qa + k, halfH - 1) * w;
            sl[k] = ld1(row + lL); sh[k] = ld1(row + halfW + lH);
        }
#pragma unroll
        for (int k = 0; k < PER + 2; k++) {
            const int row = (halfH + min(max(qa - 1 + k, 0), max(nhigh - 1, 0))) * w;
            dl[k] = (nhigh > 0) ? ld1h(row + lL) : 0; dh[k] = (nhigh > 0) ? ld1h(row + halfW + lH) : 0;
        }
        if (lane >= halfW) {
#pragma unroll
            for (int k = 0; k <= PER; k++) sl[k] = 0;
#pragma unroll
            for (int k = 0; k < PER + 2; k++) dl[k] = 0;
        }
        if (lane >= w - halfW) {
#pragma unroll
            for (int k = 0; k <= PER; k++) sh[k] = 0;
#pragma unroll
            for (int k = 0; k < PER + 2; k++) dh[k] = 0;
        }
        if (h < 2) { finish(0, sl[0], sh[0]); return; }
        // undo update: e_q = s_q - ((d_{q-1} + d_q + 2) >> 2); d_{-1} := d_0; no d_q (odd height, last row): d_q := d_{q-1}
        int el[PER + 1], eh[PER + 1];
#pragma unroll
        for (int k = 0; k <= PER; k++) {
            const int q = qa + k;
            const bool has_d = q < nhigh;
            int al = dl[k], ah = dh[k];
            const int bl = has_d ? dl[k + 1] : al, bh = has_d ? dh[k + 1] : ah;
            if (q == 0) { al = bl; ah = bh; }
            el[k] = wsub(sl[k], avg2(al, bl));
            eh[k] = wsub(sh[k], avg2(ah, bh));
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int q = qa + k;
            if (q >= qhi) break;
            const bool has_next = q + 1 < halfH;
            finish(2 * q, el[k], eh[k]);
            if (q < nhigh)       // undo predict: o_q = d_q + ((e_q + e_{q+1}) >> 1); no e_{q+1}: o_q = d_q + e_q
                finish(2 * q + 1, wadd(dl[k + 1], has_next ? avg1(el[k], el[k + 1]) : el[k]), wadd(dh[k + 1], has_next ? avg1(eh[k], eh[k + 1]) : eh[k]));
        }
    }
}
// The fast shapes (see lds_fwd_level_fast): lane = column pair, several bands per wave, 8-byte stores of the finished rows.
template <int PER>
__device__ __forceinline__ void lds_inv_level_fast(const int32_t *prev, const int32_t *gcoef, const int32_t *gcoef_hi, int32_t *dst, int w, int h, int n_next,
                                                   int wave, int lane, bool coef_lds, bool dst_lds, int qlo, int qhi) {
    const int HW = w >> 1, halfH = (h + 1) >> 1, nhigh = h - halfH;       // h >= 2: nhigh >= 1
    const int R = 64 / HW;
    const int cp = lane & (HW - 1), g = lane / HW;
    const bool first = cp == 0, last = cp == HW - 1;
    auto ld1 = [&](int idx) { return (idx < n_next) ? ld_lds(prev + idx) : (coef_lds ? ld_lds(gcoef + idx) : ld_glb(gcoef + idx)); };
    auto ld1h = [&](int idx) { return coef_lds ? ld_lds(gcoef_hi + idx) : ld_glb(gcoef_hi + idx); };      // high-pass rows (never below n_next)
    auto finish = [&](int ro, int lo, int hi) {
        int dl = from_left(hi);
        if (first) dl = hi;
        const int e = wsub(lo, avg2(dl, hi));
        const int er = from_right(e);
        const int o = wadd(hi, last ? e : avg1(e, er));
        return (tl_v2i){e, o};
    };
    auto put = [&](int ro, const tl_v2i &v) {
        if (dst_lds) *(lds_tl_v2i_t *)(dst + ro * w + 2 * cp) = v;
        else *(__attribute__((address_space(1))) tl_v2i *)(dst + ro * w + 2 * cp) = v;
    };
    for (int q0 = qlo + wave * R * PER; q0 < qhi; q0 += TAIL_WAVES * R * PER) {
        const int qa = q0 + g * PER;
        // (rows clamped to what the range needs -- s rows up to qhi, d rows qlo - 1 .. qhi: the caller may have staged only those)
        int sl[PER + 1], sh[PER + 1], dl[PER + 2], dh[PER + 2];
#pragma unroll
        for (int k = 0; k <= PER; k++) {
            const int row = min(qa + k, min(halfH - 1, qhi)) * w + cp;
            sl[k] = ld1(row); sh[k] = ld1(row + HW);
        }
#pragma unroll
        for (int k = 0; k < PER + 2; k++) {
            const int row = (halfH + min(max(qa - 1 + k, 0), min(nhigh - 1, qhi))) * w + cp;
            dl[k] = ld1h(row); dh[k] = ld1h(row + HW);
        }
        int el[PER + 1], eh[PER + 1];
#pragma unroll
        for (int k = 0; k <= PER; k++) {
            const int q = qa + k;
            const bool has_d = q < nhigh;
            int al = dl[k], ah = dh[k];
            const int bl = has_d ? dl[k + 1] : al, bh = has_d ? dh[k + 1] : ah;
            if (q == 0) { al = bl; ah = bh; }
            el[k] = wsub(sl[k], avg2(al, bl));
            eh[k] = wsub(sh[k], avg2(ah, bh));
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int q = qa + k;
            const bool has_next = q + 1 < halfH;
            // (the DPP shifts inside finish() need every lane: compute first, store under the lane's own condition)
            const tl_v2i re = finish(2 * q, el[k], eh[k]);
            const tl_v2i ro = finish(2 * q + 1, wadd(dl[k + 1], has_next ? avg1(el[k], el[k + 1]) : el[k]), wadd(dh[k + 1], has_next ? avg1(eh[k], eh[k + 1]) : eh[k]));
            if (q < qhi) {
                put(2 * q, re);
                if (q < nhigh) put(2 * q + 1, ro);
            }
        }
    }
}
// pair-rows [qlo, qhi) of the level (the whole level: 0, ceil(h / 2)): output rows 2 qlo .. 2 qhi - 1
__device__ __forceinline__ void tail_inv_level(const int32_t *prev, const int32_t *gcoef, const int32_t *gcoef_hi, int32_t *dst, int w, int h, int n_next,
                                               int wave, int lane, bool coef_lds, bool dst_lds, int qlo, int qhi) {
    const int nq = qhi - qlo;
    if (nq <= 0) return;
    if (tail_level_fast_shape(w, h)) {
        const int per_band = (nq + TAIL_WAVES * (128 / w) - 1) / (TAIL_WAVES * (128 / w));
        if (per_band > 2) lds_inv_level_fast<4>(prev, gcoef, gcoef_hi, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, qlo, qhi);
        else if (per_band > 1) lds_inv_level_fast<2>(prev, gcoef, gcoef_hi, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, qlo, qhi);
        else lds_inv_level_fast<1>(prev, gcoef, gcoef_hi, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, qlo, qhi);
        return;
    }
    if (nq > 2 * TAIL_WAVES) lds_inv_level<4>(prev, gcoef, gcoef_hi, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, qlo, qhi);
    else if (nq > TAIL_WAVES) lds_inv_level<2>(prev, gcoef, gcoef_hi, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, qlo, qhi);
    else lds_inv_level<1>(prev, gcoef, gcoef_hi, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, qlo, qhi);
}
__device__ __forceinline__ void tail_inv_level(const int32_t *prev, const int32_t *gcoef, int32_t *dst, int w, int h, int n_next,
                                               int wave, int lane, bool coef_lds, bool dst_lds) {
    tail_inv_level(prev, gcoef, gcoef, dst, w, h, n_next, wave, lane, coef_lds, dst_lds, 0, (h + 1) >> 1);
}

__global__ __launch_bounds__(64 * TAIL_WAVES) void dwt53_tail_inv_kernel(const TailPlane *__restrict__ planes, const int32_t *__restrict__ coef,
                                                             int32_t *__restrict__ scr) {
    extern __shared__ __attribute__((aligned(16))) int32_t tail_lds[];
    const TailPlane P = planes[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ws[32], hs[32];
    ws[0] = P.w; hs[0] = P.h;
    for (int l = 1; l <= P.nlev; l++) { ws[l] = (ws[l - 1] + 1) >> 1; hs[l] = (hs[l - 1] + 1) >> 1; }
    const int n1 = ws[1] * hs[1];
    int32_t *bufA = tail_lds, *bufB = tail_lds + ((n1 + 3) & ~3);      // X_{l0+1} (largest LDS-resident), X_{l0+2}, ...
    const int32_t *gcoef = coef + P.coef_off;
    // level index l counts from l0: X_l goes to bufA when (l odd), bufB when (l even, l>0), global scratch when l == 0
    for (int l = P.nlev - 1; l >= 0; l--) {
        const int n_next = (l == P.nlev - 1) ? 0 : ws[l + 1] * hs[l + 1];
        const int32_t *prev = ((l + 1) & 1) ? bufA : bufB;
        int32_t *dst = (l == 0) ? scr + P.scr_off : ((l & 1) ? bufA : bufB);
        tail_inv_level(prev, gcoef, dst, ws[l], hs[l], n_next, wave, lane, false, l != 0);
        lds_barrier();
    }
}

hipError_t launch_dwt53_tail_fwd(hipStream_t s, const TailPlane *planes, int nplanes, size_t lds_bytes, const int32_t *scr,
                                 int32_t *coef, hipEvent_t ev0, hipEvent_t ev1) {
    if (nplanes <= 0) return hipSuccess;
    if (lds_bytes > 64 * 1024) {   // above the default dynamic-LDS limit: raise it (gfx950 has 160 KiB per workgroup)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dwt53_tail_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipExtLaunchKernelGGL(dwt53_tail_fwd_kernel, dim3(nplanes), dim3(64 * TAIL_WAVES), lds_bytes, s, ev0, ev1, 0, planes, scr, coef);
    return hipGetLastError();
}
hipError_t launch_dwt53_tail_inv(hipStream_t s, const TailPlane *planes, int nplanes, size_t lds_bytes, const int32_t *coef,
                                 int32_t *scr, hipEvent_t ev0, hipEvent_t ev1) {
    if (nplanes <= 0) return hipSuccess;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dwt53_tail_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipExtLaunchKernelGGL(dwt53_tail_inv_kernel, dim3(nplanes), dim3(64 * TAIL_WAVES), lds_bytes, s, ev0, ev1, 0, planes, coef, scr);
    return hipGetLastError();
}

#include "dwt53_l0pix.inc"
#include "dwt53_deep.inc"
#include "dwt53_plane_wg.inc"

// ================================================================================
// launchers
// ================================================================================
template <int NW>
static hipError_t fwd_wg_go(hipStream_t s, const LevelLaunch &L, const int32_t *src, int32_t *out, int32_t *nxt, int dc) {
    const uint32_t *pix = reinterpret_cast<const uint32_t *>(src);
#define J2K_WG(FL) hipExtLaunchKernelGGL((dwt53_fwd_rgba8_wg_kernel<NW, FL, 6>), dim3(L.njobs), dim3(NW * 64), 0, s, L.ev_start, L.ev_stop, 0, \
                                         L.jobs, L.njobs, L.planes, pix, out, nxt, dc, L.pix_stride)
    // with a fused part the event pair brackets both launches (start on the first, stop on the second)
    const hipEvent_t ev_a = L.ev_start, ev_b = L.ev_stop;
    if (L.njobs2 > 0) {
#define J2K_WG2(W, FL) hipExtLaunchKernelGGL((dwt53_fwd_rgba8_wg2_kernel<W, FL, 5>), dim3(L.njobs2), dim3(W * 64), 0, s, ev_a, L.njobs > 0 ? nullptr : ev_b, 0, \
                                             L.jobs2, L.njobs2, L.planes, L.planes1, pix, out, L.nxt1, dc, L.pix_stride)
        if (L.wg2_waves == 16) { if (L.wg_store == 0) J2K_WG2(16, 0); else J2K_WG2(16, 1); }
        else if (L.wg2_waves == 10) { if (L.wg_store == 0) J2K_WG2(10, 0); else J2K_WG2(10, 1); }
        else { if (L.wg_store == 0) J2K_WG2(8, 0); else J2K_WG2(8, 1); }
#undef J2K_WG2
        if (L.njobs <= 0) return hipGetLastError();
    }
    const hipEvent_t ev_s = L.njobs2 > 0 ? nullptr : ev_a;
#undef J2K_WG
#define J2K_WG(FL) hipExtLaunchKernelGGL((dwt53_fwd_rgba8_wg_kernel<NW, FL, 6>), dim3(L.njobs), dim3(NW * 64), 0, s, ev_s, ev_b, 0, \
                                         L.jobs, L.njobs, L.planes, pix, out, nxt, dc, L.pix_stride)
    switch (L.wg_store) {
        case 0: J2K_WG(0); break;
        case 2: J2K_WG(2); break;
        case 4: J2K_WG(4); break;
        default: J2K_WG(1); break;
    }
#undef J2K_WG
    return hipGetLastError();
}
template <int CPL, int NC, bool VEC>
static hipError_t fwd_go(hipStream_t s, const LevelLaunch &L, const int32_t *src, int32_t *out, int32_t *nxt, int dc) {
    const int blocks = (L.njobs + 3) / 4;
    if constexpr (CPL == 8 && (NC == 3 || NC == 1) && VEC) {
        if (L.pix_stride > 0) {   // packed RGBA8 / Gray16 frame: its own instantiation, so the planar kernel is untouched
            if (L.pf) hipExtLaunchKernelGGL((dwt53_fwd_kernel<CPL, NC, VEC, true, true>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0,
                                            L.jobs, L.njobs, L.planes, src, out, nxt, dc, L.pix_stride);
            else hipExtLaunchKernelGGL((dwt53_fwd_kernel<CPL, NC, VEC, false, true>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0,
                                       L.jobs, L.njobs, L.planes, src, out, nxt, dc, L.pix_stride);
            return hipGetLastError();
        }
    }
    if (L.pix_stride > 0) return hipErrorInvalidValue;
    if (L.pf) hipExtLaunchKernelGGL((dwt53_fwd_kernel<CPL, NC, VEC, true, false>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0,
                                    L.jobs, L.njobs, L.planes, src, out, nxt, dc, 0);
    else hipExtLaunchKernelGGL((dwt53_fwd_kernel<CPL, NC, VEC, false, false>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0,
                               L.jobs, L.njobs, L.planes, src, out, nxt, dc, 0);
    return hipGetLastError();
}
template <int CPL, int NC, bool VEC>
static hipError_t inv_go(hipStream_t s, const LevelLaunch &L, const int32_t *coef, const int32_t *prev, int32_t *dst, int dc, int fin) {
    const int blocks = (L.njobs + 3) / 4;
    if constexpr (CPL == 8 && (NC == 3 || NC == 1) && VEC) {
        if (L.pix_stride > 0) {
            hipExtLaunchKernelGGL((dwt53_inv_kernel<CPL, NC, VEC, true>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0, L.jobs, L.njobs, L.planes, coef, prev, dst, dc, fin, L.pix_stride);
            return hipGetLastError();
        }
    }
    if (L.pix_stride > 0) return hipErrorInvalidValue;
    hipExtLaunchKernelGGL((dwt53_inv_kernel<CPL, NC, VEC, false>), dim3(blocks), dim3(256), 0, s, L.ev_start, L.ev_stop, 0, L.jobs, L.njobs, L.planes, coef, prev, dst, dc, fin, 0);
    return hipGetLastError();
}

#define J2K_DISPATCH(FN, ...)                                                        \
    do {                                                                             \
        if (L.njobs <= 0) return hipSuccess;                                         \
        if (!L.vec) {                                                                \
            if (L.ncomp == 3) return FN<2, 3, false>(__VA_ARGS__);                   \
            return FN<2, 1, false>(__VA_ARGS__);                                     \
        }                                                                            \
        if (L.ncomp == 3) {                                                          \
            if (L.cpl == 8) return FN<8, 3, true>(__VA_ARGS__);                      \
            if (L.cpl == 4) return FN<4, 3, true>(__VA_ARGS__);                      \
            return FN<2, 3, true>(__VA_ARGS__);                                      \
        }                                                                            \
        if (L.cpl == 8) return FN<8, 1, true>(__VA_ARGS__);                          \
        if (L.cpl == 4) return FN<4, 1, true>(__VA_ARGS__);                          \
        return FN<2, 1, true>(__VA_ARGS__);                                          \
    } while (0)

hipError_t launch_dwt53_fwd(hipStream_t s, const LevelLaunch &L, const int32_t *src, int32_t *out, int32_t *nxt, int dc_shift) {
    if (L.wg_waves > 0) {      // packed RGBA8 level 0, workgroup form (dwt53_l0pix.inc); geometry checked by the plan
        if (L.njobs <= 0 && L.njobs2 <= 0) return hipSuccess;
        if (L.pix_stride <= 0 || L.ncomp != 3) return hipErrorInvalidValue;
        if (L.wg_waves == 4) return fwd_wg_go<4>(s, L, src, out, nxt, dc_shift);
        if (L.wg_waves == 8) return fwd_wg_go<8>(s, L, src, out, nxt, dc_shift);
        return hipErrorInvalidValue;
    }
    if (L.pwaves > 0 && L.pnjobs > 0 && (L.ncomp == 1 || L.pix_stride <= 0 || L.pix_src == 4)) {     // planes in workgroup form (dwt53_plane_wg.inc)
#define J2K_PWG(NW, NC, SRC, MULTI, WPE) hipExtLaunchKernelGGL((dwt53_fwd_plane_wg_kernel<NW, NC, SRC, MULTI, WPE>), dim3(L.pnjobs), dim3(NW * 64), 0, s, L.ev_start, L.ev_stop, 0, \
                                                 L.pjobs, L.pnjobs, L.planes, (const void *)src, out, nxt, dc_shift, L.pix_stride, (int64_t)L.comp_elems)
#define J2K_PWGM(NW, NC, SRC, WPE) do { if (L.pmulti) J2K_PWG(NW, NC, SRC, true, WPE); else J2K_PWG(NW, NC, SRC, false, WPE); } while (0)
        if (L.pix_stride > 0 && L.pix_src != 1) {     // Gray8 / a channel of a four-channel pixel / RGBA64 (four waves per workgroup only)
            if (L.pwaves != 4 || L.comp_elems <= 0) return hipErrorInvalidValue;
            if (L.ncomp == 3) { if (L.pix_src != 4) return hipErrorInvalidValue; J2K_PWGM(4, 3, 4, 4); }
            else if (L.pix_src == 2) J2K_PWGM(4, 1, 2, 8);
            else if (L.pix_src == 3) J2K_PWGM(4, 1, 3, 8);
            else if (L.pix_src == 4) J2K_PWGM(4, 1, 4, 8);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
#define J2K_PWG2(NW) do { if (L.ncomp == 3) { if (L.pmulti) J2K_PWG(NW, 3, 0, true, 4); else J2K_PWG(NW, 3, 0, false, 5); } \
                          else if (L.pix_stride > 0) J2K_PWGM(NW, 1, 1, 8); \
                          else J2K_PWGM(NW, 1, 0, 8); } while (0)
        if (L.pwaves == 8) J2K_PWG2(8); else J2K_PWG2(4);
#undef J2K_PWG2
#undef J2K_PWGM
#undef J2K_PWG
        return hipGetLastError();
    }
    J2K_DISPATCH(fwd_go, s, L, src, out, nxt, dc_shift);
}
hipError_t launch_dwt53_inv(hipStream_t s, const LevelLaunch &L, const int32_t *coef, const int32_t *prev, int32_t *dst,
                            int dc_shift, int final_level) {
    if (L.wg_waves > 0) {      // level 0 straight to a packed RGBA8 frame, workgroup form (dwt53_l0pix.inc)
        if (L.njobs <= 0) return hipSuccess;
        if (L.pix_stride <= 0 || L.ncomp != 3 || !final_level) return hipErrorInvalidValue;
        uint32_t *pix = reinterpret_cast<uint32_t *>(dst);
#define J2K_INVWG(NW, WPE) hipExtLaunchKernelGGL((dwt53_inv_rgba8_wg_kernel<NW, WPE>), dim3(L.njobs), dim3(NW * 64), 0, s, L.ev_start, L.ev_stop, 0, \
                                             L.jobs, L.njobs, L.planes, coef, prev, pix, dc_shift, L.pix_stride)
        const int wpe = L.wg_store;    // (inverse: occupancy variant, J2K_L0_INV_WPE: 5 = everything in registers, 6 / 7 = odd row parked in LDS)
        if (L.wg_waves == 4) { if (wpe == 5) J2K_INVWG(4, 5); else if (wpe == 7) J2K_INVWG(4, 7); else J2K_INVWG(4, 6); }
        else if (L.wg_waves == 8) { if (wpe == 5) J2K_INVWG(8, 5); else if (wpe == 7) J2K_INVWG(8, 7); else J2K_INVWG(8, 6); }
        else return hipErrorInvalidValue;
#undef J2K_INVWG
        return hipGetLastError();
    }
    if (L.pwaves > 0 && L.pnjobs > 0 && (L.ncomp == 1 || L.pix_stride <= 0 || L.pix_src == 4)) {     // planes in workgroup form (dwt53_plane_wg.inc)
        if (L.pix_stride > 0 && !final_level) return hipErrorInvalidValue;
#define J2K_PWG(NW, NC, DST, MULTI, WPE) hipExtLaunchKernelGGL((dwt53_inv_plane_wg_kernel<NW, NC, DST, MULTI, WPE>), dim3(L.pnjobs), dim3(NW * 64), 0, s, L.ev_start, L.ev_stop, 0, \
                                                 L.pjobs, L.pnjobs, L.planes, coef, prev, (void *)dst, dc_shift, final_level, L.pix_stride, (int64_t)L.comp_elems)
#define J2K_PWGM(NW, NC, DST, WPE) do { if (L.pmulti) J2K_PWG(NW, NC, DST, true, WPE); else J2K_PWG(NW, NC, DST, false, WPE); } while (0)
        if (L.pix_stride > 0 && L.pix_src != 1) {
            if (L.pwaves != 4 || L.comp_elems <= 0) return hipErrorInvalidValue;
            if (L.ncomp == 3) { if (L.pix_src != 4) return hipErrorInvalidValue; if (L.pmulti) J2K_PWG(4, 3, 4, true, 3); else J2K_PWG(4, 3, 4, false, 5); }
            else if (L.pix_src == 2) J2K_PWGM(4, 1, 2, 8);
            else if (L.pix_src == 3) J2K_PWGM(4, 1, 3, 8);
            else if (L.pix_src == 4) J2K_PWGM(4, 1, 4, 8);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
#define J2K_PWG2(NW) do { if (L.ncomp == 3) { if (L.pmulti) J2K_PWG(NW, 3, 0, true, 3); else J2K_PWG(NW, 3, 0, false, 5); } \
                          else if (L.pix_stride > 0) J2K_PWGM(NW, 1, 1, 8); \
                          else J2K_PWGM(NW, 1, 0, 8); } while (0)
        if (L.pwaves == 8) J2K_PWG2(8); else J2K_PWG2(4);
#undef J2K_PWG2
#undef J2K_PWGM
#undef J2K_PWG
        return hipGetLastError();
    }
    J2K_DISPATCH(inv_go, s, L, coef, prev, dst, dc_shift, final_level);
}

}  // namespace j2k
