/*
 * j2k_oracle.c -- CPU restatement (plain C) of the go-jpeg2000 hot path.
 * TEST INFRASTRUCTURE ONLY -- see j2k_oracle.h.  Build: -O2 -fwrapv -ffp-contract=off.
 *
 * Go semantics reproduced on purpose (SURVEY.md "Numerical-semantics checklist"):
 *   - int32 >> is arithmetic (floor); signed overflow wraps (-fwrapv);
 *   - int32(float64) truncates toward zero; rounding is the explicit +-0.5;
 *   - no FMA contraction (amd64 Go never fuses): -ffp-contract=off;
 *   - unsigned shifts by >= the operand width yield 0 (shl32/shr64 helpers).
 */
#include "j2k_oracle.h"
#include "ht_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Go: an unsigned shift by >= the operand width yields 0 (x86/GPU hardware would mask the count) */
static inline uint32_t shl32(uint32_t x, uint32_t n) { return n >= 32 ? 0u : x << n; }
static inline uint64_t shl64(uint64_t x, uint64_t n) { return n >= 64 ? 0ull : x << n; }
static inline uint64_t shr64(uint64_t x, uint64_t n) { return n >= 64 ? 0ull : x >> n; }

/* ======================================================================== */
/* internal/mct/mct.go                                                       */
/* ======================================================================== */

void orc_dc_shift_fwd(int32_t *d, size_t n, int precision) {          /* mct.go:96-101 */
    int32_t shift = (int32_t)((uint32_t)1 << (precision - 1));
    for (size_t i = 0; i < n; i++) d[i] -= shift;
}

void orc_dc_shift_inv(int32_t *d, size_t n, int precision) {          /* mct.go:113-118 */
    int32_t shift = (int32_t)((uint32_t)1 << (precision - 1));
    for (size_t i = 0; i < n; i++) d[i] += shift;
}

void orc_rct_fwd(int32_t *r, int32_t *g, int32_t *b, size_t n) {      /* mct.go:28-38 */
    for (size_t i = 0; i < n; i++) {
        int32_t y = (r[i] + 2 * g[i] + b[i]) >> 2;
        int32_t u = b[i] - g[i];
        int32_t v = r[i] - g[i];
        r[i] = y; g[i] = u; b[i] = v;
    }
}

void orc_rct_inv(int32_t *y, int32_t *u, int32_t *v, size_t n) {      /* mct.go:56-66 */
    for (size_t i = 0; i < n; i++) {
        int32_t g = y[i] - ((u[i] + v[i]) >> 2);
        int32_t r = v[i] + g;
        int32_t b = u[i] + g;
        y[i] = r; u[i] = g; v[i] = b;
    }
}

void orc_ict_fwd(double *r, double *g, double *b, size_t n) {         /* mct.go:14-24 */
    for (size_t i = 0; i < n; i++) {
        double y  = 0.299 * r[i] + 0.587 * g[i] + 0.114 * b[i];
        double cb = -0.16875 * r[i] - 0.33126 * g[i] + 0.5 * b[i];
        double cr = 0.5 * r[i] - 0.41869 * g[i] - 0.08131 * b[i];
        r[i] = y; g[i] = cb; b[i] = cr;
    }
}

void orc_ict_inv(double *y, double *cb, double *cr, size_t n) {       /* mct.go:43-53 */
    for (size_t i = 0; i < n; i++) {
        double r = y[i] + 1.402 * cr[i];
        double g = y[i] - 0.34413 * cb[i] - 0.71414 * cr[i];
        double b = y[i] + 1.772 * cb[i];
        y[i] = r; cb[i] = g; cr[i] = b;
    }
}

/* ======================================================================== */
/* internal/dwt/dwt.go                                                       */
/* ======================================================================== */

static void deinterleave_i32(int32_t *d, int n, int32_t *tmp) {       /* dwt.go:265-284 */
    int half = (n + 1) / 2, i, j;
    for (i = 0, j = 0; i < n; i += 2, j++) tmp[j] = d[i];
    for (i = 1, j = half; i < n; i += 2, j++) tmp[j] = d[i];
    memcpy(d, tmp, (size_t)n * sizeof(int32_t));
}

static void interleave_i32(int32_t *d, int n, int32_t *tmp) {         /* dwt.go:287-306 */
    int half = (n + 1) / 2, i, j;
    memcpy(tmp, d, (size_t)n * sizeof(int32_t));
    for (i = 0, j = 0; j < half; i += 2, j++) d[i] = tmp[j];
    for (i = 1, j = half; j < n; i += 2, j++) d[i] = tmp[j];
}

static void deinterleave_f64(double *d, int n, double *tmp) {         /* dwt.go:309-326 */
    int half = (n + 1) / 2, i, j;
    for (i = 0, j = 0; i < n; i += 2, j++) tmp[j] = d[i];
    for (i = 1, j = half; i < n; i += 2, j++) tmp[j] = d[i];
    memcpy(d, tmp, (size_t)n * sizeof(double));
}

static void interleave_f64(double *d, int n, double *tmp) {           /* dwt.go:329-346 */
    int half = (n + 1) / 2, i, j;
    memcpy(tmp, d, (size_t)n * sizeof(double));
    for (i = 0, j = 0; j < half; i += 2, j++) d[i] = tmp[j];
    for (i = 1, j = half; j < n; i += 2, j++) d[i] = tmp[j];
}

static void fwd53_core(int32_t *d, int n, int32_t *tmp) {             /* dwt.go:73-118 */
    int i;
    if (n < 2) return;
    for (i = 1; i < n - 1; i += 2) d[i] -= (d[i - 1] + d[i + 1]) >> 1;
    if ((n & 1) == 0) d[n - 1] -= d[n - 2];
    d[0] += (d[1] + d[1] + 2) >> 2;
    for (i = 2; i < n - 1; i += 2) d[i] += (d[i - 1] + d[i + 1] + 2) >> 2;
    if ((n & 1) != 0) d[n - 1] += (d[n - 2] + d[n - 2] + 2) >> 2;
    deinterleave_i32(d, n, tmp);
}

static void inv53_core(int32_t *d, int n, int32_t *tmp) {             /* dwt.go:122-147 */
    int i;
    if (n < 2) return;
    interleave_i32(d, n, tmp);
    d[0] -= (d[1] + d[1] + 2) >> 2;
    for (i = 2; i < n - 1; i += 2) d[i] -= (d[i - 1] + d[i + 1] + 2) >> 2;
    if ((n & 1) != 0) d[n - 1] -= (d[n - 2] + d[n - 2] + 2) >> 2;
    for (i = 1; i < n - 1; i += 2) d[i] += (d[i - 1] + d[i + 1]) >> 1;
    if ((n & 1) == 0) d[n - 1] += d[n - 2];
}

/* dwt.go:150-157 -- the reference's literal constants, not full-precision ISO values */
static const double alpha97 = -1.586134342059924;
static const double beta97  = -0.052980118572961;
static const double gamma97 = 0.882911075530934;
static const double delta97 = 0.443506852043971;
static const double k97     = 1.230174104914001;
static const double k97Inv  = 0.812893066115961;

static void fwd97_core(double *d, int n, double *tmp) {               /* dwt.go:161-210 */
    int i;
    if (n < 2) return;
    for (i = 1; i < n - 1; i += 2) d[i] += alpha97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) == 0) d[n - 1] += 2 * alpha97 * d[n - 2];
    d[0] += 2 * beta97 * d[1];
    for (i = 2; i < n - 1; i += 2) d[i] += beta97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) != 0) d[n - 1] += 2 * beta97 * d[n - 2];
    for (i = 1; i < n - 1; i += 2) d[i] += gamma97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) == 0) d[n - 1] += 2 * gamma97 * d[n - 2];
    d[0] += 2 * delta97 * d[1];
    for (i = 2; i < n - 1; i += 2) d[i] += delta97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) != 0) d[n - 1] += 2 * delta97 * d[n - 2];
    for (i = 0; i < n; i += 2) d[i] *= k97Inv;
    for (i = 1; i < n; i += 2) d[i] *= k97;
    deinterleave_f64(d, n, tmp);
}

static void inv97_core(double *d, int n, double *tmp) {               /* dwt.go:213-262 */
    int i;
    if (n < 2) return;
    interleave_f64(d, n, tmp);
    for (i = 0; i < n; i += 2) d[i] *= k97;
    for (i = 1; i < n; i += 2) d[i] *= k97Inv;
    d[0] -= 2 * delta97 * d[1];
    for (i = 2; i < n - 1; i += 2) d[i] -= delta97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) != 0) d[n - 1] -= 2 * delta97 * d[n - 2];
    for (i = 1; i < n - 1; i += 2) d[i] -= gamma97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) == 0) d[n - 1] -= 2 * gamma97 * d[n - 2];
    d[0] -= 2 * beta97 * d[1];
    for (i = 2; i < n - 1; i += 2) d[i] -= beta97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) != 0) d[n - 1] -= 2 * beta97 * d[n - 2];
    for (i = 1; i < n - 1; i += 2) d[i] -= alpha97 * (d[i - 1] + d[i + 1]);
    if ((n & 1) == 0) d[n - 1] -= 2 * alpha97 * d[n - 2];
}

void orc_fwd53_1d(int32_t *d, int n) {
    if (n < 2) return;
    int32_t *tmp = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    fwd53_core(d, n, tmp);
    free(tmp);
}
void orc_inv53_1d(int32_t *d, int n) {
    if (n < 2) return;
    int32_t *tmp = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    inv53_core(d, n, tmp);
    free(tmp);
}
void orc_fwd97_1d(double *d, int n) {
    if (n < 2) return;
    double *tmp = (double *)malloc((size_t)n * sizeof(double));
    fwd97_core(d, n, tmp);
    free(tmp);
}
void orc_inv97_1d(double *d, int n) {
    if (n < 2) return;
    double *tmp = (double *)malloc((size_t)n * sizeof(double));
    inv97_core(d, n, tmp);
    free(tmp);
}

/* Forward2D53: all rows, then all columns (gather / transform / scatter); the
 * 4-way unrolling of dwt.go:357-406 does not change any result. */
void orc_fwd53_2d(int32_t *d, int w, int h) {                         /* dwt.go:356-407 */
    int m = w > h ? w : h, x, y;
    int32_t *tmp = (int32_t *)malloc((size_t)(m + 1) * sizeof(int32_t));
    int32_t *col = (int32_t *)malloc((size_t)(h + 1) * sizeof(int32_t));
    for (y = 0; y < h; y++) fwd53_core(d + (size_t)y * w, w, tmp);
    for (x = 0; x < w; x++) {
        for (y = 0; y < h; y++) col[y] = d[(size_t)y * w + x];
        fwd53_core(col, h, tmp);
        for (y = 0; y < h; y++) d[(size_t)y * w + x] = col[y];
    }
    free(tmp); free(col);
}

void orc_inv53_2d(int32_t *d, int w, int h) {                         /* dwt.go:410-429 */
    int m = w > h ? w : h, x, y;
    int32_t *tmp = (int32_t *)malloc((size_t)(m + 1) * sizeof(int32_t));
    int32_t *col = (int32_t *)malloc((size_t)(h + 1) * sizeof(int32_t));
    for (x = 0; x < w; x++) {
        for (y = 0; y < h; y++) col[y] = d[(size_t)y * w + x];
        inv53_core(col, h, tmp);
        for (y = 0; y < h; y++) d[(size_t)y * w + x] = col[y];
    }
    for (y = 0; y < h; y++) inv53_core(d + (size_t)y * w, w, tmp);
    free(tmp); free(col);
}

void orc_fwd97_2d(double *d, int w, int h) {                          /* dwt.go:432-451 */
    int m = w > h ? w : h, x, y;
    double *tmp = (double *)malloc((size_t)(m + 1) * sizeof(double));
    double *col = (double *)malloc((size_t)(h + 1) * sizeof(double));
    for (y = 0; y < h; y++) fwd97_core(d + (size_t)y * w, w, tmp);
    for (x = 0; x < w; x++) {
        for (y = 0; y < h; y++) col[y] = d[(size_t)y * w + x];
        fwd97_core(col, h, tmp);
        for (y = 0; y < h; y++) d[(size_t)y * w + x] = col[y];
    }
    free(tmp); free(col);
}

void orc_inv97_2d(double *d, int w, int h) {                          /* dwt.go:454-473 */
    int m = w > h ? w : h, x, y;
    double *tmp = (double *)malloc((size_t)(m + 1) * sizeof(double));
    double *col = (double *)malloc((size_t)(h + 1) * sizeof(double));
    for (x = 0; x < w; x++) {
        for (y = 0; y < h; y++) col[y] = d[(size_t)y * w + x];
        inv97_core(col, h, tmp);
        for (y = 0; y < h; y++) d[(size_t)y * w + x] = col[y];
    }
    for (y = 0; y < h; y++) inv97_core(d + (size_t)y * w, w, tmp);
    free(tmp); free(col);
}

/* Level l>0 re-interprets the contiguous prefix data[0 : w_l*h_l] as a dense
 * w_l x h_l matrix of stride w_l (NOT the Mallat LL sub-rectangle). */
void orc_decompose53(int32_t *d, int w, int h, int levels) {          /* dwt.go:524-531 */
    for (int l = 0; l < levels; l++) {
        orc_fwd53_2d(d, w, h);
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
}

void orc_reconstruct53(int32_t *d, int w, int h, int levels) {        /* dwt.go:534-548 */
    if (levels <= 0) return;
    int *ws = (int *)malloc(sizeof(int) * (size_t)levels), *hs = (int *)malloc(sizeof(int) * (size_t)levels);
    for (int l = 0; l < levels; l++) { ws[l] = w; hs[l] = h; w = (w + 1) / 2; h = (h + 1) / 2; }
    for (int l = levels - 1; l >= 0; l--) orc_inv53_2d(d, ws[l], hs[l]);
    free(ws); free(hs);
}

void orc_decompose97(double *d, int w, int h, int levels) {           /* dwt.go:551-558 */
    for (int l = 0; l < levels; l++) {
        orc_fwd97_2d(d, w, h);
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
}

void orc_reconstruct97(double *d, int w, int h, int levels) {         /* dwt.go:561-573 */
    if (levels <= 0) return;
    int *ws = (int *)malloc(sizeof(int) * (size_t)levels), *hs = (int *)malloc(sizeof(int) * (size_t)levels);
    for (int l = 0; l < levels; l++) { ws[l] = w; hs[l] = h; w = (w + 1) / 2; h = (h + 1) / 2; }
    for (int l = levels - 1; l >= 0; l--) orc_inv97_2d(d, ws[l], hs[l]);
    free(ws); free(hs);
}

/* ======================================================================== */
/* caller glue                                                               */
/* ======================================================================== */

/* Go's int32(float64), amd64: the compiler lowers the conversion (SSA op Cvt64Fto32) to CVTTSD2SL -- truncation toward zero,
 * and the x86 "integer indefinite" 0x80000000 for NaN and for every value whose truncation does not fit int32.  (Round 2
 * believed in a 64-bit convert + truncation there and restated that in cs_go_int32 and in the product's go_int32, while the
 * plain C casts of this file happened to compile to the same CVTTSD2SI r32 as Go's: two answers for one conversion.  A plain
 * cast of an out-of-range double is undefined in C, so it is spelled out now, once, for every conversion of this path.)
 * A value in (-2^31 - 1, -2^31] truncates to -2^31, which is the indefinite value anyway: one comparison is enough. */
static int32_t go_int32(double v) {
    return (v < 2147483648.0 && v > -2147483649.0) ? (int32_t)v : INT32_MIN;     /* NaN: both comparisons false */
}
static int32_t round_half_away(double v) {       /* encoder.go:238-242, tcd.go:527-531 */
    return v >= 0 ? go_int32(v + 0.5) : go_int32(v - 0.5);
}

void orc_preprocess(int32_t **planes, int ncomp, int w, int h, int precision,
                    int lossless, int num_resolutions, int quality) { /* encoder.go:216-281 */
    size_t n = (size_t)w * (size_t)h;
    for (int c = 0; c < ncomp; c++) orc_dc_shift_fwd(planes[c], n, precision);
    if (ncomp >= 3) {
        if (lossless) {
            orc_rct_fwd(planes[0], planes[1], planes[2], n);
        } else {
            double *f[3];
            for (int c = 0; c < 3; c++) {
                f[c] = (double *)malloc(n * sizeof(double));
                for (size_t i = 0; i < n; i++) f[c][i] = (double)planes[c][i];
            }
            orc_ict_fwd(f[0], f[1], f[2], n);
            for (int c = 0; c < 3; c++) {
                for (size_t i = 0; i < n; i++) planes[c][i] = round_half_away(f[c][i]);
                free(f[c]);
            }
        }
    }
    int levels = num_resolutions - 1;
    if (levels <= 0) levels = 5;
    for (int c = 0; c < ncomp; c++) {
        if (lossless) {
            orc_decompose53(planes[c], w, h, levels);
        } else {
            double *f = (double *)malloc(n * sizeof(double));
            for (size_t i = 0; i < n; i++) f[i] = (double)planes[c][i];
            orc_decompose97(f, w, h, levels);
            int q = quality;
            if (q <= 0) q = 100;
            double step = 1.0 / (double)q;
            for (size_t i = 0; i < n; i++) {
                double v = f[i];
                planes[c][i] = v >= 0 ? go_int32(v / step + 0.5) : go_int32(v / step - 0.5);
            }
            free(f);
        }
    }
}

void orc_tcd_forward_dwt(int32_t *d, int w, int h, int levels, int reversible) { /* tcd.go:508-534 */
    if (reversible) { orc_decompose53(d, w, h, levels); return; }
    size_t n = (size_t)w * (size_t)h;
    double *f = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) f[i] = (double)d[i];
    orc_decompose97(f, w, h, levels);
    for (size_t i = 0; i < n; i++) d[i] = round_half_away(f[i]);
    free(f);
}

void orc_tcd_inverse_dwt(int32_t *d, int w, int h, int levels, int reversible) { /* tcd.go:416-437 */
    if (reversible) { orc_reconstruct53(d, w, h, levels); return; }
    size_t n = (size_t)w * (size_t)h;
    double *f = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) f[i] = (double)d[i];
    orc_reconstruct97(f, w, h, levels);
    for (size_t i = 0; i < n; i++) d[i] = go_int32(f[i] + 0.5);       /* trunc: negatives round toward + */
    free(f);
}

/* ---- colorspace.go:54-501 ---------------------------------------------------------------------- */
static int32_t cs_go_int32(double v) { return go_int32(v); }       /* Go int32(float64) on amd64: see go_int32 */
static int32_t cs_clamp_to_int32(double v, double lo, double hi) {   /* colorspace.go:483-491 */
    if (v < lo) return cs_go_int32(lo);
    if (v > hi) return cs_go_int32(hi);
    return cs_go_int32(v + 0.5);
}
static double cs_clamp_f64(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double cs_lab_inverse_f(double t) {                           /* :293-299 */
    const double delta = 6.0 / 29.0;
    if (t > delta) return t * t * t;
    return 3 * delta * delta * (t - 4.0 / 29.0);
}
static double cs_srgb_gamma(double linear) {                         /* :302-307 */
    if (linear <= 0.0031308) return 12.92 * linear;
    return 1.055 * pow(linear, 1.0 / 2.4) - 0.055;
}

static double cs_srgb_inverse_gamma(double encoded) {               /* :310-315 (only the reference's tests call it) */
    if (encoded <= 0.04045) return encoded / 12.92;
    return pow((encoded + 0.055) / 1.055, 2.4);
}
double orc_pin_srgb_gamma(double linear) { return cs_srgb_gamma(linear); }
double orc_pin_srgb_inverse_gamma(double encoded) { return cs_srgb_inverse_gamma(encoded); }

void orc_convert_colorspace(int cs, int32_t **planes, int ncomp, size_t n, int precision) {
    const int need4 = (cs == 5 || cs == 11);
    if (ncomp < (need4 ? 4 : 3)) return;
    const double maxVal = (double)(int32_t)(((uint32_t)1 << precision) - 1);
    const double halfVal = (double)(int32_t)((uint32_t)1 << (precision - 1));
    int32_t *p0 = planes[0], *p1 = planes[1], *p2 = planes[2];
    for (size_t i = 0; i < n; i++) {
        double r, g, b;
        switch (cs) {
        case 3: case 16: case 17: case 4: {                          /* sYCC :92-116, YPbPr :429-452, e-sYCC :456-480 */
            double y = (double)p0[i], cb = (double)p1[i] - halfVal, cr = (double)p2[i] - halfVal;
            r = y + 1.5748 * cr; g = y - 0.1873 * cb - 0.4681 * cr; b = y + 1.8556 * cb;
            break;
        }
        case 7: case 8: {                                            /* BT.601 :119-142 */
            double y = (double)p0[i], cb = (double)p1[i] - halfVal, cr = (double)p2[i] - halfVal;
            r = y + 1.402 * cr; g = y - 0.344136 * cb - 0.714136 * cr; b = y + 1.772 * cb;
            break;
        }
        case 9: case 11: {                                           /* PhotoYCC :145-169, YCCK :218-247 */
            double scale = maxVal / 255.0;
            double y = (double)p0[i] / scale, c1 = (double)p1[i] / scale - 156.0, c2 = (double)p2[i] / scale - 156.0;
            r = y + 1.3584 * c2; g = y - 0.4302 * c1 - 0.7915 * c2; b = y + 2.2179 * c1;
            if (cs == 11) {
                double k = (double)planes[3][i] / maxVal;
                r = r * scale * (1 - k); g = g * scale * (1 - k); b = b * scale * (1 - k);
            } else { r = r * scale; g = g * scale; b = b * scale; }
            break;
        }
        case 10: {                                                   /* CMY :172-189 */
            int32_t mv = (int32_t)(((uint32_t)1 << precision) - 1);
            p0[i] = (int32_t)((uint32_t)mv - (uint32_t)p0[i]); p1[i] = (int32_t)((uint32_t)mv - (uint32_t)p1[i]);
            p2[i] = (int32_t)((uint32_t)mv - (uint32_t)p2[i]);
            continue;
        }
        case 5: {                                                    /* CMYK :192-215 */
            double c = (double)p0[i] / maxVal, m = (double)p1[i] / maxVal, y = (double)p2[i] / maxVal, k = (double)planes[3][i] / maxVal;
            r = (1 - c) * (1 - k) * maxVal; g = (1 - m) * (1 - k) * maxVal; b = (1 - y) * (1 - k) * maxVal;
            break;
        }
        case 12: case 13: {                                          /* CIELab :250-290, CIEJab :319-359 */
            double L = (double)p0[i] / maxVal * 100.0, a = (double)p1[i] / maxVal * 255.0 - 128.0, bb = (double)p2[i] / maxVal * 255.0 - 128.0;
            double fy = (L + 16.0) / 116.0, fx = a / 500.0 + fy, fz = fy - bb / 200.0;
            double x = 0.96422 * cs_lab_inverse_f(fx), y = 1.0 * cs_lab_inverse_f(fy), z = 0.82521 * cs_lab_inverse_f(fz);
            double rl = 3.2404542 * x - 1.5371385 * y - 0.4985314 * z;
            double gl = -0.9692660 * x + 1.8760108 * y + 0.0415560 * z;
            double bl = 0.0556434 * x - 0.2040259 * y + 1.0572252 * z;
            r = cs_srgb_gamma(rl) * maxVal; g = cs_srgb_gamma(gl) * maxVal; b = cs_srgb_gamma(bl) * maxVal;
            break;
        }
        case 14: {                                                   /* e-sRGB :362-388 */
            double er = (double)p0[i] / maxVal * 1.25 - 0.25, eg = (double)p1[i] / maxVal * 1.25 - 0.25, eb = (double)p2[i] / maxVal * 1.25 - 0.25;
            r = cs_srgb_gamma(cs_clamp_f64(er, 0, 1)) * maxVal; g = cs_srgb_gamma(cs_clamp_f64(eg, 0, 1)) * maxVal;
            b = cs_srgb_gamma(cs_clamp_f64(eb, 0, 1)) * maxVal;
            break;
        }
        case 15: {                                                   /* ROMM-RGB :391-426 */
            double rr = pow((double)p0[i] / maxVal, 1.8), gr = pow((double)p1[i] / maxVal, 1.8), br = pow((double)p2[i] / maxVal, 1.8);
            double x = 0.7977 * rr + 0.1352 * gr + 0.0313 * br;
            double y = 0.2880 * rr + 0.7119 * gr + 0.0001 * br;
            double z = 0.0000 * rr + 0.0000 * gr + 0.8249 * br;
            double rl = 3.2404542 * x - 1.5371385 * y - 0.4985314 * z;
            double gl = -0.9692660 * x + 1.8760108 * y + 0.0415560 * z;
            double bl = 0.0556434 * x - 0.2040259 * y + 1.0572252 * z;
            r = cs_srgb_gamma(cs_clamp_f64(rl, 0, 1)) * maxVal; g = cs_srgb_gamma(cs_clamp_f64(gl, 0, 1)) * maxVal;
            b = cs_srgb_gamma(cs_clamp_f64(bl, 0, 1)) * maxVal;
            break;
        }
        default: return;                                             /* :86-89: no conversion */
        }
        p0[i] = cs_clamp_to_int32(r, 0, maxVal);
        p1[i] = cs_clamp_to_int32(g, 0, maxVal);
        p2[i] = cs_clamp_to_int32(b, 0, maxVal);
    }
}

/* ---- RawEncoder / RawDecoder, mqc.go:516-600 ------------------------------------------------- */
long orc_raw_encode(const uint8_t *bits, size_t n, uint8_t *out, size_t cap) {
    uint32_t c = 0; int ct = 8; size_t pos = 0;
    for (size_t i = 0; i < n; i++) {                                  /* EncodeBit, mqc.go:577-589 */
        ct--;
        c = c + ((uint32_t)(bits[i] & 1) << ct);
        if (ct == 0) {
            if (pos >= cap) return -1;
            out[pos++] = (uint8_t)c;
            ct = 8;
            if ((uint8_t)c == 0xFF) ct = 7;
            c = 0;
        }
    }
    if (ct < 8) { if (pos >= cap) return -1; out[pos++] = (uint8_t)c; } /* Flush, mqc.go:592-599 */
    return (long)pos;
}

void orc_raw_decode(const uint8_t *data, size_t len, size_t n, uint8_t *bits) {
    size_t pos = 0; uint8_t c = 0; int ct = 0;
    for (size_t i = 0; i < n; i++) {                                  /* DecodeBit, mqc.go:535-557 */
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

/* ---- pixels: encoder.go:79-213, decoder.go:417-588 ------------------------------------------ */
static int32_t go_mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }   /* Go int32 multiply wraps */
static int be16p(const uint8_t *p) { return (p[0] << 8) | p[1]; }

int orc_extract_image_data(int format, const uint8_t *pix, size_t stride, int w, int h, int target_precision, int32_t **planes) {
    static const int comps[6] = {1, 1, 3, 3, 4, 4}, prec[6] = {8, 16, 8, 16, 8, 16};
    if (format < 0 || format > 5) return 0;
    const int nc = comps[format];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *row = pix + (size_t)y * stride;
            const size_t idx = (size_t)y * w + x;
            switch (format) {
            case 0: planes[0][idx] = row[x]; break;                                           /* GrayAt(x,y).Y */
            case 1: planes[0][idx] = be16p(row + 2 * x); break;                               /* Gray16At */
            case 2: case 4:                                                                   /* RGBAAt / NRGBAAt */
                for (int c = 0; c < nc; c++) planes[c][idx] = row[4 * x + c];
                break;
            default:                                                                          /* RGBA64At / NRGBA64At */
                for (int c = 0; c < nc; c++) planes[c][idx] = be16p(row + 8 * x + 2 * c);
            }
        }
    if (target_precision > 0 && target_precision <= 16 && target_precision != prec[format]) {  /* encoder.go:196-210 */
        const int32_t srcMax = (int32_t)((1 << prec[format]) - 1), dstMax = (int32_t)((1 << target_precision) - 1);
        for (int c = 0; c < nc; c++)
            for (size_t i = 0; i < (size_t)w * h; i++) planes[c][i] = go_mul32(planes[c][i], dstMax) / srcMax;
    }
    return nc;
}

int orc_create_image(int32_t *const *planes, int ncomp, int precision, int w, int h, uint8_t *pix, size_t stride) {
    if (ncomp != 1 && ncomp != 3 && ncomp != 4) return 0;                                     /* decoder.go:583-585 */
    const int32_t maxVal = (int32_t)((1 << precision) - 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const size_t idx = (size_t)y * w + x;
            uint8_t *row = pix + (size_t)y * stride;
            int32_t v[4] = {0, 0, 0, 0};
            for (int c = 0; c < ncomp; c++) {
                int32_t t = planes[c][idx];
                if (t < 0) t = 0;
                if (t > maxVal) t = maxVal;
                if (precision <= 8) { if (precision != 8) t = go_mul32(t, 255) / maxVal; }
                else t = go_mul32(t, 65535) / maxVal;
                v[c] = t;
            }
            if (ncomp == 1) {
                if (precision <= 8) row[x] = (uint8_t)v[0];                                    /* color.Gray{Y: uint8(v)} */
                else { const uint16_t u = (uint16_t)v[0]; row[2 * x] = (uint8_t)(u >> 8); row[2 * x + 1] = (uint8_t)u; }
            } else if (precision <= 8) {
                row[4 * x] = (uint8_t)v[0]; row[4 * x + 1] = (uint8_t)v[1]; row[4 * x + 2] = (uint8_t)v[2];
                row[4 * x + 3] = ncomp == 4 ? (uint8_t)v[3] : 255;
            } else {
                const uint16_t o[4] = {(uint16_t)v[0], (uint16_t)v[1], (uint16_t)v[2], ncomp == 4 ? (uint16_t)v[3] : (uint16_t)65535};
                for (int c = 0; c < 4; c++) { row[8 * x + 2 * c] = (uint8_t)(o[c] >> 8); row[8 * x + 2 * c + 1] = (uint8_t)o[c]; }
            }
        }
    return ncomp;
}

void orc_postprocess(int32_t **planes, int ncomp, size_t n, int precision,
                     int reversible, int mct, int is_signed) {        /* decoder.go:321-348 */
    if (mct && ncomp >= 3) {
        if (reversible) {
            orc_rct_inv(planes[0], planes[1], planes[2], n);
        } else {
            double *f[3];
            for (int c = 0; c < 3; c++) {
                f[c] = (double *)malloc(n * sizeof(double));
                for (size_t i = 0; i < n; i++) f[c][i] = (double)planes[c][i];
            }
            orc_ict_inv(f[0], f[1], f[2], n);
            for (int c = 0; c < 3; c++) {
                for (size_t i = 0; i < n; i++) planes[c][i] = go_int32(f[c][i] + 0.5);
                free(f[c]);
            }
        }
    }
    if (!is_signed)
        for (int c = 0; c < ncomp; c++) orc_dc_shift_inv(planes[c], n, precision);
}

/* ======================================================================== */
/* internal/entropy/mqc.go -- MQ coder                                       */
/* ======================================================================== */

/* ISO/IEC 15444-1 Table C.2 (47 rows: Qe, NMPS, NLPS, SWITCH).  The reference
 * carries the OpenJPEG-style 94-entry form (mqc.go:21-116): state 2i has MPS 0,
 * state 2i+1 has MPS 1; it is generated here by rule and checked entry by entry
 * against the reference literal in tests/test_oracle_tables.py. */
static const uint16_t iso_qe[47] = {
    0x5601, 0x3401, 0x1801, 0x0AC1, 0x0521, 0x0221, 0x5601, 0x5401, 0x4801, 0x3801,
    0x3001, 0x2401, 0x1C01, 0x1601, 0x5601, 0x5401, 0x5101, 0x4801, 0x3801, 0x3401,
    0x3001, 0x2801, 0x2401, 0x2201, 0x1C01, 0x1801, 0x1601, 0x1401, 0x1201, 0x1101,
    0x0AC1, 0x09C1, 0x08A1, 0x0521, 0x0441, 0x02A1, 0x0221, 0x0141, 0x0111, 0x0085,
    0x0049, 0x0025, 0x0015, 0x0009, 0x0005, 0x0001, 0x5601};
static const uint8_t iso_nmps[47] = {
    1, 2, 3, 4, 5, 38, 7, 8, 9, 10, 11, 12, 13, 29, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24,
    25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 45, 46};
static const uint8_t iso_nlps[47] = {
    1, 6, 9, 12, 29, 33, 6, 14, 14, 14, 17, 18, 20, 21, 14, 14, 15, 16, 17, 18, 19, 19, 20, 21,
    22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 46};
static const uint8_t iso_switch[47] = {
    1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static uint32_t mqQe[94];
static uint8_t mqNMPS[94], mqNLPS[94];
static uint8_t lutZCCtx[4 * 256], lutSignCtx[256], lutSignPred[256];
static int tables_ready = 0;

enum { CtxZC0 = 0, CtxSC0 = 9, CtxMag0 = 14, CtxMag1 = 15, CtxMag2 = 16, CtxRL = 17, CtxUni = 18,
       NumContexts = 19 };                                            /* mqc.go:135-166 */
enum { BandLL = 0, BandHL = 1, BandLH = 2, BandHH = 3 };              /* t1.go:125-130 */

static void init_tables(void) {
    if (tables_ready) return;
    for (int i = 0; i < 47; i++) {
        for (int m = 0; m < 2; m++) {
            int s = 2 * i + m;
            mqQe[s] = iso_qe[i];
            mqNMPS[s] = (uint8_t)(2 * iso_nmps[i] + m);
            mqNLPS[s] = (uint8_t)(2 * iso_nlps[i] + (iso_switch[i] ? 1 - m : m));
        }
    }
    /* t1_luts.go:34-110 -- ZC contexts (note the HL h/v swap) */
    for (int band = 0; band < 4; band++) {
        for (int p = 0; p < 256; p++) {
            int w = p & 1, e = (p >> 1) & 1, n = (p >> 2) & 1, s = (p >> 3) & 1;
            int d = ((p >> 4) & 1) + ((p >> 5) & 1) + ((p >> 6) & 1) + ((p >> 7) & 1);
            int hh = w + e, v = n + s, ctx = 0;
            if (band == BandHL) { int t = hh; hh = v; v = t; }
            if (band == BandHH) {
                int hv = hh + v;
                if (hv >= 3) ctx = 8;
                else if (hv == 2) ctx = d >= 2 ? 7 : (d >= 1 ? 6 : 5);
                else if (hv == 1) ctx = d >= 2 ? 4 : 3;
                else ctx = d >= 2 ? 2 : (d >= 1 ? 1 : 0);
            } else {
                if (hh == 2) ctx = 8;
                else if (hh == 1) ctx = v >= 1 ? 7 : (d >= 1 ? 6 : 5);
                else if (v == 2) ctx = 4;
                else if (v == 1) ctx = d >= 1 ? 3 : 2;
                else if (d >= 2) ctx = 1;
                else ctx = 0;
            }
            lutZCCtx[band * 256 + p] = (uint8_t)ctx;
        }
    }
    /* t1_luts.go:153-230 -- sign context / prediction from packed (sig,chi) x WENS */
    for (int i = 0; i < 256; i++) {
        int wSig = i & 1, wChi = (i >> 1) & 1, eSig = (i >> 2) & 1, eChi = (i >> 3) & 1;
        int nSig = (i >> 4) & 1, nChi = (i >> 5) & 1, sSig = (i >> 6) & 1, sChi = (i >> 7) & 1;
        int hc = 0, vc = 0, pred = 0, ctx = 0;
        if (wSig) hc += wChi ? -1 : 1;
        if (eSig) hc += eChi ? -1 : 1;
        if (nSig) vc += nChi ? -1 : 1;
        if (sSig) vc += sChi ? -1 : 1;
        if (hc < 0) { pred = 1; hc = -hc; }
        if (hc == 0 && vc < 0) { pred = 1; vc = -vc; }
        if (hc == 1) ctx = vc == 1 ? 4 : (vc == 0 ? 2 : 1);
        else if (hc == 0) ctx = vc == 1 ? 1 : 0;
        else if (hc == 2) ctx = 3;
        lutSignCtx[i] = (uint8_t)ctx;
        lutSignPred[i] = (uint8_t)pred;
    }
    tables_ready = 1;
}

void orc_mq_table(uint32_t qe[94], uint8_t nmps[94], uint8_t nlps[94]) {
    init_tables();
    memcpy(qe, mqQe, sizeof(mqQe)); memcpy(nmps, mqNMPS, 94); memcpy(nlps, mqNLPS, 94);
}
void orc_t1_luts(uint8_t zc[1024], uint8_t sc[256], uint8_t sp[256]) {
    init_tables();
    memcpy(zc, lutZCCtx, 1024); memcpy(sc, lutSignCtx, 256); memcpy(sp, lutSignPred, 256);
}

/* ---- MQ encoder: mqc.go:169-349 == t1_fast.go:11-34 + t1_fast5.go:878-898 -- */
typedef struct {
    uint32_t A, C, CT;
    uint8_t *buf;          /* buf[0] is the 0 sentinel ("bp[-1]" in OpenJPEG terms) */
    size_t cap;            /* capacity of buf */
    size_t bp;
    int overflow;
    uint8_t ctx[NumContexts];
} mq_enc;

static void mq_enc_init(mq_enc *e, uint8_t *buf, size_t cap) {       /* mqc.go:185-201 */
    e->A = 0x8000; e->C = 0; e->CT = 12;
    e->buf = buf; e->cap = cap; e->bp = 0; e->overflow = 0;
    buf[0] = 0;
    memset(e->ctx, 0, sizeof(e->ctx));
    e->ctx[CtxUni] = 92;
}

static void mq_byte_out(mq_enc *e) {                                 /* mqc.go:270-310, t1_fast.go:11-34 */
    if (e->bp + 1 >= e->cap) { e->overflow = 1; e->CT = 8; return; }
    if (e->buf[e->bp] == 0xFF) {
        e->bp++; e->buf[e->bp] = (uint8_t)(e->C >> 20); e->C &= 0xFFFFF; e->CT = 7;
        return;
    }
    if ((e->C & 0x8000000) == 0) {
        e->bp++; e->buf[e->bp] = (uint8_t)(e->C >> 19); e->C &= 0x7FFFF; e->CT = 8;
        return;
    }
    e->buf[e->bp]++;
    if (e->buf[e->bp] == 0xFF) {
        e->C &= 0x7FFFFFF;
        e->bp++; e->buf[e->bp] = (uint8_t)(e->C >> 20); e->C &= 0xFFFFF; e->CT = 7;
        return;
    }
    e->bp++; e->buf[e->bp] = (uint8_t)(e->C >> 19); e->C &= 0x7FFFF; e->CT = 8;
}

static void mq_renorm_enc(mq_enc *e) {                               /* mqc.go:258-267 */
    while ((e->A & 0x8000) == 0) {
        e->A <<= 1; e->C <<= 1; e->CT--;
        if (e->CT == 0) mq_byte_out(e);
    }
}

static void mq_encode(mq_enc *e, int ctx, int decision) {            /* mqc.go:224-255 */
    uint8_t st = e->ctx[ctx];
    uint32_t qe = mqQe[st];
    uint8_t mps = st & 1;
    e->A -= qe;
    if ((uint8_t)decision == mps) {
        if ((e->A & 0x8000) == 0) {
            if (e->A < qe) e->A = qe; else e->C += qe;
            e->ctx[ctx] = mqNMPS[st];
            mq_renorm_enc(e);
        } else {
            e->C += qe;
        }
    } else {
        if (e->A < qe) e->C += qe; else e->A = qe;
        e->ctx[ctx] = mqNLPS[st];
        mq_renorm_enc(e);
    }
}

/* Flush (mqc.go:313-332): returns the length of buf[1:endPos] (0 == nil). */
static size_t mq_flush(mq_enc *e) {
    uint32_t tempC = e->C + e->A;                                    /* setbits, mqc.go:335-341 */
    e->C |= 0xFFFF;
    if (e->C >= tempC) e->C -= 0x8000;
    e->C <<= e->CT; mq_byte_out(e);
    e->C <<= e->CT; mq_byte_out(e);
    size_t end = e->bp + 1;
    if (end > 0 && e->buf[end - 1] == 0xFF) end--;
    return end > 1 ? end - 1 : 0;
}

long orc_mq_encode(const uint8_t *ctx, const uint8_t *dec, size_t n, uint8_t *out, size_t cap) {
    init_tables();
    size_t bcap = n * 2 + 64;
    uint8_t *buf = (uint8_t *)malloc(bcap);
    mq_enc e; mq_enc_init(&e, buf, bcap);
    for (size_t i = 0; i < n; i++) mq_encode(&e, ctx[i], dec[i]);
    size_t len = mq_flush(&e);
    long r = (long)len;
    if (e.overflow || len > cap) r = -1; else memcpy(out, buf + 1, len);
    free(buf);
    return r;
}

/* ---- MQ decoder: mqc.go:352-497 -------------------------------------------- */
typedef struct {
    uint32_t C, A, CT;
    long bp;
    const uint8_t *data;
    long len;
    uint8_t ctx[NumContexts];
} mq_dec;

static void mq_byte_in(mq_dec *d) {                                  /* mqc.go:402-439 */
    if (d->bp < 0) d->bp = 0;
    if (d->bp >= d->len) { d->C += 0xFF00; d->CT = 8; return; }
    uint8_t next = (d->bp + 1 < d->len) ? d->data[d->bp + 1] : 0xFF;
    if (d->data[d->bp] == 0xFF) {
        if (next > 0x8F) { d->C += 0xFF00; d->CT = 8; }
        else { d->bp++; d->C += (uint32_t)next << 9; d->CT = 7; }
    } else {
        d->bp++; d->C += (uint32_t)next << 8; d->CT = 8;
    }
}

static void mq_dec_init(mq_dec *d, const uint8_t *data, size_t len) { /* mqc.go:370-399 */
    d->A = 0x8000; d->C = 0; d->CT = 0; d->data = data; d->len = (long)len; d->bp = -1;
    memset(d->ctx, 0, sizeof(d->ctx));
    d->ctx[CtxUni] = 92;
    if (len == 0) d->C = (uint32_t)0xFF << 16;
    else { d->bp = 0; d->C = (uint32_t)data[0] << 16; }
    mq_byte_in(d);
    d->C <<= 7;
    d->CT -= 7;
    d->A = 0x8000;
}

static void mq_renorm_dec(mq_dec *d) {                               /* mqc.go:488-497 */
    while ((d->A & 0x8000) == 0) {
        if (d->CT == 0) mq_byte_in(d);
        d->A <<= 1; d->C <<= 1; d->CT--;
    }
}

static int mq_decode(mq_dec *d, int ctx) {                           /* mqc.go:443-485 */
    uint8_t st = d->ctx[ctx];
    uint32_t qe = mqQe[st];
    int mps = st & 1, decision;
    d->A -= qe;
    if ((d->C >> 16) < qe) {
        if (d->A < qe) { d->A = qe; decision = mps; d->ctx[ctx] = mqNMPS[st]; }
        else { d->A = qe; decision = 1 - mps; d->ctx[ctx] = mqNLPS[st]; }
        mq_renorm_dec(d);
        return decision;
    }
    d->C -= qe << 16;
    if ((d->A & 0x8000) == 0) {
        if (d->A < qe) { decision = 1 - mps; d->ctx[ctx] = mqNLPS[st]; }
        else { decision = mps; d->ctx[ctx] = mqNMPS[st]; }
        mq_renorm_dec(d);
        return decision;
    }
    return mps;
}

void orc_mq_decode(const uint8_t *bytes, size_t nbytes, const uint8_t *ctx, size_t n, uint8_t *dec_out) {
    init_tables();
    mq_dec d; mq_dec_init(&d, bytes, nbytes);
    for (size_t i = 0; i < n; i++) dec_out[i] = (uint8_t)mq_decode(&d, ctx[i]);
}

/* ======================================================================== */
/* internal/entropy/t1.go + t1_fast5.go -- EBCOT-like T1                     */
/* ======================================================================== */

enum { T1Sig = 1, T1Visit = 2, T1Refine = 4, T1SignNeg = 8,
       T1SigN = 16, T1SigS = 32, T1SigE = 64, T1SigW = 128 };         /* t1.go:74-91 */

typedef struct {
    int w, h, stride, band;
    int32_t *data;      /* magnitudes */
    uint8_t *flags;     /* (w+2)*(h+2), 1-sample border */
} t1_state;

static inline int zc_packed(const uint8_t *f, int stride) {          /* t1_fast5.go:118-125 */
    return (f[-1] & T1Sig) | ((f[1] & T1Sig) << 1) | ((f[-stride] & T1Sig) << 2) |
           ((f[stride] & T1Sig) << 3) | ((f[-stride - 1] & T1Sig) << 4) |
           ((f[-stride + 1] & T1Sig) << 5) | ((f[stride - 1] & T1Sig) << 6) |
           ((f[stride + 1] & T1Sig) << 7);
}

static inline int sc_index(uint8_t fW, uint8_t fE, uint8_t fN, uint8_t fS) { /* t1_fast5.go:171-181 */
    return (fW & T1Sig) | (((fW & T1SignNeg) >> 3) << 1) | ((fE & T1Sig) << 2) |
           (((fE & T1SignNeg) >> 3) << 3) | ((fN & T1Sig) << 4) | (((fN & T1SignNeg) >> 3) << 5) |
           ((fS & T1Sig) << 6) | (((fS & T1SignNeg) >> 3) << 7);
}

static inline void set_significant(t1_state *t, uint8_t *f, int x, int y) { /* t1_fast5.go:233-245 */
    *f |= T1Sig;
    if (y > 0) f[-t->stride] |= T1SigS;
    if (y < t->h - 1) f[t->stride] |= T1SigN;
    if (x > 0) f[-1] |= T1SigE;
    if (x < t->w - 1) f[1] |= T1SigW;
}

static void enc_sign(t1_state *t, mq_enc *e, uint8_t *f, uint8_t fW, uint8_t fE, uint8_t fN, uint8_t fS) {
    int sci = sc_index(fW, fE, fN, fS);
    int ctx = lutSignCtx[sci] + CtxSC0, pred = lutSignPred[sci];
    int sign = (*f & T1SignNeg) ? 1 : 0;
    (void)t;
    mq_encode(e, ctx, sign ^ pred);
}

long orc_t1_encode(const int32_t *src, int w, int h, int band,
                   uint8_t *out, size_t cap, int *numbps_out) {
    init_tables();
    if (numbps_out) *numbps_out = 0;
    if (w <= 0 || h <= 0) return 0;
    t1_state t;
    t.w = w; t.h = h; t.stride = w + 2; t.band = band;
    size_t n = (size_t)w * (size_t)h;
    t.data = (int32_t *)malloc(n * sizeof(int32_t));
    t.flags = (uint8_t *)calloc((size_t)(w + 2) * (size_t)(h + 2), 1);
    /* SetData (t1.go:292-304) */
    for (size_t i = 0; i < n; i++) {
        int32_t v = src[i];
        if (v < 0) {
            v = (int32_t)(0u - (uint32_t)v);
            t.flags[(i / (size_t)w + 1) * (size_t)t.stride + (i % (size_t)w + 1)] |= T1SignNeg;
        }
        t.data[i] = v;
    }
    /* EncodeFast5 (t1_fast5.go:13-28) */
    int32_t maxVal = 0;
    for (size_t i = 0; i < n; i++) if (t.data[i] > maxVal) maxVal = t.data[i];
    if (maxVal == 0) { free(t.data); free(t.flags); return 0; }
    int numBPS = 0;
    while (maxVal > 0) { numBPS++; maxVal >>= 1; }
    if (numbps_out) *numbps_out = numBPS;

    size_t est = n * 2 + 1024;                                       /* t1_fast5.go:47-50 */
    if (est < 16384) est = 16384;
    uint8_t *buf = (uint8_t *)malloc(est);
    mq_enc e; mq_enc_init(&e, buf, est);
    const int stride = t.stride, bandOff = band * 256;

    for (int bp = numBPS - 1; bp >= 0; bp--) {
        const int32_t bit = (int32_t)((uint32_t)1 << bp);
        /* ---- significance propagation, raster order (t1_fast5.go:72-249) ---- */
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                uint8_t *f = t.flags + (size_t)(y + 1) * stride + x + 1;
                uint8_t fv = *f;
                if (fv & T1Sig) continue;
                uint8_t fW = 0, fE = 0, fN = 0, fS = 0, fNW, fNE, fSW, fSE;
                if ((fv & (T1SigN | T1SigS | T1SigE | T1SigW)) == 0) {
                    fNW = f[-stride - 1]; fNE = f[-stride + 1]; fSW = f[stride - 1]; fSE = f[stride + 1];
                    if (((fNW | fNE | fSW | fSE) & T1Sig) == 0) continue;
                } else {
                    fW = f[-1]; fE = f[1]; fN = f[-stride]; fS = f[stride];
                    fNW = f[-stride - 1]; fNE = f[-stride + 1]; fSW = f[stride - 1]; fSE = f[stride + 1];
                }
                int sig = (int)((t.data[(size_t)y * w + x] >> bp) & 1);
                int packed = (fW & T1Sig) | ((fE & T1Sig) << 1) | ((fN & T1Sig) << 2) | ((fS & T1Sig) << 3) |
                             ((fNW & T1Sig) << 4) | ((fNE & T1Sig) << 5) | ((fSW & T1Sig) << 6) | ((fSE & T1Sig) << 7);
                mq_encode(&e, lutZCCtx[bandOff + packed], sig);
                if (sig) {
                    enc_sign(&t, &e, f, fW, fE, fN, fS);
                    set_significant(&t, f, x, y);
                }
                *f |= T1Visit;
            }
        }
        /* ---- magnitude refinement, raster order (t1_fast5.go:252-335) ---- */
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                uint8_t *f = t.flags + (size_t)(y + 1) * stride + x + 1;
                uint8_t fv = *f;
                if ((fv & T1Sig) == 0 || (fv & T1Visit) != 0) continue;
                int ref = (t.data[(size_t)y * w + x] & bit) ? 1 : 0;
                int ctx;
                if ((fv & T1Refine) == 0) {
                    uint8_t o = f[-1] | f[1] | f[-stride] | f[stride] | f[-stride - 1] | f[-stride + 1] |
                                f[stride - 1] | f[stride + 1];
                    ctx = (o & T1Sig) ? CtxMag1 : CtxMag0;
                } else {
                    ctx = CtxMag2;
                }
                mq_encode(&e, ctx, ref);
                *f |= T1Refine;
            }
        }
        /* ---- cleanup, 4-row stripes, column by column (t1_fast5.go:338-876) ---- */
        for (int y = 0; y < h; y += 4) {
            for (int x = 0; x < w; x++) {
                int canRL = (y + 4 <= h);
                if (canRL) {
                    for (int yy = 0; yy < 4; yy++) {
                        uint8_t *f = t.flags + (size_t)(y + yy + 1) * stride + x + 1;
                        if (*f & (T1Sig | T1Visit)) { canRL = 0; break; }
                        uint8_t o = f[-1] | f[1] | f[-stride] | f[stride] | f[-stride - 1] | f[-stride + 1] |
                                    f[stride - 1] | f[stride + 1];
                        if (o & T1Sig) { canRL = 0; break; }
                    }
                }
                if (canRL) {
                    int firstSig = -1;
                    for (int i = 0; i < 4; i++)
                        if (t.data[(size_t)(y + i) * w + x] & bit) { firstSig = i; break; }
                    mq_encode(&e, CtxRL, firstSig >= 0 ? 1 : 0);
                    if (firstSig < 0) continue;
                    mq_encode(&e, CtxUni, (firstSig >> 1) & 1);
                    mq_encode(&e, CtxUni, firstSig & 1);
                    {
                        int yy = y + firstSig;
                        uint8_t *f = t.flags + (size_t)(yy + 1) * stride + x + 1;
                        enc_sign(&t, &e, f, f[-1], f[1], f[-stride], f[stride]);
                        set_significant(&t, f, x, yy);
                    }
                    for (int i = firstSig + 1; i < 4; i++) {
                        int yy = y + i;
                        uint8_t *f = t.flags + (size_t)(yy + 1) * stride + x + 1;
                        int sig = (t.data[(size_t)yy * w + x] & bit) ? 1 : 0;
                        uint8_t fW = f[-1], fE = f[1], fN = f[-stride], fS = f[stride];
                        mq_encode(&e, lutZCCtx[bandOff + zc_packed(f, stride)], sig);
                        if (sig) {
                            enc_sign(&t, &e, f, fW, fE, fN, fS);
                            set_significant(&t, f, x, yy);
                        }
                    }
                    continue;
                }
                int yEnd = y + 4 > h ? h : y + 4;
                for (int yy = y; yy < yEnd; yy++) {
                    uint8_t *f = t.flags + (size_t)(yy + 1) * stride + x + 1;
                    uint8_t fv = *f;
                    if (fv & T1Visit) { *f &= (uint8_t)~T1Visit; continue; }
                    if (fv & T1Sig) continue;
                    int sig = (t.data[(size_t)yy * w + x] & bit) ? 1 : 0;
                    uint8_t fW = f[-1], fE = f[1], fN = f[-stride], fS = f[stride];
                    mq_encode(&e, lutZCCtx[bandOff + zc_packed(f, stride)], sig);
                    if (sig) {
                        enc_sign(&t, &e, f, fW, fE, fN, fS);
                        set_significant(&t, f, x, yy);
                    }
                }
            }
        }
    }
    size_t len = mq_flush(&e);                                       /* t1_fast5.go:878-898 */
    long r = (long)len;
    if (e.overflow || len > cap) r = -1; else memcpy(out, buf + 1, len);
    free(buf); free(t.data); free(t.flags);
    return r;
}

/* ---- T1.Decode (t1.go:1261-1410) ------------------------------------------ */
static inline int has_sig_neighbor(const uint8_t *f, int stride) {   /* t1.go:1087-1092 */
    return ((f[-1] | f[1] | f[-stride] | f[stride] | f[-stride - 1] | f[-stride + 1] | f[stride - 1] |
             f[stride + 1]) & T1Sig) != 0;
}

static void dec_sign(mq_dec *d, uint8_t *f, int stride) {            /* t1.go:1322-1328, 387-460 */
    int sci = sc_index(f[-1], f[1], f[-stride], f[stride]);
    int sign = mq_decode(d, lutSignCtx[sci] + CtxSC0) ^ lutSignPred[sci];
    if (sign) *f |= T1SignNeg;
}

static inline int mr_context(const uint8_t *f, int stride) {         /* t1.go:463-479 */
    if ((*f & T1Refine) == 0) return has_sig_neighbor(f, stride) ? CtxMag1 : CtxMag0;
    return CtxMag2;
}

void orc_t1_decode(const uint8_t *bytes, size_t nbytes, int numBPS, int band,
                   int w, int h, int32_t *out) {
    init_tables();
    if (w <= 0 || h <= 0) return;
    t1_state t;
    t.w = w; t.h = h; t.stride = w + 2; t.band = band;
    size_t n = (size_t)w * (size_t)h;
    t.data = (int32_t *)calloc(n, sizeof(int32_t));
    t.flags = (uint8_t *)calloc((size_t)(w + 2) * (size_t)(h + 2), 1);
    mq_dec d; mq_dec_init(&d, bytes, nbytes);
    const int stride = t.stride, bandOff = band * 256;

    for (int bp = numBPS - 1; bp >= 0; bp--) {
        const int32_t bit = (int32_t)shl32(1, (uint32_t)bp);         /* Go: int32(1)<<bp is 0 for bp>=32 */
        for (int y = 0; y < h; y++)                                  /* t1.go:1295-1319 */
            for (int x = 0; x < w; x++) {
                uint8_t *f = t.flags + (size_t)(y + 1) * stride + x + 1;
                if (*f & T1Sig) continue;
                if (!has_sig_neighbor(f, stride)) continue;
                int sig = mq_decode(&d, lutZCCtx[bandOff + zc_packed(f, stride)]);
                if (sig) {
                    t.data[(size_t)y * w + x] = bit;
                    dec_sign(&d, f, stride);
                    set_significant(&t, f, x, y);
                }
                *f |= T1Visit;
            }
        for (int y = 0; y < h; y++)                                  /* t1.go:1331-1347 */
            for (int x = 0; x < w; x++) {
                uint8_t *f = t.flags + (size_t)(y + 1) * stride + x + 1;
                if ((*f & T1Sig) == 0 || (*f & T1Visit) != 0) continue;
                if (mq_decode(&d, mr_context(f, stride))) t.data[(size_t)y * w + x] |= bit;
                *f |= T1Refine;
            }
        for (int y = 0; y < h; y += 4)                               /* t1.go:1350-1410 */
            for (int x = 0; x < w; x++) {
                int canRL = (y + 4 <= h);                            /* t1.go:1195-1208 */
                if (canRL)
                    for (int yy = y; yy < y + 4; yy++) {
                        uint8_t *f = t.flags + (size_t)(yy + 1) * stride + x + 1;
                        if ((*f & (T1Sig | T1Visit)) || has_sig_neighbor(f, stride)) { canRL = 0; break; }
                    }
                if (canRL) {                                         /* decodeRunLength */
                    if (mq_decode(&d, CtxRL) == 0) continue;
                    int pos = mq_decode(&d, CtxUni) << 1;
                    pos |= mq_decode(&d, CtxUni);
                    {
                        uint8_t *f = t.flags + (size_t)(y + pos + 1) * stride + x + 1;
                        t.data[(size_t)(y + pos) * w + x] = bit;
                        dec_sign(&d, f, stride);
                        set_significant(&t, f, x, y + pos);
                    }
                    for (int i = pos + 1; i < 4 && y + i < h; i++) {
                        uint8_t *f = t.flags + (size_t)(y + i + 1) * stride + x + 1;
                        if (mq_decode(&d, lutZCCtx[bandOff + zc_packed(f, stride)])) {
                            t.data[(size_t)(y + i) * w + x] = bit;
                            dec_sign(&d, f, stride);
                            set_significant(&t, f, x, y + i);
                        }
                    }
                    continue;
                }
                for (int yy = y; yy < y + 4 && yy < h; yy++) {
                    uint8_t *f = t.flags + (size_t)(yy + 1) * stride + x + 1;
                    if (*f & T1Visit) { *f &= (uint8_t)~T1Visit; continue; }
                    if (*f & T1Sig) continue;
                    if (mq_decode(&d, lutZCCtx[bandOff + zc_packed(f, stride)])) {
                        t.data[(size_t)yy * w + x] = bit;
                        dec_sign(&d, f, stride);
                        set_significant(&t, f, x, yy);
                    }
                }
            }
    }
    for (size_t i = 0; i < n; i++) {                                 /* t1.go:1281-1289 */
        int neg = t.flags[(i / (size_t)w + 1) * (size_t)stride + (i % (size_t)w + 1)] & T1SignNeg;
        out[i] = neg ? (int32_t)(0u - (uint32_t)t.data[i]) : t.data[i];
    }
    free(t.data); free(t.flags);
}

/* ======================================================================== */
/* internal/entropy/ht.go -- the reference's "HT" block coder, bug for bug   */
/* ======================================================================== */

static const uint16_t vlcTbl0[1024] = J2K_HT_VLC_TBL0_INIT;           /* ht_luts.go:18  */
static const uint16_t vlcTbl1[1024] = J2K_HT_VLC_TBL1_INIT;           /* ht_luts.go:152 */


size_t orc_ht_bound(int w, int h) {
    size_t maxSize = (size_t)w * (size_t)h * 2;
    if (maxSize < 64) maxSize = 64;
    return maxSize / 2 + maxSize / 4 + maxSize / 2 + 2;
}

typedef struct {
    uint8_t *data; long len; long pos; uint64_t tmp; long bits; uint8_t last; int fault;
} ht_writer;

static void vlc_write(ht_writer *v, uint32_t val, uint32_t nbits) {  /* ht.go:1266-1286 */
    v->tmp |= shl64((uint64_t)val, (uint64_t)v->bits);
    v->bits += (long)nbits;
    while (v->bits >= 8) {
        uint8_t b = (uint8_t)(v->tmp & 0xFF);
        if (v->last > 0x8F && (b & 0x7F) == 0x7F) b &= 0x7F;
        if (v->pos < 0) { v->fault = 1; return; }
        v->data[v->pos] = b; v->pos--; v->last = b;
        v->tmp >>= 8; v->bits -= 8;
    }
}

static void vlc_flush(ht_writer *v) {                                /* ht.go:1289-1300 */
    while (v->bits > 0) {
        uint8_t b = (uint8_t)(v->tmp & 0xFF);
        if (v->pos < 0) { v->fault = 1; return; }
        v->data[v->pos] = b; v->pos--;
        v->tmp >>= 8; v->bits -= 8;
        if (v->bits < 0) v->bits = 0;
    }
}

static void magsgn_write(ht_writer *m, uint32_t val, uint32_t nbits) { /* ht.go:1303-1327 */
    m->tmp |= shl64((uint64_t)val, (uint64_t)m->bits);
    m->bits += (long)nbits;
    while (m->bits >= 8) {
        uint8_t b = (uint8_t)(m->tmp & 0xFF);
        if (m->pos >= m->len) { m->fault = 1; return; }
        if (m->last == 0xFF) {
            b &= 0x7F;
            m->data[m->pos] = b; m->pos++;
            m->tmp >>= 7; m->bits -= 7;
        } else {
            m->data[m->pos] = b; m->pos++;
            m->tmp >>= 8; m->bits -= 8;
        }
        m->last = b;
    }
}

static void magsgn_flush(ht_writer *m) {                             /* ht.go:1330-1341 */
    while (m->bits > 0) {
        uint8_t b = (uint8_t)(m->tmp & 0xFF);
        if (m->pos >= m->len) { m->fault = 1; return; }
        m->data[m->pos] = b; m->pos++;
        m->tmp >>= 8; m->bits -= 8;
        if (m->bits < 0) m->bits = 0;
    }
}

static void encode_vlc_quad(ht_writer *v, uint8_t context, uint8_t rho, int initial) { /* ht.go:1199-1226 */
    const uint16_t *tbl = initial ? vlcTbl0 : vlcTbl1;
    for (uint32_t cwd = 0; cwd < 128; cwd++) {
        uint16_t entry = tbl[((uint32_t)context << 7) | cwd];
        uint16_t elen = entry & 0x0F, erho = (entry >> 4) & 0x0F;
        if ((uint8_t)erho == rho && elen > 0) { vlc_write(v, cwd, elen); return; }
    }
    vlc_write(v, 0, 1);
}

static void encode_uvlc_one(ht_writer *v, uint32_t u) {              /* ht.go:1242-1249 */
    if (u <= 1) vlc_write(v, 1, 1);
    else if (u <= 2) vlc_write(v, 2, 2);
    else { vlc_write(v, 0, 3); vlc_write(v, u - 3, 5); }
}

long orc_ht_encode(const int32_t *data, int w, int h, int band, uint8_t *out, size_t cap) {
    (void)band;
    if (w <= 0 || h <= 0) return 0;
    size_t n = (size_t)w * (size_t)h;
    int32_t maxMag = 0;                                              /* ht.go:947-960 */
    for (size_t i = 0; i < n; i++) {
        int32_t v = data[i];
        if (v < 0) v = (int32_t)(0u - (uint32_t)v);                  /* -MinInt32 wraps to MinInt32 (< 0) */
        if (v > maxMag) maxMag = v;
    }
    if (maxMag == 0) return 0;
    size_t maxSize = n * 2;
    if (maxSize < 64) maxSize = 64;
    size_t melLen = maxSize / 4;                                     /* ht.go:978, 1019: never fed -> zeros */
    ht_writer vlc, ms;
    memset(&vlc, 0, sizeof(vlc)); memset(&ms, 0, sizeof(ms));
    vlc.len = (long)(maxSize / 2); vlc.data = (uint8_t *)calloc((size_t)vlc.len, 1); vlc.pos = vlc.len - 1;
    ms.len = (long)(maxSize / 2); ms.data = (uint8_t *)calloc((size_t)ms.len, 1); ms.pos = 0;
    int quadCols = (w + 3) / 4;
    uint8_t *sigma1 = (uint8_t *)calloc((size_t)quadCols + 2, 1);

    for (int y = 0; y < h && !vlc.fault && !ms.fault; y += 4) {      /* ht.go:1054-1195: only row y of each stripe */
        int initial = (y == 0);
        for (int qx = 0; qx < quadCols && !vlc.fault && !ms.fault; qx += 2) {
            uint8_t rho = 0, rho2 = 0;
            for (int i = 0; i < 4 && qx * 4 + i < w; i++)
                if (data[(size_t)y * w + qx * 4 + i] != 0) rho |= (uint8_t)(1 << i);
            for (int i = 0; i < 4 && (qx + 1) * 4 + i < w; i++)
                if (data[(size_t)y * w + (qx + 1) * 4 + i] != 0) rho2 |= (uint8_t)(1 << i);
            uint8_t context = 0;
            if (initial) { if (qx > 0) context = sigma1[qx - 1] >> 4; }
            else context = sigma1[qx] >> 4;
            encode_vlc_quad(&vlc, context, rho, initial);
            sigma1[qx] = rho;
            uint8_t context2 = (uint8_t)((rho >> 2) | (sigma1[qx + 1] >> 4));
            encode_vlc_quad(&vlc, context2, rho2, initial);
            sigma1[qx + 1] = rho2;
            int uOff1 = rho != 0, uOff2 = rho2 != 0;
            if (uOff1 || uOff2) {                                    /* ht.go:1105-1142 */
                uint32_t u1 = 1, u2 = 1;
                for (int i = 0; i < 4 && qx * 4 + i < w; i++) {
                    int32_t v = data[(size_t)y * w + qx * 4 + i];
                    if (v < 0) v = (int32_t)(0u - (uint32_t)v);
                    if ((uint32_t)v >= shl32(1, u1)) u1++;
                }
                for (int i = 0; i < 4 && (qx + 1) * 4 + i < w; i++) {
                    int32_t v = data[(size_t)y * w + (qx + 1) * 4 + i];
                    if (v < 0) v = (int32_t)(0u - (uint32_t)v);
                    if ((uint32_t)v >= shl32(1, u2)) u2++;
                }
                uint32_t mode = (uOff1 ? 1u : 0u) | (uOff2 ? 2u : 0u);
                if (mode == 1) encode_uvlc_one(&vlc, u1);            /* ht.go:1229-1263 */
                else if (mode == 2) encode_uvlc_one(&vlc, u2);
                else { encode_uvlc_one(&vlc, u1); encode_uvlc_one(&vlc, u2); }
            }
            for (int q = 0; q < 2; q++) {                            /* ht.go:1145-1193 */
                uint8_t r = q ? rho2 : rho;
                int base = (qx + q) * 4;
                for (int i = 0; i < 4 && base + i < w; i++) {
                    if (!(r & (1 << i))) continue;
                    int32_t v = data[(size_t)y * w + base + i];
                    uint32_t sign = 0;
                    if (v < 0) { sign = 1; v = (int32_t)(0u - (uint32_t)v); }
                    uint32_t mag = (uint32_t)v, emb = 1;
                    if (mag >= 0x80000000u) { ms.fault = 1; break; }  /* ht.go:1159: Go loop never terminates */
                    while (mag >= shl32(1, emb)) emb++;
                    magsgn_write(&ms, mag & (shl32(1, emb - 1) - 1), emb - 1);
                    magsgn_write(&ms, sign, 1);
                }
            }
        }
    }
    if (!vlc.fault) vlc_flush(&vlc);                                 /* ht.go:1008-1010 (melFlush is a no-op) */
    if (!ms.fault) magsgn_flush(&ms);
    long r;
    if (vlc.fault || ms.fault) {
        r = -2;
    } else {
        size_t magLen = (size_t)ms.pos;                              /* ht.go:1018-1042 */
        size_t vlcLen = (size_t)(vlc.len - vlc.pos - 1);
        size_t scup = melLen + vlcLen + 2;
        size_t total = magLen + scup;
        if (total > cap) r = -1;
        else {
            memcpy(out, ms.data, magLen);
            memset(out + magLen, 0, melLen);
            for (size_t i = 0; i < vlcLen; i++) out[magLen + melLen + i] = vlc.data[(size_t)vlc.len - 1 - i];
            out[total - 2] = (uint8_t)(scup >> 8);
            out[total - 1] = (uint8_t)(scup & 0xFF);
            r = (long)total;
        }
    }
    free(vlc.data); free(ms.data); free(sigma1);
    return r;
}

/* ---- HTDecoder (ht.go:93-150, 153-195, 276-519, 583-864) -------------------- */
typedef struct { const uint8_t *data; long len; long pos; uint64_t tmp; uint32_t bits; long size; int unstuff; } ht_rev;
typedef struct { const uint8_t *data; long len; long pos; uint64_t tmp; uint32_t bits; int unstuff; long size; uint32_t x; } ht_fwd;

static int ht_init_mel(const uint8_t *data, long len, long lcup, long scup) { /* ht.go:153-195 */
    long pos = lcup - scup, size = scup - 1, bits = 0;
    uint64_t tmp = 0; int unstuff = 0;
    long num = 4 - (pos & 3);
    if (num > 4) num = 4;
    for (long i = 0; i < num && size > 0; i++) {
        if (unstuff && pos < len && data[pos] > 0x8F) return 0;
        uint8_t b;
        if (size > 0 && pos < len) { b = data[pos]; pos++; size--; } else b = 0xFF;
        if (size == 1) b |= 0x0F;
        long dBits = unstuff ? 7 : 8;
        tmp = (tmp << dBits) | (uint64_t)b;
        bits += dBits;
        unstuff = (b == 0xFF);
    }
    tmp = shl64(tmp, (uint64_t)(64 - bits));
    (void)tmp;
    return 1;
}

static void ht_rev_read(ht_rev *v) {                                 /* ht.go:317-378 */
    if (v->bits > 32) return;
    uint32_t val = 0;
    if (v->size > 3) {
        long p = v->pos - 3;
        if (p >= 0 && p + 3 < v->len)
            val = (uint32_t)v->data[p] | (uint32_t)v->data[p + 1] << 8 | (uint32_t)v->data[p + 2] << 16 |
                  (uint32_t)v->data[p + 3] << 24;
        v->pos -= 4; v->size -= 4;
    } else if (v->size > 0) {
        int i = 24;
        while (v->size > 0) {
            if (v->pos >= 0 && v->pos < v->len) { val |= (uint32_t)v->data[v->pos] << i; v->pos--; }
            v->size--; i -= 8;
        }
    }
    uint32_t tmp = val >> 24, bits = 8;
    if (v->unstuff && ((val >> 24) & 0x7F) == 0x7F) bits = 7;
    int unstuff = (val >> 24) > 0x8F;
    tmp |= ((val >> 16) & 0xFF) << bits;
    bits += (unstuff && ((val >> 16) & 0x7F) == 0x7F) ? 7 : 8;
    unstuff = ((val >> 16) & 0xFF) > 0x8F;
    tmp |= ((val >> 8) & 0xFF) << bits;
    bits += (unstuff && ((val >> 8) & 0x7F) == 0x7F) ? 7 : 8;
    unstuff = ((val >> 8) & 0xFF) > 0x8F;
    tmp |= (val & 0xFF) << bits;
    bits += (unstuff && (val & 0x7F) == 0x7F) ? 7 : 8;
    v->unstuff = (val & 0xFF) > 0x8F;
    v->tmp |= shl64((uint64_t)tmp, v->bits);
    v->bits += bits;
}

static uint32_t ht_rev_fetch(ht_rev *v) {                            /* ht.go:381-389 */
    if (v->bits < 32) { ht_rev_read(v); if (v->bits < 32) ht_rev_read(v); }
    return (uint32_t)v->tmp;
}
static void ht_rev_advance(ht_rev *v, uint32_t n) {                  /* ht.go:392-396 */
    v->tmp = shr64(v->tmp, n); v->bits -= n;
}

static void ht_init_vlc(ht_rev *v, const uint8_t *data, long len, long lcup, long scup) { /* ht.go:276-314 */
    v->data = data; v->len = len; v->pos = lcup - 2; v->size = scup - 2; v->tmp = 0; v->bits = 0; v->unstuff = 0;
    if (v->pos >= 0 && v->pos < len) {
        uint8_t b = data[v->pos];
        v->pos--;
        v->tmp = (uint64_t)(b >> 4);
        v->bits = 4 - (uint32_t)((v->tmp & 7) >> 2);
        v->unstuff = (b | 0x0F) > 0x8F;
    }
    long num = 1 + (v->pos & 3);
    if (num > v->size) num = v->size;
    for (long i = 0; i < num; i++) {
        uint8_t b = 0;
        if (v->pos >= 0 && v->pos < len) { b = data[v->pos]; v->pos--; }
        uint32_t dBits = (v->unstuff && (b & 0x7F) == 0x7F) ? 7 : 8;
        v->tmp |= shl64((uint64_t)b, v->bits);
        v->bits += dBits;
        v->unstuff = b > 0x8F;
    }
    v->size -= num;
    ht_rev_read(v);
}

static void ht_fwd_read(ht_fwd *f) {                                 /* ht.go:432-501 */
    if (f->bits > 32) return;
    uint32_t val = 0;
    if (f->size > 3) {
        if (f->pos + 3 < f->len)
            val = (uint32_t)f->data[f->pos] | (uint32_t)f->data[f->pos + 1] << 8 |
                  (uint32_t)f->data[f->pos + 2] << 16 | (uint32_t)f->data[f->pos + 3] << 24;
        f->pos += 4; f->size -= 4;
    } else if (f->size > 0) {
        if (f->x != 0) val = 0xFFFFFFFFu;
        int i = 0;
        while (f->size > 0) {
            if (f->pos < f->len) {
                uint32_t v = f->data[f->pos];
                uint32_t m = ~((uint32_t)0xFF << i);
                val = (val & m) | (v << i);
                f->pos++;
            }
            f->size--; i += 8;
        }
    } else {
        if (f->x != 0) val = 0xFFFFFFFFu;
    }
    uint32_t bits = f->unstuff ? 7 : 8;
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
    f->unstuff = ((val >> 24) & 0xFF) == 0xFF;
    f->tmp |= shl64((uint64_t)t, f->bits);
    f->bits += bits;
}

static uint32_t ht_fwd_fetch(ht_fwd *f) {                            /* ht.go:504-512 */
    if (f->bits < 32) { ht_fwd_read(f); if (f->bits < 32) ht_fwd_read(f); }
    return (uint32_t)f->tmp;
}
static void ht_fwd_advance(ht_fwd *f, uint32_t n) {                  /* ht.go:515-519 */
    f->tmp = shr64(f->tmp, n); f->bits -= n;
}

static void ht_init_magsgn(ht_fwd *f, const uint8_t *data, long len, long size) { /* ht.go:399-429 */
    f->data = data; f->len = len; f->pos = 0; f->size = size; f->tmp = 0; f->bits = 0; f->unstuff = 0; f->x = 0xFF;
    long num = 4 - (f->pos & 3);
    for (long i = 0; i < num; i++) {
        uint8_t b;
        if (f->size > 0 && f->pos < len) { b = data[f->pos]; f->pos++; f->size--; } else b = (uint8_t)f->x;
        uint32_t dBits = f->unstuff ? 7 : 8;
        f->tmp |= shl64((uint64_t)b, f->bits);
        f->bits += dBits;
        f->unstuff = (b == 0xFF);
    }
    ht_fwd_read(f);
}

static const uint8_t uvlc_dec[8] = {                                 /* ht.go:718-727 == 809-818 */
    3 | (5 << 2) | (5 << 5), 1 | (0 << 2) | (1 << 5), 2 | (0 << 2) | (2 << 5), 1 | (0 << 2) | (1 << 5),
    3 | (1 << 2) | (3 << 5), 1 | (0 << 2) | (1 << 5), 2 | (0 << 2) | (2 << 5), 1 | (0 << 2) | (1 << 5)};

static uint32_t decode_uvlc(uint32_t vlc, uint32_t mode, uint32_t u[2], int initial) { /* ht.go:716-864 */
    uint32_t consumed = 0;
    if (mode == 0) { u[0] = 1; u[1] = 1; }
    else if (mode <= 2) {
        uint8_t t = uvlc_dec[vlc & 7];
        uint32_t pl = t & 3; vlc >>= pl; consumed += pl;
        uint32_t sl = (t >> 2) & 7; consumed += sl;
        uint32_t val = (uint32_t)(t >> 5) + (vlc & (shl32(1, sl) - 1));
        if (mode == 1) { u[0] = val + 1; u[1] = 1; } else { u[0] = 1; u[1] = val + 1; }
    } else if (mode == 3) {
        uint8_t t1 = uvlc_dec[vlc & 7];
        uint32_t pl1 = t1 & 3; vlc >>= pl1; consumed += pl1;
        if (initial && pl1 > 2) {                                    /* ht.go:756-764 (initial rows only) */
            u[1] = (vlc & 1) + 2; consumed++; vlc >>= 1;
            uint32_t sl = (t1 >> 2) & 7; consumed += sl;
            u[0] = (uint32_t)(t1 >> 5) + (vlc & (shl32(1, sl) - 1)) + 1;
        } else {
            uint8_t t2 = uvlc_dec[vlc & 7];
            uint32_t pl2 = t2 & 3; vlc >>= pl2; consumed += pl2;
            uint32_t sl1 = (t1 >> 2) & 7; consumed += sl1;
            u[0] = (uint32_t)(t1 >> 5) + (vlc & (shl32(1, sl1) - 1)) + 1;
            vlc >>= sl1;
            uint32_t sl2 = (t2 >> 2) & 7; consumed += sl2;
            u[1] = (uint32_t)(t2 >> 5) + (vlc & (shl32(1, sl2) - 1)) + 1;
        }
    }
    return consumed;
}

int orc_ht_decode(const uint8_t *bytes, size_t nbytes, int num_bitplanes, int band,
                  int w, int h, int32_t *out) {
    (void)num_bitplanes; (void)band;
    if (w <= 0 || h <= 0) return 0;
    size_t n = (size_t)w * (size_t)h;
    memset(out, 0, n * sizeof(int32_t));
    long len = (long)nbytes;
    if (len < 2) return 0;                                           /* ht.go:94-100 */
    long scup = (long)bytes[len - 1] + ((long)(bytes[len - 2] & 0x0F) << 8);
    if (scup < 2 || scup > len) return 0;                            /* ht.go:104-111 */
    long lcup = len;
    if (!ht_init_mel(bytes, len, lcup, scup)) return 0;              /* ht.go:117-122 */
    ht_rev vlc; ht_fwd ms;
    ht_init_vlc(&vlc, bytes, len, lcup, scup);
    ht_init_magsgn(&ms, bytes, len, lcup - scup);
    int quadCols = (w + 3) / 4;
    uint8_t *sigma1 = (uint8_t *)calloc((size_t)quadCols + 2, 1);
    uint8_t *lineState = (uint8_t *)calloc((size_t)quadCols + 2, 1);

    for (int y = 0; y < h; y += 4) {                                 /* ht.go:589-711 */
        int initial = (y == 0);
        const uint16_t *tbl = initial ? vlcTbl0 : vlcTbl1;
        for (int qx = 0; qx < quadCols; qx += 2) {
            uint32_t vlcVal = ht_rev_fetch(&vlc);
            uint8_t context = 0;
            if (initial) { if (qx > 0) context = sigma1[qx - 1] >> 4; }
            else context = (uint8_t)((sigma1[qx] >> 4) | (lineState[qx] >> 4));
            uint16_t qinf = tbl[((uint32_t)context << 7) | (vlcVal & 0x7F)];
            uint16_t vlcLen = qinf & 0x0F, rho = (qinf >> 4) & 0x0F, uOff1 = (qinf >> 3) & 1;
            ht_rev_advance(&vlc, vlcLen);
            vlcVal = ht_rev_fetch(&vlc);
            uint8_t context2 = (uint8_t)((uint8_t)(rho >> 2) | (sigma1[qx + 1] >> 4));
            uint16_t qinf2 = tbl[((uint32_t)context2 << 7) | (vlcVal & 0x7F)];
            uint16_t vlcLen2 = qinf2 & 0x0F, rho2 = (qinf2 >> 4) & 0x0F, uOff2 = (qinf2 >> 3) & 1;
            ht_rev_advance(&vlc, vlcLen2);
            sigma1[qx] = (uint8_t)rho; sigma1[qx + 1] = (uint8_t)rho2;
            uint32_t u[2];
            uint32_t mode = (uint32_t)((uOff1 << 1) | uOff2);
            if (mode > 0) {
                vlcVal = ht_rev_fetch(&vlc);
                uint32_t consumed = decode_uvlc(vlcVal, mode, u, initial);
                ht_rev_advance(&vlc, consumed);
            } else { u[0] = 1; u[1] = 1; }
            for (int q = 0; q < 2; q++) {                            /* ht.go:661-710 */
                uint16_t r = q ? rho2 : rho;
                uint32_t emb = u[q];
                int base = (qx + q) * 4;
                for (int i = 0; i < 4 && base + i < w; i++) {
                    if (!(r & (1 << i))) continue;
                    uint32_t magVal = ht_fwd_fetch(&ms);
                    uint32_t mag = (magVal & (shl32(1, emb) - 1)) + shl32(1, emb - 1);
                    ht_fwd_advance(&ms, emb);
                    uint32_t sign = ht_fwd_fetch(&ms) & 1;
                    ht_fwd_advance(&ms, 1);
                    size_t idx = (size_t)y * w + base + i;
                    if (idx < n) out[idx] = sign ? (int32_t)(0u - mag) : (int32_t)mag;
                }
            }
        }
    }
    free(sigma1); free(lineState);
    return 0;
}

/* ======================================================================== */
/* encoder.encodeTile job list + sequential body                             */
/* ======================================================================== */

/* The job list in two window modes.  windows == 0: the reference's (encoder.go:597-673: every band's blocks are cut from the
 * TOP LEFT of the plane, so windows overlap and most of the plane is never coded).  windows == 1: this library's closed-loop
 * mode (NOT the reference): the same comp -> res -> band -> cby -> cbx order, but band b of resolution r is the Mallat
 * rectangle of decomposition level l = numRes-1-r of the plane (w_0 = w, w_{l+1} = ceil(w_l / 2); LL = [0,w_L) x [0,h_L);
 * HL = [w_{l+1},w_l) x [0,h_{l+1}); LH = [0,w_{l+1}) x [h_{l+1},h_l); HH = the rest), cut into cb_w x cb_h blocks from the
 * band's own origin -- the rectangles partition the plane, so that a decoder can put every sample back. */
size_t orc_enumerate_blocks2(int ncomp, int w, int h, int num_resolutions,
                             int cb_w, int cb_h, int windows, orc_block *out, size_t cap) { /* encoder.go:597-673 */
    int numRes = num_resolutions;
    if (numRes <= 0) numRes = 6;
    if (cb_w <= 0) cb_w = 64;
    if (cb_h <= 0) cb_h = 64;
    size_t k = 0;
    for (int c = 0; c < ncomp; c++)
        for (int r = 0; r < numRes; r++) {
            int numBands = r == 0 ? 1 : 3;
            for (int b = 0; b < numBands; b++) {
                int band = r == 0 ? BandLL : (b == 0 ? BandHL : (b == 1 ? BandLH : BandHH));
                int bx0 = 0, by0 = 0, bw, bh;
                if (!windows) {
                    long scale = 1L << (numRes - 1 - r);
                    bw = (int)(((long)w + scale - 1) / scale); bh = (int)(((long)h + scale - 1) / scale);
                    if (r > 0) { bw = (bw + 1) / 2; bh = (bh + 1) / 2; }
                } else {
                    int lv = r == 0 ? numRes - 1 : numRes - 1 - r;   /* dims of level lv, and of the next one */
                    int wl = w, hl = h;
                    for (int i = 0; i < lv; i++) { wl = (wl + 1) / 2; hl = (hl + 1) / 2; }
                    int wn = (wl + 1) / 2, hn = (hl + 1) / 2;
                    if (r == 0) { bw = wl; bh = hl; }
                    else if (band == BandHL) { bx0 = wn; bw = wl - wn; bh = hn; }
                    else if (band == BandLH) { by0 = hn; bw = wn; bh = hl - hn; }
                    else { bx0 = wn; by0 = hn; bw = wl - wn; bh = hl - hn; }
                }
                for (int cby = 0; cby * cb_h < bh; cby++)
                    for (int cbx = 0; cbx * cb_w < bw; cbx++) {
                        int aw = cb_w, ah = cb_h, sx = cbx * cb_w, sy = cby * cb_h;
                        if (sx + aw > bw) aw = bw - sx;
                        if (sy + ah > bh) ah = bh - sy;
                        if (k < cap) {
                            out[k].comp = c; out[k].res = r; out[k].band = band;
                            out[k].x0 = bx0 + sx; out[k].y0 = by0 + sy; out[k].w = aw; out[k].h = ah;
                        }
                        k++;
                    }
            }
        }
    return k;
}
size_t orc_enumerate_blocks(int ncomp, int w, int h, int num_resolutions,
                            int cb_w, int cb_h, orc_block *out, size_t cap) {
    return orc_enumerate_blocks2(ncomp, w, h, num_resolutions, cb_w, cb_h, 0, out, cap);
}

void orc_extract_block(const int32_t *plane, int plane_w, int plane_h,
                       const orc_block *b, int32_t *dst) {           /* encoder.go:763-795 */
    for (int y = 0; y < b->h; y++)
        for (int x = 0; x < b->w; x++) {
            int sx = b->x0 + x, sy = b->y0 + y;
            dst[(size_t)y * b->w + x] = (sx < plane_w && sy < plane_h) ? plane[(size_t)sy * plane_w + sx] : 0;
        }
}

long orc_encode_tile_blocks2(int32_t *const *planes, int ncomp, int w, int h,
                             int num_resolutions, int cb_w, int cb_h, int coder, int windows,
                             uint8_t *out, size_t cap, uint32_t *lens, uint8_t *numbps) { /* encoder.go:677-688 */
    size_t nj = orc_enumerate_blocks2(ncomp, w, h, num_resolutions, cb_w, cb_h, windows, NULL, 0);
    orc_block *jobs = (orc_block *)malloc((nj ? nj : 1) * sizeof(orc_block));
    orc_enumerate_blocks2(ncomp, w, h, num_resolutions, cb_w, cb_h, windows, jobs, nj);
    size_t total = 0;
    long status = 0;
    for (size_t j = 0; j < nj && status >= 0; j++) {
        size_t bn = (size_t)jobs[j].w * (size_t)jobs[j].h;
        int32_t *blk = (int32_t *)malloc((bn ? bn : 1) * sizeof(int32_t));
        orc_extract_block(planes[jobs[j].comp], w, h, &jobs[j], blk);
        int nb = 0;
        long r;
        if (coder == 0) {
            r = orc_t1_encode(blk, jobs[j].w, jobs[j].h, jobs[j].band, out + total, cap - total, &nb);
        } else {
            r = orc_ht_encode(blk, jobs[j].w, jobs[j].h, jobs[j].band, out + total, cap - total);
            if (r > 0) {                                             /* bit-length of max |x| (ht.go:962-966) */
                uint32_t m = 0;
                for (size_t i = 0; i < bn; i++) {
                    uint32_t a = blk[i] < 0 ? 0u - (uint32_t)blk[i] : (uint32_t)blk[i];
                    if (a > m) m = a;
                }
                while (m) { nb++; m >>= 1; }
            }
        }
        free(blk);
        if (r < 0) { status = r; break; }
        if (lens) lens[j] = (uint32_t)r;
        if (numbps) numbps[j] = (uint8_t)nb;
        total += (size_t)r;
    }
    free(jobs);
    return status < 0 ? status : (long)total;
}
long orc_encode_tile_blocks(int32_t *const *planes, int ncomp, int w, int h,
                            int num_resolutions, int cb_w, int cb_h, int coder,
                            uint8_t *out, size_t cap, uint32_t *lens, uint8_t *numbps) {
    return orc_encode_tile_blocks2(planes, ncomp, w, h, num_resolutions, cb_w, cb_h, coder, 0, out, cap, lens, numbps);
}

/* The decode body the reference leaves as a placeholder (decoder.go:375-411), composed from the functions it does have:
 * TileDecoder.DecodeCodeBlock (tcd.go:393-413: NewT1(w, h).Decode(data, numBPS, band) / a fresh HT decoder; a block without
 * data is skipped and its samples stay 0) for every job of the list, each decoded block put at its window of the (zeroed)
 * component plane.  With windows == 1 every sample is written exactly once; with the reference's overlapping windows a later
 * job overwrites an earlier one.  bytes = the jobs' data end to end in job order. */
int orc_decode_tile_blocks(const uint8_t *bytes, const uint32_t *lens, const uint8_t *numbps, int ncomp, int w, int h,
                           int num_resolutions, int cb_w, int cb_h, int coder, int windows, int32_t *const *planes) {
    size_t nj = orc_enumerate_blocks2(ncomp, w, h, num_resolutions, cb_w, cb_h, windows, NULL, 0);
    orc_block *jobs = (orc_block *)malloc((nj ? nj : 1) * sizeof(orc_block));
    orc_enumerate_blocks2(ncomp, w, h, num_resolutions, cb_w, cb_h, windows, jobs, nj);
    for (int c = 0; c < ncomp; c++) memset(planes[c], 0, (size_t)w * h * sizeof(int32_t));
    size_t pos = 0;
    int status = 0;
    for (size_t j = 0; j < nj; j++) {
        const orc_block *b = &jobs[j];
        size_t bn = (size_t)b->w * (size_t)b->h;
        if (lens[j] == 0) continue;                                  /* tcd.go:394-396 */
        int32_t *blk = (int32_t *)calloc(bn ? bn : 1, sizeof(int32_t));
        if (coder == 0) orc_t1_decode(bytes + pos, lens[j], numbps[j], b->band, b->w, b->h, blk);
        else if (orc_ht_decode(bytes + pos, lens[j], numbps[j], b->band, b->w, b->h, blk) != 0) status = -2;
        pos += lens[j];
        for (int y = 0; y < b->h; y++)
            for (int x = 0; x < b->w; x++) {
                int sx = b->x0 + x, sy = b->y0 + y;
                if (sx < w && sy < h) planes[b->comp][(size_t)sy * w + sx] = blk[(size_t)y * b->w + x];
            }
        free(blk);
    }
    free(jobs);
    return status;
}

/* ======================================================================== */
/* Pins: the oracle's internals exposed at the granularity the reference's   */
/* OWN unit tests probe (internal/entropy/coverage_test.go, t1_test.go), so  */
/* that every numeric vector those tests hold is reproduced by               */
/* tests/test_oracle_reference_pins.py.  Test infrastructure only.           */
/* ======================================================================== */

/* mqByteOutLocal(buf, bp, c) -> (bp, c, ct)   t1_fast.go:11-34
 * == mqByteOutCommon / mqByteOutRare          mq_inline.go:103-134 */
void orc_pin_mq_byte_out(uint8_t *buf, size_t buflen, long bp, uint32_t c,
                         long *bp_out, uint32_t *c_out, uint32_t *ct_out) {
    mq_enc e;
    e.A = 0x8000; e.C = c; e.CT = 0; e.buf = buf; e.cap = buflen + 1; e.bp = (size_t)bp; e.overflow = 0;
    mq_byte_out(&e);
    *bp_out = (long)e.bp; *c_out = e.C; *ct_out = e.CT;
}
/* mqNeedsSlowPath   mq_inline.go:96-98 */
int orc_pin_mq_needs_slow_path(uint8_t buf_byte, uint32_t c) { return buf_byte == 0xFF || (c & 0x8000000u) != 0; }

/* MQDecoder.byteIn on an explicit state   mqc.go:402-439 (endCounter counts the 0xFF00 feeds) */
void orc_pin_mq_byte_in(const uint8_t *data, long len, long *bp, uint32_t *C, uint32_t *CT, int *end_counter) {
    mq_dec d; memset(&d, 0, sizeof d);
    d.data = data; d.len = len; d.bp = *bp; d.C = *C; d.CT = *CT; d.A = 0x8000;
    const uint32_t c0 = d.C; const long bp0 = d.bp < 0 ? 0 : d.bp;
    mq_byte_in(&d);
    /* the two branches that feed 0xFF00 are the ones that bump endCounter (mqc.go:410-412, 425-428) */
    if (d.bp == bp0 && d.C == c0 + 0xFF00u && d.CT == 8) (*end_counter)++;
    *bp = d.bp; *C = d.C; *CT = d.CT;
}

/* getSignContextParams   mq_inline.go:26-65.  NOT on the hot path (lutSC is never read by the coders); restated
 * only because coverage_test.go:961-1008 pins its values -- including its 10/11/12/14 constants, which are not
 * CtxSC0..4 (= 9..13). */
void orc_pin_sign_context_params(int hc, int vc, int *ctx, int *xorbit) {
    int x = 0;
    if (hc < 0) { x = 1; hc = -hc; }
    if (hc == 0 && vc < 0) { x = 1; vc = -vc; }
    if (hc > 1) hc = 1;
    if (vc < 0) vc = -vc;
    if (vc > 1) vc = 1;
    if (hc == 1) *ctx = vc == 1 ? 14 : 12;
    else if (hc == 0) *ctx = vc == 0 ? 10 : 11;
    else *ctx = 10;
    *xorbit = x;
}
/* getSignContrib / clampContrib   mq_inline.go:70-91 */
int orc_pin_sign_contrib(int flag) { return (flag & T1Sig) == 0 ? 0 : ((flag & T1SignNeg) ? -1 : 1); }
int orc_pin_clamp_contrib(int c) { return c < -2 ? -2 : (c > 2 ? 2 : c); }
/* lutSCCtx[(hc+2)*5 + (vc+2)] = (ctx << 1) | pred   t1_luts.go:112-150 (getSCContextFast :240-258 clamps first) */
int orc_pin_lut_sc_ctx(int hc, int vc) {
    hc = orc_pin_clamp_contrib(hc); vc = orc_pin_clamp_contrib(vc);
    int h = hc, v = vc, pred = 0, ctx = 0;
    if (h < 0) { pred = 1; h = -h; }
    if (h == 0 && v < 0) { pred = 1; v = -v; }
    if (h == 1) ctx = v == 1 ? CtxSC0 + 4 : (v == 0 ? CtxSC0 + 2 : CtxSC0 + 1);
    else if (h == 0) ctx = v == 1 ? CtxSC0 + 1 : (v == 0 ? CtxSC0 : 0);
    else if (h == 2) ctx = CtxSC0 + 3;
    return (ctx << 1) | pred;
}

/* Flag-array helpers on a caller-owned (w+2)*(h+2) array, as NewT1(w,h).flags (t1.go:94-122, flagIndex :307-309). */
static uint8_t *pin_at(uint8_t *flags, int w, int x, int y) { return flags + (size_t)(y + 1) * (size_t)(w + 2) + x + 1; }
/* updateNeighborFlags   t1.go:328-345 == setSignificant's neighbour part, t1_fast5.go:233-245 */
void orc_pin_update_neighbor_flags(uint8_t *flags, int w, int h, int x, int y) {
    t1_state t; t.w = w; t.h = h; t.stride = w + 2; t.band = 0; t.data = NULL; t.flags = flags;
    uint8_t *f = pin_at(flags, w, x, y);
    const uint8_t keep = *f;
    set_significant(&t, f, x, y);
    *f = keep;                        /* updateNeighborFlags does not touch the sample's own T1Sig */
}
int orc_pin_has_sig_neighbor(uint8_t *flags, int w, int h, int x, int y) {          /* t1.go:1087-1092 */
    (void)h; return has_sig_neighbor(pin_at(flags, w, x, y), w + 2);
}
int orc_pin_mr_context(uint8_t *flags, int w, int h, int x, int y) {                /* t1.go:463-479 */
    (void)h; return mr_context(pin_at(flags, w, x, y), w + 2);
}
int orc_pin_zc_context(uint8_t *flags, int w, int h, int x, int y, int band) {      /* t1.go:312-384 == LUT t1_luts.go:34-110 */
    (void)h; init_tables(); return lutZCCtx[band * 256 + zc_packed(pin_at(flags, w, x, y), w + 2)];
}
void orc_pin_sc_context(uint8_t *flags, int w, int h, int x, int y, int *ctx, int *pred) { /* t1.go:387-460 */
    (void)h; init_tables();
    uint8_t *f = pin_at(flags, w, x, y);
    const int s = w + 2, sci = sc_index(f[-1], f[1], f[-s], f[s]);
    *ctx = lutSignCtx[sci] + CtxSC0; *pred = lutSignPred[sci];
}
/* canUseRunLength(x, y, bp)   t1.go:1195-1208 == canUseRunLengthInlined :774-813 == the test inside orc_t1_encode */
int orc_pin_can_use_run_length(uint8_t *flags, int w, int h, int x, int y) {
    if (y + 4 > h) return 0;
    for (int yy = 0; yy < 4; yy++) {
        uint8_t *f = pin_at(flags, w, x, y + yy);
        if (*f & (T1Sig | T1Visit)) return 0;
        if (has_sig_neighbor(f, w + 2)) return 0;
    }
    return 1;
}

/* ======================================================================== */
/* encoder.createTileHeader (encoder.go:746-760) -- SURVEY 8f rank 1          */
/* ======================================================================== */
size_t orc_create_tile_header(int tile_idx, const uint8_t *tile_data, size_t len, uint8_t *out) {
    const uint32_t psot = (uint32_t)(14 + len);             /* uint32(14 + len(tileData)) */
    const uint16_t isot = (uint16_t)tile_idx;               /* uint16(tileIdx) */
    out[0] = 0xFF; out[1] = 0x90;                           /* codestream.SOT, markers.go:9 */
    out[2] = 0; out[3] = 10;                                /* sotLength */
    out[4] = (uint8_t)(isot >> 8); out[5] = (uint8_t)isot;
    out[6] = (uint8_t)(psot >> 24); out[7] = (uint8_t)(psot >> 16); out[8] = (uint8_t)(psot >> 8); out[9] = (uint8_t)psot;
    out[10] = 0;                                            /* tile-part index */
    out[11] = 1;                                            /* number of tile-parts */
    out[12] = 0xFF; out[13] = 0x93;                         /* codestream.SOD, markers.go:10 */
    for (size_t i = 0; i < len; i++) out[14 + i] = tile_data[i];
    return 14 + len;
}
