// colorspace.hip -- decode-side colour conversions to sRGB: the elementwise follow-on of the
// inverse component transform.
//
// Replaces (reference, mrjoshuak/go-jpeg2000): colorspace.go:54-90 getColorConversion and the fourteen
// convert*ToRGB functions it selects (:92-480), clampToInt32 / clampFloat64 (:483-501).  float64 arithmetic in the
// reference's association, compiled -ffp-contract=off; the matrix conversions are bit-exact, the four that go through
// math.Pow (CIELab, CIEJab, e-sRGB, ROMM-RGB) depend on the pow implementation and are tested to +-1 code value.
#include "j2k_internal.h"

namespace j2k {

// Go int32(float64) on amd64 = CVTTSD2SL: truncation toward zero, 0x80000000 for NaN and for anything that does not fit
// (v_cvt_i32_f64 saturates instead: right at the negative end, INT_MAX at the positive one, 0 for NaN)
__device__ __forceinline__ int cs_int32(double v) { return v < 2147483648.0 ? (int)v : (int)0x80000000; }
__device__ __forceinline__ int clamp_to_int32(double v, double lo, double hi) {      // colorspace.go:483-491
    if (v < lo) return cs_int32(lo);
    if (v > hi) return cs_int32(hi);
    return cs_int32(v + 0.5);
}
__device__ __forceinline__ double clamp_f64(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }   // :494-501
__device__ __forceinline__ double lab_inverse_f(double t) {                        // :293-299
    const double delta = 6.0 / 29.0;
    if (t > delta) return t * t * t;
    return 3 * delta * delta * (t - 4.0 / 29.0);
}
__device__ __forceinline__ double srgb_gamma(double linear) {                      // :302-307
    if (linear <= 0.0031308) return 12.92 * linear;
    return 1.055 * pow(linear, 1.0 / 2.4) - 0.055;
}
__device__ __forceinline__ void xyz_to_srgb(double x, double y, double z, double &r, double &g, double &b) {
    r = 3.2404542 * x - 1.5371385 * y - 0.4985314 * z;
    g = -0.9692660 * x + 1.8760108 * y + 0.0415560 * z;
    b = 0.0556434 * x - 0.2040259 * y + 1.0572252 * z;
}

__global__ __launch_bounds__(256) void colorspace_kernel(int cs, int32_t *__restrict__ p0, int32_t *__restrict__ p1, int32_t *__restrict__ p2,
                                                         const int32_t *__restrict__ p3, size_t n, int precision) {
    const double maxVal = (double)(int)((1u << precision) - 1);
    const double halfVal = (double)(int)(1u << (precision - 1));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c0 = p0[i], c1 = p1[i], c2 = p2[i];
        double r, g, b;
        switch (cs) {
        case J2K_CS_SYCC: case J2K_CS_YPBPR60: case J2K_CS_YPBPR50: case J2K_CS_EYCC: {   // :92-116, :429-452, :456-480
            const double y = (double)c0, cb = (double)c1 - halfVal, cr = (double)c2 - halfVal;
            r = y + 1.5748 * cr; g = y - 0.1873 * cb - 0.4681 * cr; b = y + 1.8556 * cb;
            break;
        }
        case J2K_CS_YCBCR2: case J2K_CS_YCBCR3: {                                         // :119-142
            const double y = (double)c0, cb = (double)c1 - halfVal, cr = (double)c2 - halfVal;
            r = y + 1.402 * cr; g = y - 0.344136 * cb - 0.714136 * cr; b = y + 1.772 * cb;
            break;
        }
        case J2K_CS_PHOTOYCC: case J2K_CS_YCCK: {                                         // :145-169, :218-247
            const double scale = maxVal / 255.0;
            const double y = (double)c0 / scale, k1 = (double)c1 / scale - 156.0, k2 = (double)c2 / scale - 156.0;
            r = y + 1.3584 * k2; g = y - 0.4302 * k1 - 0.7915 * k2; b = y + 2.2179 * k1;
            if (cs == J2K_CS_YCCK) {
                const double k = (double)p3[i] / maxVal;
                r = r * scale * (1 - k); g = g * scale * (1 - k); b = b * scale * (1 - k);
            } else {
                r = r * scale; g = g * scale; b = b * scale;
            }
            break;
        }
        case J2K_CS_CMY: {                                                                  // :172-189 (integer, no clamp)
            const int mv = (int)((1u << precision) - 1);
            p0[i] = (int)((unsigned)mv - (unsigned)c0); p1[i] = (int)((unsigned)mv - (unsigned)c1); p2[i] = (int)((unsigned)mv - (unsigned)c2);
            continue;
        }
        case J2K_CS_CMYK: {                                                                 // :192-215
            const double c = (double)c0 / maxVal, m = (double)c1 / maxVal, y = (double)c2 / maxVal, k = (double)p3[i] / maxVal;
            r = (1 - c) * (1 - k) * maxVal; g = (1 - m) * (1 - k) * maxVal; b = (1 - y) * (1 - k) * maxVal;
            break;
        }
        case J2K_CS_CIELAB: case J2K_CS_CIEJAB: {                                           // :250-290, :319-359
            const double L = (double)c0 / maxVal * 100.0, a = (double)c1 / maxVal * 255.0 - 128.0, bb = (double)c2 / maxVal * 255.0 - 128.0;
            const double fy = (L + 16.0) / 116.0, fx = a / 500.0 + fy, fz = fy - bb / 200.0;
            const double x = 0.96422 * lab_inverse_f(fx), y = 1.0 * lab_inverse_f(fy), z = 0.82521 * lab_inverse_f(fz);
            double rl, gl, bl;
            xyz_to_srgb(x, y, z, rl, gl, bl);
            r = srgb_gamma(rl) * maxVal; g = srgb_gamma(gl) * maxVal; b = srgb_gamma(bl) * maxVal;
            break;
        }
        case J2K_CS_ESRGB: {                                                                // :362-388
            const double er = (double)c0 / maxVal * 1.25 - 0.25, eg = (double)c1 / maxVal * 1.25 - 0.25, eb = (double)c2 / maxVal * 1.25 - 0.25;
            r = srgb_gamma(clamp_f64(er, 0, 1)) * maxVal; g = srgb_gamma(clamp_f64(eg, 0, 1)) * maxVal; b = srgb_gamma(clamp_f64(eb, 0, 1)) * maxVal;
            break;
        }
        case J2K_CS_ROMMRGB: {                                                              // :391-426
            const double rr = pow((double)c0 / maxVal, 1.8), gr = pow((double)c1 / maxVal, 1.8), br = pow((double)c2 / maxVal, 1.8);
            const double x = 0.7977 * rr + 0.1352 * gr + 0.0313 * br;
            const double y = 0.2880 * rr + 0.7119 * gr + 0.0001 * br;
            const double z = 0.0000 * rr + 0.0000 * gr + 0.8249 * br;
            double rl, gl, bl;
            xyz_to_srgb(x, y, z, rl, gl, bl);
            r = srgb_gamma(clamp_f64(rl, 0, 1)) * maxVal; g = srgb_gamma(clamp_f64(gl, 0, 1)) * maxVal; b = srgb_gamma(clamp_f64(bl, 0, 1)) * maxVal;
            break;
        }
        default: continue;                                                                  // sRGB, gray, ...: no conversion (:86-89)
        }
        p0[i] = clamp_to_int32(r, 0, maxVal);
        p1[i] = clamp_to_int32(g, 0, maxVal);
        p2[i] = clamp_to_int32(b, 0, maxVal);
    }
}

hipError_t launch_colorspace(hipStream_t s, int cs, int32_t *planes, int ncomp, size_t n, int precision) {
    if (!n) return hipSuccess;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(colorspace_kernel, dim3(blocks), dim3(256), 0, s, cs, planes, planes + n, planes + 2 * n,
                       ncomp >= 4 ? planes + 3 * n : planes, n, precision);
    return hipGetLastError();
}

}  // namespace j2k
