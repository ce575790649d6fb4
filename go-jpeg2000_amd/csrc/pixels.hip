// pixels.hip -- pixel unpack / pack at native width: the host loops on either side of the
// tile-component path.
//
// Replaces (reference, mrjoshuak/go-jpeg2000):
//   encoder.extractImageData   encoder.go:79-213   image.Gray / Gray16 / RGBA / RGBA64 / NRGBA / NRGBA64 -> planar
//                                                  int32 components (+ the Options.Precision rescale, :196-210)
//   decoder.createImage        decoder.go:417-588  planar int32 -> image.Gray / Gray16 / RGBA / RGBA64 (clamp, rescale)
//
// The byte layouts are Go's image package: Pix is row-major with `stride` bytes per row; 16-bit samples are
// big-endian; RGBA = R,G,B,A bytes.  Arithmetic is Go int32: the rescale multiplies wrap (65535 * 65535 overflows)
// and `/` truncates toward zero, exactly as written in the reference.
#include "j2k_internal.h"

namespace j2k {

__device__ __forceinline__ int be16(const uint8_t *p) { return (int)p[0] << 8 | (int)p[1]; }
__device__ __forceinline__ int go_muldiv(int v, int a, int b) { return (int)((uint32_t)v * (uint32_t)a) / b; }   // int32 wrap, trunc

__global__ __launch_bounds__(256) void unpack_pixels_kernel(const uint8_t *__restrict__ pix, size_t stride, int format, int w, int h,
                                                            int src_max, int dst_max, int32_t *__restrict__ planes) {
    const size_t n = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / (size_t)w), x = (int)(i - (size_t)y * w);
        const uint8_t *row = pix + (size_t)y * stride;
        int v[4] = {0, 0, 0, 0};
        int nc = 1;
        switch (format) {
        case J2K_PIX_GRAY8: v[0] = row[x]; break;
        case J2K_PIX_GRAY16: v[0] = be16(row + 2 * (size_t)x); break;
        case J2K_PIX_RGBA8: {        // alpha ignored (encoder.go:107)
            const uint32_t p = *reinterpret_cast<const uint32_t *>(row + 4 * (size_t)x);   // Pix rows of image.RGBA are 4-byte aligned when stride % 4 == 0
            v[0] = p & 0xFF; v[1] = (p >> 8) & 0xFF; v[2] = (p >> 16) & 0xFF; nc = 3;
            break;
        }
        case J2K_PIX_NRGBA8: {
            const uint32_t p = *reinterpret_cast<const uint32_t *>(row + 4 * (size_t)x);
            v[0] = p & 0xFF; v[1] = (p >> 8) & 0xFF; v[2] = (p >> 16) & 0xFF; v[3] = p >> 24; nc = 4;
            break;
        }
        case J2K_PIX_RGBA64: {
            const uint8_t *q = row + 8 * (size_t)x;
            v[0] = be16(q); v[1] = be16(q + 2); v[2] = be16(q + 4); nc = 3;
            break;
        }
        default: {                   // J2K_PIX_NRGBA64
            const uint8_t *q = row + 8 * (size_t)x;
            v[0] = be16(q); v[1] = be16(q + 2); v[2] = be16(q + 4); v[3] = be16(q + 6); nc = 4;
            break;
        }
        }
        for (int c = 0; c < nc; c++) {
            int t = v[c];
            if (dst_max != src_max) t = go_muldiv(t, dst_max, src_max);       // encoder.go:196-210
            planes[(size_t)c * n + i] = t;
        }
    }
}

// decoder.createImage: ncomp 1 -> Gray (precision <= 8) or Gray16; 3 / 4 -> RGBA (precision <= 8) or RGBA64
__global__ __launch_bounds__(256) void pack_pixels_kernel(const int32_t *__restrict__ planes, int ncomp, int precision, int w, int h,
                                                          uint8_t *__restrict__ pix, size_t stride) {
    const size_t n = (size_t)w * h;
    const int max_val = (int)((1u << precision) - 1);
    const bool wide = precision > 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / (size_t)w), x = (int)(i - (size_t)y * w);
        uint8_t *row = pix + (size_t)y * stride;
        int v[4];
        for (int c = 0; c < ncomp; c++) {
            int t = planes[(size_t)c * n + i];
            t = t < 0 ? 0 : (t > max_val ? max_val : t);
            if (wide) t = go_muldiv(t, 65535, max_val);
            else if (precision != 8) t = go_muldiv(t, 255, max_val);
            v[c] = t;
        }
        if (ncomp == 1) {
            if (wide) { row[2 * (size_t)x] = (uint8_t)((uint32_t)v[0] >> 8); row[2 * (size_t)x + 1] = (uint8_t)v[0]; }   // uint16(v), big-endian
            else row[x] = (uint8_t)v[0];
        } else {
            const int a = ncomp == 4 ? v[3] : (wide ? 65535 : 255);
            if (wide) {
                uint8_t *q = row + 8 * (size_t)x;
                const int o[4] = {v[0], v[1], v[2], a};
                for (int c = 0; c < 4; c++) { q[2 * c] = (uint8_t)((uint32_t)o[c] >> 8); q[2 * c + 1] = (uint8_t)o[c]; }
            } else {
                *reinterpret_cast<uint32_t *>(row + 4 * (size_t)x) =
                    ((uint32_t)v[0] & 0xFF) | ((uint32_t)v[1] & 0xFF) << 8 | ((uint32_t)v[2] & 0xFF) << 16 | ((uint32_t)a & 0xFF) << 24;
            }
        }
    }
}

hipError_t launch_unpack_pixels(hipStream_t s, const uint8_t *pix, size_t stride, int format, int w, int h, int src_max, int dst_max,
                                int32_t *planes) {
    const size_t n = (size_t)w * h;
    if (!n) return hipSuccess;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(unpack_pixels_kernel, dim3(blocks), dim3(256), 0, s, pix, stride, format, w, h, src_max, dst_max, planes);
    return hipGetLastError();
}

hipError_t launch_pack_pixels(hipStream_t s, const int32_t *planes, int ncomp, int precision, int w, int h, uint8_t *pix, size_t stride) {
    const size_t n = (size_t)w * h;
    if (!n) return hipSuccess;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(pack_pixels_kernel, dim3(blocks), dim3(256), 0, s, planes, ncomp, precision, w, h, pix, stride);
    return hipGetLastError();
}

}  // namespace j2k
