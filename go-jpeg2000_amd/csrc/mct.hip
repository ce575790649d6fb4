// mct.hip -- stand-alone elementwise kernels for the reference's internal/mct functions.
// (The frame pipeline never launches these: DC shift and RCT/ICT are fused into the
// level-0 DWT kernels.  They exist for the one-call-per-reference-function host ABI.)
//
//   mct.DCLevelShiftForward/Inverse  internal/mct/mct.go:96-101, 113-118
//   mct.ForwardRCT / InverseRCT      mct.go:28-38, 56-66
//   mct.ForwardICT / InverseICT      mct.go:14-24, 43-53   (f64, no FMA contraction)
#include "j2k_internal.h"

namespace j2k {

__global__ __launch_bounds__(256) void add_const_kernel(int32_t *__restrict__ d, size_t n, int delta) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        d[i] = (int)((unsigned)d[i] + (unsigned)delta);
}

__global__ __launch_bounds__(256) void rct_fwd_kernel(int32_t *__restrict__ r, int32_t *__restrict__ g,
                                                      int32_t *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned R = (unsigned)r[i], G = (unsigned)g[i], B = (unsigned)b[i];
        r[i] = (int)(R + 2u * G + B) >> 2;
        g[i] = (int)(B - G);
        b[i] = (int)(R - G);
    }
}

__global__ __launch_bounds__(256) void rct_inv_kernel(int32_t *__restrict__ y, int32_t *__restrict__ u,
                                                      int32_t *__restrict__ v, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned Y = (unsigned)y[i], U = (unsigned)u[i], V = (unsigned)v[i];
        const unsigned G = Y - (unsigned)((int)(U + V) >> 2);
        y[i] = (int)(V + G);
        u[i] = (int)G;
        v[i] = (int)(U + G);
    }
}

// Go/amd64 never fuses a*b+c: this file is compiled with -ffp-contract=off and the
// products are additionally kept in separate statements.
__global__ __launch_bounds__(256) void ict_fwd_kernel(double *__restrict__ r, double *__restrict__ g,
                                                      double *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double R = r[i], G = g[i], B = b[i];
        const double y = 0.299 * R + 0.587 * G + 0.114 * B;
        const double cb = -0.16875 * R - 0.33126 * G + 0.5 * B;
        const double cr = 0.5 * R - 0.41869 * G - 0.08131 * B;
        r[i] = y; g[i] = cb; b[i] = cr;
    }
}

__global__ __launch_bounds__(256) void ict_inv_kernel(double *__restrict__ y, double *__restrict__ cb,
                                                      double *__restrict__ cr, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double Y = y[i], Cb = cb[i], Cr = cr[i];
        const double R = Y + 1.402 * Cr;
        const double G = Y - 0.34413 * Cb - 0.71414 * Cr;
        const double B = Y + 1.772 * Cb;
        y[i] = R; cb[i] = G; cr[i] = B;
    }
}

static inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

hipError_t launch_add_const(hipStream_t s, int32_t *d, size_t n, int delta) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(add_const_kernel, dim3(grid_for(n)), dim3(256), 0, s, d, n, delta);
    return hipGetLastError();
}
hipError_t launch_rct(hipStream_t s, int32_t *a, int32_t *b, int32_t *c, size_t n, int inverse) {
    if (!n) return hipSuccess;
    if (inverse) hipLaunchKernelGGL(rct_inv_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, n);
    else hipLaunchKernelGGL(rct_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, n);
    return hipGetLastError();
}
hipError_t launch_ict(hipStream_t s, double *a, double *b, double *c, size_t n, int inverse) {
    if (!n) return hipSuccess;
    if (inverse) hipLaunchKernelGGL(ict_inv_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, n);
    else hipLaunchKernelGGL(ict_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, n);
    return hipGetLastError();
}

}  // namespace j2k
