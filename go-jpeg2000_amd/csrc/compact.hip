// compact.hip -- turns the per-job worst-case slots into the dense byte stream the reference
// builds with `tileData = append(tileData, encoded...)` (encoder.go:684): exclusive scan of the
// job lengths, then one wave per job gathers its bytes.
#include "j2k_internal.h"

namespace j2k {

// single workgroup; eight lengths per thread per pass (one pass up to 8192 jobs), every load issued before the first
// use, one barrier per pass; offs[n] = total
__global__ __launch_bounds__(1024) void scan_lens_kernel(const uint32_t *__restrict__ lens, int n, uint64_t *__restrict__ offs) {
    __shared__ uint64_t wave_sum[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t carry = 0;
    int par = 0;
    for (int base = 0; base < n; base += 8192, par ^= 1) {
        const int i0 = base + tid * 8;
        uint32_t v[8];
        if (i0 + 8 <= n) {
            const uint4 a = *reinterpret_cast<const uint4 *>(lens + i0), b = *reinterpret_cast<const uint4 *>(lens + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (i0 + k < n) ? lens[i0 + k] : 0u;
        }
        uint64_t t = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) t += v[k];
        uint64_t x = t;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_sum[par][wv] = x;
        __syncthreads();
        uint64_t pre = carry, all = carry;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t ws = wave_sum[par][k];
            if (k < wv) pre += ws;
            all += ws;
        }
        uint64_t o = pre + x - t;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k < n) offs[i0 + k] = o;
            o += v[k];
        }
        carry = all;
    }
    if (tid == 0) offs[n] = carry;
}

// one wavefront per job: the slot is 16-byte aligned, the destination is not -- destination-aligned 16-byte stores
// whose dwords are assembled from two aligned source dwords (v_alignbyte)
__global__ __launch_bounds__(256) void gather_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ slots,
                                                     const uint32_t *__restrict__ lens, const uint64_t *__restrict__ offs,
                                                     uint8_t *__restrict__ stream) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= njobs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t len = lens[j];
    const uint8_t *src = slots + jobs[j].out_off;   // 16-byte aligned
    uint8_t *dst = stream + offs[j];
    if (len == 0) return;
    // head bytes up to 4-byte alignment of dst, then dwords, then the tail
    uint32_t head = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);
    if (head > len) head = len;
    if (lane < (int)head) dst[lane] = src[lane];
    const uint32_t nd = (len - head) >> 2;          // destination dword i holds source bytes head + 4i .. head + 4i + 3
    const uint32_t *S = reinterpret_cast<const uint32_t *>(src);
    const uint32_t last = (len - 1) >> 2;            // last source dword that holds job bytes
    uint32_t *D = reinterpret_cast<uint32_t *>(dst + head);
    const uint32_t sh = 8 * head;
    for (uint32_t i = 4 * lane; i < nd; i += 256) {
        const uint4 a = *reinterpret_cast<const uint4 *>(S + i);               // i is a multiple of 4: aligned
        const uint32_t e = S[min(i + 4, last)];
        uint32_t d0 = a.x, d1 = a.y, d2 = a.z, d3 = a.w;
        if (head) {
            d0 = (uint32_t)(((uint64_t)a.y << 32 | a.x) >> sh);
            d1 = (uint32_t)(((uint64_t)a.z << 32 | a.y) >> sh);
            d2 = (uint32_t)(((uint64_t)a.w << 32 | a.z) >> sh);
            d3 = (uint32_t)(((uint64_t)e << 32 | a.w) >> sh);
        }
        if (i + 4 <= nd) {
            D[i] = d0; D[i + 1] = d1; D[i + 2] = d2; D[i + 3] = d3;
        } else {
            D[i] = d0;
            if (i + 1 < nd) D[i + 1] = d1;
            if (i + 2 < nd) D[i + 2] = d2;
        }
    }
    const uint32_t done = head + 4 * nd;
    if (done + lane < len) dst[done + lane] = src[done + lane];
}

hipError_t launch_compact(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *slots, const uint32_t *lens,
                          uint64_t *offs, uint8_t *stream, void *) {
    hipLaunchKernelGGL(scan_lens_kernel, dim3(1), dim3(1024), 0, s, lens, njobs, offs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || njobs <= 0) return e;
    hipLaunchKernelGGL(gather_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, jobs, njobs, slots, lens, offs, stream);
    return hipGetLastError();
}

}  // namespace j2k
