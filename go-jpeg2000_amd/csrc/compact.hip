// compact.hip -- turns the per-job worst-case slots into the dense byte stream the reference
// builds with `tileData = append(tileData, encoded...)` (encoder.go:684): exclusive scan of the
// job lengths, then one wave per job gathers its bytes.
#include "j2k_internal.h"

namespace j2k {

// single workgroup, chunked inclusive scan with a running carry; offs[n] = total
__global__ __launch_bounds__(1024) void scan_lens_kernel(const uint32_t *__restrict__ lens, int n, uint64_t *__restrict__ offs) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint64_t v = (i < n) ? lens[i] : 0;
        uint64_t x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_sum[wv] = x;
        __syncthreads();
        uint64_t pre = carry_s;
        for (int k = 0; k < wv; k++) pre += wave_sum[k];
        if (i < n) offs[i] = pre + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (tid == 0) offs[n] = carry_s;
}

__global__ __launch_bounds__(256) void gather_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ slots,
                                                     const uint32_t *__restrict__ lens, const uint64_t *__restrict__ offs,
                                                     uint8_t *__restrict__ stream) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= njobs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t len = lens[j];
    const uint8_t *src = slots + jobs[j].out_off;   // 16-byte aligned
    uint8_t *dst = stream + offs[j];
    // head bytes up to 4-byte alignment of dst, then dwords, then the tail
    uint32_t head = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);
    if (head > len) head = len;
    if (lane < (int)head) dst[lane] = src[lane];
    const uint32_t nd = (len - head) >> 2;
    for (uint32_t i = lane; i < nd; i += 64) {
        const uint8_t *s = src + head + 4 * i;
        const uint32_t v = (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16 | (uint32_t)s[3] << 24;
        *reinterpret_cast<uint32_t *>(dst + head + 4 * i) = v;
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
