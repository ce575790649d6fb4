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

// n bytes src -> dst, any alignment on both sides, one wavefront: bytes up to the destination's next 16-byte boundary,
// then aligned 16-byte stores of (possibly unaligned) 16-byte loads, then the tail bytes
__device__ __forceinline__ void copy_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t n, int lane) {
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15));
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t nv = (n - head) >> 4;
    for (uint32_t i = lane; i < nv; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, src + head + 16 * (size_t)i, 16);
        *reinterpret_cast<uint4 *>(dst + head + 16 * (size_t)i) = v;
    }
    const uint32_t done = head + (nv << 4);
    if (done + lane < n) dst[done + lane] = src[done + lane];
}
__device__ __forceinline__ void zero_run(uint8_t *__restrict__ dst, uint32_t n, int lane) {
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15));
    if ((uint32_t)lane < head) dst[lane] = 0;
    const uint32_t nv = (n - head) >> 4;
    for (uint32_t i = lane; i < nv; i += 64) *reinterpret_cast<uint4 *>(dst + head + 16 * (size_t)i) = make_uint4(0, 0, 0, 0);
    const uint32_t done = head + (nv << 4);
    if (done + lane < n) dst[done + lane] = 0;
}

// one wavefront per job.  maglens != NULL (HT blocks coded by j2k_plan_encode_stream): the slot holds
// MagSgn | <hole> | VLC | SCUP -- the MEL segment of max(64, 2wh)/4 zero bytes (ht.go:978, 1019) was never written to the
// slot and is produced here as zeros, so two thirds of a 64x64 block's bytes are neither stored twice nor read back.
__global__ __launch_bounds__(256) void gather_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ slots,
                                                     const uint32_t *__restrict__ lens, const uint64_t *__restrict__ offs,
                                                     uint8_t *__restrict__ stream, const uint32_t *__restrict__ maglens) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= njobs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t len = lens[j];
    if (len == 0) return;
    const BlockJob J = jobs[j];
    const uint8_t *src = slots + J.out_off;
    uint8_t *dst = stream + offs[j];
    if (!maglens) {
        copy_bytes(dst, src, len, lane);
        return;
    }
    const uint32_t mag = maglens[j];
    const size_t nsamp = (size_t)J.w * J.h;
    const uint32_t mel = (uint32_t)((nsamp * 2 < 64 ? 64 : nsamp * 2) / 4);
    copy_bytes(dst, src, mag, lane);
    zero_run(dst + mag, mel, lane);
    copy_bytes(dst + mag + mel, src + mag + mel, len - mag - mel, lane);
}

hipError_t launch_compact(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *slots, const uint32_t *lens,
                          uint64_t *offs, uint8_t *stream, const uint32_t *maglens) {
    hipLaunchKernelGGL(scan_lens_kernel, dim3(1), dim3(1024), 0, s, lens, njobs, offs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || njobs <= 0) return e;
    hipLaunchKernelGGL(gather_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, jobs, njobs, slots, lens, offs, stream, maglens);
    return hipGetLastError();
}

}  // namespace j2k
