// compact.hip -- turns the per-job worst-case slots into the dense byte stream the reference
// builds with `tileData = append(tileData, encoded...)` (encoder.go:684): exclusive scan of the
// job lengths, then one wave per job gathers its bytes.
#include "j2k_internal.h"
#include <algorithm>

namespace j2k {

// single workgroup; eight lengths per thread per pass (one pass up to 8192 jobs), every load issued before the first
// use, one barrier per pass; offs[n] = total
// mels / toffs (both or neither): also the exclusive scan of the TRANSPORT lengths (a block's length without its
// mels[j] bytes of MEL zero run; see the pack kernels below) -- the same pass, a second running sum.
template <bool TWO>
__global__ __launch_bounds__(1024) void scan_lens_kernel(const uint32_t *__restrict__ lens, int n, uint64_t *__restrict__ offs,
                                                         const uint32_t *__restrict__ mels, uint64_t *__restrict__ toffs) {
    __shared__ uint64_t wave_sum[2][16], wave_sum2[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t carry = 0, carry2 = 0;
    int par = 0;
    for (int base = 0; base < n; base += 8192, par ^= 1) {
        const int i0 = base + tid * 8;
        uint32_t v[8], m[8];
        if (i0 + 8 <= n) {
            const uint4 a = *reinterpret_cast<const uint4 *>(lens + i0), b = *reinterpret_cast<const uint4 *>(lens + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (i0 + k < n) ? lens[i0 + k] : 0u;
        }
        if (TWO && i0 + 8 <= n) {                       // both vectors in flight with the lens above: no serialised loads
            const uint4 a = *reinterpret_cast<const uint4 *>(mels + i0), b = *reinterpret_cast<const uint4 *>(mels + i0 + 4);
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) m[k] = (TWO && i0 + k < n) ? mels[i0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = v[k] ? m[k] : 0u;
        uint64_t t = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { t += v[k]; if (TWO) t2 += v[k] - m[k]; }
        uint64_t x = t, x2 = t2;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
            if (TWO) {
                const uint64_t y2 = __shfl_up(x2, o);
                if (lane >= o) x2 += y2;
            }
        }
        if (lane == 63) { wave_sum[par][wv] = x; if (TWO) wave_sum2[par][wv] = x2; }
        __syncthreads();
        uint64_t pre = carry, all = carry, pre2 = carry2, all2 = carry2;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t ws = wave_sum[par][k], ws2 = TWO ? wave_sum2[par][k] : 0;
            if (k < wv) { pre += ws; pre2 += ws2; }
            all += ws; all2 += ws2;
        }
        uint64_t o = pre + x - t, o2 = pre2 + x2 - t2;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k < n) { offs[i0 + k] = o; if (TWO) toffs[i0 + k] = o2; }
            o += v[k]; o2 += v[k] - m[k];
        }
        carry = all; carry2 = all2;
    }
    if (tid == 0) { offs[n] = carry; if (TWO) toffs[n] = carry2; }
}

// (copy_bytes / zero_run / gather_job: j2k_internal.h -- the packet coder gathers from slots too)
__global__ __launch_bounds__(256) void gather_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ slots,
                                                     const uint32_t *__restrict__ lens, const uint64_t *__restrict__ offs,
                                                     uint8_t *__restrict__ stream, const uint32_t *__restrict__ maglens) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= njobs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t len = lens[j];
    if (len == 0) return;
    gather_job(jobs[j], slots, stream + offs[j], len, maglens != nullptr, maglens ? maglens[j] : 0u, lane);
}

// The same with the exclusive scan of the lengths inside (up to 8192 jobs, no transport offsets): every workgroup sums the
// lengths before its four jobs itself -- at most eight 16-byte loads per thread from a 28 KB array that stays in L2, issued
// together with the wave's own job / length / MagSgn-length loads, so the sum costs the gather no extra round trip -- and
// the single-workgroup scan launch that used to stand between the block coder and the gather (5.5 us for 7005 lengths,
// nearly all of it launch and latency) is gone.  Also writes offs[0..njobs] for the decoder and the caller.
__global__ __launch_bounds__(256) void gather_scan_kernel(const BlockJob *__restrict__ jobs, int njobs, const uint8_t *__restrict__ slots,
                                                          const uint32_t *__restrict__ lens, uint64_t *__restrict__ offs,
                                                          uint8_t *__restrict__ stream, const uint32_t *__restrict__ maglens) {
    __shared__ uint64_t s_wave[4];
    __shared__ uint32_t s_len[4];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int j0 = blockIdx.x * 4, j = j0 + wv;
    const bool mine = j < njobs;
    // this wave's own inputs first (wave-uniform addresses)
    const uint32_t len = mine ? lens[j] : 0u;
    const uint32_t mag = (mine && maglens) ? maglens[j] : 0u;
    BlockJob J = jobs[mine ? j : 0];
    // lengths of the jobs before this workgroup: element i of pass p is lens[1024 p + 4 tid + i]
    uint64_t part = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int i0 = 1024 * p + 4 * tid;
        if (i0 + 4 <= j0) {
            const uint4 v = *reinterpret_cast<const uint4 *>(lens + i0);
            part += (uint64_t)v.x + v.y + v.z + v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (i0 + i < j0) part += lens[i0 + i];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) { s_wave[wv] = part; s_len[wv] = len; }
    __syncthreads();
    uint64_t off = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (k < wv) off += s_len[k];
    if (!mine) return;
    if (lane == 0) {
        offs[j] = off;
        if (j == njobs - 1) offs[njobs] = off + len;
    }
    if (len == 0) return;
    gather_job(J, slots, stream + off, len, maglens != nullptr, mag, lane);
}

// ---- transport form of a stream (multi-GPU gather) ----
// The reference's HT block is  MagSgn | MEL | VLC | SCUP  with MEL = max(64, 2wh)/4 ZERO bytes (ht.go:978, 1019): two
// thirds of a 64x64 block.  A rank that sends its stream to the root over one xGMI link sends the blocks without those
// runs, preceded by the per-block arrays the root needs to put them back (and to assemble packets): the pack is
//   [total bytes u64 | payload bytes u64 | n u32 | mel flag u32 | pad to 64] lens u32[n] | maglens u32[n] | numbps u8[n] |
//   offs u64[n+1] | toffs u64[n+1] | payload      (every section 16-byte aligned)
// and j2k_plan_unpack_stream at the root rebuilds the dense stream, byte for byte.
#define UNPACK_BATCH 16
__host__ __device__ inline size_t pack_a16(size_t x) { return (x + 15) & ~size_t(15); }
struct PackLayout { size_t lens, mag, nb, offs, toffs, payload; };
__host__ __device__ inline PackLayout pack_layout(size_t n) {
    PackLayout L;
    L.lens = 64;
    L.mag = pack_a16(L.lens + 4 * n);
    L.nb = pack_a16(L.mag + 4 * n);
    L.offs = pack_a16(L.nb + n);
    L.toffs = pack_a16(L.offs + 8 * (n + 1));
    L.payload = pack_a16(L.toffs + 8 * (n + 1));
    return L;
}
size_t pack_header_bytes(size_t n) { return pack_layout(n).payload; }

__device__ __forceinline__ uint32_t mel_bytes(const BlockJob &J) {
    const size_t nsamp = (size_t)J.w * J.h;
    return (uint32_t)((nsamp * 2 < 64 ? 64 : nsamp * 2) / 4);
}

// one wavefront per block: its header entries and its bytes; toffs = the exclusive scan of the transport lengths that
// j2k_plan_encode_stream's scan left in the plan (for a stream without MEL runs: offs itself)
__global__ __launch_bounds__(256) void pack_kernel(const BlockJob *__restrict__ jobs, int n, const uint8_t *__restrict__ stream,
                                                   const uint64_t *__restrict__ offs, const uint64_t *__restrict__ toffs,
                                                   const uint32_t *__restrict__ lens, const uint32_t *__restrict__ maglens,
                                                   const uint8_t *__restrict__ numbps, uint8_t *__restrict__ pack) {
    const PackLayout L = pack_layout((size_t)n);
    const int has_mel = maglens != nullptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t *h = reinterpret_cast<uint64_t *>(pack);
        h[0] = ((uint64_t)L.payload + toffs[n] + 15) & ~uint64_t(15);     // packs laid end to end stay 16-byte aligned
        h[1] = toffs[n];
        reinterpret_cast<uint32_t *>(pack)[4] = (uint32_t)n;
        reinterpret_cast<uint32_t *>(pack)[5] = (uint32_t)has_mel;
    }
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j > n) return;
    const int lane = threadIdx.x & 63;
    const uint64_t off = offs[j], toff = toffs[j];
    if (lane == 0) {
        reinterpret_cast<uint64_t *>(pack + L.offs)[j] = off;
        reinterpret_cast<uint64_t *>(pack + L.toffs)[j] = toff;
    }
    if (j == n) return;
    const uint32_t len = lens[j];
    const uint32_t mag = has_mel ? maglens[j] : len;
    if (lane == 0) {
        reinterpret_cast<uint32_t *>(pack + L.lens)[j] = len;
        reinterpret_cast<uint32_t *>(pack + L.mag)[j] = mag;
        pack[L.nb + j] = numbps[j];
    }
    if (len == 0) return;
    const uint32_t mel = has_mel ? mel_bytes(jobs[j]) : 0u;
    const uint8_t *src = stream + off;
    uint8_t *dst = pack + L.payload + toff;
    copy_bytes(dst, src, mag, lane);
    copy_bytes(dst + mag, src + mag + mel, len - mag - mel, lane);
}
// per-job MEL run length (plan build): max(64, 2wh) / 4
__global__ void mel_table_kernel(const BlockJob *__restrict__ jobs, int n, uint32_t *__restrict__ mels) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) mels[j] = mel_bytes(jobs[j]);
}
// root side: pack -> dense stream + offs / lens / numbps, up to UNPACK_BATCH packs of the same geometry per launch
// (blockIdx.y = which pack: at N = 8 the root rebuilds 7 peers' streams per frame slot, and one launch keeps far more
// copies in flight than seven).  fault: 4 = a pack does not belong to this plan / its pieces do not fit
struct UnpackBatch {
    const uint8_t *pack[UNPACK_BATCH];
    uint64_t pack_bytes[UNPACK_BATCH];     // bytes the caller really holds at pack[i] (>= the header: checked on the host)
    uint8_t *stream[UNPACK_BATCH];
    uint64_t *offs[UNPACK_BATCH];
    uint32_t *lens[UNPACK_BATCH];
    uint8_t *numbps[UNPACK_BATCH];
};
__global__ __launch_bounds__(256) void unpack_kernel(const BlockJob *__restrict__ jobs, int n, UnpackBatch B, uint64_t stream_cap,
                                                     int *__restrict__ fault) {
    const uint8_t *__restrict__ pack = B.pack[blockIdx.y];
    uint8_t *__restrict__ stream = B.stream[blockIdx.y];
    uint64_t *__restrict__ offs = B.offs[blockIdx.y];
    uint32_t *__restrict__ lens = B.lens[blockIdx.y];
    uint8_t *__restrict__ numbps = B.numbps[blockIdx.y];
    const PackLayout L = pack_layout((size_t)n);
    // header: the block count must be this plan's, and the payload size the pack claims must fit what the caller holds
    const uint64_t avail = B.pack_bytes[blockIdx.y] - L.payload;          // host: pack_bytes >= L.payload
    const uint64_t paylen = reinterpret_cast<const uint64_t *>(pack)[1];
    if (reinterpret_cast<const uint32_t *>(pack)[4] != (uint32_t)n || paylen > avail) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(fault, 4);
        return;
    }
    const int has_mel = (int)reinterpret_cast<const uint32_t *>(pack)[5];
    const uint64_t *poffs = reinterpret_cast<const uint64_t *>(pack + L.offs);
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j > n) return;
    const int lane = threadIdx.x & 63;
    if (lane == 0) offs[j] = poffs[j];
    if (j == n) return;
    const uint32_t len = reinterpret_cast<const uint32_t *>(pack + L.lens)[j];
    if (lane == 0) { lens[j] = len; numbps[j] = pack[L.nb + j]; }
    if (len == 0) return;
    const uint32_t mag = reinterpret_cast<const uint32_t *>(pack + L.mag)[j];
    const uint32_t mel = has_mel ? mel_bytes(jobs[j]) : 0u;
    const uint64_t toff = reinterpret_cast<const uint64_t *>(pack + L.toffs)[j];
    // a pack is foreign input: nothing is copied unless the block's pieces fit the pack and the stream.  Every test is
    // written so that no sum of untrusted 64-bit fields can wrap (an offset of 2^64 - k would pass `off + len > end`)
    const uint64_t pj = poffs[j], pn = poffs[n];
    if (mag > len || mel > len - mag || toff > paylen || (uint64_t)(len - mel) > paylen - toff ||
        pn > stream_cap || pj > pn || (uint64_t)len > pn - pj) {
        if (lane == 0) atomicMax(fault, 4);
        return;
    }
    const uint8_t *src = pack + L.payload + toff;
    uint8_t *dst = stream + pj;
    copy_bytes(dst, src, mag, lane);
    zero_run(dst + mag, mel, lane);
    copy_bytes(dst + mag + mel, src + mag, len - mag - mel, lane);
}

hipError_t launch_scan(hipStream_t s, const uint32_t *lens, int njobs, uint64_t *offs, const uint32_t *mels, uint64_t *toffs) {
    if (mels && toffs) hipLaunchKernelGGL(scan_lens_kernel<true>, dim3(1), dim3(1024), 0, s, lens, njobs, offs, mels, toffs);
    else hipLaunchKernelGGL(scan_lens_kernel<false>, dim3(1), dim3(1024), 0, s, lens, njobs, offs, mels, toffs);
    return hipGetLastError();
}
hipError_t launch_pack(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs, const uint64_t *toffs,
                       const uint32_t *lens, const uint8_t *numbps, const uint32_t *maglens, uint8_t *pack) {
    hipLaunchKernelGGL(pack_kernel, dim3((njobs + 1 + 3) / 4), dim3(256), 0, s, jobs, njobs, stream, offs, toffs, lens, maglens, numbps, pack);
    return hipGetLastError();
}
hipError_t launch_mel_table(hipStream_t s, const BlockJob *jobs, int njobs, uint32_t *mels) {
    if (njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(mel_table_kernel, dim3((njobs + 255) / 256), dim3(256), 0, s, jobs, njobs, mels);
    return hipGetLastError();
}
hipError_t launch_unpack(hipStream_t s, const BlockJob *jobs, int njobs, int count, const uint8_t *const *packs, const size_t *pack_bytes,
                         uint8_t *const *streams,
                         size_t stream_cap, uint64_t *const *offs, uint32_t *const *lens, uint8_t *const *numbps, int *fault) {
    for (int c0 = 0; c0 < count; c0 += UNPACK_BATCH) {
        UnpackBatch B{};
        const int m = count - c0 < UNPACK_BATCH ? count - c0 : UNPACK_BATCH;
        for (int i = 0; i < m; i++) {
            B.pack[i] = packs[c0 + i]; B.pack_bytes[i] = (uint64_t)pack_bytes[c0 + i]; B.stream[i] = streams[c0 + i]; B.offs[i] = offs[c0 + i]; B.lens[i] = lens[c0 + i];
            B.numbps[i] = numbps[c0 + i];
        }
        hipLaunchKernelGGL(unpack_kernel, dim3((njobs + 1 + 3) / 4, m), dim3(256), 0, s, jobs, njobs, B, (uint64_t)stream_cap, fault);
    }
    return hipGetLastError();
}

// ---- tile-parts on the device (multi-tile codestream assembly) ----
// encoder.createTileHeader (encoder.go:746-760) for every tile of a dense stream at once: tile t of the plan (index tile_first
// + t) is the bytes of its jobs, offs[job0[t]] .. offs[job0[t+1]]; its tile-part is SOT (FF90, Lsot = 10, Isot = uint16(index),
// Psot = uint32(14 + len), TPsot = 0, TNsot = 1), SOD (FF93), then the data -- laid end to end, tile t at 14 t + its
// stream offset.  One workgroup per (tile, 64 KiB chunk); the first chunk's first lanes write the header.  out_len[0] =
// total bytes.  (The host call j2k_assemble_tiles does the same from host memory; this one saves the host pass and lets one
// D2H copy carry the finished tile-parts.)
__global__ __launch_bounds__(256) void assemble_tiles_kernel(const uint8_t *__restrict__ stream, const uint64_t *__restrict__ offs,
                                                              const int *__restrict__ job0, int ntiles, int tile_first,
                                                              uint8_t *__restrict__ out, uint64_t *__restrict__ out_len) {
    const int t = blockIdx.x;
    const uint64_t base = offs[job0[0]], o0 = offs[job0[t]], o1 = offs[job0[t + 1]];
    const uint64_t len = o1 - o0;
    uint8_t *dst = out + (o0 - base) + 14ull * (uint64_t)t;
    if (blockIdx.y == 0 && threadIdx.x < 14) {
        const uint32_t idx = (uint32_t)(tile_first + t) & 0xFFFFu, psot = (uint32_t)(14 + len);
        const uint8_t hdr[14] = {0xFF, 0x90, 0x00, 0x0A, (uint8_t)(idx >> 8), (uint8_t)idx, (uint8_t)(psot >> 24), (uint8_t)(psot >> 16),
                                 (uint8_t)(psot >> 8), (uint8_t)psot, 0x00, 0x01, 0xFF, 0x93};
        dst[threadIdx.x] = hdr[threadIdx.x];
    }
    if (out_len && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *out_len = (offs[job0[ntiles]] - base) + 14ull * (uint64_t)ntiles;
    const uint64_t c0 = (uint64_t)blockIdx.y << 16;
    if (c0 >= len) return;
    const uint32_t n = (uint32_t)(len - c0 < 65536 ? len - c0 : 65536);
    const uint8_t *src = stream + o0 + c0;
    uint8_t *d = dst + 14 + c0;
    // bytes up to the destination's next 16-byte boundary, aligned 16-byte stores of (unaligned) 16-byte loads, tail bytes
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)d & 15)) & 15));
    if (threadIdx.x < head) d[threadIdx.x] = src[threadIdx.x];
    const uint32_t nv = (n - head) >> 4;
    for (uint32_t i = threadIdx.x; i < nv; i += 256) {
        uint4 v;
        __builtin_memcpy(&v, src + head + 16 * (size_t)i, 16);
        *reinterpret_cast<uint4 *>(d + head + 16 * (size_t)i) = v;
    }
    const uint32_t done = head + (nv << 4);
    if (done + threadIdx.x < n) d[done + threadIdx.x] = src[done + threadIdx.x];
}
hipError_t launch_assemble_tiles(hipStream_t s, const uint8_t *stream, const uint64_t *offs, const int *job0, int ntiles, int tile_first,
                                 uint64_t max_tile_bytes, uint8_t *out, uint64_t *out_len) {
    if (ntiles <= 0) return hipSuccess;
    const unsigned chunks = (unsigned)std::max<uint64_t>(1, (max_tile_bytes + 65535) >> 16);
    hipLaunchKernelGGL(assemble_tiles_kernel, dim3(ntiles, chunks), dim3(256), 0, s, stream, offs, job0, ntiles, tile_first, out, out_len);
    return hipGetLastError();
}

hipError_t launch_compact(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *slots, const uint32_t *lens,
                          uint64_t *offs, uint8_t *stream, const uint32_t *maglens, const uint32_t *mels, uint64_t *toffs) {
    if (njobs > 0 && njobs <= 8192 && !(mels && toffs) && ((uintptr_t)lens & 15) == 0) {
        hipLaunchKernelGGL(gather_scan_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, jobs, njobs, slots, lens, offs, stream, maglens);
        return hipGetLastError();
    }
    hipError_t e = launch_scan(s, lens, njobs, offs, mels, toffs);
    if (e != hipSuccess || njobs <= 0) return e;
    hipLaunchKernelGGL(gather_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, jobs, njobs, slots, lens, offs, stream, maglens);
    return hipGetLastError();
}

}  // namespace j2k
