// l0_probe: what does the MI355X memory system allow for the packed-pixel level-0 shape (4 B read : 12 B written per
// pixel, 1 KiB row segments, three dense per-tile planes)?  Every variant is timed per launch with the dispatch's own
// start/stop events (hipExtLaunchKernelGGL), at a cache-resident footprint (F = 1 frame, 133 MB) and at an HBM footprint
// (F = 8 frames rotating, 1.06 GB).  No arithmetic that matters: this is the ceiling for ANY level-0 kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int W = 3584, H = 2048, T = 512;     // 7 x 4 full tiles
constexpr int TX = W / T, TY = H / T, NT = TX * TY;
constexpr size_t PX = (size_t)W * H;

// ---- P0: plain copy, 16 B per lane, grid-stride, U loads in flight -----------------------------------------
template <int U, bool NTS>
__global__ __launch_bounds__(256) void copy_k(const v4i* __restrict__ s, v4i* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x, st = (size_t)gridDim.x * 256 * U;
  for (; i + 256 * (U - 1) < n; i += st) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = s[i + 256 * u];
#pragma unroll
    for (int u = 0; u < U; u++) { if (NTS) __builtin_nontemporal_store(v[u], &d[i + 256 * u]); else d[i + 256 * u] = v[u]; }
  }
}
// ---- P1: write only / read only -----------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void fill_k(v4i* __restrict__ d, size_t n, int val) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x, st = (size_t)gridDim.x * 256 * U;
  const v4i v = {val, val + 1, val + 2, val + 3};
  for (; i + 256 * (U - 1) < n; i += st) {
#pragma unroll
    for (int u = 0; u < U; u++) d[i + 256 * u] = v;
  }
}
template <int U>
__global__ __launch_bounds__(256) void read_k(const v4i* __restrict__ s, int* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x, st = (size_t)gridDim.x * 256 * U;
  v4i acc = {0, 0, 0, 0};
  for (; i + 256 * (U - 1) < n; i += st) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = s[i + 256 * u];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[0] = 1;
}
// ---- P2: 1:3 mix, contiguous elementwise (unpack RGBA8 -> three int32 planes of the whole frame) ---------------
template <int U>
__global__ __launch_bounds__(256) void mix_k(const v4i* __restrict__ s, v4i* __restrict__ d, size_t n4) {
  // n4 = pixels / 4 ; plane k at d + k * n4
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x, st = (size_t)gridDim.x * 256 * U;
  for (; i + 256 * (U - 1) < n4; i += st) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = s[i + 256 * u];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int k = 0; k < 3; k++) d[(size_t)k * n4 + i + 256 * u] = (v[u] >> (8 * k)) & 255;
  }
}
// ---- P3: the tile shape.  One wave = R consecutive source rows of one 512-wide tile (8 px per lane = two 16-B loads per
// row), all 2R loads issued first; stores de-interleaved: source row r -> row (r>>1) or 256 + (r>>1) of each of the three
// dense 512 x 512 planes of the tile, low half = even pixels, high half = odd pixels (1 KiB segments, 2 KiB per row).
template <int R, bool NTS, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void tile_k(const int* __restrict__ src, int* __restrict__ dst, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  constexpr int BPT = T / R;                      // bands per tile
  const int t = wave / BPT, b = wave % BPT;
  const int tx = t % TX, ty = t / TX;
  v4i a[R], c[R];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const v4i* p = (const v4i*)(src + ((size_t)ty * T + b * R + i) * W + tx * T + lane * 8);
    a[i] = p[0]; c[i] = p[1];
  }
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = b * R + i;
    const int ro = (r & 1) ? T / 2 + (r >> 1) : (r >> 1);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      int* q = dst + ((size_t)(t * 3 + k) * T + ro) * T + lane * 4;
      const v4i lo = {(a[i].x >> (8 * k)) & 255, (a[i].z >> (8 * k)) & 255, (c[i].x >> (8 * k)) & 255, (c[i].z >> (8 * k)) & 255};
      const v4i hi = {(a[i].y >> (8 * k)) & 255, (a[i].w >> (8 * k)) & 255, (c[i].y >> (8 * k)) & 255, (c[i].w >> (8 * k)) & 255};
      if (NTS) { __builtin_nontemporal_store(lo, (v4i*)q); __builtin_nontemporal_store(hi, (v4i*)(q + T / 2)); }
      else { *(v4i*)q = lo; *(v4i*)(q + T / 2) = hi; }
    }
  }
}
// ---- P4: marching wave with a prefetch ring: band of B pair-rows, D pair-rows of packed pixels in flight ------------
template <int D, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void march_k(const int* __restrict__ src, int* __restrict__ dst, int band, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int bpt = (T / 2 + band - 1) / band;
  const int t = wave / bpt, b = wave % bpt;
  const int tx = t % TX, ty = t / TX;
  const int q0 = b * band, q1 = min(q0 + band, T / 2);
  const int* base = src + ((size_t)ty * T) * W + tx * T + lane * 8;
  v4i ring[D][4];
#pragma unroll
  for (int d = 0; d < D; d++) {
    const int q = min(q0 + d, T / 2 - 1);
    const v4i* p0 = (const v4i*)(base + (size_t)(2 * q) * W); const v4i* p1 = (const v4i*)(base + (size_t)(2 * q + 1) * W);
    ring[d][0] = p0[0]; ring[d][1] = p0[1]; ring[d][2] = p1[0]; ring[d][3] = p1[1];
  }
  v4i carry = {0, 0, 0, 0};
  for (int q = q0; q < q1; q += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      if (q + d < q1) {
        const v4i x0 = ring[d][0], x1 = ring[d][1], x2 = ring[d][2], x3 = ring[d][3];
        {   // refill this slot before the stores of this pair-row
          const int qn = min(q + d + D, T / 2 - 1);
          const v4i* p0 = (const v4i*)(base + (size_t)(2 * qn) * W); const v4i* p1 = (const v4i*)(base + (size_t)(2 * qn + 1) * W);
          ring[d][0] = p0[0]; ring[d][1] = p0[1]; ring[d][2] = p1[0]; ring[d][3] = p1[1];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          int* ql = dst + ((size_t)(t * 3 + k) * T + (q + d)) * T + lane * 4;
          int* qh = dst + ((size_t)(t * 3 + k) * T + T / 2 + (q + d)) * T + lane * 4;
          const v4i lo0 = {(x0.x >> (8 * k)) & 255, (x0.z >> (8 * k)) & 255, (x1.x >> (8 * k)) & 255, (x1.z >> (8 * k)) & 255};
          const v4i hi0 = {(x0.y >> (8 * k)) & 255, (x0.w >> (8 * k)) & 255, (x1.y >> (8 * k)) & 255, (x1.w >> (8 * k)) & 255};
          const v4i lo1 = {(x2.x >> (8 * k)) & 255, (x2.z >> (8 * k)) & 255, (x3.x >> (8 * k)) & 255, (x3.z >> (8 * k)) & 255};
          const v4i hi1 = {(x2.y >> (8 * k)) & 255, (x2.w >> (8 * k)) & 255, (x3.y >> (8 * k)) & 255, (x3.w >> (8 * k)) & 255};
          *(v4i*)ql = lo0 + carry; *(v4i*)(ql + T / 2) = hi0;
          *(v4i*)qh = lo1 - lo0; *(v4i*)(qh + T / 2) = hi1 - hi0;
          carry = lo1;
        }
      }
    }
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
};

int main(int argc, char** argv) {
  const int FMAX = 8;
  int *src, *dst, *sink, *cbuf;
  CK(hipMalloc(&src, PX * 4 * FMAX)); CK(hipMalloc(&dst, PX * 12 * FMAX)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cbuf, PX * 8 * FMAX)); CK(hipMemset(cbuf, 3, PX * 8 * FMAX));
  {
    std::vector<int> h(PX);
    for (size_t i = 0; i < PX; i++) h[i] = (int)((i * 2654435761u) >> 3);
    for (int f = 0; f < FMAX; f++) CK(hipMemcpy(src + f * PX, h.data(), PX * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemset(dst, 0, PX * 12 * FMAX));
  const int IT = 24;
  std::vector<Timer> ev(IT);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto report = [&](const std::string& name, int F, double bytes, auto launch) {
    for (int i = 0; i < 4; i++) launch(i % F, (hipEvent_t) nullptr, (hipEvent_t) nullptr);
    for (int i = 0; i < IT; i++) launch(i % F, ev[i].a, ev[i].b);
    CK(hipStreamSynchronize(st));
    std::vector<float> us(IT);
    for (int i = 0; i < IT; i++) { float ms; CK(hipEventElapsedTime(&ms, ev[i].a, ev[i].b)); us[i] = ms * 1e3f; }
    std::sort(us.begin(), us.end());
    double avg = 0; for (float u : us) avg += u; avg /= IT;
    printf("%-44s F=%d  avg %7.2f us  med %7.2f  min %7.2f  -> %6.0f GB/s (avg)  %6.0f (min)\n", name.c_str(), F, avg, us[IT / 2], us[0], bytes / avg / 1e3, bytes / us[0] / 1e3);
    fflush(stdout);
  };
#define LAUNCH(kern, grid, ...) [&](int f, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a, b, 0, __VA_ARGS__); }
  const size_t n16 = PX * 16 / 16 / 2;   // copy: read 8 B/px-equivalent, write 8 B/px-equivalent = same 16 B/px total
  for (int F : {1, FMAX}) {
    printf("==== footprint: %d frame(s) = %.0f MB ====\n", F, F * PX * 16 / 1e6);
    // P0 copy (same total bytes as one level-0 launch: 16 B/px)
    for (int grid : {1024, 2048, 4096, 16384}) {
      report("copy U=1 grid=" + std::to_string(grid), F, PX * 16.0, LAUNCH((copy_k<1, false>), grid, (const v4i*)(cbuf + (size_t)f * PX * 2), (v4i*)(dst + (size_t)f * PX * 3), n16));
      report("copy U=4 grid=" + std::to_string(grid), F, PX * 16.0, LAUNCH((copy_k<4, false>), grid, (const v4i*)(cbuf + (size_t)f * PX * 2), (v4i*)(dst + (size_t)f * PX * 3), n16));
    }
    report("copy U=8 grid=2048", F, PX * 16.0, LAUNCH((copy_k<8, false>), 2048, (const v4i*)(cbuf + (size_t)f * PX * 2), (v4i*)(dst + (size_t)f * PX * 3), n16));
    report("copy U=4 grid=2048 nt-store", F, PX * 16.0, LAUNCH((copy_k<4, true>), 2048, (const v4i*)(cbuf + (size_t)f * PX * 2), (v4i*)(dst + (size_t)f * PX * 3), n16));
    // P1
    for (int grid : {2048, 8192}) {
      report("fill 12 B/px U=4 grid=" + std::to_string(grid), F, PX * 12.0, LAUNCH((fill_k<4>), grid, (v4i*)(dst + (size_t)f * PX * 3), PX * 12 / 16, f));
      report("read 12 B/px U=4 grid=" + std::to_string(grid), F, PX * 12.0, LAUNCH((read_k<4>), grid, (const v4i*)(dst + (size_t)f * PX * 3), sink, PX * 12 / 16));
    }
    // P2
    for (int grid : {2048, 8192}) {
      report("mix 4:12 contiguous U=1 grid=" + std::to_string(grid), F, PX * 16.0, LAUNCH((mix_k<1>), grid, (const v4i*)(src + (size_t)f * PX), (v4i*)(dst + (size_t)f * PX * 3), PX / 4));
      report("mix 4:12 contiguous U=4 grid=" + std::to_string(grid), F, PX * 16.0, LAUNCH((mix_k<4>), grid, (const v4i*)(src + (size_t)f * PX), (v4i*)(dst + (size_t)f * PX * 3), PX / 4));
    }
    // P3
#define TILE(R, NTS, WPE) report("tile R=" #R " nts=" #NTS " wpe=" #WPE, F, PX * 16.0, LAUNCH((tile_k<R, NTS, WPE>), (NT * (T / R) + 3) / 4, src + (size_t)f * PX, dst + (size_t)f * PX * 3, NT * (T / R)))
    TILE(1, false, 8); TILE(2, false, 8); TILE(4, false, 8); TILE(8, false, 8); TILE(8, false, 4); TILE(16, false, 4); TILE(4, true, 8); TILE(2, false, 4); TILE(4, false, 4); TILE(4, false, 2);
    // P4
#define MARCH(D, WPE, BAND) report("march D=" #D " wpe=" #WPE " band=" #BAND, F, PX * 16.0, LAUNCH((march_k<D, WPE>), (NT * ((T / 2 + BAND - 1) / BAND) + 3) / 4, src + (size_t)f * PX, dst + (size_t)f * PX * 3, BAND, NT * ((T / 2 + BAND - 1) / BAND)))
    MARCH(1, 4, 3); MARCH(1, 4, 5); MARCH(1, 4, 8); MARCH(2, 4, 4); MARCH(2, 4, 8); MARCH(2, 4, 16); MARCH(4, 4, 8); MARCH(4, 4, 16); MARCH(2, 8, 8); MARCH(4, 8, 16); MARCH(2, 2, 8); MARCH(4, 2, 16); MARCH(4, 4, 32); MARCH(4, 4, 64);
  }
  return 0;
}
