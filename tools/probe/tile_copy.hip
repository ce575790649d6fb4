// Probe: does the DWT level-0 ACCESS PATTERN (2 KB row chunks of a 512-wide tile inside a 3840-wide frame, rows
// de-interleaved at the store) cost HBM bandwidth compared with a linear copy of the same bytes?
// One wavefront = one tile x band of BAND rows x 3 components, 32 B per lane per row (like CPL=8).
#include <hip/hip_runtime.h>
#include <cstdio>
#define W 3584
#define H 2048
#define T 512
__global__ __launch_bounds__(256) void tile_copy(const int* __restrict__ src, int* __restrict__ dst, int mode, int band, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int bands = T / band, tiles = (W / T) * (H / T);
  const int f = wave / (tiles * bands), t = (wave / bands) % tiles, b = wave % bands;
  const int tx = t % (W / T), ty = t / (W / T);
  const size_t fsz = (size_t)3 * W * H;
  for (int r = b * band; r < (b + 1) * band; r++) {
    int4 v[3][2];
    for (int k = 0; k < 3; k++) {
      const int* p = (mode & 2) ? src + f * fsz + ((size_t)(t * 3 + k) * T + r) * T + lane * 8
                                : src + f * fsz + ((size_t)k * H + ty * T + r) * W + tx * T + lane * 8;
      v[k][0] = *(const int4*)p; v[k][1] = *(const int4*)(p + 4);
    }
    const int ro = (mode & 1) ? r : ((r & 1) ? T / 2 + (r >> 1) : (r >> 1));
    for (int k = 0; k < 3; k++) {
      int* q = dst + f * fsz + ((size_t)(t * 3 + k) * T + ro) * T + lane * 8;
      *(int4*)q = v[k][0]; *(int4*)(q + 4) = v[k][1];
    }
  }
}
__global__ __launch_bounds__(256) void copy4(const int4* __restrict__ s, int4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) d[i] = s[i];
}
int main() {
  const int F = 8;
  const size_t fsz = (size_t)3 * W * H, bytes = fsz * 4 * F;
  int *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 5;
  for (int w = 0; w < 2; w++) copy4<<<65536, 256>>>((const int4*)s, (int4*)d, bytes / 16);
  hipEventRecord(e0);
  for (int k = 0; k < it; k++) copy4<<<65536, 256>>>((const int4*)s, (int4*)d, bytes / 16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("linear copy, %zu MB footprint: %.0f GB/s\n", 2 * bytes >> 20, 2.0 * bytes / (ms / it * 1e-3) / 1e9);
  for (int perframe = 0; perframe < 2; perframe++)
  for (int band : {8, 16, 64})
    for (int mode = 0; mode < 4; mode++) {
      const int nw1 = (W / T) * (H / T) * (T / band), nw = perframe ? nw1 : nw1 * F;
      auto go = [&]() { if (perframe) { for (int f = 0; f < F; f++) tile_copy<<<(nw + 3) / 4, 256>>>(s + f * fsz, d + f * fsz, mode, band, nw); }
                        else tile_copy<<<(nw + 3) / 4, 256>>>(s, d, mode, band, nw); };
      go(); go();
      hipEventRecord(e0);
      for (int k = 0; k < it; k++) go();
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s band=%2d src=%s dst=%s : %.0f GB/s  (%.1f us per 88 MB frame)\n", perframe ? "launch/frame" : "one launch  ", band, (mode & 2) ? "tile-planar " : "frame-strided",
             (mode & 1) ? "linear rows " : "deinterleaved", 2.0 * bytes / (ms / it * 1e-3) / 1e9, ms / it / F * 1e3);
    }
  return 0;
}
