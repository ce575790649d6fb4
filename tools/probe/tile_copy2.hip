// Probe 2: marching-wave copy (the DWT level-0 decomposition) -- how do rows in flight per wave (R), non-temporal
// access and band height change the achieved HBM rate?  Footprint 1.3 GB (8 frames), one launch per frame.
#include <hip/hip_runtime.h>
#include <cstdio>
#define W 3584
#define H 2048
#define T 512
typedef int v4i __attribute__((ext_vector_type(4)));
template <int R, int NT>
__global__ __launch_bounds__(256) void tile_copy(const int* __restrict__ src, int* __restrict__ dst, int band, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int bands = T / band, tiles = (W / T) * (H / T);
  const int t = (wave / bands) % tiles, b = wave % bands;
  const int tx = t % (W / T), ty = t / (W / T);
  for (int r0 = b * band; r0 < (b + 1) * band; r0 += R) {
    v4i v[R][3][2];
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const v4i* p = (const v4i*)(src + ((size_t)k * H + ty * T + r0 + i) * W + tx * T + lane * 8);
        if (NT & 1) { v[i][k][0] = __builtin_nontemporal_load(p); v[i][k][1] = __builtin_nontemporal_load(p + 1); }
        else { v[i][k][0] = p[0]; v[i][k][1] = p[1]; }
      }
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = r0 + i, ro = (r & 1) ? T / 2 + (r >> 1) : (r >> 1);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        v4i* q = (v4i*)(dst + ((size_t)(t * 3 + k) * T + ro) * T + lane * 8);
        if (NT & 2) { __builtin_nontemporal_store(v[i][k][0], q); __builtin_nontemporal_store(v[i][k][1], q + 1); }
        else { q[0] = v[i][k][0]; q[1] = v[i][k][1]; }
      }
    }
  }
}
template <int R, int NT>
void run(int* s, int* d, int band) {
  const int F = 8; const size_t fsz = (size_t)3 * W * H;
  const int nw = (W / T) * (H / T) * (T / band);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto go = [&]() { for (int f = 0; f < F; f++) tile_copy<R, NT><<<(nw + 3) / 4, 256>>>(s + f * fsz, d + f * fsz, band, nw); };
  go(); go();
  (void)hipEventRecord(e0);
  const int it = 5;
  for (int k = 0; k < it; k++) go();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("R=%d NT=%d band=%3d waves=%5d : %.0f GB/s (%.1f us per 176 MB frame)\n", R, NT, band, nw, 2.0 * fsz * 4 * F / (ms / it * 1e-3) / 1e9, ms / it / F * 1e3);
}
int main() {
  const size_t bytes = (size_t)3 * W * H * 4 * 8;
  int *s, *d; (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes); (void)hipMemset(s, 1, bytes); (void)hipMemset(d, 0, bytes);
  for (int band : {4, 8, 16, 32}) {
    run<1, 0>(s, d, band); run<2, 0>(s, d, band); run<4, 0>(s, d, band);
    run<2, 1>(s, d, band); run<2, 2>(s, d, band); run<2, 3>(s, d, band); run<4, 3>(s, d, band);
  }
  return 0;
}
