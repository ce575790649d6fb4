// Probe 3: marching-wave copy -- does the ORDER in which wavefronts sweep the frame matter (DRAM page locality)?
// order 0: tile-major (all bands of tile 0, then tile 1, ...)   order 1: frame-row-major (band b of every tile in a tile row, then band b+1)
#include <hip/hip_runtime.h>
#include <cstdio>
#define W 3584
#define H 2048
#define T 512
typedef int v4i __attribute__((ext_vector_type(4)));
template <int R>
__global__ __launch_bounds__(256) void tile_copy(const int* __restrict__ src, int* __restrict__ dst, int band, int nwaves, int order) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int bands = T / band, TX = W / T;
  int t, b;
  if (order == 0) { t = wave / bands; b = wave % bands; }
  else { const int ty = wave / (bands * TX), rem = wave % (bands * TX); b = rem / TX; t = ty * TX + rem % TX; }
  const int tx = t % TX, ty = t / TX;
  for (int r0 = b * band; r0 < (b + 1) * band; r0 += R) {
    v4i v[R][3][2];
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const v4i* p = (const v4i*)(src + ((size_t)k * H + ty * T + r0 + i) * W + tx * T + lane * 8);
        v[i][k][0] = p[0]; v[i][k][1] = p[1];
      }
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = r0 + i, ro = (r & 1) ? T / 2 + (r >> 1) : (r >> 1);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        v4i* q = (v4i*)(dst + ((size_t)(t * 3 + k) * T + ro) * T + lane * 8);
        q[0] = v[i][k][0]; q[1] = v[i][k][1];
      }
    }
  }
}
template <int R>
void run(int* s, int* d, int band, int order) {
  const int F = 8; const size_t fsz = (size_t)3 * W * H;
  const int nw = (W / T) * (H / T) * (T / band);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto go = [&]() { for (int f = 0; f < F; f++) tile_copy<R><<<(nw + 3) / 4, 256>>>(s + f * fsz, d + f * fsz, band, nw, order); };
  go(); go();
  (void)hipEventRecord(e0);
  const int it = 5;
  for (int k = 0; k < it; k++) go();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("R=%d band=%3d waves=%6d order=%s : %.0f GB/s (%.1f us per 176 MB frame)\n", R, band, nw, order ? "frame-row-major" : "tile-major     ", 2.0 * fsz * 4 * F / (ms / it * 1e-3) / 1e9, ms / it / F * 1e3);
}
int main() {
  const size_t bytes = (size_t)3 * W * H * 4 * 8;
  int *s, *d; (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes); (void)hipMemset(s, 1, bytes); (void)hipMemset(d, 0, bytes);
  for (int order = 0; order < 2; order++) {
    run<1>(s, d, 1, order); run<2>(s, d, 2, order); run<2>(s, d, 4, order); run<2>(s, d, 8, order);
  }
  return 0;
}
