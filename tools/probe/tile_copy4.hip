// Probe 4: the packed-pixel level-0 shape -- read 4 B/px (one RGBA8 dword), write 12 B/px (three int32 planes,
// de-interleaved rows), marching wavefronts, band of 10 rows.  What does the memory system allow for 16 B/px?
#include <hip/hip_runtime.h>
#include <cstdio>
#define W 3584
#define H 2048
#define T 512
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const int* __restrict__ src, int* __restrict__ dst, int band, int nwaves) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const int bands = (T + band - 1) / band, TX = W / T;
  const int t = wave / bands, b = wave % bands;
  const int tx = t % TX, ty = t / TX;
  for (int r = b * band; r < min((b + 1) * band, T); r++) {
    const v4i* p = (const v4i*)(src + ((size_t)ty * T + r) * W + tx * T + lane * 8);
    const v4i a = p[0], c = p[1];
    const int ro = (r & 1) ? T / 2 + (r >> 1) : (r >> 1);
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      v4i* q = (v4i*)(dst + ((size_t)(t * 3 + kk) * T + ro) * T + lane * 8);
      v4i x = {(a.x >> (8 * kk)) & 255, (a.y >> (8 * kk)) & 255, (a.z >> (8 * kk)) & 255, (a.w >> (8 * kk)) & 255};
      v4i y = {(c.x >> (8 * kk)) & 255, (c.y >> (8 * kk)) & 255, (c.z >> (8 * kk)) & 255, (c.w >> (8 * kk)) & 255};
      q[0] = x; q[1] = y;
    }
  }
}
int main() {
  const int F = 8;
  const size_t px = (size_t)W * H;
  int *s, *d; (void)hipMalloc(&s, px * 4 * F); (void)hipMalloc(&d, px * 12 * F); (void)hipMemset(s, 7, px * 4 * F); (void)hipMemset(d, 0, px * 12 * F);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int band : {4, 6, 10, 16}) {
    const int nw = (W / T) * (H / T) * ((T + band - 1) / band);
    auto go = [&]() { for (int f = 0; f < F; f++) k<<<(nw + 3) / 4, 256>>>(s + f * px, d + f * px * 3, band, nw); };
    go(); go();
    (void)hipEventRecord(e0);
    const int it = 5;
    for (int i = 0; i < it; i++) go();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / it / F * 1e3;
    printf("band=%2d waves=%5d : %.1f us per frame (%.1f MB: %.0f GB/s)  -> a 3840x2160 frame: %.1f us\n", band, nw, us, px * 16 / 1e6, px * 16 / us / 1e3,
           us * (3840.0 * 2160) / px);
  }
  return 0;
}
