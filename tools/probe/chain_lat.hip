// Probe: cost of dependent chains for ONE wavefront (what bounds the HT VLC walk): dependent VALU ops,
// dependent LDS reads (b32 / u16, random addresses), a taken branch per iteration, scattered global stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void k(int mode, int iters, uint32_t* out, long long* cyc) {
  __shared__ uint32_t tab[8192];
  __shared__ uint16_t tab16[16384];
  for (int i = threadIdx.x; i < 8192; i += 64) tab[i] = (i * 2654435761u) >> 7;
  for (int i = threadIdx.x; i < 16384; i += 64) tab16[i] = (uint16_t)((i * 2654435761u) >> 11);
  __syncthreads();
  uint32_t x = threadIdx.x * 977 + 13;
  long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) { for (int i = 0; i < iters; i++) { x = x * 3 + 1; x ^= x >> 3; x += 7; x ^= x << 2; } }                 // 6-7 dependent VALU
  if (mode == 1) { for (int i = 0; i < iters; i++) { x = tab[x & 8191]; } }                                                // dependent b32 LDS
  if (mode == 2) { for (int i = 0; i < iters; i++) { x = tab16[x & 16383] + i; } }                                         // dependent u16 LDS
  if (mode == 3) { for (int i = 0; i < iters; i++) { x = tab[x & 8191]; x = tab16[(x >> 3) & 16383] + i; } }               // two dependent LDS
  if (mode == 4) { for (int i = 0; i < iters; i++) { x = x * 3 + 1; if (x & 0x80000000u) x = tab[x & 8191]; x ^= x >> 3; } } // divergent branch
  if (mode == 5) { for (int i = 0; i < iters; i++) { x = x * 3 + 1; x ^= x >> 3; out[(size_t)threadIdx.x * 4096 + (i & 1023)] = x; } } // scattered store
  if (mode == 6) { uint64_t y = x; for (int i = 0; i < iters; i++) { y = (y >> (y & 7)) + 0x9E3779B97F4A7C15ull; y = (y << (i & 3)) ^ (y >> 11); } x = (uint32_t)y; } // 64-bit shifts
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[mode] = t1 - t0;
}
int main() {
  uint32_t* out; long long* cyc; (void)hipMalloc(&out, 64 * 4096 * 4 + 4096); (void)hipMalloc(&cyc, 64);
  const char* names[] = {"7 dependent VALU", "dependent ds_read_b32", "dependent ds_read_u16", "b32 then u16 LDS", "VALU + divergent branch w/ LDS", "3 VALU + scattered global store", "64-bit variable shifts x4"};
  for (int m = 0; m < 7; m++) {
    const int iters = 2000;
    k<<<1, 64>>>(m, iters, out, cyc); k<<<1, 64>>>(m, iters, out, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc + m, 8, hipMemcpyDeviceToHost);
    printf("%-36s : %.1f memtime ticks / iteration (100 MHz ticks? x24 = cycles: %.0f)\n", names[m], (double)c / iters, (double)c / iters);
  }
  return 0;
}
