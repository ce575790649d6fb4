// Practical HBM roofline on this box: device-to-device copy at the bench's footprint (100 MB -> 100 MB) and larger.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void copy4(const int4* __restrict__ s, int4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) d[i] = s[i];
}
__global__ __launch_bounds__(256) void copy4nt(const int4* __restrict__ s, int4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i* ss = reinterpret_cast<const v4i*>(s); v4i* dd = reinterpret_cast<v4i*>(d);
  for (; i < n; i += st) { v4i v = __builtin_nontemporal_load(&ss[i]); __builtin_nontemporal_store(v, &dd[i]); }
}
int main() {
  for (size_t mb : {100, 400, 1600}) {
    size_t bytes = mb << 20, n = bytes / 16;
    int4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; nt++) for (int grid : {2048, 8192, 65536}) {
      for (int w = 0; w < 3; w++) { if (nt) copy4nt<<<grid, 256>>>(s, d, n); else copy4<<<grid, 256>>>(s, d, n); }
      hipEventRecord(e0);
      const int it = 20;
      for (int k = 0; k < it; k++) { if (nt) copy4nt<<<grid, 256>>>(s, d, n); else copy4<<<grid, 256>>>(s, d, n); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("copy %4zu MB -> %4zu MB  nt=%d grid=%6d : %.1f us  %.0f GB/s (read+write)\n", mb, mb, nt, grid, ms / it * 1e3, 2.0 * bytes / (ms / it * 1e-3) / 1e9);
    }
    hipFree(s); hipFree(d);
  }
  return 0;
}
