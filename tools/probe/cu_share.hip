// Probe: where do the workgroups of CONCURRENT small grids land, and what does a chain of dependent scalar instructions pay when
// another chain shares its CU / its SIMD?  (c1gpu: twenty-four frames in flight, each a grid of 21 workgroups whose one busy wave
// is an MQ chain; the kernel trace shows two populations of durations, 90 and 150 ms.)
//   K streams each launch a grid of G workgroups of 256 threads with `lds` bytes of LDS; in every workgroup ONE wave runs the
//   chain: wave 0 (mode 0) or the wave on the SIMD a per-CU counter names (mode 1); chain = SALU (kind 0) or VALU (kind 1).
//   Per workgroup: XCC, HW_ID of the chain wave, begin / end on the 100 MHz clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#define REP8(x) x x x x x x x x
#define HWREG(id) ((id) | 31 << 11)
__device__ uint32_t g_rot[4096];
struct Rec { uint32_t xcc, hw, simd_chain, pad; long long t0, t1; };
__global__ __launch_bounds__(256) void k(int mode, int kind, int iters, Rec *rec, uint32_t *sink) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t wsimd[4], slot;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg(HWREG(4)), xcc = (uint32_t)__builtin_amdgcn_s_getreg(HWREG(20)) & 15u;
  if (lane == 0) wsimd[wv] = (hw >> 4) & 3u;
  if (threadIdx.x == 0) { slot = mode ? atomicAdd(&g_rot[(xcc << 8 | ((hw >> 8) & 255u)) & 4095u], 1u) & 3u : 99u; lds[0] = hw; }
  __syncthreads();
  int mine = 0;
  for (int i = (int)(blockDim.x >> 6) - 1; i >= 0; i--) if (wsimd[i] == slot) mine = i;
  if (wv != __builtin_amdgcn_readfirstlane(mine)) return;
  uint32_t s = __builtin_amdgcn_readfirstlane(iters * 977 + 13), v = threadIdx.x * 977 + 13;
  const long long t0 = wall_clock64();
  if (kind == 0) for (int i = 0; i < iters; i++) asm volatile(REP8("s_add_u32 %0, %0, 0x1234567\n s_xor_b32 %0, %0, 0x55aa\n") : "+s"(s) : : "scc");
  else for (int i = 0; i < iters; i++) asm volatile(REP8("v_add_u32 %0, 0x1234567, %0\n v_xor_b32 %0, 0x55aa, %0\n") : "+v"(v));
  const long long t1 = wall_clock64();
  if (lane == 0) { Rec r; r.xcc = xcc; r.hw = hw; r.simd_chain = (hw >> 4) & 3u; r.pad = 0; r.t0 = t0; r.t1 = t1; rec[blockIdx.x] = r; sink[blockIdx.x] = s ^ v; }
}
int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 24, G = argc > 2 ? atoi(argv[2]) : 21, lds = argc > 3 ? atoi(argv[3]) : 60000, iters = argc > 4 ? atoi(argv[4]) : 200000;
  const int threads = argc > 5 ? atoi(argv[5]) : 256;     // 64: one wave per workgroup (no choice of SIMD: mode 1 = mode 0)
  Rec *rec; uint32_t *sink;
  (void)hipMalloc(&rec, sizeof(Rec) * K * G); (void)hipMalloc(&sink, 4 * K * G);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  std::vector<hipStream_t> st(K);
  for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int kind = 0; kind < (getenv("CU_SHARE_GRIDS") ? 1 : 2); kind++)
    for (int mode = 0; mode < 2; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(k, dim3(G), dim3(threads), lds, st[i], threads == 64 ? 0 : mode, kind, iters, rec + i * G, sink + i * G);
        (void)hipDeviceSynchronize();
      }
      std::vector<Rec> h(K * G);
      (void)hipMemcpy(h.data(), rec, sizeof(Rec) * K * G, hipMemcpyDeviceToHost);
      // co-residents: workgroups on the same CU (same SIMD) whose chains overlap mine for more than half of my time
      std::map<int, std::vector<double>> by_same_simd, by_same_cu;
      std::map<uint32_t, int> cu_used;
      double lo = 1e30, hi = 0;
      for (int a = 0; a < K * G; a++) {
        const uint32_t cua = h[a].xcc << 8 | ((h[a].hw >> 8) & 255u);
        cu_used[cua]++;
        int nc = 0, ns = 0;
        for (int b = 0; b < K * G; b++) {
          if (a == b) continue;
          const uint32_t cub = h[b].xcc << 8 | ((h[b].hw >> 8) & 255u);
          if (cua != cub) continue;
          const long long ov = std::min(h[a].t1, h[b].t1) - std::max(h[a].t0, h[b].t0);
          if (ov * 2 < h[a].t1 - h[a].t0) continue;
          nc++;
          if (h[a].simd_chain == h[b].simd_chain) ns++;
        }
        const double ms = (h[a].t1 - h[a].t0) * 1e-5;
        by_same_cu[nc].push_back(ms); by_same_simd[ns].push_back(ms);
        lo = std::min(lo, ms); hi = std::max(hi, ms);
      }
      printf("%s chain, %s: %d grids x %d workgroups, %d bytes LDS: %zu CUs used, chain %.2f .. %.2f ms (%.2f ns per op alone)\n", kind ? "VALU" : "SALU",
             mode ? "elected SIMD" : "wave 0", K, G, lds, cu_used.size(), lo, hi, lo * 1e6 / (iters * 16.0));
      if (getenv("CU_SHARE_GRIDS")) {
        long long tb = h[0].t0; for (auto &r : h) tb = std::min(tb, r.t0);
        for (int i = 0; i < K; i++) {
          long long a = h[i * G].t0, b = h[i * G].t1; double m = 0;
          for (int j = 0; j < G; j++) { a = std::min(a, h[i * G + j].t0); b = std::max(b, h[i * G + j].t1); m += (h[i * G + j].t1 - h[i * G + j].t0) * 1e-5 / G; }
          printf("   grid %2d: begins %8.2f ms, ends %8.2f ms, mean chain %7.2f ms\n", i, (a - tb) * 1e-5, (b - tb) * 1e-5, m);
        }
      }
      for (auto &kv : by_same_cu) { double s = 0; for (double x : kv.second) s += x; printf("   %d other chains on my CU: %4zu workgroups, mean %.2f ms\n", kv.first, kv.second.size(), s / kv.second.size()); }
      for (auto &kv : by_same_simd) { double s = 0; for (double x : kv.second) s += x; printf("   %d other chains on my SIMD: %4zu workgroups, mean %.2f ms\n", kv.first, kv.second.size(), s / kv.second.size()); }
      fflush(stdout);
    }
  return 0;
}
