// Probe: what ONE wavefront pays per dependent instruction on the scalar unit, the vector unit and the paths between them
// (v_readlane / v_writelane / s_load / LDS through readfirstlane) -- the budget of a code-block coder that runs one MQ chain
// per wavefront with wave-uniform control (t1_big: blocks above 64 x 64).  ns per op from the 100 MHz wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
__global__ __launch_bounds__(64) void k(int mode, int iters, uint32_t* out, const uint32_t* __restrict__ ctab, double* ns) {
  __shared__ uint32_t tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) tab[i] = (i * 2654435761u) >> 7;
  __syncthreads();
  uint32_t vt = (threadIdx.x * 2654435761u) >> 9;     // a table across the lanes
  uint32_t s = __builtin_amdgcn_readfirstlane(iters * 977 + 13);
  uint32_t v = threadIdx.x * 977 + 13;
  uint64_t m = 0x123456789abcdef1ull ^ (uint64_t)iters << 40;
  m = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)m);
  long long t0 = wall_clock64();
  int ops = 8;
  if (mode == 0) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_add_u32 %0, %0, 0x1234567\n s_xor_b32 %0, %0, 0x55aa\n") : "+s"(s) : : "scc"); ops = 16; }
  if (mode == 1) for (int i = 0; i < iters; i++) { asm volatile(REP8("v_add_u32 %0, 0x1234567, %0\n v_xor_b32 %0, 0x55aa, %0\n") : "+v"(v)); ops = 16; }
  if (mode == 2) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_and_b32 %0, %0, 63\n s_nop 0\n v_readlane_b32 %0, %1, %0\n") : "+s"(s) : "v"(vt) : "scc"); ops = 8; }
  if (mode == 3) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_and_b32 m0, %0, 63\n s_add_u32 %0, %0, 0x9e37\n v_writelane_b32 %1, %0, m0\n s_nop 1\n v_readlane_b32 %0, %1, m0\n") : "+s"(s), "+v"(vt) : : "m0", "scc"); ops = 8; }
  if (mode == 4) { const uint32_t* p = ctab; for (int i = 0; i < iters; i++) { asm volatile(REP8("s_and_b32 %0, %0, 0xffc\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+s"(s) : "s"(p) : "scc"); ops = 8; } }
  if (mode == 5) for (int i = 0; i < iters; i++) { uint32_t a;
#pragma unroll
    for (int r = 0; r < 8; r++) { a = tab[s & 4095]; s = __builtin_amdgcn_readfirstlane(a); } ops = 8; }
  if (mode == 6) for (int i = 0; i < iters; i++) { uint32_t b;
      asm volatile(REP8("s_ff1_i32_b64 %1, %0\n s_lshl_b64 %0, %0, 1\n s_lshl_b64 %0, %0, %1\n s_or_b64 %0, %0, 0x5\n") : "+s"(m), "=&s"(b) : : "scc"); ops = 32; s ^= b; }
  if (mode == 7) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n s_add_u32 %0, %0, 3\n") : "+s"(s) : : "scc"); ops = 8; }   // taken branch each time
  if (mode == 8) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n s_add_u32 %0, %0, 3\n") : "+s"(s) : : "scc"); ops = 8; }   // not-taken branch
  if (mode == 9) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_lshr_b32 %1, %0, 3\n s_and_b32 %1, %1, 7\n s_bfe_u32 %0, %0, 0x100005\n s_add_u32 %0, %0, %1\n s_mul_i32 %0, %0, 0x9e3779b1\n") : "+s"(s), "+s"(ops) : : "scc"); ops = 40; }
  if (mode == 10) for (int i = 0; i < iters; i++) { asm volatile(REP8("v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 5\n v_mov_b32 %1, %0\n") : "+s"(s), "+v"(v) : : "scc"); ops = 8; }   // s->v->s round trip
  if (mode == 11) for (int i = 0; i < iters; i++) { asm volatile(REP8("s_add_u32 %0, %0, 7\n v_add_u32 %1, 3, %1\n s_xor_b32 %0, %0, 0x11\n v_xor_b32 %1, 5, %1\n") : "+s"(s), "+v"(v) : : "scc"); ops = 32; }   // independent s and v streams interleaved
  long long t1 = wall_clock64();
  out[threadIdx.x] = v ^ s ^ vt ^ (uint32_t)m ^ (uint32_t)(m >> 32);
  if (threadIdx.x == 0) ns[mode] = (double)(t1 - t0) * 10.0 / ((double)iters * ops);
}
int main() {
  uint32_t* out; uint32_t* ctab; double* ns; (void)hipMalloc(&out, 4096); (void)hipMalloc(&ns, 256); (void)hipMalloc(&ctab, 4096 * 4);
  uint32_t h[1024]; for (int i = 0; i < 1024; i++) h[i] = ((i * 2654435761u) >> 5); (void)hipMemcpy(ctab, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[] = {"dependent SALU op (s_add/s_xor)", "dependent VALU op (v_add/v_xor)", "s_and + v_readlane(s idx) chain (per lookup)", "v_writelane + v_readlane via m0 (per update)",
                         "dependent s_load_dword (scalar cache, per load)", "LDS lookup via readfirstlane (per lookup)", "s_ff1_b64/s_lshl_b64/s_xor/s_or (per op)", "taken scalar branch (per cmp+branch+add)",
                         "not-taken scalar branch (per cmp+branch+2 adds)", "s_lshr/and/bfe/add/mul chain (per op)", "readfirstlane + s_add + v_mov round trip", "independent SALU + VALU interleaved (per op)"};
  for (int mth = 0; mth < 12; mth++) {
    const int iters = 4000;
    k<<<1, 64>>>(mth, iters, out, ctab, ns); k<<<1, 64>>>(mth, iters, out, ctab, ns);
    (void)hipDeviceSynchronize();
    double c; (void)hipMemcpy(&c, ns + mth, 8, hipMemcpyDeviceToHost);
    printf("%-52s : %6.2f ns\n", names[mth], c); fflush(stdout);
  }
  return 0;
}
