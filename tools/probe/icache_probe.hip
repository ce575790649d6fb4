// icache_probe.hip -- what does a wave pay for code it executes for the first time?  (gfx950)
// A kernel whose body is N straight-line VALU instructions (independent adds, no memory), executed TWICE in a loop by every
// wave; wave 0 of each workgroup stamps the constant 100 MHz clock around each pass.  Pass 1 pays the instruction fetch
// (cold instruction cache), pass 2 does not.  Grid: one 1024-thread workgroup per CU (240 workgroups), as the deep 5-3 kernel.
// Before each timed launch a "polluter" kernel with a different large body runs, as the other kernels of a frame would.
// build: hipcc -O3 --offload-arch=gfx950 -o icache_probe icache_probe.hip ; run: ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP256(x) REP16(REP16(x))
#define REP1024(x) REP4(REP256(x))
template <int VARIANT>
__global__ __launch_bounds__(1024) void body_kernel(unsigned long long *stamps, int *out, int passes) {
    int a = threadIdx.x, b = blockIdx.x, c = 3, d = 7;
    for (int p = 0; p < passes; p++) {
        const unsigned long long t0 = wall_clock64();
        // 4096 instructions = 16 KB (4 bytes each... v_add_u32 e32), four independent chains
        REP1024(asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) { stamps[(blockIdx.x * 4 + p) * 2] = t0; stamps[(blockIdx.x * 4 + p) * 2 + 1] = t1; }
        if (VARIANT == 1) { a ^= p; }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d;
}
int main() {
    const int G = 240;
    unsigned long long *st; int *out;
    hipMalloc(&st, G * 8 * sizeof(unsigned long long)); hipMalloc(&out, G * 1024 * sizeof(int));
    std::vector<unsigned long long> h(G * 8);
    for (int it = 0; it < 4; it++) {
        hipMemset(st, 0, G * 8 * sizeof(unsigned long long));
        hipLaunchKernelGGL(body_kernel<1>, dim3(G), dim3(1024), 0, 0, st, out, 1);      // polluter: another 16 KB body
        hipLaunchKernelGGL(body_kernel<0>, dim3(G), dim3(1024), 0, 0, st, out, 2);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, G * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<long> p0, p1;
        for (int g = 0; g < G; g++) { p0.push_back((long)(h[(g * 4 + 0) * 2 + 1] - h[(g * 4 + 0) * 2])); p1.push_back((long)(h[(g * 4 + 1) * 2 + 1] - h[(g * 4 + 1) * 2])); }
        std::sort(p0.begin(), p0.end()); std::sort(p1.begin(), p1.end());
        printf("launch %d: 16 KB of straight-line VALU code, 16 waves per CU: first pass min/med/max %ld/%ld/%ld ticks (10 ns), second pass %ld/%ld/%ld\n", it,
               p0.front(), p0[G / 2], p0.back(), p1.front(), p1[G / 2], p1.back());
    }
    return 0;
}
