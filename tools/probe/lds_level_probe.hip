// lds_level_probe.hip -- how long does ONE LDS-resident 5-3 level take in a 16-wave workgroup, by itself?
// Includes the product source (device routines tail_fwd_level / tail_inv_level) and times them with the constant 100 MHz
// clock: REPS back-to-back runs of the same level per workgroup, grid = 1 or 240 workgroups.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../go-jpeg2000_amd/csrc -o lds_level_probe lds_level_probe.hip
#include "../../go-jpeg2000_amd/csrc/dwt53.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace j2k;
#define REPS 6
__global__ __launch_bounds__(1024) void probe_kernel(int32_t *gout, unsigned long long *st, int w, int h, int dir, int nostore) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    int32_t *bufA = lds, *bufB = lds + ((w * h + 3) & ~3), *bufC = bufB + ((w * h + 3) & ~3);
    for (int i = threadIdx.x; i < w * h; i += 1024) { bufA[i] = i * 7 + blockIdx.x; bufC[i] = i * 3; }
    __syncthreads();
    const int wn = (w + 1) >> 1, hn = (h + 1) >> 1;
    int32_t *g = gout + (size_t)blockIdx.x * w * h;
    for (int r = 0; r < REPS; r++) {
        const unsigned long long t0 = wall_clock64();
        const int d = (dir < 2) ? dir : ((0x2C >> r) & 1);          // dir 2: the sequence F F I I F I -- is a slow first pass about the CODE or about the moment?
        if (d == 0) tail_fwd_level(bufA, bufB, g, w, h, nostore ? w * h : wn * hn, wave, lane);
        else tail_inv_level(bufB, bufC, bufA, w, h, wn * hn, wave, lane, true, true);
        lds_barrier();
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) st[blockIdx.x * REPS + r] = t1 - t0;
    }
}
int main() {
    int32_t *gout; unsigned long long *st;
    hipMalloc(&gout, 240 * 128 * 128 * 4); hipMalloc(&st, 240 * REPS * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    std::vector<unsigned long long> h(240 * REPS);
    const int shapes[][2] = {{64, 64}, {128, 64}};
    for (auto &sh : shapes) for (int dir = 0; dir < 3; dir++) for (int nostore = 0; nostore < (dir == 0 ? 2 : 1); nostore++) for (int grid : {1, 240}) {
        const int w = sh[0], hh = sh[1];
        const size_t ldsb = (size_t)(((w * hh + 3) & ~3) * 3 + 16) * 4;
        if (ldsb > 150 * 1024) continue;
        for (int it = 0; it < 2; it++) hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(1024), ldsb, 0, gout, st, w, hh, dir, nostore);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, grid * REPS * 8, hipMemcpyDeviceToHost);
        printf("%3dx%-3d %s%s grid %3d:", w, hh, dir == 2 ? "FFIIFI" : (dir ? "inv" : "fwd"), nostore ? " (all output to LDS)" : "", grid);
        for (int r = 0; r < REPS; r++) {
            std::vector<unsigned long long> v;
            for (int g = 0; g < grid; g++) v.push_back(h[g * REPS + r]);
            std::sort(v.begin(), v.end());
            printf("  %llu", v[v.size() / 2]);
        }
        printf("   (median over workgroups, 10 ns ticks, rep 0..%d)\n", REPS - 1);
    }
    return 0;
}
