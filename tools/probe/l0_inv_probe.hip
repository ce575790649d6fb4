// l0_inv_probe: what does the memory system allow for the INVERSE level-0 byte mix (12 B read : 4 B written per pixel:
// per pair-row of a 512-column tile, a low-pass and a high-pass row of three dense int32 planes in, two rows of packed RGBA8
// out)?  No arithmetic that matters, no halo, variants of the wave / workgroup structure.  Per-launch dispatch events,
// footprint F frames rotating (F = 1: cache-resident, F = 8: HBM).
// build: hipcc -O3 --offload-arch=gfx950 -o l0_inv_probe l0_inv_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int W = 3584, H = 2048, T = 512;
constexpr int TX = W / T, TY = H / T, NT = TX * TY;
constexpr size_t PX = (size_t)W * H;
constexpr int PROWS = NT * (T / 2);            // pair-rows in a frame

// MODE 0: wave = pair-row, 12 loads in flight, 4 nt stores (32 B per lane per row: two 16-byte pieces 16 B apart... as the product)
// MODE 1: the same, stores lane-contiguous per instruction (lane l writes bytes [16 l, 16 l + 16) and [1024 + 16 l, ...))
// MODE 2: MODE 0 + an LDS exchange and two barriers per 4-wave workgroup (the product's synchronisation shape)
// MODE 3: read only
// MODE 4: MODE 0 with plain stores
// XCD: workgroup b takes pair-row group (b % 8) * chunk + b / 8
template <int MODE, int XCD>
__global__ __launch_bounds__(256) void inv_mix_k(const int* __restrict__ coef, unsigned* __restrict__ pix, int* __restrict__ sink, int ngroups) {
  __shared__ v4i slot[4][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int grp = blockIdx.x;
  if (XCD == 1) { const int chunk = (ngroups + 7) / 8; grp = (blockIdx.x % 8) * chunk + blockIdx.x / 8; if (grp >= ngroups) return; }
  if (XCD == 2) {        // whole tiles dealt round-robin to the XCDs: XCD x takes tiles x, x + 8, ...
    const int gpt = T / 2 / 4, x = blockIdx.x % 8, i = blockIdx.x / 8;
    const int tile = (i / gpt) * 8 + x;
    if (tile >= NT) return;
    grp = tile * gpt + i % gpt;
  }
  if (XCD == 3) {        // chunked, each XCD's chunk rotated by a different amount
    const int chunk = (ngroups + 7) / 8, x = blockIdx.x % 8;
    grp = x * chunk + (blockIdx.x / 8 + x * 37) % chunk; if (grp >= ngroups) return;
  }
  const int pr = grp * 4 + wv;                  // pair-row index in the frame's tile-major order
  const int tile = pr / (T / 2), q = pr % (T / 2);
  const int* base = coef + (size_t)tile * 3 * T * T;
  v4i a[12];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int* p = base + (size_t)k * T * T;
    a[4 * k + 0] = *reinterpret_cast<const v4i*>(p + q * T + lane * 4);
    a[4 * k + 1] = *reinterpret_cast<const v4i*>(p + q * T + T / 2 + lane * 4);
    a[4 * k + 2] = *reinterpret_cast<const v4i*>(p + (T / 2 + q) * T + lane * 4);
    a[4 * k + 3] = *reinterpret_cast<const v4i*>(p + (T / 2 + q) * T + T / 2 + lane * 4);
  }
  v4i r0 = a[0] + a[4] + a[8], r1 = a[1] + a[5] + a[9], r2 = a[2] + a[6] + a[10], r3 = a[3] + a[7] + a[11];
  if (MODE == 2) {
    slot[wv][lane] = r0;
    __syncthreads();
    r1 += slot[(wv + 1) & 3][lane];
    __syncthreads();
    slot[wv][lane] = r1;
    __syncthreads();
    r2 += slot[(wv + 3) & 3][lane];
  }
  if (MODE == 3) { if ((r0 + r1 + r2 + r3).x == 0x12345678) sink[0] = 1; return; }
  const int tx = tile % TX, ty = tile / TX;
  unsigned* o = pix + (size_t)(ty * T + 2 * q) * W + tx * T;
  if (MODE == 1) {
    __builtin_nontemporal_store(r0, reinterpret_cast<v4i*>(o + lane * 4));
    __builtin_nontemporal_store(r1, reinterpret_cast<v4i*>(o + 256 + lane * 4));
    __builtin_nontemporal_store(r2, reinterpret_cast<v4i*>(o + W + lane * 4));
    __builtin_nontemporal_store(r3, reinterpret_cast<v4i*>(o + W + 256 + lane * 4));
  } else if (MODE == 4) {
    *reinterpret_cast<v4i*>(o + lane * 8) = r0; *reinterpret_cast<v4i*>(o + lane * 8 + 4) = r1;
    *reinterpret_cast<v4i*>(o + W + lane * 8) = r2; *reinterpret_cast<v4i*>(o + W + lane * 8 + 4) = r3;
  } else {
    __builtin_nontemporal_store(r0, reinterpret_cast<v4i*>(o + lane * 8));
    __builtin_nontemporal_store(r1, reinterpret_cast<v4i*>(o + lane * 8 + 4));
    __builtin_nontemporal_store(r2, reinterpret_cast<v4i*>(o + W + lane * 8));
    __builtin_nontemporal_store(r3, reinterpret_cast<v4i*>(o + W + lane * 8 + 4));
  }
}
template <int MODE, int XCD>
static void run(const char* name, int F, int* coef, unsigned* pix, int* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ngroups = PROWS / 4;
  const int grid = XCD == 2 ? ((NT + 7) / 8) * 8 * (T / 2 / 4) : (XCD ? ((ngroups + 7) / 8) * 8 : ngroups);
  std::vector<float> t;
  for (int it = 0; it < 40; it++) {
    const int f = it % F;
    hipExtLaunchKernelGGL((inv_mix_k<MODE, XCD>), dim3(grid), dim3(256), 0, 0, e0, e1, 0, coef + (size_t)f * 3 * PX, pix + (size_t)f * PX, sink, ngroups);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 8) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  const double bytes = (MODE == 3 ? 12.0 : 16.0) * PX;
  printf("%-58s F=%d  median %.1f us  min %.1f  -> %.2f TB/s\n", name, F, t[t.size() / 2], t[0], bytes / t[t.size() / 2] * 1e-6);
}
int main() {
  const int FMAX = 8;
  int* coef; unsigned* pix; int* sink;
  CK(hipMalloc(&coef, FMAX * 3 * PX * 4)); CK(hipMalloc(&pix, FMAX * PX * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(coef, 1, FMAX * 3 * PX * 4)); CK(hipMemset(pix, 0, FMAX * PX * 4));
  printf("frame %dx%d (%d tiles), %.1f MB moved per launch\n", W, H, NT, 16.0 * PX * 1e-6);
  for (int F : {1, 8}) {
    run<0, 0>("wave = pair-row, stores as the product (2 x 16 B per lane)", F, coef, pix, sink);
    run<0, 1>("  + XCD-chunked job order", F, coef, pix, sink);
    run<1, 0>("stores lane-contiguous per instruction", F, coef, pix, sink);
    run<1, 1>("  + XCD-chunked job order", F, coef, pix, sink);
    run<1, 2>("  + whole tiles round-robin over the XCDs", F, coef, pix, sink);
    run<1, 3>("  + XCD chunks, rotated starts", F, coef, pix, sink);
    run<2, 1>("product's barriers + LDS exchange, XCD order", F, coef, pix, sink);
    run<2, 2>("product's barriers + LDS exchange, tiles round-robin", F, coef, pix, sink);
    run<4, 1>("plain stores, XCD order", F, coef, pix, sink);
    run<3, 0>("read only", F, coef, pix, sink);
    run<3, 1>("read only, XCD order", F, coef, pix, sink);
  }
  return 0;
}
