// j2k_internal.h -- shared between the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/j2kgfx.h"

namespace j2k {

// One tile-component (or RCT/ICT group of three) at one decomposition level.
// All offsets are in ELEMENTS relative to the base pointer the launch passes.
struct DwtPlane {
    int64_t src_off[3];   // level input: comp k at src + src_off[k], row stride src_stride
    int64_t out_off[3];   // final coefficient plane of comp k (dense, stride = w of level 0... see n_next)
    int64_t nxt_off[3];   // scratch that receives the prefix [0, n_next) = input of the next level
    int32_t src_stride;
    int32_t w, h;         // dims of this level: dense w x h matrix
    int32_t n_next;       // w_{l+1}*h_{l+1}; linear idx < n_next -> nxt, else -> out (0 on the last level)
    int32_t out_stride;   // inverse level 0 only: row stride of the destination frame
    int32_t pad_;
};

// One wavefront's work: a column strip x a band of pair-rows of one plane.
struct DwtJob {
    int32_t plane;
    int32_t col0;         // first OWNED input column (multiple of CPL); 0 for the first strip
    int32_t prow0;        // first pair-row of the band
    int32_t nprow;        // pair-rows in the band (low 16 bits) | J2K_LINK_* (forward 5-3 only)
};
// forward 5-3: this band and its neighbour in the same workgroup exchange their halo rows through LDS (dwt53.hip)
#define J2K_LINK_UP 0x10000
#define J2K_LINK_DOWN 0x20000

// One code-block job (device form).
struct BlockJob {
    int64_t src_off;      // element offset of the window origin inside the coefficient buffer
    int64_t out_off;      // byte offset of this job's slot / element offset of its decoded block
    int32_t stride;       // plane width
    int32_t w, h;
    int32_t band;
};

// One distinct block window of the HT encoder (j2k_plan_encode_stream): the job + the jobs that read the same window
struct HtUJob {
    BlockJob J;
    int32_t jid;          // the job that is coded (first of its list)
    int32_t alias_off;    // its list of job ids in the plan's alias_ids table (its own id first)
    int32_t nalias;
    int32_t pad_;
};

// One tile-component for the fused "tail" kernels: decomposition levels l0..l0+nlev-1 run inside LDS.
struct TailPlane {
    int64_t scr_off;      // forward: level-l0 input prefix in scratch; inverse: where X_{l0} is written
    int64_t coef_off;     // final coefficient plane
    int32_t w, h;         // dims at level l0
    int32_t nlev;
    int32_t pad_;
};

// launch wrappers (dwt53.hip, dwt97.hip, mct.hip, ht.hip, t1.hip)
struct LevelLaunch {
    const DwtJob *jobs;   // device
    int njobs;
    const DwtPlane *planes;  // device
    int cpl;              // columns per lane: 2, 4 or 8
    int vec;              // 1: every plane satisfies the vector-access alignment rules
    int ncomp;            // 1 or 3 (3 = fused colour transform on level 0)
    int pf;               // forward 5-3: software-prefetch variant
    int pix_stride;       // level 0: > 0 = the frame is packed pixels with this row stride in PIXELS (RGBA8 for a triple unless pix_src says otherwise)
    int pix_src;          // ... which pixels (dwt53_plane_wg.inc SRC / DST: 1 Gray16, 2 Gray8, 3 a byte of a four-byte pixel, 4 a 16-bit sample of an eight-byte pixel)
    long long comp_elems; // ... W * H of the frame (plane offsets are component * W * H + y0 * W + x0)
    int wg_waves;         // > 0: `jobs` is the per-WORKGROUP table of dwt53_fwd_rgba8_wg_kernel (dwt53_l0pix.inc): wg_waves
                          // wavefronts per workgroup, wg_waves - 1 pair-rows each; level 0 of an RGBA8 frame only
    int wg_store;         // its final-coefficient store flavour (0 plain, 1 nt, ...)
    // levels 0 + 1 in one launch (dwt53_fwd_rgba8_wg2_kernel): jobs2 = the bands of the top half of every plane (wg2_waves waves
    // each), planes1 = the level-1 plane table (three per level-0 plane), nxt1 = the scratch that feeds level 2; `jobs` then
    // holds only the remaining bands
    const DwtJob *jobs2;
    int njobs2, wg2_waves;
    const DwtPlane *planes1;
    int32_t *nxt1;
    const DwtJob *pjobs;  // single-component planes in workgroup form (dwt53_plane_wg.inc): per-workgroup table, pwaves waves each
    int pnjobs, pwaves, pmulti;
    hipEvent_t ev_start, ev_stop;   // non-null: the dispatch itself stamps these (hipExtLaunchKernelGGL) -- the kernel's own
                                    // begin / end, without the launch gap an event pair around the launch would include
};

hipError_t launch_dwt53_fwd(hipStream_t s, const LevelLaunch &L, const int32_t *src, int32_t *out,
                            int32_t *nxt, int dc_shift);
hipError_t launch_dwt53_inv(hipStream_t s, const LevelLaunch &L, const int32_t *coef, const int32_t *prev,
                            int32_t *dst, int dc_shift, int final_level);
hipError_t launch_dwt53_tail_fwd(hipStream_t s, const TailPlane *planes, int nplanes, size_t lds_bytes, const int32_t *scr,
                                 int32_t *coef, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t launch_dwt53_tail_inv(hipStream_t s, const TailPlane *planes, int nplanes, size_t lds_bytes, const int32_t *coef,
                                 int32_t *scr, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// every level below level 0 in one launch (dwt53_deep.inc); ev0 / ev1: optional begin / end events stamped by the dispatch
hipError_t launch_dwt53_deep_fwd(hipStream_t s, const DwtJob *jobs, int njobs, const TailPlane *planes, size_t lds_bytes, const int32_t *scr,
                                 int32_t *coef, hipEvent_t ev0, hipEvent_t ev1);
hipError_t launch_dwt53_deep_inv(hipStream_t s, const DwtJob *jobs, int njobs, const TailPlane *planes, size_t lds_bytes, const int32_t *coef,
                                 int32_t *scr, hipEvent_t ev0, hipEvent_t ev1);

hipError_t launch_dwt53_mega_fwd(hipStream_t s, const DwtJob *jobs, int njobs, const TailPlane *tplanes, const DwtPlane *l0planes, size_t lds_bytes,
                                 const int32_t *scr, int32_t *coef, const uint32_t *pix, int32_t *nxt0, int dc_shift, int pix_stride,
                                 hipEvent_t ev0, hipEvent_t ev1);
hipError_t launch_dwt53_mega_inv(hipStream_t s, const DwtJob *jobs, int njobs, const TailPlane *tplanes, const DwtPlane *l0planes, size_t lds_bytes,
                                 const int32_t *coef, int32_t *scr, const int32_t *prev0, uint32_t *pix, int dc_shift, int pix_stride,
                                 hipEvent_t ev0, hipEvent_t ev1);
hipError_t launch_unpack_pixels(hipStream_t s, const uint8_t *pix, size_t stride, int format, int w, int h, int src_max, int dst_max,
                                int32_t *planes);
hipError_t launch_colorspace(hipStream_t s, int cs, int32_t *planes, int ncomp, size_t n, int precision);
hipError_t launch_pack_pixels(hipStream_t s, const int32_t *planes, int ncomp, int precision, int w, int h, uint8_t *pix, size_t stride);

// Tier-2 packets on device buffers (t2dev.hip)
hipError_t launch_t2_fill_cbs(hipStream_t s, long n, const uint64_t *offs, const uint32_t *lens, const uint8_t *numbps, int mb, int ht, j2k_t2_dev_cb *cbs, uint64_t *reset = nullptr);
hipError_t launch_t2_encode_tile_parts(hipStream_t s, const j2k_t2_dev_packet *packets, long npackets, const j2k_t2_dev_cb *cbs, uint64_t ncbs, const uint8_t *data,
                                       int sop, int eph, uint8_t *out, uint64_t cap, uint64_t *offs, void *ws, uint64_t *result, const int32_t *ptile,
                                       const int *tile_packet0, int ntiles, int tile_first, uint64_t *tile_offs, int *status, const BlockJob *slot_jobs,
                                       const uint32_t *maglens, int ht, int body_slices);
size_t t2_dev_workspace(long npackets);
hipError_t launch_t2_encode_packets(hipStream_t s, const j2k_t2_dev_packet *packets, long npackets, const j2k_t2_dev_cb *cbs, uint64_t ncbs, const uint8_t *data,
                                    int sop, int eph, int delay_in, uint8_t *out, uint64_t cap, uint64_t *offs, void *ws, uint64_t *result);

// n bytes src -> dst, any alignment on both sides, one wavefront: bytes up to the destination's next 16-byte boundary,
// then aligned 16-byte stores of (possibly unaligned) 16-byte loads, then the tail bytes
__device__ __forceinline__ void copy_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t n, int lane) {
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15));
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t nv = (n - head) >> 4;
    for (uint32_t i = lane; i < nv; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, src + head + 16 * (size_t)i, 16);
        *reinterpret_cast<uint4 *>(dst + head + 16 * (size_t)i) = v;
    }
    const uint32_t done = head + (nv << 4);
    if (done + lane < n) dst[done + lane] = src[done + lane];
}
__device__ __forceinline__ void zero_run(uint8_t *__restrict__ dst, uint32_t n, int lane) {
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15));
    if ((uint32_t)lane < head) dst[lane] = 0;
    const uint32_t nv = (n - head) >> 4;
    for (uint32_t i = lane; i < nv; i += 64) *reinterpret_cast<uint4 *>(dst + head + 16 * (size_t)i) = make_uint4(0, 0, 0, 0);
    const uint32_t done = head + (nv << 4);
    if (done + lane < n) dst[done + lane] = 0;
}

// one wavefront per job.  maglens != NULL (HT blocks coded by j2k_plan_encode_stream): the slot holds
// MagSgn | <hole> | VLC | SCUP -- the MEL segment of max(64, 2wh)/4 zero bytes (ht.go:978, 1019) was never written to the
// slot and is produced here as zeros, so two thirds of a 64x64 block's bytes are neither stored twice nor read back.
__device__ __forceinline__ void gather_job(const BlockJob &J, const uint8_t *__restrict__ slots, uint8_t *__restrict__ dst, uint32_t len,
                                           bool ht, uint32_t mag, int lane) {
    const uint8_t *src = slots + J.out_off;
    if (!ht) {
        copy_bytes(dst, src, len, lane);
        return;
    }
    const size_t nsamp = (size_t)J.w * J.h;
    const uint32_t mel = (uint32_t)((nsamp * 2 < 64 ? 64 : nsamp * 2) / 4);
    copy_bytes(dst, src, mag, lane);
    zero_run(dst + mag, mel, lane);
    copy_bytes(dst + mag + mel, src + mag + mel, len - mag - mel, lane);
}

// an environment variable of the tuning set: read only when J2K_TUNING=1 (see j2k_ctx_create)
inline const char *tuning_env(const char *name) {
    static const int on = [] { const char *t = getenv("J2K_TUNING"); return (t && t[0] == '1' && !t[1]) ? 1 : 0; }();
    return on ? getenv(name) : nullptr;
}

// DEV BUILDS ONLY (make CXXFLAGS+=-DJ2K_DEV; env J2K_DEV_SKIP = bit mask): launches left out to measure what each kernel
// costs with several frames in flight (`J2K_DEV_SKIP=<mask> bash tools/ab.sh ...` on a -DJ2K_DEV build).  Results are wrong with any bit set, so the shipped library
// does not have the switch at all: g_dev_skip is the constant 0 and every test of it folds away.
#ifdef J2K_DEV
extern int g_dev_skip, g_dev_dup;      // J2K_DEV_DUP: the same bits, launches issued TWICE (every one is idempotent: results stay valid)
inline int dev_reps(int bit) { return (g_dev_skip & bit) ? 0 : ((g_dev_dup & bit) ? 2 : 1); }
#else
constexpr int g_dev_skip = 0;
constexpr int dev_reps(int) { return 1; }
#endif
}  // namespace j2k
