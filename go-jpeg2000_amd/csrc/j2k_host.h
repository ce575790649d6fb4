// j2k_host.h -- what the host-side files of the C ABI share (j2k_ctx.cpp, j2k_planbuild.cpp, j2k_stages.cpp, j2k_frame.cpp,
// j2k_hostcalls.cpp; round 4 had all of it in one 2 200-line j2k_abi.cpp): the kernels' launch wrappers, the status helpers, the
// context's staging slots and the plan builder.  Host-side orchestration only: geometry -> device job tables, launches on the
// context's HIP stream, host <-> device staging.  All arithmetic lives in the .hip kernels; there is no CPU fallback.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

#include "j2k_plan.h"

namespace j2k {
#ifdef J2K_DEV
int g_dev_skip = 0, g_dev_dup = 0;
#endif
hipError_t launch_add_const(hipStream_t s, int32_t *d, size_t n, int delta);
hipError_t launch_rct(hipStream_t s, int32_t *a, int32_t *b, int32_t *c, size_t n, int inverse);
hipError_t launch_ict(hipStream_t s, double *a, double *b, double *c, size_t n, int inverse);
hipError_t launch_dwt97_fwd(hipStream_t s, const LevelLaunch &L, const void *src, int src_is_f64, int32_t *out_i32,
                            double *out_f64, double *nxt, int dc_shift, int quant, double step, int mct);
hipError_t launch_dwt97_inv(hipStream_t s, const LevelLaunch &L, const void *coef, int coef_is_f64, const double *prev,
                            void *dst, int dc_shift, int final_level, int dst_mode, int mct);
hipError_t launch_ht_encode(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *slots,
                            uint32_t *lens, uint8_t *numbps, int *fault, uint32_t *maglens = nullptr, const HtUJob *utab = nullptr, int nunique = 0,
                            const int *alias_ids = nullptr);
hipError_t launch_ht_decode(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs,
                            const uint32_t *lens, int32_t *decoded, uint32_t *scratch, int coded_rows_only = 0, const BlockJob *placed_jobs = nullptr);
size_t ht_decode_scratch_words(int njobs);
hipError_t launch_ht_encode_stream(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *stream,
                                   uint64_t *offs, uint32_t *lens, uint8_t *numbps, uint64_t *status, uint32_t epoch, int *fault);
int ht_fast_max_samples();
hipError_t launch_t1_encode(hipStream_t s, const BlockJob *jobs, int njobs, const int32_t *coef, uint8_t *slots,
                            uint32_t *lens, uint8_t *numbps, uint8_t *work, size_t work_per_job, int *fault, int max_dim,
                            uint8_t *sym, size_t sym_stride, uint32_t *nsyms, int lanes, uint8_t *bigsym = nullptr, const uint64_t *bigsym_off = nullptr,
                            uint32_t *bignsyms = nullptr);
size_t t1_sym_stride(int planes);
hipError_t launch_t1_decode(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs,
                            const uint32_t *lens, const uint8_t *numbps, int32_t *decoded, uint8_t *work,
                            size_t work_per_job, int max_dim, int general_only, uint8_t *split_ws, int sig_lanes, int throughput = 0);
size_t t1_dec_split_bytes(size_t njobs);
hipError_t launch_mq_encode(hipStream_t s, const uint8_t *ctxs, const uint8_t *decs, size_t n, uint8_t *out, size_t cap, uint32_t *out_len, int *fault);
hipError_t launch_mq_decode(hipStream_t s, const uint8_t *data, size_t len, const uint8_t *ctxs, size_t n, uint8_t *decs, int *fault);
hipError_t launch_raw_encode(hipStream_t s, const uint8_t *bits, size_t n, uint8_t *out, size_t cap, uint32_t *out_len, int *fault);
hipError_t launch_raw_decode(hipStream_t s, const uint8_t *data, size_t len, size_t n, uint8_t *bits);
size_t t1_work_bytes(int w, int h);
size_t t1_flag_bytes(int w, int h);
hipError_t launch_compact(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *slots, const uint32_t *lens,
                          uint64_t *offs, uint8_t *stream, const uint32_t *maglens, const uint32_t *mels = nullptr, uint64_t *toffs = nullptr);
hipError_t launch_mel_table(hipStream_t s, const BlockJob *jobs, int njobs, uint32_t *mels);
hipError_t launch_assemble_tiles(hipStream_t s, const uint8_t *stream, const uint64_t *offs, const int *job0, int ntiles, int tile_first,
                                 uint64_t max_tile_bytes, uint8_t *out, uint64_t *out_len);
// t2dec.hip
size_t t2_chain_bytes();
hipError_t launch_t2_tile_chains(hipStream_t s, const uint8_t *cs, uint64_t len, const uint64_t *tile_offs, int ntiles, int tile_first,
                                 const int *tile_packet0, void *chains);
hipError_t launch_t2_decode_packets(hipStream_t s, void *chains, int nchains, const j2k_t2_dev_packet *packets, long npackets, j2k_t2_dev_cb *cbs, uint64_t ncbs,
                                    const uint8_t *data, int sop, int eph, int clean, uint64_t *body_base, int *frame_status);
void t2_make_chain(void *dst, uint64_t len, long npackets, const j2k_t2_dec_state &st);
void t2_read_chain(const void *src, j2k_t2_dec_state &st, int &status, long &done);
size_t t2_par_workspace(long npackets, int ntiles);
hipError_t launch_t2_decode_tiles(hipStream_t s, void *chains, int ntiles, const int *tile_packet0, const j2k_t2_dev_packet *packets, long npackets, j2k_t2_dev_cb *cbs,
                                  uint64_t ncbs, const uint8_t *data, uint64_t len, int sop, int eph, uint64_t *body_base, int *frame_status, void *ws,
                                  int ht, int mb, uint64_t *offs, uint32_t *lens, uint8_t *numbps);
hipError_t launch_t2_blocks(hipStream_t s, long n, const j2k_t2_dev_cb *cbs, int ht, int mb, uint64_t total, uint64_t *offs, uint32_t *lens, uint8_t *numbps, int *status);
hipError_t launch_place_blocks(hipStream_t s, const BlockJob *src_jobs, const BlockJob *dec_jobs, int njobs, int max_h, const int32_t *decoded, int32_t *coeff, int ystep = 1);
hipError_t launch_scan(hipStream_t s, const uint32_t *lens, int njobs, uint64_t *offs, const uint32_t *mels, uint64_t *toffs);
size_t pack_header_bytes(size_t n);
hipError_t launch_pack(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs, const uint64_t *toffs,
                       const uint32_t *lens, const uint8_t *numbps, const uint32_t *maglens, uint8_t *pack);
hipError_t launch_unpack(hipStream_t s, const BlockJob *jobs, int njobs, int count, const uint8_t *const *packs, const size_t *pack_bytes,
                         uint8_t *const *streams, size_t stream_cap, uint64_t *const *offs, uint32_t *const *lens, uint8_t *const *numbps, int *fault);
}  // namespace j2k

// ---- status / errors (j2k_ctx.cpp) ----
int fail(j2k_ctx *ctx, int status, const char *msg);
int fail_hip(j2k_ctx *ctx, hipError_t e, const char *where);
#define HIPCHK(ctx, call)                                      \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call); \
    } while (0)
// contexts of this process that have built an MQ-coder plan: two or more = frames in flight (throughput settings of the MQ kernels)
extern std::atomic<int> g_mq_ctxs;
bool mq_throughput_mode();
int check_fault(j2k_ctx *ctx);
bool profile_pair(j2k_ctx *ctx, int tag, hipEvent_t &e0, hipEvent_t &e1);
int stage_reserve(j2k_ctx *ctx, int slot, size_t bytes);

// ---- plans (j2k_planbuild.cpp) ----
static inline int64_t align4(int64_t v) { return (v + 3) & ~int64_t(3); }
template <typename T>
static int upload(j2k_ctx *ctx, T **dptr, const std::vector<T> &v) {
    *dptr = nullptr;
    if (v.empty()) return J2K_OK;
    HIPCHK(ctx, hipMalloc((void **)dptr, v.size() * sizeof(T)));
    HIPCHK(ctx, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return J2K_OK;
}
int build_plan(j2k_ctx *ctx, const j2k::PlanSpec &S, j2k_plan **out);
int cached_plan(j2k_ctx *ctx, const j2k::PlanSpec &S, j2k_plan **out);
void plan_t2_packets(const j2k_plan *P, int layer, std::vector<j2k_t2_dev_packet> &out, std::vector<int> *tile_packet0);

// ---- stages (j2k_stages.cpp) ----
struct T1Workspace { size_t off_nsyms, off_sym, stride, total; };
T1Workspace t1_workspace(const j2k_ctx *ctx, size_t n, size_t wpj);
int ensure(j2k_ctx *ctx, void **p, size_t bytes);
struct PixIO { int stride = 0, single = 0, triple = 0; };
int plan_forward_impl(j2k_plan *P, const void *d_frame, void *d_coeff, PixIO pix = PixIO());
int plan_encode_private_slots(j2k_plan *P, const int32_t *d_coeff, uint32_t *d_lens, uint8_t *d_numbps);
int plan_encode_frame_from_coeff(j2k_plan *P, const int32_t *d_coeff, uint32_t *d_lens, uint8_t *d_numbps, int sop, int eph, uint8_t *d_out, size_t cap,
                                 uint64_t *d_tile_offs);
int plan_inverse_impl(j2k_plan *P, const void *d_coeff, void *d_frame, PixIO pix = PixIO());
