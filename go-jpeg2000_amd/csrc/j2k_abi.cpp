// j2k_abi.cpp -- the C ABI of libj2kgfx.so (declared in include/j2kgfx.h).
// Host-side orchestration only: geometry -> device job tables, launches on the
// context's HIP stream, host<->device staging for the one-call-per-reference-function
// entry points.  All arithmetic lives in the .hip kernels; there is no CPU fallback.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

#include "j2k_plan.h"

using namespace j2k;

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
                            const uint32_t *lens, int32_t *decoded, uint32_t *scratch, int coded_rows_only = 0);
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
                                 uint64_t max_tile_bytes, uint8_t *out, uint64_t *out_len, uint64_t cap = 0, uint64_t *tile_offs = nullptr,
                                 int *status = nullptr);
// t2dec.hip
size_t t2_chain_bytes();
hipError_t launch_t2_tile_chains(hipStream_t s, const uint8_t *cs, uint64_t len, const uint64_t *tile_offs, int ntiles, int tile_first,
                                 const int *tile_packet0, void *chains);
hipError_t launch_t2_decode_packets(hipStream_t s, void *chains, int nchains, const j2k_t2_dev_packet *packets, long npackets, j2k_t2_dev_cb *cbs, uint64_t ncbs,
                                    const uint8_t *data, int sop, int eph, int clean, uint64_t *body_base, int *frame_status);
void t2_make_chain(void *dst, uint64_t len, long npackets, const j2k_t2_dec_state &st);
void t2_read_chain(const void *src, j2k_t2_dec_state &st, int &status, long &done);
hipError_t launch_t2_blocks(hipStream_t s, long n, const j2k_t2_dev_cb *cbs, int ht, int mb, uint64_t total, uint64_t *offs, uint32_t *lens, uint8_t *numbps);
hipError_t launch_place_blocks(hipStream_t s, const BlockJob *src_jobs, const BlockJob *dec_jobs, int njobs, int max_h, const int32_t *decoded, int32_t *coeff);
hipError_t launch_scan(hipStream_t s, const uint32_t *lens, int njobs, uint64_t *offs, const uint32_t *mels, uint64_t *toffs);
size_t pack_header_bytes(size_t n);
hipError_t launch_pack(hipStream_t s, const BlockJob *jobs, int njobs, const uint8_t *stream, const uint64_t *offs, const uint64_t *toffs,
                       const uint32_t *lens, const uint8_t *numbps, const uint32_t *maglens, uint8_t *pack);
hipError_t launch_unpack(hipStream_t s, const BlockJob *jobs, int njobs, int count, const uint8_t *const *packs, const size_t *pack_bytes,
                         uint8_t *const *streams, size_t stream_cap, uint64_t *const *offs, uint32_t *const *lens, uint8_t *const *numbps, int *fault);
}  // namespace j2k

// ------------------------------------------------------------------------------
// status / errors
// ------------------------------------------------------------------------------
static int fail(j2k_ctx *ctx, int status, const char *msg) {
    if (ctx) ctx->last_error = msg ? msg : "";
    return status;
}
static int fail_hip(j2k_ctx *ctx, hipError_t e, const char *where) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
    if (ctx) ctx->last_error = buf;
    return J2K_ERR_HIP;
}
#define HIPCHK(ctx, call)                                      \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call); \
    } while (0)

extern "C" const char *j2k_ctx_last_error(j2k_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

// ------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------
// contexts of this process that have built an MQ-coder plan: two or more = frames in flight, the MQ kernels then take their
// throughput settings (t1.hip: blocks per wavefront of the encode chains, plane-stepped decoder), one = latency settings
static std::atomic<int> g_mq_ctxs{0};
static bool mq_throughput_mode() { return g_mq_ctxs.load(std::memory_order_relaxed) >= 2; }
// ---- tuning options: which of the measured kernel forms a context takes.  The defaults (j2k_plan.h) are what the benchmarks
// measured best; a host sets them through j2k_ctx_set_option BEFORE it creates plans on the context.
struct CtxOption { const char *name; bool (*set)(j2k_ctx *, long); };
static const std::vector<CtxOption> &ctx_options() {
#define OPT(name, cond, stmt) CtxOption{name, [](j2k_ctx *c, long v) -> bool { (void)c; if (!(cond)) return false; stmt; return true; }}
    static const std::vector<CtxOption> T = {
        OPT("plane_wg3", v == 0 || v == 1, c->plane_wg3 = v != 0),
        OPT("pix_fuse", v >= 0 && v <= 2, c->pix_fuse = (int)v),
        OPT("plane_wg", v == 0 || v == 4 || v == 8, c->plane_wg = (int)v),
        OPT("l0_fuse", v == 0 || v == 8 || v == 10 || v == 16, c->l0_fuse = (int)v),
        OPT("l0_wg", v == 0 || v == 4 || v == 8, c->l0_wg = (int)v),
        OPT("plane_wg97", v == 0 || v == 8, c->plane_wg97 = (int)v),
        OPT("l0_wg97_inv", v == 0 || v == 6 || v == 8 || v == 10 || v == 12, c->l0_wg97_inv = (int)v),
        OPT("l0_wg97", v == 0 || (v >= 6 && v <= 16 && v % 2 == 0), c->l0_wg97 = (int)v),
        OPT("l0_xcd", v == 0 || v == 1, c->l0_xcd = v != 0),
        OPT("l0_deal", v == 0 || v == 1, c->l0_deal = v != 0),
        OPT("ht_alias", v == 0 || v == 1, c->ht_alias = v != 0),
        OPT("l0_inv_wpe", v >= 5 && v <= 7, c->l0_inv_wpe = (int)v),
        OPT("l0_wg_inv", v == 0 || v == 1, c->l0_wg_inv = v != 0),
        OPT("l0_wg_invw", v == 0 || v == 4 || v == 8, c->l0_wg_invw = (int)v),
        OPT("l0_xcd_group", v >= 0 && v <= 4096, c->l0_xcd_group = (int)v),
        OPT("l0_store", v == 0 || v == 1 || v == 2 || v == 4, c->l0_store = (int)v),
        OPT("fuse_compact", v == 0 || v == 1, c->fuse_compact = v != 0),
        OPT("deep", v == 0 || v == 1, c->use_deep = v != 0),
        OPT("deep_min_planes", v >= 0 && v <= 1000000, c->deep_min_planes = (int)v),
        OPT("deep_mid", v == 0 || v == 1, c->deep_mid = v != 0),
        OPT("deep_mid_inv", v >= 0 && v <= 2, c->deep_mid_inv = (int)v),
        OPT("mega", v >= 0 && v <= 2, c->mega = (int)v),
        OPT("t1_split", v == 0 || v == 1, c->t1_split = v != 0),
        OPT("t1_sym_mb", v >= 0, c->t1_sym_mb = v),
        OPT("t1_dec_general", v == 0 || v == 1, c->t1_dec_general = v != 0),
        OPT("t1_dec_split", v >= -1, c->t1_dec_split = (int)v),
        OPT("t1_dec_lanes", v >= 0 && v <= 2, c->t1_dec_lanes = (int)v),
        OPT("t1_lanes", v >= 0 && v <= 64, c->t1_lanes = (int)v),
    };
#undef OPT
    return T;
}
extern "C" int j2k_ctx_set_option(j2k_ctx *ctx, const char *name, long value) {
    if (!ctx || !name) return J2K_ERR_INVALID_ARG;
    for (const CtxOption &o : ctx_options())
        if (!strcmp(o.name, name)) return o.set(ctx, value) ? J2K_OK : fail(ctx, J2K_ERR_INVALID_ARG, "j2k_ctx_set_option: value out of range");
    return fail(ctx, J2K_ERR_INVALID_ARG, "j2k_ctx_set_option: no such option");
}

extern "C" int j2k_ctx_create(int device, j2k_ctx **out) {
    if (!out) return J2K_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return J2K_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return J2K_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return J2K_ERR_NO_DEVICE;
    j2k_ctx *ctx = new j2k_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return J2K_ERR_HIP;
    }
#ifdef J2K_DEV
    if (const char *e = getenv("J2K_DEV_SKIP")) j2k::g_dev_skip = (int)strtol(e, nullptr, 0);
    if (const char *e = getenv("J2K_DEV_DUP")) j2k::g_dev_dup = (int)strtol(e, nullptr, 0);
#endif
    // The environment decides NOTHING unless J2K_TUNING=1 is set (tests, tools/ab.sh, bench.py's A/B modes): a host process does
    // not inherit kernel choices from variables it never heard of.  With it, every option of
    // j2k_ctx_set_option is read from J2K_<NAME IN UPPER CASE>; values out of range are ignored, as before.
    if (j2k::tuning_env("J2K_TUNING"))
        for (const CtxOption &o : ctx_options()) {
            std::string var = "J2K_";
            for (const char *c = o.name; *c; c++) var += (char)toupper((unsigned char)*c);
            if (const char *e = getenv(var.c_str())) (void)o.set(ctx, atol(e));
        }
    *out = ctx;
    return J2K_OK;
}

extern "C" void j2k_ctx_destroy(j2k_ctx *ctx) {
    if (!ctx) return;
    if (ctx->counted_mq) g_mq_ctxs.fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (j2k_plan *p : ctx->cache) j2k_plan_destroy(p);
    for (int i = 0; i < 5; i++)
        if (ctx->stage[i]) (void)hipFree(ctx->stage[i]);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ---- HIP graphs: a recorded sequence of plan calls replayed with one launch --------------------------------------------
// (No counterpart in the reference.)  A frame's pipeline is a dozen or two dependent kernel launches on one stream; for small
// frames each is too short to hide the next one's launch, and the host pays for every one.  Between capture_begin and
// capture_end the asynchronous plan calls of this context (j2k_plan_forward* / encode_stream / decode_blocks / inverse* /
// assemble) are recorded instead of run; the graph replays them on the context's stream with the same device pointers.
// The calls must have run once before (workspaces sized, lazy tables uploaded): nothing may allocate or synchronise while
// the stream captures.
struct j2k_graph { j2k_ctx *ctx; hipGraph_t graph; hipGraphExec_t exec; bool arms_fault; };
extern "C" int j2k_ctx_capture_begin(j2k_ctx *ctx) {
    if (!ctx || ctx->capturing) return J2K_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->fault_armed_before_capture = ctx->fault_armed;
    HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return J2K_OK;
}
extern "C" int j2k_ctx_capture_end(j2k_ctx *ctx, j2k_graph **out) {
    if (!ctx || !out || !ctx->capturing) return J2K_ERR_INVALID_ARG;
    *out = nullptr;
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    HIPCHK(ctx, hipStreamEndCapture(ctx->stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return fail_hip(ctx, e, "hipGraphInstantiate"); }
    *out = new j2k_graph{ctx, g, ex, ctx->fault_armed};
    ctx->fault_armed = ctx->fault_armed_before_capture;      // nothing ran yet
    return J2K_OK;
}
extern "C" int j2k_graph_launch(j2k_graph *G) {
    if (!G || !G->ctx || G->ctx->capturing) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = G->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipGraphLaunch(G->exec, ctx->stream));
    if (G->arms_fault) ctx->fault_armed = true;
    return J2K_OK;
}
extern "C" void j2k_graph_destroy(j2k_graph *G) {
    if (!G) return;
    if (G->ctx) { (void)hipSetDevice(G->ctx->device); (void)hipStreamSynchronize(G->ctx->stream); }
    if (G->exec) (void)hipGraphExecDestroy(G->exec);
    if (G->graph) (void)hipGraphDestroy(G->graph);
    delete G;
}

// The block-encode kernels report inputs outside the reference's domain (Go panic) or a slot overflow in a STICKY
// device word (first int of stage[3]): it is armed by every encode launch, read and cleared at the next
// synchronisation point (j2k_ctx_sync or a synchronous call), so the asynchronous plan calls fail loudly too.
static int check_fault(j2k_ctx *ctx) {
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->fault_armed || !ctx->stage[3]) return J2K_OK;
    int f = 0;
    HIPCHK(ctx, hipMemcpy(&f, ctx->stage[3], sizeof(int), hipMemcpyDeviceToHost));
    ctx->fault_armed = false;
    if (!f) return J2K_OK;
    HIPCHK(ctx, hipMemsetAsync(ctx->stage[3], 0, sizeof(int), ctx->stream));   // ordered with the next launches on this stream
    if (f == 1) return fail(ctx, J2K_ERR_GO_PANIC, "block coder: input on which the reference panics (stream buffer overrun / MinInt32)");
    if (f == 4) return fail(ctx, J2K_ERR_INVALID_ARG, "unpack_stream: the pack was not made by a plan of this geometry");
    // the MQ coder ran past the reference's own mqBuf size (j2k_block_bound): the Go code indexes out of range there
    return fail(ctx, J2K_ERR_GO_PANIC, "block coder: the block needs more bytes than the reference's own buffer holds (t1_fast5.go:47-56: index out of range)");
}

extern "C" int j2k_ctx_sync(j2k_ctx *ctx) {
    if (!ctx) return J2K_ERR_INVALID_ARG;
    return check_fault(ctx);
}
extern "C" void *j2k_ctx_stream(j2k_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int j2k_ctx_profile_enable(j2k_ctx *ctx, int on) {
    if (!ctx) return J2K_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->profile = on < 0 ? 0 : on;          // 1: the forward level-0 dispatch only; 2: every 5-3 transform dispatch, tagged
    ctx->ev_used = 0;
    return J2K_OK;
}
extern "C" int j2k_ctx_profile_read(j2k_ctx *ctx, int64_t *launches, double *total_ms) {
    if (!ctx || !launches || !total_ms) return J2K_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        if (ctx->ev_tag[i / 2] != 0) continue;          // the forward level-0 dispatches (the roofline kernel)
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        tot += ms; n++;
    }
    *launches = n;
    *total_ms = tot;
    ctx->ev_used = 0;
    return J2K_OK;
}
// Sum over the stamped dispatches carrying `tag` (0 forward level 0, 1 forward deeper levels, 2 inverse level 0, 3 inverse
// deeper levels) since the last j2k_ctx_profile_read / _enable; does not reset (call per tag, then j2k_ctx_profile_read).
extern "C" int j2k_ctx_profile_read_tag(j2k_ctx *ctx, int tag, int64_t *launches, double *total_ms) {
    if (!ctx || !launches || !total_ms) return J2K_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        if (ctx->ev_tag[i / 2] != tag) continue;
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        tot += ms; n++;
    }
    *launches = n;
    *total_ms = tot;
    return J2K_OK;
}
// next free event of the pool (grows on demand, capped), or nullptr
static hipEvent_t profile_event(j2k_ctx *ctx) {
    if (!ctx->profile || ctx->capturing || ctx->ev_used >= 8192) return nullptr;
    if (ctx->ev_used >= ctx->ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev.push_back(e);
    }
    return ctx->ev[ctx->ev_used++];
}
// an event pair for one dispatch, or {nullptr, nullptr}: tag 0 in any profile mode, the other tags in mode 2 only
static bool profile_pair(j2k_ctx *ctx, int tag, hipEvent_t &e0, hipEvent_t &e1) {
    e0 = e1 = nullptr;
    if (!ctx->profile || (tag != 0 && ctx->profile < 2) || ctx->capturing || ctx->ev_used + 2 > 8192) return false;
    const size_t pair = ctx->ev_used / 2;
    e0 = profile_event(ctx);
    e1 = e0 ? profile_event(ctx) : nullptr;
    if (!e1) { e0 = nullptr; ctx->ev_used = pair * 2; return false; }
    if (ctx->ev_tag.size() <= pair) ctx->ev_tag.resize(pair + 1, 0);
    ctx->ev_tag[pair] = tag;
    return true;
}

static int stage_reserve(j2k_ctx *ctx, int slot, size_t bytes) {
    if (ctx->stage_bytes[slot] >= bytes) return J2K_OK;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: a workspace would have to grow -- run the same calls once before j2k_ctx_capture_begin");
    if (ctx->stage[slot]) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(ctx->stage[slot]));
        ctx->stage[slot] = nullptr;
        ctx->stage_bytes[slot] = 0;
    }
    size_t cap = std::max<size_t>(bytes, 1 << 20);
    HIPCHK(ctx, hipMalloc(&ctx->stage[slot], cap));
    ctx->stage_bytes[slot] = cap;
    // slot 3 holds the sticky fault word (check_fault): cleared ON THE CONTEXT'S STREAM, i.e. before any kernel that may
    // set it -- a hipMemset on the null stream is not ordered with a non-blocking stream and could clear a fault afterwards
    if (slot == 3) HIPCHK(ctx, hipMemsetAsync(ctx->stage[slot], 0, cap, ctx->stream));
    return J2K_OK;
}

// ------------------------------------------------------------------------------
// plan construction
// ------------------------------------------------------------------------------
bool PlanSpec::operator==(const PlanSpec &o) const {
    return W == o.W && H == o.H && C == o.C && tile_w == o.tile_w && tile_h == o.tile_h && levels == o.levels &&
           frame_h == o.frame_h && wavelet == o.wavelet && precision == o.precision && dc_shift == o.dc_shift && dc_shift_inv == o.dc_shift_inv && mct == o.mct && quant == o.quant && quality == o.quality &&
           num_res_jobs == o.num_res_jobs && cb_w == o.cb_w && cb_h == o.cb_h && coder == o.coder &&
           tile_first == o.tile_first && tile_count == o.tile_count && frame_is_f64 == o.frame_is_f64 && closed_loop == o.closed_loop;
}

static inline int64_t align4(int64_t v) { return (v + 3) & ~int64_t(3); }
static inline int pick_cpl(int maxw) { return maxw >= 384 ? 8 : (maxw >= 192 ? 4 : 2); }

extern "C" size_t j2k_block_bound(int coder, int w, int h) {
    size_t n = (size_t)std::max(w, 0) * (size_t)std::max(h, 0);
    if (coder == J2K_CODER_HT) {               // ht.go:969-996: MagSgn + MEL + VLC buffers + SCUP
        size_t maxSize = std::max<size_t>(n * 2, 64);
        return maxSize / 2 + maxSize / 4 + maxSize / 2 + 2;
    }
    // t1_fast5.go:47-56: mqBuf has width*height*2 + 1024 bytes but never fewer than 16384 (a fresh T1; a pooled one may have
    // more left over from a larger block, which is history, not geometry).  A block that needs more than this is an index
    // panic in the reference; one that fits must be coded: the 16384 floor matters for deep 64x64 blocks (16-bit noise
    // after five lifting levels is ~9.3 KB against 2wh + 1024 = 9216; found by tools/fuzz_gpu.py).
    return std::max<size_t>(n * 2 + 1024, 16384);
}

template <typename T>
static int upload(j2k_ctx *ctx, T **dptr, const std::vector<T> &v) {
    *dptr = nullptr;
    if (v.empty()) return J2K_OK;
    HIPCHK(ctx, hipMalloc((void **)dptr, v.size() * sizeof(T)));
    HIPCHK(ctx, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return J2K_OK;
}

// halo = lanes per side that only feed their neighbours (1 for 5-3; 9-7: 1 if a lane holds >= 2 pairs, else 2)
static void make_jobs(std::vector<DwtJob> &jobs, int plane, int w, int h, int cpl, int band, int halo) {
    const int halfH = (h + 1) / 2;
    int col0 = 0;
    for (;;) {
        const int c_base = col0 - (col0 ? halo * cpl : 0);
        for (int pr = 0; pr < std::max(halfH, 1); pr += band) jobs.push_back(DwtJob{plane, col0, pr, band});
        if (c_base + 64 * cpl >= w) break;
        col0 = c_base + (64 - halo) * cpl;
    }
}

static int build_plan(j2k_ctx *ctx, const PlanSpec &S, j2k_plan **out) {
    if (S.W <= 0 || S.H <= 0 || S.C <= 0 || S.levels < 0 || S.levels > 32) return fail(ctx, J2K_ERR_INVALID_ARG, "bad geometry");
    if ((int64_t)S.W * S.H >= (int64_t)1 << 31) return fail(ctx, J2K_ERR_INVALID_ARG, "plane too large");
    j2k_plan *P = new j2k_plan();
    P->ctx = ctx;
    P->spec = S;
    // a batch (j2k_params.frame_rows): S.H / fh frames stacked vertically, the tile grid starts again at every frame
    const int fh = S.frame_h > 0 ? S.frame_h : S.H;
    const int tw = S.tile_w > 0 ? S.tile_w : S.W, th = S.tile_h > 0 ? S.tile_h : fh;
    const int tiles_y_frame = (fh + th - 1) / th;
    P->tiles_x = (S.W + tw - 1) / tw;
    P->tiles_y = tiles_y_frame * (S.H / fh);
    const int ntiles_all = P->tiles_x * P->tiles_y;
    P->tile_first = std::min(std::max(S.tile_first, 0), ntiles_all);
    P->tile_count = S.tile_count > 0 ? std::min(S.tile_count, ntiles_all - P->tile_first) : ntiles_all - P->tile_first;
    const int L = S.levels;
    const bool triple = S.mct && S.C >= 3;
    const int esz = S.wavelet == W97 ? 8 : 4;

    // ---- tile-components, coefficient + scratch offsets -------------------------
    int64_t coef = 0, sa = 0, sb = 0;
    for (int tl = 0; tl < P->tile_count; tl++) {
        const int t = P->tile_first + tl;
        const int tx = t % P->tiles_x, ty = t / P->tiles_x;
        const int x0 = tx * tw, y0 = (ty / tiles_y_frame) * fh + (ty % tiles_y_frame) * th;
        const int w = std::min(tw, S.W - x0), h = std::min(th, fh - (ty % tiles_y_frame) * th);
        const int w1 = (w + 1) / 2, h1 = (h + 1) / 2, w2 = (w1 + 1) / 2, h2 = (h1 + 1) / 2;
        int c = 0;
        while (c < S.C) {
            Group g{};
            g.tile = tl; g.comp0 = c; g.nc = (triple && c == 0) ? 3 : 1;
            g.x0 = x0; g.y0 = y0; g.w = w; g.h = h;
            for (int k = 0; k < g.nc; k++) {
                g.coef_off[k] = coef; coef += align4((int64_t)w * h);
                g.scrA_off[k] = sa; sa += align4((int64_t)w1 * h1);
                g.scrB_off[k] = sb; sb += align4((int64_t)w2 * h2);
                const int64_t d[7] = {t, c + k, x0, y0, w, h, g.coef_off[k]};
                P->plane_desc.insert(P->plane_desc.end(), d, d + 7);
            }
            P->groups.push_back(g);
            c += g.nc;
        }
    }
    P->coeff_elems = coef; P->scrA_elems = sa; P->scrB_elems = sb;

    // ---- fused LDS tail for the small levels (5-3 only) --------------------------------
    if (S.wavelet == W53 && ctx->use_tail && L >= 3) {
        for (int l0 = 1; l0 <= L - 2 && P->tail_l0 < 0; l0++) {
            bool ok = true;
            for (const Group &g : P->groups) {
                int w = g.w, h = g.h;
                for (int i = 0; i < l0; i++) { w = (w + 1) / 2; h = (h + 1) / 2; }
                if (w > 128 || (int64_t)w * h > 16384) ok = false;
            }
            if (ok) P->tail_l0 = l0;
        }
        if (P->tail_l0 >= 0) {
            const int l0 = P->tail_l0;
            std::vector<TailPlane> tp;
            for (const Group &g : P->groups) {
                int w = g.w, h = g.h;
                for (int i = 0; i < l0; i++) { w = (w + 1) / 2; h = (h + 1) / 2; }
                const int w1 = (w + 1) / 2, h1 = (h + 1) / 2, w2 = (w1 + 1) / 2, h2 = (h1 + 1) / 2;
                P->tail_lds_fwd = std::max(P->tail_lds_fwd, (size_t)(((w * h + 3) & ~3) + w1 * h1 + 8) * 4);
                P->tail_lds_inv = std::max(P->tail_lds_inv, (size_t)(((w1 * h1 + 3) & ~3) + w2 * h2 + 8) * 4);
                for (int k = 0; k < g.nc; k++) {
                    TailPlane T{};
                    T.scr_off = ((l0 & 1) ? g.scrA_off : g.scrB_off)[k];   // where level l0's input prefix lives
                    T.coef_off = g.coef_off[k];
                    T.w = w; T.h = h; T.nlev = L - l0;
                    tp.push_back(T);
                }
            }
            P->ntail = (int)tp.size();
            int r = upload(ctx, &P->d_tail, tp);
            if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
        }
    }

    // ---- every level below level 0 in one launch per direction (dwt53_deep.inc) ------------------
    // levels deep_l0 .. L-1, deep_l0 = the level above the first one that fits LDS: it streams from memory in the same workgroups
    int lds_l0 = -1;        // the first level whose input fits the LDS buffers (the tail above needs two such levels, this one)
    if (S.wavelet == W53) {
        for (int l0 = 1; l0 <= L - 1 && lds_l0 < 0; l0++) {
            bool fits = true;
            for (const Group &g : P->groups) {
                int w = g.w, h = g.h;
                for (int i = 0; i < l0; i++) { w = (w + 1) / 2; h = (h + 1) / 2; }
                if (w > 128 || (int64_t)w * h > 16384) fits = false;
            }
            if (fits) lds_l0 = l0;
        }
    }
    if (lds_l0 >= 2 && ctx->use_deep) {
        const int l0 = lds_l0 - 1;
        bool ok = true;
        std::vector<TailPlane> tp;
        std::vector<DwtJob> deep_f, flat_f, deep_i, flat_i;      // forward / inverse: the deep and mid jobs own different level-l0 rows
        size_t lds = 0, lds_f = 0;
        for (const Group &g : P->groups) {
            int w = g.w, h = g.h;
            for (int i = 0; i < l0; i++) { w = (w + 1) / 2; h = (h + 1) / 2; }
            if (w < 8 || w > 256 || (w % 4) || h < 2 || h > 256) { ok = false; break; }
            const int w1 = w / 2, h1 = (h + 1) / 2, w2 = (w1 + 1) / 2, h2 = (h1 + 1) / 2;
            if (w1 > 128 || (int64_t)w1 * h1 > 16384) { ok = false; break; }
            const int halfH = h1, T = (halfH + 1) / 2, T1 = (T + 1) / 2;   // pair-rows of level l0; of level l0+1; those feeding level l0+2
            // the top half split once more (dwt53_deep.inc): level l0+1 must take the pair-column LDS routines, whole 16-byte runs
            const bool has_mid = ctx->deep_mid && T >= 2 && (w % 8) == 0 && (w1 & 1) == 0 && (64 % (w1 / 2)) == 0 && h1 >= 2;
            const size_t a_ints = has_mid ? (size_t)(2 * T1 + 2) * w1 : (size_t)w1 * h1;
            const size_t n1a = (a_ints + 3) & ~size_t(3), n2a = (size_t)((w2 * h2 + 3) & ~3), nc = (size_t)((w1 * h1 + 3) & ~3), slots = 16 * 64 * 4;   // (ints)
            lds_f = std::max(lds_f, (std::max(2 * slots, n2a) + n1a + 8) * 4);          // forward: slotE, slotD (later bufB) | bufA
            const size_t n1i = (has_mid && ctx->deep_mid_inv) ? n1a : nc;                 // (the inverse takes the split only on request)
            // inverse, deep + mid + flat (J2K_DEEP_MID_INV=1, the default since round 4): the staged coefficients as compact runs in the
            // places that are free when they are needed (dwt53_deep_inv_body, DEEP_COMPACT) -- bufA | X = bufB + the low-pass rows beyond
            // |X_{l0+2}| (mid: its low-pass rows) | RB = the high-pass rows: 69 KB for a 256 x 256 plane instead of 113, two workgroups per CU
            const size_t nn1 = (L - l0 > 2) ? (size_t)w2 * h2 : 0;
            const bool compact = has_mid && ctx->deep_mid_inv == 1 && (w1 % 4) == 0 && (nn1 % 4) == 0;
            if (compact) {
                const size_t lowtail = (size_t)std::max<int64_t>((int64_t)std::min(T1 + 2, T) * w1 - (int64_t)nn1, 0), lowmid = (size_t)(T - T1) * w1;
                const size_t xsize = (std::max(std::max(n2a + lowtail, lowmid), slots) + 3) & ~size_t(3);
                const size_t rb = (size_t)std::max(std::min(T + T1 + 2, h1) - T, h1 - (T + T1 - 1)) * w1;
                lds = std::max(lds, (n1a + xsize + rb + 8) * 4);
            } else
            lds = std::max(lds, (n1i + n2a + std::max(nc, slots) + 8) * 4);             // inverse: bufA | bufB | bufC (= slotE later)
            const int flag = has_mid ? 0x10000 : 0;
            const int cflag = compact ? 0x20000 : 0;
            for (int k = 0; k < g.nc; k++) {
                const int64_t so = ((l0 & 1) ? g.scrA_off : g.scrB_off)[k];
                if ((so % 4) || (g.coef_off[k] % 4)) ok = false;
                TailPlane T_{};
                T_.scr_off = so; T_.coef_off = g.coef_off[k];
                T_.w = w; T_.h = h; T_.nlev = L - l0;
                const int pi = (int)tp.size();
                tp.push_back(T_);
                // The inverse launch cannot share a CU between two workgroups (113 KB of LDS, 95 registers), so a third job per
                // plane waits for a CU and the split gains nothing there (J2K_DEEP_MID_INV: 0 = deep + flat, 16.2 us on a C2 frame;
                // 1 = deep + mid + flat, 16.5; 2 = deep + mid, each rebuilding half of the bottom rows first, 18.2).  The forward
                // launch fits two per CU (65 KB, 64 registers): 16.3 -> 13.8 us.
                const int mid_inv = has_mid ? ctx->deep_mid_inv : 0;
                if (has_mid) {
                    deep_f.push_back(DwtJob{pi, 1, 0, std::min(T1 + 1, T) | flag});
                    deep_f.push_back(DwtJob{pi, 3, T1 - 1, (T - (T1 - 1)) | flag});
                } else deep_f.push_back(DwtJob{pi, 1, 0, T});
                if (mid_inv == 2) {
                    const int nf = halfH - T, fa = (nf + 1) / 2;
                    deep_i.push_back(DwtJob{pi, 1 | fa << 8, 0, T1 | flag});
                    deep_i.push_back(DwtJob{pi, 3 | (nf - fa) << 8 | fa << 16, T1, (T - T1) | flag});
                } else if (mid_inv == 1) {
                    deep_i.push_back(DwtJob{pi, 1, 0, T1 | flag | cflag});
                    deep_i.push_back(DwtJob{pi, 3, T1, (T - T1) | flag | cflag});
                } else deep_i.push_back(DwtJob{pi, 1, 0, T});
                for (int q = T; q < halfH; q += 64) {
                    flat_f.push_back(DwtJob{pi, 0, q, std::min(64, halfH - q)});
                    if (mid_inv != 2) flat_i.push_back(DwtJob{pi, 0, q, std::min(64, halfH - q)});
                }
            }
        }
        // One deep chain per plane: with a handful of planes (C5: one 2048 x 2048 plane per frame) the per-level launches, which
        // spread every level over the whole device, are at least as fast (C5 at eight frames in flight: 76 Gpixel/s with the chain, 72-80 without)
        if (ok && (int)tp.size() < ctx->deep_min_planes) ok = false;
        if (ok && !tp.empty()) {
            // deep jobs first (the longest chains), mid jobs behind them, then the flat ones
            auto order = [](std::vector<DwtJob> &d, const std::vector<DwtJob> &f) {
                std::stable_sort(d.begin(), d.end(), [](const DwtJob &a, const DwtJob &b) { return (a.col0 & 0xff) < (b.col0 & 0xff); });
                d.insert(d.end(), f.begin(), f.end());
            };
            P->deep_jobs_host = deep_f; P->flat_jobs_host = flat_f;      // (J2K_MEGA's merged launches put level-0 bands between them)
            P->deep_jobs_host_inv = deep_i; P->flat_jobs_host_inv = flat_i;
            order(deep_f, flat_f); order(deep_i, flat_i);
            P->deep_l0 = l0; P->ndeep_jobs = (int)deep_f.size(); P->ndeep_jobs_inv = (int)deep_i.size(); P->deep_lds = lds; P->deep_lds_fwd = lds_f;
            int r = upload(ctx, &P->d_deep_planes, tp);
            if (r == J2K_OK) r = upload(ctx, &P->d_deep_jobs, deep_f);
            if (r == J2K_OK) r = upload(ctx, &P->d_deep_jobs_inv, deep_i);
            if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
        }
    }

    // XCD-aware order for per-workgroup job tables: workgroup b runs on XCD b % 8, in table order.  Each XCD gets a contiguous
    // chunk of the FULL bands (neighbouring bands share halo rows: L2 hits) followed by a chunk of the SHORT ones (the last
    // band of a plane, fewer live waves): a frame whose full bands fill the device a whole number of times then ends with
    // the short workgroups instead of one more round of full ones (4K, 512-tiles, 5-row bands: 2040 full + 40 short workgroups
    // on 512 slots).  `is_short(job)`.
    auto deal_xcd = [](std::vector<DwtJob> &wj, auto is_short) {
        std::vector<DwtJob> full, shrt;
        for (const DwtJob &j : wj) (is_short(j) ? shrt : full).push_back(j);
        std::vector<std::vector<DwtJob>> per(8);
        for (std::vector<DwtJob> *v : {&full, &shrt}) {
            const size_t chunk = (v->size() + 7) / 8;
            for (size_t i = 0; i < v->size(); i++) per[i / std::max<size_t>(chunk, 1)].push_back((*v)[i]);
        }
        size_t m = 0;
        for (auto &v : per) m = std::max(m, v.size());
        std::vector<DwtJob> perm(m * 8, DwtJob{-1, 0, 0, 0});
        for (int x = 0; x < 8; x++)
            for (size_t i = 0; i < per[x].size(); i++) perm[i * 8 + x] = per[x][i];
        wj.swap(perm);
    };

    // ---- per-level launch tables --------------------------------------------------
    for (int cls = 0; cls < 2; cls++) { P->fwd[cls].resize(L); P->inv[cls].resize(L); }
    for (int dir = 0; dir < 2; dir++) {
        for (int l = 0; l < L; l++) {
            for (int cls = 0; cls < 2; cls++) {
                std::vector<DwtPlane> planes;
                std::vector<int> pw, ph;
                bool vec_ok = !ctx->force_novec;
                int maxw = 0;
                for (const Group &g : P->groups) {
                    const bool as_triple = (g.nc == 3 && l == 0);
                    if ((cls == 1) != as_triple) continue;
                    int w = g.w, h = g.h;
                    for (int i = 0; i < l; i++) { w = (w + 1) / 2; h = (h + 1) / 2; }
                    const int wn = (w + 1) / 2, hn = (h + 1) / 2;
                    const int nplanes_here = as_triple ? 1 : g.nc;
                    for (int k0 = 0; k0 < nplanes_here; k0++) {
                        DwtPlane D{};
                        const int kn = as_triple ? 3 : 1;
                        for (int k = 0; k < kn; k++) {
                            const int kk = as_triple ? k : k0;
                            const int64_t frame_off = (int64_t)(g.comp0 + kk) * S.H * S.W + (int64_t)g.y0 * S.W + g.x0;
                            const int64_t *scr_in = (l & 1) ? g.scrA_off : g.scrB_off;    // where level l's input prefix lives
                            const int64_t *scr_out = (l & 1) ? g.scrB_off : g.scrA_off;   // where level l's output prefix goes
                            if (dir == 0) {  // forward
                                D.src_off[k] = (l == 0) ? frame_off : scr_in[kk];
                                D.out_off[k] = g.coef_off[kk];
                                D.nxt_off[k] = scr_out[kk];
                            } else {         // inverse: X_l lives where the forward input of level l lived
                                D.src_off[k] = g.coef_off[kk];
                                D.nxt_off[k] = scr_out[kk];               // X_{l+1}
                                D.out_off[k] = (l == 0) ? frame_off : scr_in[kk];
                            }
                        }
                        D.src_stride = (dir == 0 && l == 0) ? S.W : w;
                        D.out_stride = S.W;
                        D.w = w; D.h = h;
                        D.n_next = (l == L - 1) ? 0 : wn * hn;
                        planes.push_back(D);
                        pw.push_back(w); ph.push_back(h);
                        maxw = std::max(maxw, w);
                    }
                }
                LevelTab &T = (dir == 0 ? P->fwd : P->inv)[cls][l];
                T.ncomp = cls ? 3 : 1;
                T.nplanes = (int)planes.size();
                if (planes.empty()) continue;
                int cpl = pick_cpl(maxw);
                if (l == 0 && ctx->cpl0 > 0 && S.wavelet == W53) cpl = ctx->cpl0;   // tuning knob J2K_CPL0
                if (S.wavelet == W97) cpl = (cls == 1) ? 2 : (maxw >= 192 ? 4 : 2);   // f64: 2 or 4 columns per lane
                for (size_t i = 0; i < planes.size() && vec_ok; i++) {
                    const DwtPlane &D = planes[i];
                    if (D.w % cpl) vec_ok = false;
                    if (l == 0 && (S.W % cpl)) vec_ok = false;
                    for (int k = 0; k < 3; k++)
                        if ((D.src_off[k] % 4) || (D.out_off[k] % 4) || (D.nxt_off[k] % 4)) vec_ok = false;
                }
                if (S.wavelet == W97) vec_ok = false;        // the 9-7 kernels use scalar accesses
                else if (!vec_ok) cpl = 2;
                T.cpl = cpl; T.vec = vec_ok ? 1 : 0;
                const int halo = (S.wavelet == W97 && cpl < 4) ? 2 : 1;
                const int band53 = (dir == 1 && ctx->band_prows_inv > 0) ? ctx->band_prows_inv : ctx->band_prows;
                const int band = (S.wavelet == W97) ? ctx->band_prows_97 : band53;
                auto build_jobs = [&](int band_) {
                    std::vector<DwtJob> jobs;
                    for (size_t i = 0; i < planes.size(); i++) make_jobs(jobs, (int)i, pw[i], ph[i], cpl, band_, halo);
                    if (ctx->xcd_map && jobs.size() >= 64) {
                        // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), so the
                        // wavefronts of one plane -- whose bands share halo rows -- are placed in workgroups = x (mod 8):
                        // the halo re-reads then hit that XCD's L2 instead of going back to HBM.  Speed only.
                        std::vector<std::vector<DwtJob>> per(8);
                        for (const DwtJob &j : jobs) per[j.plane % 8].push_back(j);
                        size_t m = 0;
                        for (auto &v : per) m = std::max(m, v.size());
                        m = (m + 3) & ~size_t(3);
                        std::vector<DwtJob> perm(m * 8, DwtJob{-1, 0, 0, 0});
                        for (int x = 0; x < 8; x++)
                            for (size_t i = 0; i < per[x].size(); i++) perm[((i / 4) * 8 + x) * 4 + (i % 4)] = per[x][i];
                        jobs.swap(perm);
                    }
                    if (S.wavelet == W53 && (dir == 0 ? ctx->fwd_link : ctx->inv_link)) {
                        // link vertically adjacent bands that share a workgroup (4 consecutive jobs): see dwt53_fwd_kernel
                        for (size_t i = 1; i < jobs.size(); i++) {
                            if (i % 4 == 0) continue;
                            DwtJob &a = jobs[i - 1], &b = jobs[i];
                            if (a.plane < 0 || a.plane != b.plane || a.col0 != b.col0) continue;
                            const int w = pw[a.plane], h = ph[a.plane], halfH = (h + 1) / 2;
                            if (w < 2 || h < 2) continue;
                            const int an = a.nprow & 0xffff, bn = b.nprow & 0xffff;
                            if (a.prow0 + an != b.prow0) continue;
                            if (std::min(an, halfH - a.prow0) < 2 || std::min(bn, halfH - b.prow0) < 2) continue;
                            if (2 * b.prow0 + 1 >= h) continue;      // the band below must own a real odd row
                            a.nprow |= J2K_LINK_DOWN;
                            b.nprow |= J2K_LINK_UP;
                        }
                    }
                    return jobs;
                };
                for (size_t i = 0; i < planes.size(); i++) T.alg_bytes += (int64_t)2 * esz * pw[i] * ph[i] * T.ncomp;
                std::vector<DwtJob> jobs = build_jobs(band);
                T.njobs = (int)jobs.size();
                int r = upload(ctx, &T.d_planes, planes);
                if (r == J2K_OK) r = upload(ctx, &T.d_jobs, jobs);
                if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                if (S.wavelet == W53 && vec_ok && ctx->plane_wg > 0 && (cls == 0 || ctx->plane_wg3 || l == 0)) {   // both directions; cls 1 = level 0 of RGB triples: int32 planes (J2K_PLANE_WG3) or RGBA64 pixels
                    // workgroup form for single-component planes (dwt53_plane_wg.inc): whole 16-byte lanes, at least two rows
                    bool ok = true;
                    int multi = 0;
                    for (size_t i = 0; i < planes.size() && ok; i++) {
                        if (pw[i] < 16 || (pw[i] % 8) || ph[i] < 2) ok = false;
                        if (l == 0 && (S.W % 8)) ok = false;                  // packed Gray16 rows: 16-byte lanes of the frame
                        if (pw[i] > 512) multi = 1;
                    }
                    // Where it pays: a lane takes eight columns, so planes narrower than 512 leave lanes idle -- fine while the
                    // level is latency-bound (few waves: the deeper levels of one big plane), a loss against the general
                    // kernels' four-columns-per-lane variant when there are many such planes (C2's level 1: 120 planes of
                    // 256 x 256, measured 13.1 us against 10.4 us in the inverse)
                    if (ok && maxw < 512) {
                        int64_t prows = 0;
                        for (size_t i = 0; i < planes.size(); i++) prows += (ph[i] + 1) / 2;
                        if (prows > 4096) ok = false;
                    }
                    if (ok) {
                        std::vector<DwtJob> pj;
                        const int nr = ctx->plane_wg - 1;
                        for (size_t i = 0; i < planes.size(); i++)
                            for (int c0 = 0; c0 < pw[i]; c0 += 512)
                                for (int pr = 0; pr < (ph[i] + 1) / 2; pr += nr) pj.push_back(DwtJob{(int)i, c0, pr, nr});
                        if (ctx->l0_xcd && pj.size() >= 64) {                 // XCD-aware order, as for the RGBA8 kernels
                            const size_t chunk = (pj.size() + 7) / 8;
                            std::vector<DwtJob> perm(chunk * 8, DwtJob{-1, 0, 0, 0});
                            for (size_t b = 0; b < perm.size(); b++) {
                                const size_t j = (b % 8) * chunk + b / 8;
                                if (j < pj.size()) perm[b] = pj[j];
                            }
                            pj.swap(perm);
                        }
                        T.pnjobs = (int)pj.size(); T.pwaves = ctx->plane_wg; T.pmulti = multi;
                        T.p_pix_only = (cls == 1 && !ctx->plane_wg3);
                        r = upload(ctx, &T.d_pjobs, pj);
                        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                    }
                }
                if (S.wavelet == W97 && cls == 0 && ctx->plane_wg97 > 0) {
                    // single planes of the 9-7 transform in workgroup form (dwt97_l0wg.inc SRC = 1 / 2, dwt97_l0wg_inv.inc): the deeper
                    // levels (float64 scratch in, int32 coefficients), and since round 4 level 0 of one int32 component (gray frames,
                    // frames without the colour transform) and the float64 unit calls (dwt.go:432-473, 551-573): one job per (plane,
                    // band of NW - 3 pair-rows)
                    bool ok = true;
                    for (size_t i = 0; i < planes.size() && ok; i++) {
                        const DwtPlane &D = planes[i];
                        if (pw[i] < 16 || pw[i] > 512 || (pw[i] % 8) || ph[i] < 2) ok = false;
                        if ((D.src_off[0] % 4) || (D.out_off[0] % 4) || (D.nxt_off[0] % 4)) ok = false;
                        if (l == 0 && ((D.src_stride % 4) || (D.out_stride % 4))) ok = false;      // a frame's rows: 16-byte row accesses
                    }
                    if (ok) {
                        std::vector<DwtJob> pj;
                        const int nr = ctx->plane_wg97 - 3;
                        for (size_t i = 0; i < planes.size(); i++)
                            for (int pr = 0; pr < (ph[i] + 1) / 2; pr += nr) pj.push_back(DwtJob{(int)i, 0, pr, nr});
                        if (ctx->l0_xcd && pj.size() >= 64)
                            deal_xcd(pj, [&](const DwtJob &j) { return j.prow0 + nr > (ph[j.plane] + 1) / 2; });
                        T.pnjobs = (int)pj.size(); T.pwaves = ctx->plane_wg97; T.pmulti = 0;
                        r = upload(ctx, &T.d_pjobs, pj);
                        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                    }
                }
                if (dir == 0 && l == 0 && cls == 1 && S.wavelet == W53 && vec_ok && cpl == 8) {
                    // the packed-pixel forward (j2k_plan_forward_rgba8) moves a third of the bytes per row on the read side
                    // and likes shorter bands: its own job table (measured: 3 pair-rows 29.6 us, 5 pair-rows 31.9 us)
                    std::vector<DwtJob> pj = build_jobs(ctx->band_prows_pix);
                    P->fwd_pix_njobs = (int)pj.size();
                    r = upload(ctx, &P->d_fwd_pix_jobs, pj);
                    if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                    // workgroup form (dwt53_l0pix.inc): one job per workgroup of l0_wg waves = l0_wg - 1 pair-rows of one
                    // plane; its geometry contract: one 512-column strip, whole 16-byte lanes, at least two rows
                    bool wg_ok = ctx->l0_wg > 0;
                    for (size_t i = 0; i < planes.size() && wg_ok; i++)
                        if (pw[i] < 16 || pw[i] > 512 || (pw[i] % 8) || ph[i] < 2) wg_ok = false;
                    if (wg_ok) {
                        // one table per direction: the forward kernel measures best with 8 waves per workgroup (7 pair-rows:
                        // 3 halo rows per 14), the inverse with 4 (A/B on one box: forward 22.5-23.0 / 21.8-21.9 us at 4 / 8,
                        // inverse 25.5 / 26.2)
                        // top_only: just the bands that cover the low-pass rows feeding level 1 (the merged launches take the rest)
                        auto split_row = [&](size_t i, int nr) {       // first pair-row of plane i that the top bands of nr rows do not cover
                            const int halfH0 = (ph[i] + 1) / 2, tb = (halfH0 + 1) / 2;
                            return std::min(halfH0, ((tb + nr - 1) / nr) * nr);
                        };
                        auto wg_table = [&](int waves, bool top_only = false, int xcd_group = 0) {
                            std::vector<DwtJob> wj;
                            const int nr = waves - 1;
                            for (size_t i = 0; i < planes.size(); i++)
                                for (int pr = 0; pr < (top_only ? split_row(i, nr) : (ph[i] + 1) / 2); pr += nr) wj.push_back(DwtJob{(int)i, 0, pr, nr});
                            if (ctx->l0_xcd && wj.size() >= 64 && xcd_group > 0) {
                                // XCD-aware in small groups: G consecutive bands (which share halo rows) go to ONE XCD, one after the
                                // other, and the eight XCDs work on eight neighbouring groups at a time -- the halo re-reads still hit
                                // that XCD's L2 (G - 1 of G boundaries) while the device as a whole sweeps memory in order, as the
                                // plain job order does (tools/probe/l0_inv_probe.hip: 5.7 TB/s in plain order, 5.4-5.5 in any order
                                // that gives every XCD a region of its own).  Table position p runs on XCD p % 8.
                                const size_t G = (size_t)xcd_group, ng = (wj.size() + G - 1) / G, rows = (ng + 7) / 8 * G;
                                std::vector<DwtJob> perm(rows * 8, DwtJob{-1, 0, 0, 0});
                                for (size_t g = 0; g < ng; g++)
                                    for (size_t j = 0; j < G && g * G + j < wj.size(); j++) perm[((g / 8) * G + j) * 8 + g % 8] = wj[g * G + j];
                                wj.swap(perm);
                            } else if (ctx->l0_xcd && wj.size() >= 64 && ctx->l0_deal) {
                                deal_xcd(wj, [&](const DwtJob &j) { return j.prow0 + nr > (ph[j.plane] + 1) / 2; });
                            } else if (ctx->l0_xcd && wj.size() >= 64) {
                                // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), so
                                // workgroup b takes job (b % 8) * chunk + b / 8: vertically adjacent bands -- which share three
                                // halo rows -- run on one XCD at about the same time and the re-read is an L2 hit.  Speed only.
                                const size_t chunk = (wj.size() + 7) / 8;
                                std::vector<DwtJob> perm(chunk * 8, DwtJob{-1, 0, 0, 0});
                                for (size_t b = 0; b < perm.size(); b++) {
                                    const size_t j = (b % 8) * chunk + b / 8;
                                    if (j < wj.size()) perm[b] = wj[j];
                                }
                                wj.swap(perm);
                            }
                            return wj;
                        };
                        std::vector<DwtJob> wj = wg_table(ctx->l0_wg);
                        P->fwd_wg_njobs = (int)wj.size();
                        P->fwd_wg_waves = ctx->l0_wg;
                        r = upload(ctx, &P->d_fwd_wg_jobs, wj);
                        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                        const int invw = ctx->l0_wg_invw > 0 ? ctx->l0_wg_invw : ctx->l0_wg;
                        std::vector<DwtJob> ij = wg_table(invw, false, ctx->l0_xcd_group);      // (measured: inverse 25.2 -> 24.5 us at groups of 8; the forward table loses 0.5 us with it)
                        P->inv_wg_njobs = (int)ij.size();
                        P->inv_wg_waves = invw;
                        r = upload(ctx, &P->d_inv_wg_jobs, ij);
                        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                        // merged launches (dwt53_mega_*_kernel): every level below 0 + the level-0 bands that neither feed nor
                        // need them, for frames of RGB triples only whose deep launch starts at level 1
                        if (ctx->mega && P->deep_l0 == 1 && S.C == 3 && planes.size() == P->groups.size() && !P->deep_jobs_host.empty()) {
                            const int order = ctx->mega;       // 1: deep, level-0 bands, flat; 2: deep, flat, level-0 bands
                            for (int d2 = 0; d2 < 2; d2++) {
                                const int nr_top = (d2 == 0 ? ctx->l0_wg : invw) - 1;
                                std::vector<DwtJob> l0b;
                                for (size_t i = 0; i < planes.size(); i++)
                                    for (int pr = split_row(i, nr_top); pr < (ph[i] + 1) / 2; pr += 15) l0b.push_back(DwtJob{(int)i, 2, pr, 15});
                                std::vector<DwtJob> mj = d2 == 0 ? P->deep_jobs_host : P->deep_jobs_host_inv;
                                const std::vector<DwtJob> &fj = d2 == 0 ? P->flat_jobs_host : P->flat_jobs_host_inv;
                                if (order == 2) mj.insert(mj.end(), fj.begin(), fj.end());
                                mj.insert(mj.end(), l0b.begin(), l0b.end());
                                if (order != 2) mj.insert(mj.end(), fj.begin(), fj.end());
                                std::vector<DwtJob> top = wg_table(nr_top + 1, true, d2 == 1 ? ctx->l0_xcd_group : 0);
                                int64_t top_px = 0;
                                for (size_t i = 0; i < planes.size(); i++) top_px += (int64_t)std::min(2 * split_row(i, nr_top), ph[i]) * pw[i];
                                if (d2 == 0) {
                                    P->mega_fwd_njobs = (int)mj.size(); P->fwd_top_njobs = (int)top.size(); P->fwd_top_bytes = top_px * 16;
                                    r = upload(ctx, &P->d_mega_fwd_jobs, mj);
                                    if (r == J2K_OK) r = upload(ctx, &P->d_fwd_top_jobs, top);
                                } else {
                                    P->mega_inv_njobs = (int)mj.size(); P->inv_top_njobs = (int)top.size(); P->inv_top_bytes = top_px * 16;
                                    r = upload(ctx, &P->d_mega_inv_jobs, mj);
                                    if (r == J2K_OK) r = upload(ctx, &P->d_inv_top_jobs, top);
                                }
                                if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                            }
                        }
                        // levels 0 + 1 in one launch (dwt53_fwd_rgba8_wg2_kernel): needs a level 1 (levels >= 2), only RGB
                        // triples in the frame (the level-1 plane table is then three planes per level-0 plane, same order)
                        if (ctx->l0_fuse > 0 && L >= 2 && S.C == 3 && (P->tail_l0 < 0 || P->tail_l0 >= 2) && (P->deep_l0 < 0 || P->deep_l0 >= 2)) {
                            auto xcd = [&](std::vector<DwtJob> &v) {
                                if (!ctx->l0_xcd || v.size() < 64) return;
                                const size_t chunk = (v.size() + 7) / 8;
                                std::vector<DwtJob> perm(chunk * 8, DwtJob{-1, 0, 0, 0});
                                for (size_t b = 0; b < perm.size(); b++) {
                                    const size_t j = (b % 8) * chunk + b / 8;
                                    if (j < v.size()) perm[b] = v[j];
                                }
                                v.swap(perm);
                            };
                            std::vector<DwtJob> fj, rj;
                            const int nr2 = ctx->l0_fuse - 3, nr = ctx->l0_wg - 1;
                            for (size_t i = 0; i < planes.size(); i++) {
                                const int halfH = (ph[i] + 1) / 2, halfH1 = (halfH + 1) / 2;
                                int pr = 0;
                                for (; pr < halfH1; pr += nr2) fj.push_back(DwtJob{(int)i, 0, pr, nr2});
                                for (; pr < halfH; pr += nr) rj.push_back(DwtJob{(int)i, 0, pr, nr});
                            }
                            xcd(fj); xcd(rj);
                            P->fwd_wg2_njobs = (int)fj.size(); P->fwd_wg2_waves = ctx->l0_fuse; P->fwd_wg_rest_njobs = (int)rj.size();
                            r = upload(ctx, &P->d_fwd_wg2_jobs, fj);
                            if (r == J2K_OK && !rj.empty()) r = upload(ctx, &P->d_fwd_wg_rest_jobs, rj);
                            if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                        }
                    }
                }
                if (dir == 0 && l == 0 && cls == 1 && S.wavelet == W97 && S.mct && !S.frame_is_f64 && ctx->l0_wg97 > 0) {
                    // workgroup form of the lossy level 0 (dwt97_l0wg.inc): one job per (plane, band of NW - 3 pair-rows,
                    // component); the three components of a band are neighbours in the table and the whole table is dealt
                    // XCD-aware like the 5-3 one, so the rows they share are L2 hits
                    // (precision <= 16 and Quality < 8192 keep every value the kernel converts inside int32: round_half_away_inrange)
                    bool ok97 = S.precision <= 16 && S.quality > 0 && S.quality < 8192;
                    for (size_t i = 0; i < planes.size() && ok97; i++) {
                        const DwtPlane &D = planes[i];
                        if (pw[i] < 16 || pw[i] > 512 || (pw[i] % 8) || ph[i] < 2 || (S.W % 4)) ok97 = false;
                        for (int k = 0; k < 3; k++)
                            if ((D.src_off[k] % 4) || (D.out_off[k] % 4) || (D.nxt_off[k] % 4)) ok97 = false;
                    }
                    if (ok97) {
                        std::vector<DwtJob> wj;
                        const int nr = ctx->l0_wg97 - 3;
                        for (size_t i = 0; i < planes.size(); i++)
                            for (int pr = 0; pr < (ph[i] + 1) / 2; pr += nr)
                                for (int k = 0; k < 3; k++) wj.push_back(DwtJob{(int)i, k, pr, nr});
                        if (ctx->l0_xcd && wj.size() >= 64)
                            deal_xcd(wj, [&](const DwtJob &j) { return j.prow0 + nr > (ph[j.plane] + 1) / 2; });
                        P->fwd97_wg_njobs = (int)wj.size();
                        P->fwd97_wg_waves = ctx->l0_wg97;
                        r = upload(ctx, &P->d_fwd97_wg_jobs, wj);
                        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                    }
                }
                if (dir == 1 && l == 0 && cls == 1 && S.wavelet == W97 && S.mct && !S.frame_is_f64 && S.quant != Q_NONE && ctx->l0_wg97_inv > 0) {
                    // workgroup form of the lossy inverse level 0 (dwt97_l0wg_inv.inc): one job per (plane, band of NW - 3
                    // pair-rows), all three components in the workgroup; dealt XCD-aware like the forward table
                    bool ok97 = (S.W % 4) == 0;
                    for (size_t i = 0; i < planes.size() && ok97; i++) {
                        const DwtPlane &D = planes[i];
                        if (pw[i] < 16 || pw[i] > 512 || (pw[i] % 8) || ph[i] < 2) ok97 = false;
                        for (int k = 0; k < 3; k++)
                            if ((D.src_off[k] % 4) || (D.out_off[k] % 4) || (D.nxt_off[k] % 4)) ok97 = false;
                    }
                    if (ok97) {
                        std::vector<DwtJob> wj;
                        const int nr = ctx->l0_wg97_inv - 3;
                        for (size_t i = 0; i < planes.size(); i++)
                            for (int pr = 0; pr < (ph[i] + 1) / 2; pr += nr) wj.push_back(DwtJob{(int)i, 0, pr, nr});
                        if (ctx->l0_xcd && wj.size() >= 64)
                            deal_xcd(wj, [&](const DwtJob &j) { return j.prow0 + nr > (ph[j.plane] + 1) / 2; });
                        P->inv97_wg_njobs = (int)wj.size();
                        P->inv97_wg_waves = ctx->l0_wg97_inv;
                        r = upload(ctx, &P->d_inv97_wg_jobs, wj);
                        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
                    }
                }
                if (dir == 0) {
                    P->dwt_bytes += T.alg_bytes;
                    if (l == 0) P->dwt_level0_bytes += T.alg_bytes;
                }
            }
        }
    }
    if (sa) { hipError_t e = hipMalloc(&P->d_scrA, (size_t)sa * esz); if (e != hipSuccess) { j2k_plan_destroy(P); return fail_hip(ctx, e, "hipMalloc scratch A"); } }
    if (sb) { hipError_t e = hipMalloc(&P->d_scrB, (size_t)sb * esz); if (e != hipSuccess) { j2k_plan_destroy(P); return fail_hip(ctx, e, "hipMalloc scratch B"); } }

    // ---- code-block jobs: encoder.go:616-673 per tile, top-left addressing (encoder.go:763-795) ----
    // (closed-loop mode, j2k_params.closed_loop: the same order, but band b of resolution r is the Mallat rectangle of level
    //  numRes-1-r of the plane and its blocks are cut from the band's own origin -- the windows partition the plane)
    {
        int numRes = S.num_res_jobs;
        if (numRes <= 0) numRes = 6;
        const int cbw = S.cb_w > 0 ? S.cb_w : 64, cbh = S.cb_h > 0 ? S.cb_h : 64;
        std::vector<BlockJob> bj;
        int64_t slot = 0, dec = 0;
        size_t gi = 0;
        for (int tl = 0; tl < P->tile_count; tl++) {
            // groups of this tile are contiguous; collect per-component coefficient offsets
            std::vector<int64_t> coff(S.C);
            int w = 0, h = 0;
            for (; gi < P->groups.size() && P->groups[gi].tile == tl; gi++) {
                const Group &g = P->groups[gi];
                for (int k = 0; k < g.nc; k++) coff[g.comp0 + k] = g.coef_off[k];
                w = g.w; h = g.h;
            }
            for (int c = 0; c < S.C; c++)
                for (int r = 0; r < numRes; r++) {
                    const int nb = r == 0 ? 1 : 3;
                    for (int b = 0; b < nb; b++) {
                        const int band = r == 0 ? J2K_BAND_LL : (b == 0 ? J2K_BAND_HL : (b == 1 ? J2K_BAND_LH : J2K_BAND_HH));
                        int bx0 = 0, by0 = 0, bw, bh;
                        if (!S.closed_loop) {
                            const int64_t scale = (int64_t)1 << std::min(numRes - 1 - r, 40);
                            bw = (int)((w + scale - 1) / scale); bh = (int)((h + scale - 1) / scale);
                            if (r > 0) { bw = (bw + 1) / 2; bh = (bh + 1) / 2; }
                        } else {
                            int wl = w, hl = h;
                            for (int i = 0; i < (r == 0 ? numRes - 1 : numRes - 1 - r); i++) { wl = (wl + 1) / 2; hl = (hl + 1) / 2; }
                            const int wn = (wl + 1) / 2, hn = (hl + 1) / 2;
                            if (r == 0) { bw = wl; bh = hl; }
                            else if (band == J2K_BAND_HL) { bx0 = wn; bw = wl - wn; bh = hn; }
                            else if (band == J2K_BAND_LH) { by0 = hn; bw = wn; bh = hl - hn; }
                            else { bx0 = wn; by0 = hn; bw = wl - wn; bh = hl - hn; }
                        }
                        for (int cby = 0; cby * cbh < bh; cby++)
                            for (int cbx = 0; cbx * cbw < bw; cbx++) {
                                const int sx = bx0 + cbx * cbw, sy = by0 + cby * cbh;
                                const int aw = std::min(cbw, bw - cbx * cbw), ah = std::min(cbh, bh - cby * cbh);
                                j2k_block jb{tl * S.C + c, band, sx, sy, aw, ah};
                                P->blocks.push_back(jb);
                                P->block_tile.push_back(tl);
                                P->block_res.push_back(r);
                                P->max_block_h = std::max(P->max_block_h, ah);
                                P->slot_off.push_back((uint64_t)slot);
                                P->dec_off.push_back((uint64_t)dec);
                                BlockJob J{};
                                J.src_off = coff[c] + (int64_t)sy * w + sx;
                                J.out_off = slot;
                                J.stride = w; J.w = aw; J.h = ah; J.band = band;
                                bj.push_back(J);
                                slot += (int64_t)((j2k_block_bound(S.coder, aw, ah) + 15) & ~size_t(15));
                                dec += align4((int64_t)aw * ah);
                                P->block_samples += (int64_t)aw * ah;
                            }
                    }
                }
        }
        P->bytes_cap = slot; P->decoded_elems = dec;
        {   // first job of every tile of the shard (jobs are enumerated tile by tile) and the largest tile's slot bytes
            std::vector<int> job0;
            std::vector<uint64_t> tstart;
            for (size_t i = 0; i < bj.size(); i++)
                if (i == 0 || P->block_tile[i] != P->block_tile[i - 1]) { job0.push_back((int)i); tstart.push_back(P->slot_off[i]); }
            job0.push_back((int)bj.size()); tstart.push_back((uint64_t)slot);
            for (size_t t = 0; t + 1 < tstart.size(); t++) P->max_tile_bytes = std::max(P->max_tile_bytes, tstart[t + 1] - tstart[t]);
            int r0 = upload(ctx, &P->d_tile_job0, job0);
            if (r0 != J2K_OK) { j2k_plan_destroy(P); return r0; }
        }
        int r = upload(ctx, &P->d_bjobs, bj);
        if (r == J2K_OK && S.coder == J2K_CODER_HT && ctx->ht_alias) {
            // Jobs with the same window are byte-identical for the HT coder (top-left addressing + a coder that ignores the
            // band: see ht_encode_kernel): one coded job per distinct window, the others chained to it and gathering from its
            // slot.  Only j2k_plan_encode_stream uses these tables (its slot buffer is private); the per-slot API does not.
            std::map<std::tuple<int64_t, int32_t, int32_t, int32_t>, int> first;
            std::vector<int> ujobs;
            std::vector<std::vector<int>> lists;
            std::vector<BlockJob> aj = bj;
            for (size_t i = 0; i < bj.size(); i++) {
                auto key = std::make_tuple(bj[i].src_off, bj[i].stride, bj[i].w, bj[i].h);
                auto it = first.find(key);
                if (it == first.end()) { first[key] = (int)ujobs.size(); ujobs.push_back((int)i); lists.push_back({(int)i}); }
                else { lists[it->second].push_back((int)i); aj[i].out_off = bj[ujobs[it->second]].out_off; }
            }
            if (ujobs.size() < bj.size()) {
                std::vector<HtUJob> utab;
                std::vector<int> ids;
                for (size_t u = 0; u < ujobs.size(); u++) {
                    utab.push_back(HtUJob{bj[ujobs[u]], ujobs[u], (int)ids.size(), (int)lists[u].size(), 0});
                    ids.insert(ids.end(), lists[u].begin(), lists[u].end());
                }
                P->ht_nunique = (int)ujobs.size();
                r = upload(ctx, &P->d_ht_ujobs, utab);
                if (r == J2K_OK) r = upload(ctx, &P->d_ht_alias_next, ids);
                if (r == J2K_OK) r = upload(ctx, &P->d_bjobs_alias, aj);
            }
        }
        for (size_t i = 0; i < bj.size(); i++) bj[i].out_off = (int64_t)P->dec_off[i];   // decode table: dense decoded blocks
        if (r == J2K_OK) r = upload(ctx, &P->d_djobs, bj);
        if (r != J2K_OK) { j2k_plan_destroy(P); return r; }
    }
    *out = P;
    return J2K_OK;
}

extern "C" void j2k_plan_destroy(j2k_plan *P) {
    if (!P) return;
    if (P->ctx) { (void)hipSetDevice(P->ctx->device); (void)hipStreamSynchronize(P->ctx->stream); }
    for (int cls = 0; cls < 2; cls++) {
        for (auto &T : P->fwd[cls]) { if (T.d_planes) (void)hipFree(T.d_planes); if (T.d_jobs) (void)hipFree(T.d_jobs); if (T.d_pjobs) (void)hipFree(T.d_pjobs); }
        if (cls == 0 && P->d_bigsym_off) { (void)hipFree(P->d_bigsym_off); P->d_bigsym_off = nullptr; }
        for (auto &T : P->inv[cls]) { if (T.d_planes) (void)hipFree(T.d_planes); if (T.d_jobs) (void)hipFree(T.d_jobs); if (T.d_pjobs) (void)hipFree(T.d_pjobs); }
    }
    void *ptrs[] = {P->d_deep_jobs_inv, P->d_mega_fwd_jobs, P->d_mega_inv_jobs, P->d_fwd_top_jobs, P->d_inv_top_jobs, P->d_inv_wg_jobs, P->d_deep_planes, P->d_deep_jobs, P->d_tile_job0, P->d_scrA, P->d_scrB, P->d_tail, P->d_bjobs, P->d_djobs, P->d_frame, P->d_coeff, P->d_slots, P->d_stream, P->d_lens, P->d_numbps, P->d_offs, P->d_status, P->d_fwd_pix_jobs, P->d_fwd_wg2_jobs, P->d_fwd_wg_rest_jobs, P->d_fwd_wg_jobs, P->d_fwd97_wg_jobs, P->d_inv97_wg_jobs, P->d_ht_ujobs, P->d_ht_alias_next, P->d_bjobs_alias, P->d_maglens, P->d_mels, P->d_toffs,
                    P->d_t2_packets, P->d_tile_packet0, P->d_t2_cbs, P->d_t2_poffs, P->d_t2_stream, P->d_t2_ws, P->d_t2_chains, P->d_t2_body_base, P->d_frame_status,
                    P->d_cl_decoded, P->d_cl_coeff, P->d_cl_stream, P->d_cl_numbps, P->d_cl_offs, P->d_cl_lens};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete P;
}

static int spec_from_params(j2k_ctx *ctx, const j2k_params *p, PlanSpec &S) {
    if (!p) return fail(ctx, J2K_ERR_INVALID_ARG, "params == NULL");
    if (p->precision < 1 || p->precision > 31) return fail(ctx, J2K_ERR_INVALID_ARG, "precision out of range");
    S.W = p->width; S.H = p->height; S.C = p->ncomp;
    S.frame_h = p->frame_rows > 0 ? p->frame_rows : p->height;
    if (S.frame_h > 0 && p->height % S.frame_h) return fail(ctx, J2K_ERR_INVALID_ARG, "height is not a whole number of frames (frame_rows)");
    S.tile_w = p->tile_w; S.tile_h = p->tile_h;
    S.levels = p->num_resolutions - 1;
    if (S.levels <= 0) S.levels = 5;                                  // encoder.go:249-252
    S.wavelet = p->lossless ? W53 : W97;
    S.precision = p->precision;
    S.dc_shift = (int)((uint32_t)1 << (p->precision - 1));            // mct.go:97: encoder.preprocess ALWAYS shifts (encoder.go:218-220)
    S.dc_shift_inv = p->is_signed ? 0 : S.dc_shift;                   // decoder.go:344-348: only the decode side skips it for signed components
    S.mct = p->ncomp >= 3;                                            // encoder.go:223
    S.quant = p->lossless ? Q_NONE : Q_ENCODER;
    S.quality = p->quality > 0 ? p->quality : 100;                    // encoder.go:265-268
    S.num_res_jobs = p->num_resolutions > 0 ? p->num_resolutions : 6; // encoder.go:601-604
    S.cb_w = p->cb_w > 0 ? p->cb_w : 64;                              // encoder.go:608-613
    S.cb_h = p->cb_h > 0 ? p->cb_h : 64;
    S.coder = p->coder;
    S.tile_first = p->tile_first; S.tile_count = p->tile_count;
    S.closed_loop = p->closed_loop != 0;
    if (S.coder != J2K_CODER_MQ && S.coder != J2K_CODER_HT) return fail(ctx, J2K_ERR_INVALID_ARG, "coder");
    if (S.coder == J2K_CODER_MQ && !ctx->counted_mq) { ctx->counted_mq = true; g_mq_ctxs.fetch_add(1, std::memory_order_relaxed); }
    return J2K_OK;
}

extern "C" int j2k_plan_create(j2k_ctx *ctx, const j2k_params *params, j2k_plan **out) {
    if (!ctx || !out) return J2K_ERR_INVALID_ARG;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlanSpec S;
    int r = spec_from_params(ctx, params, S);
    if (r != J2K_OK) return r;
    return build_plan(ctx, S, out);
}

extern "C" int j2k_plan_get_info(const j2k_plan *P, j2k_plan_info *info) {
    if (!P || !info) return J2K_ERR_INVALID_ARG;
    info->tiles = P->tile_count;
    info->planes = (int64_t)P->plane_desc.size() / 7;
    info->blocks = (int64_t)P->blocks.size();
    info->coeff_elems = P->coeff_elems;
    info->bytes_cap = P->bytes_cap;
    info->dwt_bytes = P->dwt_bytes;
    info->dwt_level0_bytes = P->dwt_level0_bytes;
    info->block_samples = P->block_samples;
    info->decoded_elems = P->decoded_elems;
    return J2K_OK;
}
extern "C" int j2k_plan_get_blocks(const j2k_plan *P, j2k_block *blocks, size_t cap) {
    if (!P || (!blocks && cap)) return J2K_ERR_INVALID_ARG;
    if (cap < P->blocks.size()) return J2K_ERR_CAPACITY;
    if (!P->blocks.empty()) memcpy(blocks, P->blocks.data(), P->blocks.size() * sizeof(j2k_block));
    return J2K_OK;
}
extern "C" int j2k_plan_get_planes(const j2k_plan *P, int64_t *desc7, size_t cap_planes) {
    if (!P || !desc7) return J2K_ERR_INVALID_ARG;
    if (cap_planes * 7 < P->plane_desc.size()) return J2K_ERR_CAPACITY;
    memcpy(desc7, P->plane_desc.data(), P->plane_desc.size() * sizeof(int64_t));
    return J2K_OK;
}
// ---- Tier-2 packets on device buffers (t2dev.hip) -------------------------------------------------------------------
extern "C" int j2k_t2_encode_packets_device(j2k_ctx *ctx, const j2k_t2_dev_packet *d_packets, size_t npackets, const j2k_t2_dev_cb *d_cbs, size_t ncbs,
                                            const uint8_t *d_data, int sop, int eph, uint8_t *bio_delay, uint8_t *d_out, size_t cap,
                                            uint64_t *d_offs, size_t *total) {
    if (!ctx || !bio_delay || !d_offs || !total || (npackets && !d_packets) || (ncbs && !d_cbs) || (cap && !d_out)) return J2K_ERR_INVALID_ARG;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    if (npackets > ((size_t)1 << 31)) return fail(ctx, J2K_ERR_INVALID_ARG, "too many packets");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t ws = j2k::t2_dev_workspace((long)npackets);
    int r = stage_reserve(ctx, 1, ws + 64);
    if (r != J2K_OK) return r;
    uint64_t *d_res = reinterpret_cast<uint64_t *>((uint8_t *)ctx->stage[1] + ((ws + 15) & ~size_t(15)));
    HIPCHK(ctx, hipMemsetAsync(d_res, 0, 3 * sizeof(uint64_t), ctx->stream));
    HIPCHK(ctx, j2k::launch_t2_encode_packets(ctx->stream, d_packets, (long)npackets, d_cbs, (uint64_t)ncbs, d_data, sop, eph, *bio_delay ? 1 : 0, d_out, (uint64_t)cap,
                                              d_offs, ctx->stage[1], d_res));
    uint64_t res[3] = {0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (res[2]) return fail(ctx, J2K_ERR_GO_PANIC, "packet coder: a tag tree of width 0 (the reference divides by it, t2.go:328,348), or a packet whose code-blocks lie outside the table");
    *total = (size_t)res[0];
    if (res[0] > cap) return fail(ctx, J2K_ERR_CAPACITY, "packet coder: the output buffer is smaller than the packets");
    *bio_delay = res[1] ? 1 : 0;
    return J2K_OK;
}

// one packet per (tile-component, resolution) that has code-blocks: a new one where the plane or the resolution changes.  (A
// resolution whose bands are all empty -- a 1-sample-wide tile has no HL / HH -- has no jobs and so no packet: nothing the
// reference's iterator would have produced stands for it, this table is the library's own.)  Closed-loop plans get the
// J2K_T2_* flags: a new coder object at every tile's first packet.
static void plan_t2_packets(const j2k_plan *P, int layer, std::vector<j2k_t2_dev_packet> &out, std::vector<int> *tile_packet0) {
    const size_t n = P->blocks.size();
    const int cl = P->spec.closed_loop ? (J2K_T2_WIDE_LEN | J2K_T2_SEATED) : 0;
    for (size_t j = 0; j < n; j++) {
        const j2k_block &b = P->blocks[j];
        const bool new_tile = j == 0 || P->block_tile[j] != P->block_tile[j - 1];
        const bool fresh = new_tile || b.plane != P->blocks[j - 1].plane || P->block_res[j] != P->block_res[j - 1];
        if (new_tile && tile_packet0) while ((int)tile_packet0->size() <= P->block_tile[j]) tile_packet0->push_back((int)out.size());
        if (fresh) {
            int cols = 0;                                      // block columns of the precinct's first band
            for (size_t k = j; k < n && P->blocks[k].plane == b.plane && P->block_res[k] == P->block_res[j] && P->blocks[k].band == b.band && P->blocks[k].y0 == b.y0; k++) cols++;
            out.push_back(j2k_t2_dev_packet{layer, cols, cols, cl | ((cl && new_tile) ? J2K_T2_FRESH : 0), (int64_t)j, 0});
        }
        out.back().ncb++;
    }
    if (tile_packet0) while ((int)tile_packet0->size() <= P->tile_count) tile_packet0->push_back((int)out.size());
}
extern "C" int j2k_plan_t2_packets(const j2k_plan *P, int layer, j2k_t2_dev_packet *packets, size_t cap, size_t *count) {
    if (!P || !count || (cap && !packets)) return J2K_ERR_INVALID_ARG;
    std::vector<j2k_t2_dev_packet> out;
    plan_t2_packets(P, layer, out, nullptr);
    *count = out.size();
    if (cap < out.size()) return J2K_ERR_CAPACITY;
    if (!out.empty()) memcpy(packets, out.data(), out.size() * sizeof(j2k_t2_dev_packet));
    return J2K_OK;
}

extern "C" int j2k_plan_t2_fill_cbs(j2k_plan *P, int mb, const uint64_t *d_offs, const uint32_t *d_lens, const uint8_t *d_numbps, j2k_t2_dev_cb *d_cbs) {
    if (!P || !d_offs || !d_lens || !d_numbps || !d_cbs) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, j2k::launch_t2_fill_cbs(ctx->stream, (long)P->blocks.size(), d_offs, d_lens, d_numbps, mb,
                                        (P->spec.coder == J2K_CODER_HT ? 1 : 0) | (P->spec.closed_loop ? 2 : 0), d_cbs));
    return J2K_OK;
}

// PacketDecoder.DecodePacket for a run of packets by one decoder object (t2dec.hip)
extern "C" int j2k_t2_decode_packets_device(j2k_ctx *ctx, const j2k_t2_dev_packet *d_packets, size_t npackets, j2k_t2_dev_cb *d_cbs, size_t ncbs,
                                            const uint8_t *d_data, size_t len, int sop, int eph, j2k_t2_dec_state *st, size_t *packets_done) {
    if (!ctx || !st || !packets_done || (npackets && !d_packets) || (ncbs && !d_cbs) || (len && !d_data)) return J2K_ERR_INVALID_ARG;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    if (npackets > ((size_t)1 << 31)) return fail(ctx, J2K_ERR_INVALID_ARG, "too many packets");
    *packets_done = 0;
    if (!npackets) return J2K_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t cb = (j2k::t2_chain_bytes() + 15) & ~size_t(15);
    int r = stage_reserve(ctx, 1, cb + npackets * 8 + 64);
    if (r != J2K_OK) return r;
    std::vector<uint8_t> h(cb);
    j2k::t2_make_chain(h.data(), (uint64_t)len, (long)npackets, *st);
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[1], h.data(), cb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, j2k::launch_t2_decode_packets(ctx->stream, ctx->stage[1], 1, d_packets, (long)npackets, d_cbs, (uint64_t)ncbs, d_data, sop, eph, 0,
                                              reinterpret_cast<uint64_t *>((uint8_t *)ctx->stage[1] + cb), nullptr));
    HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->stage[1], cb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int status = J2K_OK;
    long done = 0;
    j2k::t2_read_chain(h.data(), *st, status, done);
    *packets_done = (size_t)done;
    if (status != J2K_OK) return fail(ctx, status, status == J2K_ERR_GO_PANIC ? "packet decoder: a tag tree of width 0 (the reference divides by it, t2.go:524,547)"
                                                                              : "packet decoder: out of header bits or body bytes, or a packet whose code-blocks lie outside the table");
    return J2K_OK;
}

// ---- the closed-loop frame codec (j2k_params.closed_loop) -------------------------------------------------------------------
static int cl_prepare(j2k_plan *P) {
    j2k_ctx *ctx = P->ctx;
    if (!P->spec.closed_loop) return fail(ctx, J2K_ERR_UNSUPPORTED, "the plan was not made with j2k_params.closed_loop: the reference's code-block windows overlap and its packets cannot be read back");
    if (P->d_t2_packets) return J2K_OK;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: the frame codec's tables are made at its first call -- run it once before j2k_ctx_capture_begin");
    std::vector<j2k_t2_dev_packet> pk;
    std::vector<int> tp0;
    plan_t2_packets(P, 0, pk, &tp0);
    const size_t n = P->blocks.size(), np = pk.size();
    // per tile: its slots + what the headers can take (inclusion 2 bits, zero bit planes 32, passes 16, length 5 + 32: 11 bytes, 13
    // with every byte stuffed; a packet: presence bit + padding, SOP, EPH)
    uint64_t max_tile = 0, total = 0;
    for (int t = 0; t < P->tile_count; t++) {
        uint64_t b = 0;
        for (int q = tp0[t]; q < tp0[t + 1]; q++)
            for (int64_t j = pk[q].cb0; j < pk[q].cb0 + pk[q].ncb; j++) b += ((j2k_block_bound(P->spec.coder, P->blocks[(size_t)j].w, P->blocks[(size_t)j].h) + 15) & ~size_t(15)) + 16;
        b += 16ull * (uint64_t)(tp0[t + 1] - tp0[t]);
        max_tile = std::max(max_tile, b);
        total += b;
    }
    P->t2_stream_cap = (size_t)total + 64;
    P->max_tile_bytes = std::max<uint64_t>(P->max_tile_bytes, max_tile);
    int r = upload(ctx, &P->d_tile_packet0, tp0);
    auto alloc = [&](void **p, size_t bytes) { if (r == J2K_OK) { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 64)); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMalloc (frame codec)"); } };
    alloc((void **)&P->d_t2_cbs, n * sizeof(j2k_t2_dev_cb));
    alloc((void **)&P->d_t2_poffs, (np + 1) * 8);
    alloc((void **)&P->d_t2_stream, P->t2_stream_cap);
    alloc(&P->d_t2_ws, ((j2k::t2_dev_workspace((long)np) + 15) & ~size_t(15)) + 64);
    alloc(&P->d_t2_chains, (size_t)P->tile_count * j2k::t2_chain_bytes());
    alloc((void **)&P->d_t2_body_base, np * 8);
    alloc((void **)&P->d_frame_status, 64);
    if (r == J2K_OK) { hipError_t e = hipMemsetAsync(P->d_frame_status, 0, 64, ctx->stream); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMemsetAsync"); }
    P->t2_npackets = (int)np;
    if (r == J2K_OK) r = upload(ctx, &P->d_t2_packets, pk);      // (last: its presence says the tables are complete)
    return r;
}

extern "C" size_t j2k_plan_frame_bound(const j2k_plan *P) {
    if (!P || !P->spec.closed_loop) return 0;
    std::vector<j2k_t2_dev_packet> pk;
    plan_t2_packets(P, 0, pk, nullptr);
    return (size_t)P->bytes_cap + 16 * P->blocks.size() + 16 * pk.size() + 14 * (size_t)P->tile_count + 64;
}

extern "C" int j2k_plan_encode_tile_parts(j2k_plan *P, const uint8_t *d_stream, const uint64_t *d_offs, const uint32_t *d_lens, const uint8_t *d_numbps,
                                          int sop, int eph, uint8_t *d_out, size_t cap, uint64_t *d_tile_offs) {
    if (!P) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (!d_stream || !d_offs || !d_lens || !d_numbps || !d_out || !d_tile_offs) return fail(ctx, J2K_ERR_INVALID_ARG, "null device pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int r = cl_prepare(P);
    if (r != J2K_OK) return r;
    const long n = (long)P->blocks.size(), np = P->t2_npackets;
    HIPCHK(ctx, j2k::launch_t2_fill_cbs(ctx->stream, n, d_offs, d_lens, d_numbps, 31, (P->spec.coder == J2K_CODER_HT ? 1 : 0) | 2, P->d_t2_cbs));
    uint64_t *d_res = reinterpret_cast<uint64_t *>((uint8_t *)P->d_t2_ws + ((j2k::t2_dev_workspace(np) + 15) & ~size_t(15)));
    HIPCHK(ctx, hipMemsetAsync(d_res, 0, 3 * sizeof(uint64_t), ctx->stream));
    HIPCHK(ctx, j2k::launch_t2_encode_packets(ctx->stream, P->d_t2_packets, np, P->d_t2_cbs, (uint64_t)n, d_stream, sop, eph, 0, P->d_t2_stream,
                                              (uint64_t)P->t2_stream_cap, P->d_t2_poffs, P->d_t2_ws, d_res));
    HIPCHK(ctx, launch_assemble_tiles(ctx->stream, P->d_t2_stream, P->d_t2_poffs, P->d_tile_packet0, P->tile_count, P->tile_first, P->max_tile_bytes,
                                      d_out, nullptr, (uint64_t)cap, d_tile_offs, P->d_frame_status));
    return J2K_OK;
}

extern "C" int j2k_plan_decode_tile_parts(j2k_plan *P, const uint8_t *d_cs, size_t len, const uint64_t *d_tile_offs, int sop, int eph,
                                          uint64_t *d_offs, uint32_t *d_lens, uint8_t *d_numbps) {
    if (!P) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (!d_cs || !d_offs || !d_lens || !d_numbps) return fail(ctx, J2K_ERR_INVALID_ARG, "null device pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int r = cl_prepare(P);
    if (r != J2K_OK) return r;
    const long n = (long)P->blocks.size();
    HIPCHK(ctx, hipMemsetAsync(P->d_t2_cbs, 0, (size_t)n * sizeof(j2k_t2_dev_cb), ctx->stream));
    HIPCHK(ctx, j2k::launch_t2_tile_chains(ctx->stream, d_cs, (uint64_t)len, d_tile_offs, P->tile_count, P->tile_first, P->d_tile_packet0, P->d_t2_chains));
    HIPCHK(ctx, j2k::launch_t2_decode_packets(ctx->stream, P->d_t2_chains, P->tile_count, P->d_t2_packets, P->t2_npackets, P->d_t2_cbs, (uint64_t)n, d_cs, sop, eph, 1,
                                              P->d_t2_body_base, P->d_frame_status));
    HIPCHK(ctx, j2k::launch_t2_blocks(ctx->stream, n, P->d_t2_cbs, P->spec.coder == J2K_CODER_HT ? 1 : 0, 31, (uint64_t)len, d_offs, d_lens, d_numbps));
    return J2K_OK;
}

extern "C" int j2k_plan_place_blocks(j2k_plan *P, const int32_t *d_decoded, int32_t *d_coeff) {
    if (!P) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (!d_decoded || !d_coeff) return fail(ctx, J2K_ERR_INVALID_ARG, "null device pointer");
    if (!P->spec.closed_loop) return fail(ctx, J2K_ERR_UNSUPPORTED, "the plan was not made with j2k_params.closed_loop: the reference's code-block windows overlap");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, j2k::launch_place_blocks(ctx->stream, P->d_bjobs, P->d_djobs, (int)P->blocks.size(), P->max_block_h, d_decoded, d_coeff));
    return J2K_OK;
}

extern "C" int j2k_plan_frame_status(j2k_plan *P) {
    if (!P) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!P->d_frame_status) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return J2K_OK; }
    int st = 0;
    HIPCHK(ctx, hipMemcpyAsync(&st, P->d_frame_status, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(P->d_frame_status, 0, sizeof st, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (st == J2K_ERR_CAPACITY) return fail(ctx, st, "frame codec: the output buffer is smaller than the tile-parts (the last entry of d_tile_offs says what they take)");
    if (st != J2K_OK) return fail(ctx, st, "frame codec: a malformed tile-part or packet (SOT fields, header bits or body bytes running out)");
    return J2K_OK;
}

static int cl_workspaces(j2k_plan *P) {
    j2k_ctx *ctx = P->ctx;
    if (P->d_cl_coeff) return J2K_OK;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: the frame codec's workspaces are made at its first call");
    int r = J2K_OK;
    const size_t n = P->blocks.size();
    auto alloc = [&](void **p, size_t bytes) { if (r == J2K_OK) { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 64)); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMalloc (frame codec)"); } };
    alloc((void **)&P->d_cl_decoded, (size_t)P->decoded_elems * 4);
    alloc((void **)&P->d_cl_stream, (size_t)P->bytes_cap);
    alloc((void **)&P->d_cl_offs, (n + 1) * 8);
    alloc((void **)&P->d_cl_lens, n * 4 + 16);
    alloc((void **)&P->d_cl_numbps, n + 16);
    alloc((void **)&P->d_cl_coeff, (size_t)P->coeff_elems * 4);
    return r;
}

extern "C" int j2k_plan_encode_frame_pixels(j2k_plan *P, int format, const void *d_pix, size_t stride, int sop, int eph, uint8_t *d_out, size_t cap,
                                            uint64_t *d_tile_offs) {
    if (!P) return J2K_ERR_INVALID_ARG;
    int r = cl_prepare(P);
    if (r == J2K_OK) r = cl_workspaces(P);
    if (r == J2K_OK) r = j2k_plan_forward_pixels(P, format, d_pix, stride, P->d_cl_coeff);
    if (r == J2K_OK) r = j2k_plan_encode_stream(P, P->d_cl_coeff, P->d_cl_stream, P->d_cl_offs, P->d_cl_lens, P->d_cl_numbps);
    if (r == J2K_OK) r = j2k_plan_encode_tile_parts(P, P->d_cl_stream, P->d_cl_offs, P->d_cl_lens, P->d_cl_numbps, sop, eph, d_out, cap, d_tile_offs);
    return r;
}

extern "C" int j2k_plan_decode_frame_pixels(j2k_plan *P, const uint8_t *d_cs, size_t len, const uint64_t *d_tile_offs, int sop, int eph, void *d_pix,
                                            size_t stride) {
    if (!P) return J2K_ERR_INVALID_ARG;
    int r = cl_prepare(P);
    if (r == J2K_OK) r = cl_workspaces(P);
    if (r == J2K_OK) r = j2k_plan_decode_tile_parts(P, d_cs, len, d_tile_offs, sop, eph, P->d_cl_offs, P->d_cl_lens, P->d_cl_numbps);
    if (r == J2K_OK) r = j2k_plan_decode_blocks(P, d_cs, P->d_cl_offs, P->d_cl_lens, P->d_cl_numbps, P->d_cl_decoded);
    if (r == J2K_OK) r = j2k_plan_place_blocks(P, P->d_cl_decoded, P->d_cl_coeff);
    if (r == J2K_OK) r = j2k_plan_inverse_pixels(P, P->d_cl_coeff, d_pix, stride);
    return r;
}

extern "C" int j2k_plan_get_decoded_offsets(const j2k_plan *P, uint64_t *offs, size_t cap) {
    if (!P || !offs) return J2K_ERR_INVALID_ARG;
    if (cap < P->dec_off.size()) return J2K_ERR_CAPACITY;
    if (!P->dec_off.empty()) memcpy(offs, P->dec_off.data(), P->dec_off.size() * sizeof(uint64_t));
    return J2K_OK;
}

// ------------------------------------------------------------------------------
// transform stages on device buffers
// ------------------------------------------------------------------------------
static LevelLaunch mk(const LevelTab &T, int pf = 0) {
    LevelLaunch L{T.d_jobs, T.njobs, T.d_planes, T.cpl, T.vec, T.ncomp, pf};
    L.pjobs = T.d_pjobs; L.pnjobs = T.p_pix_only ? 0 : T.pnjobs; L.pwaves = T.pwaves; L.pmulti = T.pmulti;
    return L;
}

// A packed-pixel frame at level 0 (j2k_plan_forward_pixels / _inverse_pixels; encoder.go:79-179, decoder.go:417-588): stride in PIXELS,
// what the single-component planes read / write (dwt53_plane_wg.inc SRC codes 1 ... 4; 0 = the frame has none) and what the RGB
// triples do (8 = the RGBA8 kernels of dwt53_l0pix.inc / the general kernels, 4 = RGBA64 through the plane kernels; 0 = none).
struct PixIO { int stride = 0, single = 0, triple = 0; };
static void pix_launch(LevelLaunch &L, const LevelTab &T, const PixIO &pix, int cls, const PlanSpec &S) {
    L.pix_stride = pix.stride;
    L.pix_src = cls ? (pix.triple == 4 ? 4 : 0) : pix.single;
    L.comp_elems = (long long)S.W * S.H;
    if (cls == 1 && pix.triple == 4) L.pnjobs = T.pnjobs;      // (a table kept for pixel sources only: mk() hides it)
}

static int plan_forward_impl(j2k_plan *P, const void *d_frame, void *d_coeff, PixIO pix = PixIO()) {
    const int pix_stride = pix.stride;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    if (((uintptr_t)d_frame & 15) || ((uintptr_t)d_coeff & 15)) return fail(ctx, J2K_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double step = 1.0 / (double)S.quality;   // encoder.go:269
    const int nlevel_launches = (P->deep_l0 >= 0) ? P->deep_l0 : ((P->tail_l0 >= 0) ? P->tail_l0 : S.levels);
    bool fused_l1 = false;             // level 1 ran inside the level-0 launch (packed RGBA8 frames, dwt53_fwd_rgba8_wg2_kernel)
    bool mega = false;                 // level 0 ran its top bands only: the merged launch below takes the rest with the deep levels
    for (int l = 0; l < nlevel_launches; l++)
    for (int rep_ = 0; rep_ < dev_reps(l == 0 ? 1 : (l == 1 ? 2 : 4)); rep_++) {      // (always once outside dev builds)
        if (l == 1 && fused_l1) continue;
        void *in = (l == 0) ? const_cast<void *>(d_frame) : ((l & 1) ? P->d_scrA : P->d_scrB);
        void *nx = (l & 1) ? P->d_scrB : P->d_scrA;
        // profiling (bench.py's roofline line): the level-0 dispatch of the RGB triples stamps its own begin / end
        // (the RGB triples' launch when the plan has any, else the single-component one; 5-3 and 9-7 alike)
        const int prof_cls = P->fwd[1][0].njobs ? 1 : 0;
        for (int cls = 0; cls < 2; cls++) {
            const LevelTab &T = P->fwd[cls][l];
            if (!T.njobs) continue;
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (l == 0 ? cls == prof_cls : S.wavelet == W53) profile_pair(ctx, l == 0 ? 0 : 1, ev0, ev1);
            if (S.wavelet == W53) {
                LevelLaunch L = mk(T, ctx->fwd_pf);
                if (l == 0 && pix_stride > 0) {         // packed frame (j2k_plan_forward_rgba8 / _pixels)
                    pix_launch(L, T, pix, cls, S);
                    const bool rgba8 = cls == 1 && pix.triple == 8;
                    if (rgba8 && P->d_fwd_pix_jobs) { L.jobs = P->d_fwd_pix_jobs; L.njobs = P->fwd_pix_njobs; }
                    if (rgba8 && P->d_fwd_wg_jobs) {   // RGBA8: the workgroup form when every plane qualifies
                        L.jobs = P->d_fwd_wg_jobs; L.njobs = P->fwd_wg_njobs; L.wg_waves = P->fwd_wg_waves; L.wg_store = ctx->l0_store;
                        if (P->d_mega_fwd_jobs && !P->d_fwd_wg2_jobs) {   // ... only its top bands: the rest runs beside the deep levels
                            L.jobs = P->d_fwd_top_jobs; L.njobs = P->fwd_top_njobs;
                            mega = true;
                        }
                        if (P->d_fwd_wg2_jobs) {           // ... with level 1 fused into the bands of the top half of every plane
                            L.jobs2 = P->d_fwd_wg2_jobs; L.njobs2 = P->fwd_wg2_njobs; L.wg2_waves = P->fwd_wg2_waves;
                            L.jobs = P->d_fwd_wg_rest_jobs; L.njobs = P->fwd_wg_rest_njobs;
                            L.planes1 = P->fwd[0][1].d_planes; L.nxt1 = (int32_t *)P->d_scrB;
                            fused_l1 = true;
                        }
                    }
                }
                if (ev1) { L.ev_start = ev0; L.ev_stop = ev1; }
                HIPCHK(ctx, launch_dwt53_fwd(ctx->stream, L, (const int32_t *)in, (int32_t *)d_coeff, (int32_t *)nx, l == 0 ? S.dc_shift : 0));
            } else {
                const int src_f64 = (l > 0) || S.frame_is_f64;
                LevelLaunch L97 = mk(T);
                if (l == 0 && cls == 0 && pix.single == 97) L97.pix_stride = pix.stride;      // image.Gray pixels (j2k_plan_forward_pixels)
                if (l == 0 && cls == prof_cls && ev1) { L97.ev_start = ev0; L97.ev_stop = ev1; }
                if (l == 0 && cls == 1 && !src_f64 && P->d_fwd97_wg_jobs) {      // the workgroup form when every plane qualifies
                    L97.jobs = P->d_fwd97_wg_jobs; L97.njobs = P->fwd97_wg_njobs; L97.wg_waves = P->fwd97_wg_waves;
                    if (pix.triple == 97) L97.pix_stride = pix.stride;           // packed RGBA8 frame (j2k_plan_forward_pixels)
                }
                HIPCHK(ctx, launch_dwt97_fwd(ctx->stream, L97, in, src_f64, (int32_t *)d_coeff, (double *)d_coeff, (double *)nx,
                                             l == 0 ? S.dc_shift : 0, S.quant, step, (cls == 1) ? 1 : 0));
            }
        }
    }
    for (int rep_ = 0; mega && rep_ < dev_reps(4); rep_++) {
        hipEvent_t e0, e1;
        profile_pair(ctx, 1, e0, e1);
        HIPCHK(ctx, launch_dwt53_mega_fwd(ctx->stream, P->d_mega_fwd_jobs, P->mega_fwd_njobs, P->d_deep_planes, P->fwd[1][0].d_planes, P->deep_lds_fwd,
                                          (const int32_t *)P->d_scrA, (int32_t *)d_coeff, (const uint32_t *)d_frame, (int32_t *)P->d_scrA,
                                          S.dc_shift, pix_stride, e0, e1));
    }
    for (int rep_ = 0; !mega && P->deep_l0 >= 0 && rep_ < dev_reps(4); rep_++) {
        hipEvent_t e0, e1;
        profile_pair(ctx, 1, e0, e1);
        HIPCHK(ctx, launch_dwt53_deep_fwd(ctx->stream, P->d_deep_jobs, P->ndeep_jobs, P->d_deep_planes, P->deep_lds_fwd,
                                          (const int32_t *)((P->deep_l0 & 1) ? P->d_scrA : P->d_scrB), (int32_t *)d_coeff, e0, e1));
    }
    for (int rep_ = 0; P->deep_l0 < 0 && P->tail_l0 >= 0 && rep_ < dev_reps(4); rep_++) {
        hipEvent_t e0, e1;
        profile_pair(ctx, 1, e0, e1);
        HIPCHK(ctx, launch_dwt53_tail_fwd(ctx->stream, P->d_tail, P->ntail, P->tail_lds_fwd,
                                          (const int32_t *)((P->tail_l0 & 1) ? P->d_scrA : P->d_scrB), (int32_t *)d_coeff, e0, e1));
    }
    return J2K_OK;
}

static int plan_inverse_impl(j2k_plan *P, const void *d_coeff, void *d_frame, PixIO pix = PixIO()) {
    const int pix_stride = pix.stride;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    if (((uintptr_t)d_frame & 15) || ((uintptr_t)d_coeff & 15)) return fail(ctx, J2K_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool mega = P->d_mega_inv_jobs && pix.triple == 8 && pix_stride > 0 && ctx->l0_wg_inv && S.wavelet == W53;
    for (int rep_ = 0; mega && rep_ < dev_reps(0x100); rep_++) {
        hipEvent_t e0, e1;
        profile_pair(ctx, 3, e0, e1);
        HIPCHK(ctx, launch_dwt53_mega_inv(ctx->stream, P->d_mega_inv_jobs, P->mega_inv_njobs, P->d_deep_planes, P->inv[1][0].d_planes, P->deep_lds,
                                          (const int32_t *)d_coeff, (int32_t *)P->d_scrA, (const int32_t *)P->d_scrA, (uint32_t *)d_frame,
                                          S.dc_shift_inv, pix_stride, e0, e1));
    }
    for (int rep_ = 0; !mega && P->deep_l0 >= 0 && rep_ < dev_reps(0x100); rep_++) {
        hipEvent_t e0, e1;
        profile_pair(ctx, 3, e0, e1);
        HIPCHK(ctx, launch_dwt53_deep_inv(ctx->stream, P->d_deep_jobs_inv, P->ndeep_jobs_inv, P->d_deep_planes, P->deep_lds, (const int32_t *)d_coeff,
                                          (int32_t *)((P->deep_l0 & 1) ? P->d_scrA : P->d_scrB), e0, e1));
    }
    for (int rep_ = 0; P->deep_l0 < 0 && P->tail_l0 >= 0 && rep_ < dev_reps(0x100); rep_++) {
        hipEvent_t e0, e1;
        profile_pair(ctx, 3, e0, e1);
        HIPCHK(ctx, launch_dwt53_tail_inv(ctx->stream, P->d_tail, P->ntail, P->tail_lds_inv, (const int32_t *)d_coeff,
                                          (int32_t *)((P->tail_l0 & 1) ? P->d_scrA : P->d_scrB), e0, e1));
    }
    for (int l = ((P->deep_l0 >= 0) ? P->deep_l0 : ((P->tail_l0 >= 0) ? P->tail_l0 : S.levels)) - 1; l >= 0; l--)
    for (int rep_ = 0; rep_ < dev_reps(l == 0 ? 0x400 : (l == 1 ? 0x200 : 0x100)); rep_++) {
        void *prev = (l & 1) ? P->d_scrB : P->d_scrA;                     // X_{l+1}
        void *dst = (l == 0) ? d_frame : ((l & 1) ? P->d_scrA : P->d_scrB);  // X_l
        for (int ci = 0; ci < 2; ci++) {
            // (a packed frame: the triples first -- their kernels write whole pixels, a fourth component then puts its bytes in)
            const int cls = (l == 0 && pix_stride > 0) ? 1 - ci : ci;
            const LevelTab &T = P->inv[cls][l];
            if (!T.njobs) continue;
            if (S.wavelet == W53) {
                LevelLaunch L = mk(T);
                if (l == 0 && pix_stride > 0) pix_launch(L, T, pix, cls, S);  // packed frame (j2k_plan_inverse_rgba8 / _pixels)
                if (l == 0 && cls == 1 && pix.triple == 8 && pix_stride > 0 && P->d_inv_wg_jobs && ctx->l0_wg_inv) {
                    // RGBA8: the workgroup form when every plane qualifies (same job table as the forward: the plane order
                    // of the inverse level table is the forward one)
                    L.jobs = P->d_inv_wg_jobs; L.njobs = P->inv_wg_njobs; L.wg_waves = P->inv_wg_waves; L.wg_store = ctx->l0_inv_wpe;
                    if (mega) { L.jobs = P->d_inv_top_jobs; L.njobs = P->inv_top_njobs; }   // the bottom bands ran beside the deep levels
                }
                profile_pair(ctx, l == 0 ? 2 : 3, L.ev_start, L.ev_stop);
                HIPCHK(ctx, launch_dwt53_inv(ctx->stream, L, (const int32_t *)d_coeff, (const int32_t *)prev, (int32_t *)dst,
                                             l == 0 ? S.dc_shift_inv : 0, l == 0));
            } else {
                // dst_mode: 0 = f64 scratch (l>0), 1 = f64 frame (unit calls), 2 = int32 frame via int32(v+0.5) (tcd.go:433-435)
                const int dst_mode = (l > 0) ? 0 : (S.frame_is_f64 ? 1 : 2);
                LevelLaunch L97 = mk(T);
                if (l == 0 && cls == 0 && pix.single == 97) L97.pix_stride = pix.stride;      // image.Gray pixels (j2k_plan_inverse_pixels)
                if (l == 0 && cls == 1 && dst_mode == 2 && P->d_inv97_wg_jobs) {   // the workgroup form when every plane qualifies
                    L97.jobs = P->d_inv97_wg_jobs; L97.njobs = P->inv97_wg_njobs; L97.wg_waves = P->inv97_wg_waves;
                    if (pix.triple == 97) L97.pix_stride = pix.stride;           // packed RGBA8 frame (j2k_plan_inverse_pixels)
                }
                HIPCHK(ctx, launch_dwt97_inv(ctx->stream, L97, d_coeff, S.quant == Q_NONE ? 1 : 0, (const double *)prev, dst,
                                             l == 0 ? S.dc_shift_inv : 0, l == 0, dst_mode, (cls == 1) ? 1 : 0));
            }
        }
    }
    return J2K_OK;
}

extern "C" int j2k_plan_forward(j2k_plan *P, const int32_t *d_frame, int32_t *d_coeff) {
    if (!P || !d_frame || !d_coeff) return J2K_ERR_INVALID_ARG;
    return plan_forward_impl(P, d_frame, d_coeff);
}
extern "C" int j2k_plan_inverse(j2k_plan *P, const int32_t *d_coeff, int32_t *d_frame) {
    if (!P || !d_frame || !d_coeff) return J2K_ERR_INVALID_ARG;
    return plan_inverse_impl(P, d_coeff, d_frame);
}

// ------------------------------------------------------------------------------
// block coding stages on device buffers
// ------------------------------------------------------------------------------
static int ensure(j2k_ctx *ctx, void **p, size_t bytes) {
    if (*p) return J2K_OK;
    HIPCHK(ctx, hipMalloc(p, std::max<size_t>(bytes, 16)));
    return J2K_OK;
}

// ---- stand-alone coders (mqc.go): host buffers through the staging slots ------------------------
// slot 0: inputs (a | b), slot 1: output, slot 3: {fault, out_len}
static int coder_call(j2k_ctx *ctx, int which, const uint8_t *a, size_t na, const uint8_t *b, size_t nb, size_t n, uint8_t *out, size_t cap,
                      size_t *out_len) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t na16 = (na + 15) & ~size_t(15);
    int r = stage_reserve(ctx, 0, na16 + nb + 64);
    if (r == J2K_OK) r = stage_reserve(ctx, 1, cap + 64);
    if (r == J2K_OK) r = stage_reserve(ctx, 3, 256);
    if (r != J2K_OK) return r;
    uint8_t *d_a = (uint8_t *)ctx->stage[0], *d_b = d_a + na16, *d_out = (uint8_t *)ctx->stage[1];
    int *d_fault = (int *)ctx->stage[3] + 16;                                    // bytes 64..: word 0 is the plans' sticky fault word
    uint32_t *d_len = (uint32_t *)ctx->stage[3] + 20;
    HIPCHK(ctx, hipMemsetAsync(d_fault, 0, 32, ctx->stream));
    if (na) HIPCHK(ctx, hipMemcpyAsync(d_a, a, na, hipMemcpyHostToDevice, ctx->stream));
    if (nb) HIPCHK(ctx, hipMemcpyAsync(d_b, b, nb, hipMemcpyHostToDevice, ctx->stream));
    switch (which) {
    case 0: HIPCHK(ctx, launch_mq_encode(ctx->stream, d_a, d_b, n, d_out, cap, d_len, d_fault)); break;
    case 1: HIPCHK(ctx, launch_mq_decode(ctx->stream, d_a, na, d_b, n, d_out, d_fault)); break;
    case 2: HIPCHK(ctx, launch_raw_encode(ctx->stream, d_a, n, d_out, cap, d_len, d_fault)); break;
    default: HIPCHK(ctx, launch_raw_decode(ctx->stream, d_a, na, n, d_out)); break;
    }
    int h[8] = {0};
    HIPCHK(ctx, hipMemcpyAsync(h, d_fault, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0] == 3) return fail(ctx, J2K_ERR_GO_PANIC, "context index out of range (the Go coder panics)");
    if (h[0] == 2) return fail(ctx, J2K_ERR_CAPACITY, "output buffer too small");
    const size_t produced = (which == 0 || which == 2) ? (size_t)(uint32_t)h[4] : n;
    if (out_len) *out_len = produced;
    if (produced) HIPCHK(ctx, hipMemcpy(out, d_out, std::min(produced, cap), hipMemcpyDeviceToHost));
    return J2K_OK;
}

extern "C" int j2k_mq_encode(j2k_ctx *ctx, const uint8_t *ctxs, const uint8_t *decisions, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (!ctx || !out_len || (n && (!ctxs || !decisions)) || (cap && !out)) return J2K_ERR_INVALID_ARG;
    return coder_call(ctx, 0, ctxs, n, decisions, n, n, out, cap, out_len);
}
extern "C" int j2k_mq_decode(j2k_ctx *ctx, const uint8_t *data, size_t len, const uint8_t *ctxs, size_t n, uint8_t *decisions) {
    if (!ctx || (len && !data) || (n && (!ctxs || !decisions))) return J2K_ERR_INVALID_ARG;
    return coder_call(ctx, 1, data, len, ctxs, n, n, decisions, n, nullptr);
}
extern "C" int j2k_raw_encode(j2k_ctx *ctx, const uint8_t *bits, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (!ctx || !out_len || (n && !bits) || (cap && !out)) return J2K_ERR_INVALID_ARG;
    return coder_call(ctx, 2, bits, n, nullptr, 0, n, out, cap, out_len);
}
extern "C" int j2k_raw_decode(j2k_ctx *ctx, const uint8_t *data, size_t len, size_t n, uint8_t *bits) {
    if (!ctx || (len && !data) || (n && !bits)) return J2K_ERR_INVALID_ARG;
    return coder_call(ctx, 3, data, len, nullptr, 0, n, bits, n, nullptr);
}

// ---- pixels at native width (encoder.go:79-213, decoder.go:417-588) ----------------------------
static const int kPixComp[6] = {1, 1, 3, 3, 4, 4}, kPixPrec[6] = {8, 16, 8, 16, 8, 16}, kPixBytes[6] = {1, 2, 4, 8, 4, 8};
extern "C" int j2k_pixels_components(int format) { return (format >= 0 && format < 6) ? kPixComp[format] : 0; }
extern "C" int j2k_pixels_precision(int format) { return (format >= 0 && format < 6) ? kPixPrec[format] : 0; }

static int pix_args_ok(j2k_ctx *ctx, int format, const void *pix, size_t stride, int w, int h, int target_precision) {
    if (!ctx || !pix || format < 0 || format >= 6 || w < 0 || h < 0 || target_precision < 0 || target_precision > 16)
        return J2K_ERR_INVALID_ARG;
    if (stride < (size_t)w * kPixBytes[format]) return J2K_ERR_INVALID_ARG;
    return J2K_OK;
}

extern "C" int j2k_unpack_pixels(j2k_ctx *ctx, int format, const void *d_pix, size_t stride, int w, int h, int target_precision,
                                 int32_t *d_planes) {
    int r = pix_args_ok(ctx, format, d_pix, stride, w, h, target_precision);
    if (r != J2K_OK || !d_planes) return ctx ? fail(ctx, J2K_ERR_INVALID_ARG, "j2k_unpack_pixels: bad argument") : J2K_ERR_INVALID_ARG;
    if ((format == J2K_PIX_RGBA8 || format == J2K_PIX_NRGBA8) && (((uintptr_t)d_pix | stride) & 3))
        return fail(ctx, J2K_ERR_INVALID_ARG, "RGBA rows must be 4-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int src_max = (1 << kPixPrec[format]) - 1;
    const int dst_max = (target_precision > 0 && target_precision != kPixPrec[format]) ? (1 << target_precision) - 1 : src_max;
    HIPCHK(ctx, launch_unpack_pixels(ctx->stream, (const uint8_t *)d_pix, stride, format, w, h, src_max, dst_max, d_planes));
    return J2K_OK;
}

extern "C" int j2k_pack_pixels(j2k_ctx *ctx, const int32_t *d_planes, int ncomp, int precision, int w, int h, void *d_pix, size_t stride) {
    if (!ctx || !d_planes || !d_pix || w < 0 || h < 0 || precision < 1 || precision > 16) return J2K_ERR_INVALID_ARG;
    if (ncomp != 1 && ncomp != 3 && ncomp != 4) return fail(ctx, J2K_ERR_UNSUPPORTED, "unsupported number of components");   // decoder.go:583-585
    const size_t bpp = (ncomp == 1 ? 1 : 4) * (precision > 8 ? 2 : 1);
    if (stride < (size_t)w * bpp) return fail(ctx, J2K_ERR_INVALID_ARG, "stride smaller than a row");
    if (ncomp != 1 && precision <= 8 && (((uintptr_t)d_pix | stride) & 3)) return fail(ctx, J2K_ERR_INVALID_ARG, "RGBA rows must be 4-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_pack_pixels(ctx->stream, d_planes, ncomp, precision, w, h, (uint8_t *)d_pix, stride));
    return J2K_OK;
}

extern "C" int j2k_extract_image_data(j2k_ctx *ctx, int format, const void *pix, size_t stride, int w, int h, int target_precision,
                                      int32_t *const *planes) {
    int r = pix_args_ok(ctx, format, pix, stride, w, h, target_precision);
    if (r != J2K_OK || !planes) return ctx ? fail(ctx, J2K_ERR_INVALID_ARG, "j2k_extract_image_data: bad argument") : J2K_ERR_INVALID_ARG;
    const size_t n = (size_t)w * h;
    if (!n) return J2K_OK;
    const int nc = kPixComp[format];
    const size_t pbytes = (((size_t)h * stride) + 15) & ~size_t(15);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    r = stage_reserve(ctx, 0, pbytes + 64);                    // pixels cross PCIe at native width
    if (r == J2K_OK) r = stage_reserve(ctx, 1, n * 4 * nc + 64);
    if (r != J2K_OK) return r;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[0], pix, (size_t)(h - 1) * stride + (size_t)w * kPixBytes[format], hipMemcpyHostToDevice, ctx->stream));
    r = j2k_unpack_pixels(ctx, format, ctx->stage[0], stride, w, h, target_precision, (int32_t *)ctx->stage[1]);
    if (r != J2K_OK) return r;
    for (int c = 0; c < nc; c++)
        HIPCHK(ctx, hipMemcpyAsync(planes[c], (int32_t *)ctx->stage[1] + (size_t)c * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return J2K_OK;
}

extern "C" int j2k_create_image(j2k_ctx *ctx, const int32_t *const *planes, int ncomp, int precision, int w, int h, void *pix, size_t stride) {
    if (!ctx || !planes || !pix || w < 0 || h < 0 || precision < 1 || precision > 16) return J2K_ERR_INVALID_ARG;
    if (ncomp != 1 && ncomp != 3 && ncomp != 4) return fail(ctx, J2K_ERR_UNSUPPORTED, "unsupported number of components");
    const size_t n = (size_t)w * h;
    if (!n) return J2K_OK;
    const size_t bpp = (ncomp == 1 ? 1 : 4) * (precision > 8 ? 2 : 1);
    if (stride < (size_t)w * bpp) return fail(ctx, J2K_ERR_INVALID_ARG, "stride smaller than a row");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int r = stage_reserve(ctx, 0, (size_t)h * stride + 64);
    if (r == J2K_OK) r = stage_reserve(ctx, 1, n * 4 * ncomp + 64);
    if (r != J2K_OK) return r;
    for (int c = 0; c < ncomp; c++)
        HIPCHK(ctx, hipMemcpyAsync((int32_t *)ctx->stage[1] + (size_t)c * n, planes[c], n * 4, hipMemcpyHostToDevice, ctx->stream));
    // bytes of a row past w*bpp (stride padding) are not the image's: copy the rows back one by one
    r = j2k_pack_pixels(ctx, (const int32_t *)ctx->stage[1], ncomp, precision, w, h, ctx->stage[0], stride);
    if (r != J2K_OK) return r;
    HIPCHK(ctx, hipMemcpy2DAsync(pix, stride, ctx->stage[0], stride, (size_t)w * bpp, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return J2K_OK;
}

// ---- colour conversions (colorspace.go:54-480) ---------------------------------------------------
static bool cs_applies(int cs, int ncomp) {
    switch (cs) {
    case J2K_CS_CMYK: case J2K_CS_YCCK: return ncomp >= 4;
    case J2K_CS_SYCC: case J2K_CS_EYCC: case J2K_CS_YCBCR2: case J2K_CS_YCBCR3: case J2K_CS_PHOTOYCC: case J2K_CS_CMY:
    case J2K_CS_CIELAB: case J2K_CS_CIEJAB: case J2K_CS_ESRGB: case J2K_CS_ROMMRGB: case J2K_CS_YPBPR60: case J2K_CS_YPBPR50:
        return ncomp >= 3;
    default: return false;
    }
}

extern "C" int j2k_convert_colorspace_device(j2k_ctx *ctx, int cs, int32_t *d_planes, int ncomp, size_t n, int precision) {
    if (!ctx || (n && !d_planes) || ncomp < 0 || precision < 1 || precision > 31) return J2K_ERR_INVALID_ARG;
    if (!cs_applies(cs, ncomp) || !n) return J2K_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_colorspace(ctx->stream, cs, d_planes, ncomp, n, precision));
    return J2K_OK;
}

extern "C" int j2k_convert_colorspace(j2k_ctx *ctx, int cs, int32_t *const *planes, int ncomp, size_t n, int precision) {
    if (!ctx || (n && ncomp > 0 && !planes) || ncomp < 0 || precision < 1 || precision > 31) return J2K_ERR_INVALID_ARG;
    if (!cs_applies(cs, ncomp) || !n) return J2K_OK;
    const int nc = (cs == J2K_CS_CMYK || cs == J2K_CS_YCCK) ? 4 : 3;      // the conversions touch only these
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int r = stage_reserve(ctx, 0, n * 4 * nc + 64);
    if (r != J2K_OK) return r;
    int32_t *d = (int32_t *)ctx->stage[0];
    for (int c = 0; c < nc; c++) HIPCHK(ctx, hipMemcpyAsync(d + (size_t)c * n, planes[c], n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, launch_colorspace(ctx->stream, cs, d, nc, n, precision));
    for (int c = 0; c < 3; c++) HIPCHK(ctx, hipMemcpyAsync(planes[c], d + (size_t)c * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return J2K_OK;
}

// Can the level-0 5-3 kernels read / write the packed pixels themselves?  bps = bytes per sample (1, 2), channels = 1 (Gray, Gray16) or 4
// (RGBA, NRGBA, RGBA64, NRGBA64).  The plan's precision must be the format's own (no rescale: encoder.go:196-210) and unsigned; single
// components need the workgroup form of dwt53_plane_wg.inc (Gray16 also has a general kernel), 8-bit triples the packed-RGBA8 kernels,
// 16-bit triples the plane kernels' NC = 3 form.
static bool pix_fusable(const j2k_plan *P, int bps, int channels, const void *d_pix, size_t stride, bool inverse, PixIO &io) {
    const PlanSpec &S = P->spec;
    const int prec = 8 * bps, pb = bps * channels;
    if (S.wavelet == W97) {
        // the lossy path -- the reference's default (jpeg2000.go:305-316) -- for image.RGBA at 8 bit: the workgroup kernels of level 0 read /
        // write the pixels (dwt97_l0wg.inc SRC 3, dwt97_l0wg_inv.inc PIX)
        if (P->ctx->pix_fuse != 1 || bps != 1 || S.frame_is_f64 || S.precision != 8 || S.quant != Q_ENCODER) return false;
        if ((inverse ? S.dc_shift_inv : S.dc_shift) != 128 || S.levels < 1 || (S.W % 8)) return false;
        if ((((uintptr_t)d_pix | stride) & 15) || stride < (size_t)S.W * (size_t)channels) return false;
        const LevelTab &T0 = (inverse ? P->inv : P->fwd)[0][0], &T1 = (inverse ? P->inv : P->fwd)[1][0];
        if (channels == 1 && S.C == 1) {                 // image.Gray: the single-plane workgroup kernels (SRC 2 / DSTI32 with a pixel stride)
            if (!T0.njobs || T1.njobs || !(T0.pnjobs > 0 && T0.pwaves == 8)) return false;
            io = PixIO();
            io.stride = (int)stride;
            io.single = 97;
            return true;
        }
        if (channels != 4 || S.C != 3 || !S.mct) return false;
        if (inverse ? !(P->d_inv97_wg_jobs && P->inv97_wg_waves == 8) : !(P->d_fwd97_wg_jobs && P->fwd97_wg_waves == 8)) return false;
        if (T0.njobs) return false;
        io = PixIO();
        io.stride = (int)(stride / 4);
        io.triple = 97;
        return true;
    }
    if (S.wavelet != W53 || S.levels < 1 || S.precision != prec || (inverse ? S.dc_shift_inv : S.dc_shift) != (1 << (prec - 1))) return false;
    if (P->tail_l0 == 0 || (S.W % 8)) return false;
    if (P->ctx->pix_fuse == 0 || (P->ctx->pix_fuse == 2 && !(bps == 1 && channels == 4 && S.C == 3) && !(bps == 2 && channels == 1))) return false;
    if ((((uintptr_t)d_pix | stride) & 15) || stride < (size_t)S.W * pb) return false;
    const LevelTab &T0 = (inverse ? P->inv : P->fwd)[0][0], &T1 = (inverse ? P->inv : P->fwd)[1][0];
    if (!T0.njobs && !T1.njobs) return false;
    io = PixIO();
    io.stride = (int)(stride / pb);
    if (T0.njobs) {
        io.single = channels == 1 ? (bps == 2 ? 1 : 2) : (bps == 2 ? 4 : 3);
        const bool wg = T0.pnjobs > 0 && !T0.p_pix_only;
        if (io.single == 1 ? !(wg || (T0.vec && T0.cpl == 8)) : !(wg && T0.pwaves == 4)) return false;
        if (io.single == 1 && T1.njobs) return false;
    }
    if (T1.njobs) {
        if (channels != 4 || !S.mct) return false;
        if (bps == 1) { if (!T1.vec || T1.cpl != 8) return false; io.triple = 8; }
        else { if (!T1.pnjobs || T1.pwaves != 4) return false; io.triple = 4; }
    }
    if (inverse && channels == 4 && !T1.njobs && S.C != 4) return false;      // three components on their own leave alpha unwritten
    // The single-plane inverse kernel writes one channel of a packed pixel by reading and rewriting the whole pixel (dwt53_plane_wg.inc,
    // DST 3 / 4): safe only while at most ONE single-component plane per pixel is in the launch -- the alpha plane beside an MCT
    // triple.  A four-channel frame whose components are all single planes (no colour transform) would lose updates.
    if (inverse && channels == 4 && T0.njobs && !(T1.njobs && S.C - 3 <= 1)) return false;
    return true;
}

extern "C" int j2k_plan_forward_rgba8(j2k_plan *P, const void *d_pix, size_t stride, int32_t *d_coeff) {
    if (!P || !d_pix || !d_coeff) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    if (S.C != 3) return fail(ctx, J2K_ERR_INVALID_ARG, "j2k_plan_forward_rgba8 needs a 3-component plan");
    if (stride < (size_t)S.W * 4 || (((uintptr_t)d_pix | stride) & 3)) return fail(ctx, J2K_ERR_INVALID_ARG, "bad RGBA8 stride / alignment");
    PixIO io;
    if (S.mct && pix_fusable(P, 1, 4, d_pix, stride, false, io)) return plan_forward_impl(P, d_pix, d_coeff, io);
    int r = stage_reserve(ctx, 0, (size_t)S.W * S.H * 3 * 4 + 64);          // int32 staging frame
    if (r != J2K_OK) return r;
    r = j2k_unpack_pixels(ctx, J2K_PIX_RGBA8, d_pix, stride, S.W, S.H, 0, (int32_t *)ctx->stage[0]);
    if (r != J2K_OK) return r;
    return plan_forward_impl(P, ctx->stage[0], d_coeff);
}

extern "C" int j2k_plan_inverse_rgba8(j2k_plan *P, const int32_t *d_coeff, void *d_pix, size_t stride) {
    if (!P || !d_pix || !d_coeff) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    if (S.C != 3) return fail(ctx, J2K_ERR_INVALID_ARG, "j2k_plan_inverse_rgba8 needs a 3-component plan");
    if (S.dc_shift_inv != 128) return fail(ctx, J2K_ERR_UNSUPPORTED, "j2k_plan_inverse_rgba8 needs an unsigned 8-bit plan");
    if (stride < (size_t)S.W * 4 || (((uintptr_t)d_pix | stride) & 3)) return fail(ctx, J2K_ERR_INVALID_ARG, "bad RGBA8 stride / alignment");
    PixIO io;
    if (S.mct && pix_fusable(P, 1, 4, d_pix, stride, true, io)) return plan_inverse_impl(P, d_coeff, d_pix, io);
    int r = stage_reserve(ctx, 0, (size_t)S.W * S.H * 3 * 4 + 64);
    if (r != J2K_OK) return r;
    r = plan_inverse_impl(P, d_coeff, ctx->stage[0]);
    if (r != J2K_OK) return r;
    return j2k_pack_pixels(ctx, (const int32_t *)ctx->stage[0], 3, 8, S.W, S.H, d_pix, stride);
}

extern "C" int j2k_plan_forward_pixels(j2k_plan *P, int format, const void *d_pix, size_t stride, int32_t *d_coeff) {
    if (!P || !d_pix || !d_coeff) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    if (format < 0 || format >= 6) return fail(ctx, J2K_ERR_INVALID_ARG, "unknown pixel format");
    if (kPixComp[format] != S.C) return fail(ctx, J2K_ERR_INVALID_ARG, "pixel format and plan disagree on the component count");
    if (stride < (size_t)S.W * kPixBytes[format]) return fail(ctx, J2K_ERR_INVALID_ARG, "pixel stride shorter than a row");
    PixIO io;
    if (pix_fusable(P, kPixPrec[format] / 8, S.C == 1 ? 1 : 4, d_pix, stride, false, io)) return plan_forward_impl(P, d_pix, d_coeff, io);
    int r = stage_reserve(ctx, 0, (size_t)S.W * S.H * S.C * 4 + 64);       // int32 staging frame
    if (r != J2K_OK) return r;
    r = j2k_unpack_pixels(ctx, format, d_pix, stride, S.W, S.H, S.precision, (int32_t *)ctx->stage[0]);
    if (r != J2K_OK) return r;
    return plan_forward_impl(P, ctx->stage[0], d_coeff);
}

extern "C" int j2k_plan_pixels_fused(const j2k_plan *P, int format, const void *d_pix, size_t stride, int inverse) {
    if (!P || !d_pix) return J2K_ERR_INVALID_ARG;
    const PlanSpec &S = P->spec;
    PixIO io;
    if (inverse) return ((S.precision == 8 || S.precision == 16) && pix_fusable(P, S.precision / 8, S.C == 1 ? 1 : 4, d_pix, stride, true, io)) ? 1 : 0;
    if (format < 0 || format >= 6 || kPixComp[format] != S.C) return J2K_ERR_INVALID_ARG;
    return pix_fusable(P, kPixPrec[format] / 8, S.C == 1 ? 1 : 4, d_pix, stride, false, io) ? 1 : 0;
}

extern "C" int j2k_plan_inverse_pixels(j2k_plan *P, const int32_t *d_coeff, void *d_pix, size_t stride) {
    if (!P || !d_pix || !d_coeff) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    // decoder.createImage picks the image type from (components, precision): Gray / Gray16, RGBA / RGBA64 (decoder.go:417-588)
    PixIO io;
    if ((S.precision == 8 || S.precision == 16) && pix_fusable(P, S.precision / 8, S.C == 1 ? 1 : 4, d_pix, stride, true, io))
        return plan_inverse_impl(P, d_coeff, d_pix, io);
    int r = stage_reserve(ctx, 0, (size_t)S.W * S.H * S.C * 4 + 64);
    if (r != J2K_OK) return r;
    r = plan_inverse_impl(P, d_coeff, ctx->stage[0]);
    if (r != J2K_OK) return r;
    return j2k_pack_pixels(ctx, (const int32_t *)ctx->stage[0], S.C, S.precision, S.W, S.H, d_pix, stride);
}

// Workspace of the T1 encoder: [serial-kernel work: wpj * n] [nsyms: n words] [lane order: n + 64 words] [symbol lists: n * stride].  The symbol
// lists cover 31 bit planes when that fits ctx->t1_sym_mb (default 8192 MiB), fewer otherwise (blocks with more planes
// take the one-kernel path on the device); ctx->t1_split = 0 turns the two-kernel path off.
struct T1Workspace { size_t off_nsyms, off_sym, stride, total; };
static T1Workspace t1_workspace(const j2k_ctx *ctx, size_t n, size_t wpj) {
    const long limit_mb = ctx->t1_sym_mb;
    const int split = ctx->t1_split;
    T1Workspace W{};
    W.off_nsyms = (wpj * n + 255) & ~size_t(255);
    W.off_sym = (W.off_nsyms + (2 * n + 64) * 4 + 255) & ~size_t(255);     // nsyms, then the lane order of the MQ lanes kernel
    W.stride = 0;
    if (split && n) {
        int planes = 31;
        const size_t limit = (size_t)limit_mb << 20;
        while (planes >= 8 && j2k::t1_sym_stride(planes) * n > limit) planes--;
        if (planes >= 8) W.stride = j2k::t1_sym_stride(planes);
    }
    W.total = W.off_sym + W.stride * n + 256;
    return W;
}

static size_t t1_work_per_job(const j2k_plan *P) {
    size_t m = 0;
    for (const j2k_block &b : P->blocks) m = std::max(m, t1_work_bytes(b.w, b.h));
    return (m + 255) & ~size_t(255);
}

extern "C" int j2k_plan_encode_blocks(j2k_plan *P, const int32_t *d_coeff, uint8_t *d_slots, uint32_t *d_lens, uint8_t *d_numbps) {
    if (!P || !d_coeff || !d_slots || !d_lens || !d_numbps) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = (int)P->blocks.size();
    if (!n) return J2K_OK;
    int r = stage_reserve(ctx, 3, 256);
    if (r != J2K_OK) return r;
    int *d_fault = (int *)ctx->stage[3];                                         // sticky word: cleared when reported (check_fault)
    ctx->fault_armed = true;
    if (P->spec.coder == J2K_CODER_HT) {
        HIPCHK(ctx, launch_ht_encode(ctx->stream, P->d_bjobs, n, d_coeff, d_slots, d_lens, d_numbps, d_fault));
    } else {
        int max_dim = 0;
        for (const j2k_block &b : P->blocks) max_dim = std::max(max_dim, std::max(b.w, b.h));
        const size_t wpj = max_dim > 64 ? t1_work_per_job(P) : 0;     // only the serial kernel (blocks > 64) needs a workspace
        const T1Workspace W = t1_workspace(ctx, (size_t)n, wpj);
        r = stage_reserve(ctx, 2, W.total);
        if (r != J2K_OK) return r;
        uint8_t *ws = (uint8_t *)ctx->stage[2];
        // blocks of 65 ... 256 columns or rows: symbol lists for the two-kernel form (t1_big.inc) -- 16 symbols of room per sample
        // (a 256 x 256 block of 8-bit noise makes 10; a block that needs more is marked and takes the SERIAL t1_encode_kernel afterwards), n words of counts in front
        uint8_t *bigsym = nullptr;
        uint32_t *bignsyms = nullptr;
        // (one MQ context alone = one frame at a time: the fused kernel's latency is 7 % shorter; several = throughput: the lists)
        // (while a graph is being captured nothing may be allocated or copied: the lists need their table from an earlier call)
        if (max_dim > 64 && (mq_throughput_mode() || j2k::tuning_env("J2K_T1_BIG_SPLIT")) && !(ctx->capturing && !P->d_bigsym_off)) {
            if (!P->d_bigsym_off) {
                std::vector<uint64_t> off((size_t)n + 1, 0);
                uint64_t room = 16;                                   // J2K_T1_BIG_SYM_ROOM: symbols of room per sample (testing the fall-back)
                if (const char *en = j2k::tuning_env("J2K_T1_BIG_SYM_ROOM")) { const long v = atol(en); if (v >= 1 && v <= 64) room = (uint64_t)v; }
                for (int j = 0; j < n; j++) {
                    const j2k_block &b = P->blocks[(size_t)j];
                    const bool big = (b.w > 64 || b.h > 64) && b.w <= 256 && b.h <= 256;
                    off[(size_t)j + 1] = off[(size_t)j] + (big ? ((room * b.w * b.h + 255) & ~uint64_t(255)) : 0);
                }
                r = upload(ctx, &P->d_bigsym_off, off);
                if (r != J2K_OK) return r;
                P->bigsym_total = (size_t)off[(size_t)n];
            }
            const size_t head = ((size_t)n * 4 + 255) & ~size_t(255);
            if (P->bigsym_total && stage_reserve(ctx, 4, head + P->bigsym_total + 256) == J2K_OK) {      // (no room for the lists at all: the one-kernel form t1_encode_big_kernel<false>)
                bignsyms = (uint32_t *)ctx->stage[4];
                bigsym = (uint8_t *)ctx->stage[4] + head;
            }
        }
        HIPCHK(ctx, launch_t1_encode(ctx->stream, P->d_bjobs, n, d_coeff, d_slots, d_lens, d_numbps, ws, wpj, d_fault, max_dim,
                                     W.stride ? ws + W.off_sym : nullptr, W.stride, (uint32_t *)(ws + W.off_nsyms),
                                     ctx->t1_lanes > 0 ? ctx->t1_lanes : (mq_throughput_mode() ? -1 : 0), bigsym, P->d_bigsym_off, bignsyms));
    }
    return J2K_OK;
}

// encode_blocks + compact in one launch when every block is on the parallel HT path; otherwise the two steps through a
// slot buffer owned by the context
extern "C" int j2k_plan_encode_stream(j2k_plan *P, const int32_t *d_coeff, uint8_t *d_stream, uint64_t *d_offs, uint32_t *d_lens,
                                      uint8_t *d_numbps) {
    if (!P || !d_coeff || !d_stream || !d_offs || !d_lens || !d_numbps) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = (int)P->blocks.size();
    if (!n) { HIPCHK(ctx, hipMemsetAsync(d_offs, 0, 8, ctx->stream)); return J2K_OK; }
    bool fused = P->spec.coder == J2K_CODER_HT && ctx->fuse_compact;
    // The one-kernel path tags its look-back status words with a launch epoch that is a KERNEL ARGUMENT: a captured launch
    // would replay the same tag, read the previous replay's words as valid and place blocks at stale offsets.
    if (fused && ctx->capturing)
        return fail(ctx, J2K_ERR_UNSUPPORTED, "capture: the one-kernel HT path (J2K_FUSE_COMPACT=1) cannot be replayed from a graph -- unset it");
    if (fused && !P->d_status) {
        P->all_blocks_fast = true;
        for (const j2k_block &b : P->blocks)
            if ((int64_t)((b.h + 3) / 4) * b.w > ht_fast_max_samples()) P->all_blocks_fast = false;
        if (P->all_blocks_fast) {
            if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: run the same calls once before j2k_ctx_capture_begin");
            HIPCHK(ctx, hipMalloc((void **)&P->d_status, (size_t)n * 8));
            HIPCHK(ctx, hipMemsetAsync(P->d_status, 0, (size_t)n * 8, ctx->stream));
        }
    }
    if (fused && P->all_blocks_fast && P->d_status) {
        int r = stage_reserve(ctx, 3, 256);
        if (r != J2K_OK) return r;
        if ((++P->epoch & 0x3FFFFF) == 0) {          // the 22-bit tag wraps: clear the words once
            HIPCHK(ctx, hipMemsetAsync(P->d_status, 0, (size_t)n * 8, ctx->stream));
            P->epoch = 1;
        }
        ctx->fault_armed = true;
        HIPCHK(ctx, launch_ht_encode_stream(ctx->stream, P->d_bjobs, n, d_coeff, d_stream, d_offs, d_lens, d_numbps, P->d_status,
                                            P->epoch, (int *)ctx->stage[3]));
        return J2K_OK;
    }
    if (!P->d_slots) {
        if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: run the same calls once before j2k_ctx_capture_begin");
        HIPCHK(ctx, hipMalloc(&P->d_slots, (size_t)P->bytes_cap + 64));
    }
    if (P->spec.coder == J2K_CODER_HT) {
        // the slot buffer is private here: the MEL zero bytes are not written into it, the gather emits them (compact.hip)
        if (!P->d_maglens) {
            if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: run the same calls once before j2k_ctx_capture_begin");
            HIPCHK(ctx, hipMalloc((void **)&P->d_maglens, (size_t)n * 4));
            HIPCHK(ctx, hipMalloc((void **)&P->d_mels, (size_t)n * 4));
            HIPCHK(ctx, hipMalloc((void **)&P->d_toffs, ((size_t)n + 1) * 8));
            HIPCHK(ctx, launch_mel_table(ctx->stream, P->d_bjobs, n, P->d_mels));
        }
        int r = stage_reserve(ctx, 3, 256);
        if (r != J2K_OK) return r;
        ctx->fault_armed = true;
        for (int rep_ = 0; rep_ < dev_reps(8); rep_++)
        HIPCHK(ctx, launch_ht_encode(ctx->stream, P->d_bjobs, n, d_coeff, (uint8_t *)P->d_slots, d_lens, d_numbps, (int *)ctx->stage[3],
                                     P->d_maglens, P->d_ht_ujobs, P->ht_nunique, P->d_ht_alias_next));
        for (int rep_ = 0; rep_ < dev_reps(16); rep_++)
        // the transport offsets (a second running sum in the scan, +4 us) only once j2k_plan_pack_stream has asked for them
        HIPCHK(ctx, launch_compact(ctx->stream, P->d_bjobs_alias ? P->d_bjobs_alias : P->d_bjobs, n, (const uint8_t *)P->d_slots, d_lens, d_offs, d_stream, P->d_maglens,
                                   P->want_toffs ? P->d_mels : nullptr, P->want_toffs ? P->d_toffs : nullptr));
        P->toffs_valid = P->want_toffs;
        P->last_stream = d_stream; P->last_lens = d_lens;
        return J2K_OK;
    }
    int r = j2k_plan_encode_blocks(P, d_coeff, (uint8_t *)P->d_slots, d_lens, d_numbps);
    if (r != J2K_OK) return r;
    return j2k_plan_compact(P, (const uint8_t *)P->d_slots, d_lens, d_offs, d_stream);
}

extern "C" size_t j2k_plan_pack_bound(const j2k_plan *P) {
    if (!P) return 0;
    return pack_header_bytes(P->blocks.size()) + (size_t)P->bytes_cap + 64;
}

extern "C" size_t j2k_plan_tile_parts_bound(const j2k_plan *P) {
    return P ? (size_t)P->bytes_cap + 14 * (size_t)P->tile_count : 0;
}
extern "C" int j2k_plan_assemble_tiles_device(j2k_plan *P, const uint8_t *d_stream, const uint64_t *d_offs, uint8_t *d_out,
                                              uint64_t *d_out_len) {
    if (!P) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (!d_stream || !d_offs || !d_out || !d_out_len) return fail(ctx, J2K_ERR_INVALID_ARG, "null device pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_assemble_tiles(ctx->stream, d_stream, d_offs, P->d_tile_job0, P->tile_count, P->tile_first,
                                      P->max_tile_bytes, d_out, d_out_len));
    return J2K_OK;
}

extern "C" int j2k_plan_pack_stream(j2k_plan *P, const uint8_t *d_stream, const uint64_t *d_offs, const uint32_t *d_lens,
                                    const uint8_t *d_numbps, uint8_t *d_pack) {
    if (!P || !d_stream || !d_offs || !d_lens || !d_numbps || !d_pack) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = (int)P->blocks.size();
    const bool ht = P->spec.coder == J2K_CODER_HT;
    if (ht && !P->d_maglens) return fail(ctx, J2K_ERR_UNSUPPORTED, "pack_stream: no j2k_plan_encode_stream (three-kernel path) ran on this plan");
    // the MagSgn lengths / transport offsets the pack needs are plan state left by the LAST encode_stream: refuse any other
    // stream (e.g. an older buffer of a rotating set) instead of silently mixing two frames' arrays
    if (ht && (P->last_stream != d_stream || P->last_lens != d_lens))
        return fail(ctx, J2K_ERR_INVALID_ARG, "pack_stream: d_stream / d_lens are not the outputs of the last j2k_plan_encode_stream on this plan");
    if (ht && !P->toffs_valid) {      // first pack on this plan: scan the transport lengths now, and with every encode from here on
        int r = stage_reserve(ctx, 1, ((size_t)n + 1) * 8 + 4096);
        if (r != J2K_OK) return r;
        HIPCHK(ctx, launch_scan(ctx->stream, d_lens, n, (uint64_t *)ctx->stage[1], P->d_mels, P->d_toffs));
        P->want_toffs = true;
        P->toffs_valid = true;
    }
    HIPCHK(ctx, launch_pack(ctx->stream, P->d_bjobs, n, d_stream, d_offs, ht ? P->d_toffs : d_offs, d_lens, d_numbps,
                            ht ? P->d_maglens : nullptr, d_pack));
    return J2K_OK;
}

extern "C" int j2k_plan_unpack_streams(j2k_plan *P, int count, const uint8_t *const *d_packs, const size_t *pack_bytes,
                                       uint8_t *const *d_streams, uint64_t *const *d_offs, uint32_t *const *d_lens, uint8_t *const *d_numbps) {
    if (!P || count < 0 || (count && (!d_packs || !pack_bytes || !d_streams || !d_offs || !d_lens || !d_numbps))) return J2K_ERR_INVALID_ARG;
    for (int i = 0; i < count; i++)
        if (!d_packs[i] || !d_streams[i] || !d_offs[i] || !d_lens[i] || !d_numbps[i]) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!count) return J2K_OK;
    const int n = (int)P->blocks.size();
    // a pack is foreign input: its header and per-block sections must lie inside the bytes the caller really has
    for (int i = 0; i < count; i++)
        if (pack_bytes[i] < pack_header_bytes((size_t)n)) return fail(ctx, J2K_ERR_INVALID_ARG, "unpack_stream: the pack is shorter than its own header");
    int r = stage_reserve(ctx, 3, 256);
    if (r != J2K_OK) return r;
    ctx->fault_armed = true;
    HIPCHK(ctx, launch_unpack(ctx->stream, P->d_bjobs, n, count, d_packs, pack_bytes, d_streams, (size_t)P->bytes_cap, d_offs, d_lens, d_numbps,
                              (int *)ctx->stage[3]));
    return J2K_OK;
}

extern "C" int j2k_plan_unpack_stream(j2k_plan *P, const uint8_t *d_pack, size_t pack_bytes, uint8_t *d_stream, uint64_t *d_offs,
                                      uint32_t *d_lens, uint8_t *d_numbps) {
    return j2k_plan_unpack_streams(P, 1, &d_pack, &pack_bytes, &d_stream, &d_offs, &d_lens, &d_numbps);
}

extern "C" int j2k_plan_compact(j2k_plan *P, const uint8_t *d_slots, const uint32_t *d_lens, uint64_t *d_offs, uint8_t *d_stream) {
    if (!P || !d_slots || !d_lens || !d_offs || !d_stream) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = (int)P->blocks.size();
    int r = stage_reserve(ctx, 1, 4096);
    if (r != J2K_OK) return r;
    HIPCHK(ctx, launch_compact(ctx->stream, P->d_bjobs, n, d_slots, d_lens, d_offs, d_stream, nullptr));
    return J2K_OK;
}

extern "C" int j2k_plan_set_decode_coded_rows_only(j2k_plan *P, int on) {
    if (!P) return J2K_ERR_INVALID_ARG;
    P->dec_coded_rows_only = on != 0;
    return J2K_OK;
}

extern "C" int j2k_plan_decode_blocks(j2k_plan *P, const uint8_t *d_stream, const uint64_t *d_offs, const uint32_t *d_lens,
                                      const uint8_t *d_numbps, int32_t *d_decoded) {
    if (!P || !d_stream || !d_offs || !d_lens || !d_numbps || !d_decoded) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = (int)P->blocks.size();
    if (!n) return J2K_OK;
    if (P->spec.coder == J2K_CODER_HT) {
        int r = stage_reserve(ctx, 2, ht_decode_scratch_words(n) * 4 + 256);
        if (r != J2K_OK) return r;
        HIPCHK(ctx, launch_ht_decode(ctx->stream, P->d_djobs, n, d_stream, d_offs, d_lens, d_decoded, (uint32_t *)ctx->stage[2],
                                     P->dec_coded_rows_only ? 1 : 0));
    } else {
        size_t wpj = 0;                                  // the one-block decoder's workspace holds the flags only
        int max_dim = 0;
        for (const j2k_block &b : P->blocks) { wpj = std::max(wpj, t1_flag_bytes(b.w, b.h)); max_dim = std::max(max_dim, std::max(b.w, b.h)); }
        wpj = (wpj + 255) & ~size_t(255);
        // plane-stepped path (t1.hip): frames of at least t1_dec_split blocks, 16-byte aligned stream (its 16-byte loads)
        // (one context alone: the one-launch kernels are a 28-30 ms chain while their 8192 wavefront slots hold the frame, the lanes
        //  decoder a 45-50 ms one whatever the frame -- an 8K frame of 27 000 blocks: 86 ms against 51, tools/check_big_mq.py)
        const int split_min = ctx->t1_dec_split >= 0 ? ctx->t1_dec_split : (mq_throughput_mode() ? 512 : 12000);
        bool split = split_min > 0 && n >= split_min && !ctx->t1_dec_general && !((uintptr_t)d_stream & 15);
        const size_t gen_bytes = (wpj * (size_t)n + 255) & ~size_t(255);
        // The lanes decoder's workspace is T1DS_STRIDE + 3.5 KB of row masks + 16 KB of plane words per block (0.55 GB for an 8K frame of
        // 27 000 blocks); if the device cannot give it, the one-launch kernels decode the frame with gen_bytes alone
        // rather than the call failing.
        int r = stage_reserve(ctx, 2, gen_bytes + 256 + (split ? j2k::t1_dec_split_bytes((size_t)n) : 0));
        if (r != J2K_OK && split) { (void)hipGetLastError(); split = false; r = stage_reserve(ctx, 2, gen_bytes + 256); }
        if (r != J2K_OK) return r;
        HIPCHK(ctx, launch_t1_decode(ctx->stream, P->d_djobs, n, d_stream, d_offs, d_lens, d_numbps, d_decoded,
                                     (uint8_t *)ctx->stage[2], wpj, max_dim, ctx->t1_dec_general,
                                     split ? (uint8_t *)ctx->stage[2] + gen_bytes : nullptr, ctx->t1_dec_lanes, mq_throughput_mode() ? 1 : 0));
    }
    return J2K_OK;
}

// ------------------------------------------------------------------------------
// host (unit) calls
// ------------------------------------------------------------------------------
static int cached_plan(j2k_ctx *ctx, const PlanSpec &S, j2k_plan **out) {
    for (j2k_plan *p : ctx->cache)
        if (p->spec == S) { *out = p; return J2K_OK; }
    j2k_plan *p = nullptr;
    int r = build_plan(ctx, S, &p);
    if (r != J2K_OK) return r;
    if (ctx->cache.size() >= 16) { j2k_plan_destroy(ctx->cache.front()); ctx->cache.erase(ctx->cache.begin()); }
    ctx->cache.push_back(p);
    *out = p;
    return J2K_OK;
}

// Runs `levels` of a single-plane transform on a host buffer, in place.
static int host_dwt(j2k_ctx *ctx, void *data, int w, int h, int levels, int wavelet, bool inverse, int quant, bool frame_f64) {
    if (!ctx) return J2K_ERR_INVALID_ARG;
    if (w < 0 || h < 0 || levels < 0) return fail(ctx, J2K_ERR_INVALID_ARG, "negative size");
    if (w == 0 || h == 0 || levels == 0) return J2K_OK;   // Go loops simply do not run
    if (!data) return fail(ctx, J2K_ERR_INVALID_ARG, "data == NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlanSpec S;
    S.W = w; S.H = h; S.C = 1; S.levels = levels; S.wavelet = wavelet; S.dc_shift = 0; S.dc_shift_inv = 0; S.mct = 0;
    S.quant = quant; S.frame_is_f64 = frame_f64; S.num_res_jobs = 1; S.cb_w = 1 << 20; S.cb_h = 1 << 20;
    j2k_plan *P = nullptr;
    int r = cached_plan(ctx, S, &P);
    if (r != J2K_OK) return r;
    const size_t n = (size_t)w * h;
    const size_t fsz = frame_f64 ? 8 : 4;
    const size_t csz = (wavelet == W97 && quant == Q_NONE) ? 8 : 4;
    r = stage_reserve(ctx, 0, n * fsz + 64);
    if (r == J2K_OK) r = stage_reserve(ctx, 1, (size_t)P->coeff_elems * csz + 64);
    if (r != J2K_OK) return r;
    void *d_frame = ctx->stage[0], *d_coef = ctx->stage[1];
    if (!inverse) {
        HIPCHK(ctx, hipMemcpyAsync(d_frame, data, n * fsz, hipMemcpyHostToDevice, ctx->stream));
        r = plan_forward_impl(P, d_frame, d_coef);
        if (r != J2K_OK) return r;
        HIPCHK(ctx, hipMemcpyAsync(data, d_coef, n * csz, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        HIPCHK(ctx, hipMemcpyAsync(d_coef, data, n * csz, hipMemcpyHostToDevice, ctx->stream));
        r = plan_inverse_impl(P, d_coef, d_frame);
        if (r != J2K_OK) return r;
        HIPCHK(ctx, hipMemcpyAsync(data, d_frame, n * fsz, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return J2K_OK;
}

extern "C" int j2k_forward53(j2k_ctx *c, int32_t *d, int n) { return host_dwt(c, d, n, 1, 1, W53, false, Q_NONE, false); }
extern "C" int j2k_inverse53(j2k_ctx *c, int32_t *d, int n) { return host_dwt(c, d, n, 1, 1, W53, true, Q_NONE, false); }
extern "C" int j2k_forward97(j2k_ctx *c, double *d, int n) { return host_dwt(c, d, n, 1, 1, W97, false, Q_NONE, true); }
extern "C" int j2k_inverse97(j2k_ctx *c, double *d, int n) { return host_dwt(c, d, n, 1, 1, W97, true, Q_NONE, true); }
extern "C" int j2k_forward2d53(j2k_ctx *c, int32_t *d, int w, int h) { return host_dwt(c, d, w, h, 1, W53, false, Q_NONE, false); }
extern "C" int j2k_inverse2d53(j2k_ctx *c, int32_t *d, int w, int h) { return host_dwt(c, d, w, h, 1, W53, true, Q_NONE, false); }
extern "C" int j2k_forward2d97(j2k_ctx *c, double *d, int w, int h) { return host_dwt(c, d, w, h, 1, W97, false, Q_NONE, true); }
extern "C" int j2k_inverse2d97(j2k_ctx *c, double *d, int w, int h) { return host_dwt(c, d, w, h, 1, W97, true, Q_NONE, true); }
extern "C" int j2k_decompose_multilevel53(j2k_ctx *c, int32_t *d, int w, int h, int l) { return host_dwt(c, d, w, h, l, W53, false, Q_NONE, false); }
extern "C" int j2k_reconstruct_multilevel53(j2k_ctx *c, int32_t *d, int w, int h, int l) { return host_dwt(c, d, w, h, l, W53, true, Q_NONE, false); }
extern "C" int j2k_decompose_multilevel97(j2k_ctx *c, double *d, int w, int h, int l) { return host_dwt(c, d, w, h, l, W97, false, Q_NONE, true); }
extern "C" int j2k_reconstruct_multilevel97(j2k_ctx *c, double *d, int w, int h, int l) { return host_dwt(c, d, w, h, l, W97, true, Q_NONE, true); }

extern "C" int j2k_tcd_apply_forward_dwt(j2k_ctx *c, int32_t *d, int w, int h, int levels, int reversible) {
    if (reversible) return host_dwt(c, d, w, h, levels, W53, false, Q_NONE, false);
    return host_dwt(c, d, w, h, levels, W97, false, Q_TCD, false);
}
extern "C" int j2k_tcd_apply_inverse_dwt(j2k_ctx *c, int32_t *d, int w, int h, int levels, int reversible) {
    if (reversible) return host_dwt(c, d, w, h, levels, W53, true, Q_NONE, false);
    return host_dwt(c, d, w, h, levels, W97, true, Q_TCD, false);
}

// ---- mct ------------------------------------------------------------------------
static int host_elementwise3(j2k_ctx *ctx, void *a, void *b, void *c, size_t n, size_t esz, int op, int arg) {
    if (!ctx) return J2K_ERR_INVALID_ARG;
    if (n == 0) return J2K_OK;
    if (!a || (op != 0 && (!b || !c))) return fail(ctx, J2K_ERR_INVALID_ARG, "NULL plane");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int np = op == 0 ? 1 : 3;
    void *h[3] = {a, b, c};
    for (int i = 0; i < np; i++) {
        int r = stage_reserve(ctx, i, n * esz);
        if (r != J2K_OK) return r;
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage[i], h[i], n * esz, hipMemcpyHostToDevice, ctx->stream));
    }
    switch (op) {
        case 0: HIPCHK(ctx, launch_add_const(ctx->stream, (int32_t *)ctx->stage[0], n, arg)); break;
        case 1: HIPCHK(ctx, launch_rct(ctx->stream, (int32_t *)ctx->stage[0], (int32_t *)ctx->stage[1], (int32_t *)ctx->stage[2], n, arg)); break;
        case 2: HIPCHK(ctx, launch_ict(ctx->stream, (double *)ctx->stage[0], (double *)ctx->stage[1], (double *)ctx->stage[2], n, arg)); break;
    }
    for (int i = 0; i < np; i++) HIPCHK(ctx, hipMemcpyAsync(h[i], ctx->stage[i], n * esz, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return J2K_OK;
}

extern "C" int j2k_dc_level_shift_forward(j2k_ctx *ctx, int32_t *d, size_t n, int precision) {
    if (precision < 1 || precision > 32) return fail(ctx, J2K_ERR_INVALID_ARG, "precision");
    return host_elementwise3(ctx, d, nullptr, nullptr, n, 4, 0, (int)(0u - ((uint32_t)1 << (precision - 1))));
}
extern "C" int j2k_dc_level_shift_inverse(j2k_ctx *ctx, int32_t *d, size_t n, int precision) {
    if (precision < 1 || precision > 32) return fail(ctx, J2K_ERR_INVALID_ARG, "precision");
    return host_elementwise3(ctx, d, nullptr, nullptr, n, 4, 0, (int)((uint32_t)1 << (precision - 1)));
}
extern "C" int j2k_forward_rct(j2k_ctx *ctx, int32_t *r, int32_t *g, int32_t *b, size_t n) { return host_elementwise3(ctx, r, g, b, n, 4, 1, 0); }
extern "C" int j2k_inverse_rct(j2k_ctx *ctx, int32_t *y, int32_t *u, int32_t *v, size_t n) { return host_elementwise3(ctx, y, u, v, n, 4, 1, 1); }
extern "C" int j2k_forward_ict(j2k_ctx *ctx, double *r, double *g, double *b, size_t n) { return host_elementwise3(ctx, r, g, b, n, 8, 2, 0); }
extern "C" int j2k_inverse_ict(j2k_ctx *ctx, double *y, double *cb, double *cr, size_t n) { return host_elementwise3(ctx, y, cb, cr, n, 8, 2, 1); }

// ---- batched block coding from host planes ----------------------------------------
extern "C" int j2k_encode_blocks(j2k_ctx *ctx, int coder, const int32_t *const *planes, const int32_t *plane_w,
                                 const int32_t *plane_h, int nplanes, const j2k_block *blocks, size_t nblocks,
                                 uint8_t *out, size_t cap, uint64_t *offs, uint32_t *lens, uint8_t *numbps, size_t *total) {
    if (!ctx) return J2K_ERR_INVALID_ARG;
    if (total) *total = 0;
    if (nblocks == 0) return J2K_OK;
    if (!planes || !plane_w || !plane_h || !blocks || nplanes <= 0) return fail(ctx, J2K_ERR_INVALID_ARG, "NULL argument");
    if (coder != J2K_CODER_MQ && coder != J2K_CODER_HT) return fail(ctx, J2K_ERR_INVALID_ARG, "coder");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<int64_t> poff(nplanes);
    int64_t tot = 0;
    for (int i = 0; i < nplanes; i++) {
        if (plane_w[i] <= 0 || plane_h[i] <= 0 || !planes[i]) return fail(ctx, J2K_ERR_INVALID_ARG, "bad plane");
        poff[i] = tot; tot += align4((int64_t)plane_w[i] * plane_h[i]);
    }
    std::vector<BlockJob> bj(nblocks);
    int64_t slot = 0;
    size_t wpj = 0;
    for (size_t j = 0; j < nblocks; j++) {
        const j2k_block &b = blocks[j];
        if (b.plane < 0 || b.plane >= nplanes || b.w <= 0 || b.h <= 0 || b.x0 < 0 || b.y0 < 0 ||
            b.x0 + b.w > plane_w[b.plane] || b.y0 + b.h > plane_h[b.plane] || b.band < 0 || b.band > 3)
            return fail(ctx, J2K_ERR_INVALID_ARG, "block window outside its plane");
        bj[j].src_off = poff[b.plane] + (int64_t)b.y0 * plane_w[b.plane] + b.x0;
        bj[j].out_off = slot;
        bj[j].stride = plane_w[b.plane]; bj[j].w = b.w; bj[j].h = b.h; bj[j].band = b.band;
        slot += (int64_t)((j2k_block_bound(coder, b.w, b.h) + 15) & ~size_t(15));
        wpj = std::max(wpj, t1_work_bytes(b.w, b.h));
    }
    wpj = (wpj + 255) & ~size_t(255);
    void *d_coef = nullptr, *d_jobs = nullptr, *d_slots = nullptr, *d_lens = nullptr, *d_nb = nullptr, *d_work = nullptr, *d_fault = nullptr;
    int status = J2K_OK;
    auto cleanup = [&]() { for (void *p : {d_coef, d_jobs, d_slots, d_lens, d_nb, d_work, d_fault}) if (p) (void)hipFree(p); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { status = fail_hip(ctx, e_, #call); cleanup(); return status; } } while (0)
    TRY(hipMalloc(&d_coef, (size_t)tot * 4 + 16));
    TRY(hipMalloc(&d_jobs, nblocks * sizeof(BlockJob)));
    TRY(hipMalloc(&d_slots, (size_t)slot + 16));
    TRY(hipMalloc(&d_lens, nblocks * 4));
    TRY(hipMalloc(&d_nb, nblocks));
    TRY(hipMalloc(&d_fault, 16));
    TRY(hipMemsetAsync(d_fault, 0, 16, ctx->stream));
    for (int i = 0; i < nplanes; i++)
        TRY(hipMemcpyAsync((int32_t *)d_coef + poff[i], planes[i], (size_t)plane_w[i] * plane_h[i] * 4, hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(d_jobs, bj.data(), nblocks * sizeof(BlockJob), hipMemcpyHostToDevice, ctx->stream));
    if (coder == J2K_CODER_HT) {
        TRY(launch_ht_encode(ctx->stream, (BlockJob *)d_jobs, (int)nblocks, (int32_t *)d_coef, (uint8_t *)d_slots, (uint32_t *)d_lens, (uint8_t *)d_nb, (int *)d_fault));
    } else {
        int max_dim = 0;
        for (size_t j = 0; j < nblocks; j++) max_dim = std::max(max_dim, std::max(bj[j].w, bj[j].h));
        if (max_dim <= 64) wpj = 0;
        const T1Workspace W = t1_workspace(ctx, nblocks, wpj);
        TRY(hipMalloc(&d_work, W.total));
        uint8_t *ws = (uint8_t *)d_work;
        TRY(launch_t1_encode(ctx->stream, (BlockJob *)d_jobs, (int)nblocks, (int32_t *)d_coef, (uint8_t *)d_slots, (uint32_t *)d_lens, (uint8_t *)d_nb,
                             ws, wpj, (int *)d_fault, max_dim, W.stride ? ws + W.off_sym : nullptr, W.stride, (uint32_t *)(ws + W.off_nsyms), ctx->t1_lanes));
    }
    std::vector<uint32_t> hl(nblocks);
    std::vector<uint8_t> hn(nblocks);
    int hf = 0;
    TRY(hipMemcpyAsync(hl.data(), d_lens, nblocks * 4, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipMemcpyAsync(hn.data(), d_nb, nblocks, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipMemcpyAsync(&hf, d_fault, 4, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipStreamSynchronize(ctx->stream));
    if (hf) { cleanup(); return fail(ctx, hf == 1 ? J2K_ERR_GO_PANIC : J2K_ERR_CAPACITY, "block coder fault"); }
    size_t pos = 0;
    for (size_t j = 0; j < nblocks; j++) {
        if (offs) offs[j] = pos;
        if (lens) lens[j] = hl[j];
        if (numbps) numbps[j] = hn[j];
        pos += hl[j];
    }
    if (total) *total = pos;
    if (pos > cap || (pos && !out)) { cleanup(); return fail(ctx, J2K_ERR_CAPACITY, "out too small"); }
    pos = 0;
    for (size_t j = 0; j < nblocks; j++) {
        if (hl[j]) TRY(hipMemcpyAsync(out + pos, (uint8_t *)d_slots + bj[j].out_off, hl[j], hipMemcpyDeviceToHost, ctx->stream));
        pos += hl[j];
    }
    TRY(hipStreamSynchronize(ctx->stream));
    cleanup();
    return J2K_OK;
}

extern "C" int j2k_decode_blocks(j2k_ctx *ctx, int coder, const uint8_t *bytes, const uint64_t *offs, const uint32_t *lens,
                                 const uint8_t *numbps, const j2k_block *blocks, size_t nblocks, int32_t *coeffs,
                                 const uint64_t *coeff_offs) {
    if (!ctx) return J2K_ERR_INVALID_ARG;
    if (nblocks == 0) return J2K_OK;
    if (!offs || !lens || !blocks || !coeffs || !coeff_offs) return fail(ctx, J2K_ERR_INVALID_ARG, "NULL argument");
    if (coder != J2K_CODER_MQ && coder != J2K_CODER_HT) return fail(ctx, J2K_ERR_INVALID_ARG, "coder");
    if (coder == J2K_CODER_MQ && !numbps) return fail(ctx, J2K_ERR_INVALID_ARG, "numbps required for T1.Decode");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t nbytes = 0;
    std::vector<BlockJob> bj(nblocks);
    int64_t dec = 0;
    size_t wpj = 0;
    for (size_t j = 0; j < nblocks; j++) {
        const j2k_block &b = blocks[j];
        if (b.w <= 0 || b.h <= 0 || b.band < 0 || b.band > 3) return fail(ctx, J2K_ERR_INVALID_ARG, "bad block");
        nbytes = std::max<size_t>(nbytes, (size_t)offs[j] + lens[j]);
        bj[j].src_off = 0; bj[j].out_off = dec; bj[j].stride = b.w; bj[j].w = b.w; bj[j].h = b.h; bj[j].band = b.band;
        dec += align4((int64_t)b.w * b.h);
        wpj = std::max(wpj, t1_flag_bytes(b.w, b.h));
    }
    if (nbytes && !bytes) return fail(ctx, J2K_ERR_INVALID_ARG, "bytes == NULL");
    wpj = (wpj + 255) & ~size_t(255);
    void *d_bytes = nullptr, *d_jobs = nullptr, *d_offs = nullptr, *d_lens = nullptr, *d_nb = nullptr, *d_dec = nullptr, *d_work = nullptr;
    int status = J2K_OK;
    auto cleanup = [&]() { for (void *p : {d_bytes, d_jobs, d_offs, d_lens, d_nb, d_dec, d_work}) if (p) (void)hipFree(p); };
    TRY(hipMalloc(&d_bytes, nbytes + 16));
    TRY(hipMalloc(&d_jobs, nblocks * sizeof(BlockJob)));
    TRY(hipMalloc(&d_offs, nblocks * 8));
    TRY(hipMalloc(&d_lens, nblocks * 4));
    TRY(hipMalloc(&d_nb, nblocks));
    TRY(hipMalloc(&d_dec, (size_t)dec * 4 + 16));
    if (nbytes) TRY(hipMemcpyAsync(d_bytes, bytes, nbytes, hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(d_jobs, bj.data(), nblocks * sizeof(BlockJob), hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(d_offs, offs, nblocks * 8, hipMemcpyHostToDevice, ctx->stream));
    TRY(hipMemcpyAsync(d_lens, lens, nblocks * 4, hipMemcpyHostToDevice, ctx->stream));
    if (numbps) TRY(hipMemcpyAsync(d_nb, numbps, nblocks, hipMemcpyHostToDevice, ctx->stream));
    else TRY(hipMemsetAsync(d_nb, 0, nblocks, ctx->stream));
    if (coder == J2K_CODER_HT) {
        TRY(hipMalloc(&d_work, ht_decode_scratch_words((int)nblocks) * 4 + 256));
        TRY(launch_ht_decode(ctx->stream, (BlockJob *)d_jobs, (int)nblocks, (uint8_t *)d_bytes, (uint64_t *)d_offs, (uint32_t *)d_lens, (int32_t *)d_dec, (uint32_t *)d_work));
    } else {
        // the same choice of decoder as j2k_plan_decode_blocks (the two entry points used to differ), and the same fall-back
        // to the one-launch kernels when the lanes decoder's workspace cannot be had
        const int split_min = ctx->t1_dec_split >= 0 ? ctx->t1_dec_split : (mq_throughput_mode() ? 512 : 12000);
        bool split = split_min > 0 && (int)nblocks >= split_min && !ctx->t1_dec_general;
        const size_t gen_bytes = (wpj * nblocks + 255) & ~size_t(255);
        if (split && hipMalloc(&d_work, gen_bytes + 256 + j2k::t1_dec_split_bytes(nblocks)) != hipSuccess) { (void)hipGetLastError(); d_work = nullptr; split = false; }
        if (!split) TRY(hipMalloc(&d_work, gen_bytes + 256));
        int max_dim = 0;
        for (size_t j = 0; j < nblocks; j++) max_dim = std::max(max_dim, std::max(bj[j].w, bj[j].h));
        TRY(launch_t1_decode(ctx->stream, (BlockJob *)d_jobs, (int)nblocks, (uint8_t *)d_bytes, (uint64_t *)d_offs, (uint32_t *)d_lens, (uint8_t *)d_nb,
                             (int32_t *)d_dec, (uint8_t *)d_work, wpj, max_dim, ctx->t1_dec_general, split ? (uint8_t *)d_work + gen_bytes : nullptr, ctx->t1_dec_lanes));
    }
    for (size_t j = 0; j < nblocks; j++)
        TRY(hipMemcpyAsync(coeffs + coeff_offs[j], (int32_t *)d_dec + bj[j].out_off, (size_t)blocks[j].w * blocks[j].h * 4, hipMemcpyDeviceToHost, ctx->stream));
    TRY(hipStreamSynchronize(ctx->stream));
    cleanup();
    return J2K_OK;
}
#undef TRY

// ---- whole shard from host planes (encoder.preprocess + encodeTile) ------------------
extern "C" int j2k_encode_frame(j2k_plan *P, int32_t *const *planes, int32_t *coeff, uint8_t *out, size_t cap,
                                size_t *out_len, uint64_t *tile_offs, uint32_t *lens, uint8_t *numbps) {
    if (!P || !planes) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    const PlanSpec &S = P->spec;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)S.W * S.H;
    const size_t nb = P->blocks.size();
    int r;
    if ((r = ensure(ctx, &P->d_frame, npx * S.C * 4)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_coeff, (size_t)P->coeff_elems * 4)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_slots, (size_t)P->bytes_cap)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_stream, (size_t)P->bytes_cap)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_lens, nb * 4)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_numbps, nb)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_offs, (nb + 1) * 8)) != J2K_OK) return r;
    for (int c = 0; c < S.C; c++) {
        if (!planes[c]) return fail(ctx, J2K_ERR_INVALID_ARG, "NULL plane");
        HIPCHK(ctx, hipMemcpyAsync((int32_t *)P->d_frame + (size_t)c * npx, planes[c], npx * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((r = plan_forward_impl(P, P->d_frame, P->d_coeff)) != J2K_OK) return r;
    if ((r = j2k_plan_encode_blocks(P, (int32_t *)P->d_coeff, (uint8_t *)P->d_slots, (uint32_t *)P->d_lens, (uint8_t *)P->d_numbps)) != J2K_OK) return r;
    if ((r = j2k_plan_compact(P, (uint8_t *)P->d_slots, (uint32_t *)P->d_lens, (uint64_t *)P->d_offs, (uint8_t *)P->d_stream)) != J2K_OK) return r;
    if ((r = check_fault(ctx)) != J2K_OK) return r;
    std::vector<uint64_t> offs(nb + 1, 0);
    if (nb) HIPCHK(ctx, hipMemcpyAsync(offs.data(), P->d_offs, (nb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (lens && nb) HIPCHK(ctx, hipMemcpyAsync(lens, P->d_lens, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (numbps && nb) HIPCHK(ctx, hipMemcpyAsync(numbps, P->d_numbps, nb, hipMemcpyDeviceToHost, ctx->stream));
    // coefficients back: single tile -> in place into planes[] like e.componentData; else into coeff
    const bool single = (P->tiles_x * P->tiles_y == 1);
    for (const Group &g : P->groups)
        for (int k = 0; k < g.nc; k++) {
            int32_t *dst = single ? planes[g.comp0 + k] : (coeff ? coeff + g.coef_off[k] : nullptr);
            if (dst) HIPCHK(ctx, hipMemcpyAsync(dst, (int32_t *)P->d_coeff + g.coef_off[k], (size_t)g.w * g.h * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t total = (size_t)offs[nb];
    if (out_len) *out_len = total;
    if (tile_offs) {
        size_t j = 0;
        for (int t = 0; t < P->tile_count; t++) {
            while (j < nb && P->block_tile[j] < t) j++;
            tile_offs[t] = j < nb ? offs[j] : total;
        }
        tile_offs[P->tile_count] = total;
    }
    if (total > cap || (total && !out)) return fail(ctx, J2K_ERR_CAPACITY, "out too small");
    if (total) HIPCHK(ctx, hipMemcpy(out, P->d_stream, total, hipMemcpyDeviceToHost));
    return J2K_OK;
}
