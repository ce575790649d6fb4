// j2k_ctx.cpp -- contexts: creation, tuning options, HIP graphs, synchronisation, kernel timing, staging slots (C ABI of libj2kgfx.so, include/j2kgfx.h; shared declarations: j2k_host.h)
#include "j2k_host.h"

using namespace j2k;

// ------------------------------------------------------------------------------
// status / errors
// ------------------------------------------------------------------------------
int fail(j2k_ctx *ctx, int status, const char *msg) {
    if (ctx) ctx->last_error = msg ? msg : "";
    return status;
}
int fail_hip(j2k_ctx *ctx, hipError_t e, const char *where) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
    if (ctx) ctx->last_error = buf;
    return J2K_ERR_HIP;
}

extern "C" const char *j2k_ctx_last_error(j2k_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

// ------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------
// contexts of this process that have built an MQ-coder plan: two or more = frames in flight, the MQ kernels then take their
// throughput settings (t1.hip: blocks per wavefront of the encode chains, plane-stepped decoder), one = latency settings
std::atomic<int> g_mq_ctxs{0};
bool mq_throughput_mode() { return g_mq_ctxs.load(std::memory_order_relaxed) >= 2; }
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
        OPT("t2_parallel", v == 0 || v == 1, c->t2_parallel = v != 0),
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
int check_fault(j2k_ctx *ctx) {
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
bool profile_pair(j2k_ctx *ctx, int tag, hipEvent_t &e0, hipEvent_t &e1) {
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

int stage_reserve(j2k_ctx *ctx, int slot, size_t bytes) {
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

