// gather.cpp -- the ONE exchange step of the multi-GPU path at the C ABI: the compressed streams of the tiles a
// rank coded travel to rank 0 for codestream assembly.  Replaces, for a host that shards a frame's tiles over N GPUs, the
// in-process append of encoder.encodeTile's results (encoder.go:690-742: the goroutine fan-out that collects every job's
// bytes into one tile buffer) by
//     ncclAllGather of the per-stream byte counts (u64 x count per rank), exclusive scan on the host -> offsets,
//     one ncclGroupStart / ncclGroupEnd: every rank != 0 ncclSend(pack, bytes, ncclUint8, 0); rank 0 posts the matching
//     ncclRecv into its receive buffer at the scanned offsets
// -- direct peer -> root transfers, each peer over its own xGMI link to rank 0 (7 links x ~153 GB/s into the root), no ring.
// What travels is whatever the caller passes; the intended payload is the transport form of j2k_plan_pack_stream (blocks without
// the reference's MEL zero runs, 8.3 instead of 20.6 MB per C2 frame), rebuilt on rank 0 with j2k_plan_unpack_streams.
//
// RCCL is bound at run time (dlopen "librccl.so.1", then "librccl.so"): libj2kgfx.so has no link-time dependency on it, a
// single-GPU host never loads it, and in a process that already holds an RCCL (PyTorch's nccl backend) the same library
// instance is used.  Transfers run on the communicator's own HIP stream, ordered behind the producing contexts by events
// (j2k_gather_streams) and ahead of the consuming one (j2k_comm_wait), so a context's kernels of the next frame do not queue
// behind the bytes of the previous one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "j2k_plan.h"

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl g_rccl;

bool rccl_load() {
    if (g_rccl.h) return true;
    // J2K_RCCL_LIB names the library instead of the two sonames (an embedder with its own RCCL build; the CPU test of the
    // "no RCCL on this host" path points it at a file that does not exist)
    const char *override_name = getenv("J2K_RCCL_LIB");
    std::string why;
    for (const char *name : {override_name ? override_name : "librccl.so.1", override_name ? override_name : "librccl.so"}) {
        g_rccl.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) break;
        const char *e = dlerror();                        // ONE call: dlerror() clears the message it returns
        why = e ? e : "dlopen failed";
    }
    if (!g_rccl.h) { g_rccl.err = std::string("RCCL not found: ") + why; return false; }
#define J2K_SYM(field, sym) \
    *reinterpret_cast<void **>(&g_rccl.field) = dlsym(g_rccl.h, sym); \
    if (!g_rccl.field) { g_rccl.err = std::string("RCCL lacks ") + sym; dlclose(g_rccl.h); g_rccl.h = nullptr; return false; }
    J2K_SYM(GetUniqueId, "ncclGetUniqueId")
    J2K_SYM(CommInitRank, "ncclCommInitRank")
    J2K_SYM(CommDestroy, "ncclCommDestroy")
    J2K_SYM(AllGather, "ncclAllGather")
    J2K_SYM(Send, "ncclSend")
    J2K_SYM(Recv, "ncclRecv")
    J2K_SYM(GroupStart, "ncclGroupStart")
    J2K_SYM(GroupEnd, "ncclGroupEnd")
    J2K_SYM(GetErrorString, "ncclGetErrorString")
#undef J2K_SYM
    return true;
}
}  // namespace

struct j2k_comm {
    j2k_ctx *ctx = nullptr;              // for the error text only (never dereferenced by destroy: the context may be gone)
    int device = 0;                      // ctx->device at creation
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;        // transfers run here
    hipEvent_t ready = nullptr;          // producers -> this stream
    hipEvent_t done = nullptr;           // this stream -> consumers (recorded after every gather)
    uint64_t *d_sizes = nullptr;         // device: my `count` sizes | world * count gathered sizes
    uint64_t *h_sizes = nullptr;         // pinned host mirror
    size_t sizes_cap = 0;                // count the size buffers were made for
    std::string last_error;
};

static int cfail(j2k_comm *c, int code, const std::string &msg) {
    if (c) { c->last_error = msg; if (c->ctx) c->ctx->last_error = msg; }
    return code;
}
#define NCCLCHK(c, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return cfail(c, J2K_ERR_HIP, std::string("RCCL: ") + g_rccl.GetErrorString(r_) + " at " #expr); } while (0)
#define HIPCHKC(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cfail(c, J2K_ERR_HIP, std::string("HIP: ") + hipGetErrorString(e_) + " at " #expr); } while (0)

extern "C" const char *j2k_comm_load_error(void) { return g_rccl.err.c_str(); }

extern "C" int j2k_comm_get_unique_id(uint8_t *id128) {
    if (!id128) return J2K_ERR_INVALID_ARG;
    if (!rccl_load()) return J2K_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return J2K_ERR_HIP;
    static_assert(sizeof(id) == J2K_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return J2K_OK;
}

extern "C" int j2k_comm_create(j2k_ctx *ctx, const uint8_t *id128, int rank, int world, j2k_comm **out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return J2K_ERR_INVALID_ARG;
    *out = nullptr;
    if (!rccl_load()) { ctx->last_error = g_rccl.err; return J2K_ERR_UNSUPPORTED; }
    j2k_comm *c = new j2k_comm();
    c->ctx = ctx; c->device = ctx->device; c->rank = rank; c->world = world;
    auto bail = [&](int code) { j2k_comm_destroy(c); return code; };
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(J2K_ERR_HIP);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { ctx->last_error = std::string("RCCL: ncclCommInitRank: ") + g_rccl.GetErrorString(r); c->comm = nullptr; return bail(J2K_ERR_HIP); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(J2K_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess) return bail(J2K_ERR_HIP);
    if (hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) return bail(J2K_ERR_HIP);
    *out = c;
    return J2K_OK;
}

extern "C" void j2k_comm_destroy(j2k_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *j2k_comm_last_error(j2k_comm *c) { return c ? c->last_error.c_str() : ""; }
extern "C" void *j2k_comm_stream(j2k_comm *c) { return c ? (void *)c->stream : nullptr; }

// see include/j2kgfx.h
extern "C" int j2k_gather_streams(j2k_comm *c, int count, const uint8_t *const *d_send, const uint64_t *send_bytes,
                                  const uint64_t *all_bytes, j2k_ctx *const *producers, int nproducers,
                                  uint8_t *d_recv, size_t recv_cap, uint64_t *recv_offs, int flags) {
    if (!c || count < 1 || !d_send || !send_bytes || !recv_offs || nproducers < 0 || (nproducers && !producers)) return J2K_ERR_INVALID_ARG;
    const bool self_loop = (flags & J2K_GATHER_SELF_LOOP) != 0;
    if (self_loop && c->world != 1) return cfail(c, J2K_ERR_INVALID_ARG, "J2K_GATHER_SELF_LOOP is for a one-rank communicator");
    const int W = c->world;
    HIPCHKC(c, hipSetDevice(c->device));
    // the packs are produced on the contexts' streams: this stream starts behind them
    for (int i = 0; i < nproducers; i++) {
        if (!producers[i]) return J2K_ERR_INVALID_ARG;
        HIPCHKC(c, hipEventRecord(c->ready, producers[i]->stream));
        HIPCHKC(c, hipStreamWaitEvent(c->stream, c->ready, 0));
    }
    for (int f = 0; f < count; f++)
        if (send_bytes[f] && !d_send[f]) return cfail(c, J2K_ERR_INVALID_ARG, "null send pointer with a non-zero size");
    // ---- byte counts of every rank's streams: given, or one ncclAllGather (u64 x (count + 1) per rank: the last word is the
    //      rank's recv_cap, of which rank 0's is the one that counts -- every rank then sees the same capacity verdict) ----
    std::vector<uint64_t> sizes((size_t)W * count);
    const bool root = c->rank == 0;
    uint64_t root_cap;
    if (all_bytes) {
        memcpy(sizes.data(), all_bytes, sizes.size() * sizeof(uint64_t));
        for (int f = 0; f < count; f++)
            if (sizes[(size_t)c->rank * count + f] != send_bytes[f]) return cfail(c, J2K_ERR_INVALID_ARG, "all_bytes disagrees with send_bytes for this rank");
        root_cap = root ? (uint64_t)recv_cap : (recv_cap ? (uint64_t)recv_cap : ~uint64_t(0));   // a peer passes rank 0's capacity, or 0 = not known
    } else {
        const size_t per = (size_t)count + 1;
        if (c->sizes_cap < per) {
            HIPCHKC(c, hipStreamSynchronize(c->stream));
            if (c->d_sizes) HIPCHKC(c, hipFree(c->d_sizes));
            if (c->h_sizes) HIPCHKC(c, hipHostFree(c->h_sizes));
            c->d_sizes = nullptr; c->h_sizes = nullptr; c->sizes_cap = 0;
            HIPCHKC(c, hipMalloc((void **)&c->d_sizes, (size_t)(W + 1) * per * sizeof(uint64_t)));
            HIPCHKC(c, hipHostMalloc((void **)&c->h_sizes, (size_t)(W + 1) * per * sizeof(uint64_t), hipHostMallocDefault));
            c->sizes_cap = per;
        }
        memcpy(c->h_sizes, send_bytes, (size_t)count * sizeof(uint64_t));
        c->h_sizes[count] = (uint64_t)recv_cap;
        HIPCHKC(c, hipMemcpyAsync(c->d_sizes, c->h_sizes, per * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        NCCLCHK(c, g_rccl.AllGather(c->d_sizes, c->d_sizes + per, per, ncclUint64, c->comm, c->stream));
        HIPCHKC(c, hipMemcpyAsync(c->h_sizes + per, c->d_sizes + per, (size_t)W * per * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHKC(c, hipStreamSynchronize(c->stream));
        for (int r = 0; r < W; r++) memcpy(&sizes[(size_t)r * count], c->h_sizes + per + (size_t)r * per, (size_t)count * sizeof(uint64_t));
        root_cap = c->h_sizes[per + count];
    }
    // exclusive scan, rank-major: stream f of rank r lands at recv_offs[r * count + f]
    uint64_t tot = 0;
    for (size_t i = 0; i < sizes.size(); i++) {
        recv_offs[i] = tot;
        if (sizes[i] > (uint64_t)1 << 40) return cfail(c, J2K_ERR_INVALID_ARG, "a gathered byte count is not plausible");   // same verdict on every rank: they all scan the same counts
        tot += (sizes[i] + 15) & ~uint64_t(15);          // every stream starts 16-byte aligned (j2k_plan_unpack_streams reads 16-byte words)
    }
    recv_offs[sizes.size()] = tot;
    // A collective must fail on every rank or on none: a root that returned before posting its receives would leave the peers'
    // ncclSend -- and with them their communicator streams, j2k_comm_wait and j2k_comm_destroy -- waiting for ever.  So the
    // peers always send; a root that cannot take the bytes (buffer too small / missing / misaligned) receives them into a
    // scratch buffer, drains its stream and THEN returns the status.  With gathered counts every rank sees rank 0's recv_cap
    // and returns the same J2K_ERR_CAPACITY; with host-provided counts a peer compares with the capacity it was given.
    int status = J2K_OK;
    const char *why = nullptr;
    if (tot > root_cap) { status = J2K_ERR_CAPACITY; why = "receive buffer too small for the gathered streams"; }
    if (root && status == J2K_OK && !d_recv) { status = J2K_ERR_INVALID_ARG; why = "rank 0 needs a receive buffer"; }
    if (root && status == J2K_OK && ((uintptr_t)d_recv & 15)) { status = J2K_ERR_INVALID_ARG; why = "receive buffer must be 16-byte aligned"; }
    uint8_t *scratch = nullptr;
    uint8_t *dst = d_recv;
    const bool transfers = W > 1 || self_loop;
    if (root && status != J2K_OK && transfers && tot) {
        if (hipMalloc((void **)&scratch, (size_t)tot) != hipSuccess)
            return cfail(c, J2K_ERR_HIP, std::string(why) + "; and no scratch memory to drain the peers' sends into -- the communicator is unusable");
        dst = scratch;
    }
    if (root && status != J2K_OK && !scratch) return cfail(c, status, why);      // nothing is in flight towards this rank
    // ---- the transfers: one group, direct peer -> root ----
    NCCLCHK(c, g_rccl.GroupStart());
    ncclResult_t gr = ncclSuccess;
    if (!root || self_loop)
        for (int f = 0; f < count && gr == ncclSuccess; f++)
            if (send_bytes[f]) gr = g_rccl.Send(d_send[f], (size_t)send_bytes[f], ncclUint8, 0, c->comm, c->stream);
    if (root) {
        for (int r = self_loop ? 0 : 1; r < W && gr == ncclSuccess; r++)
            for (int f = 0; f < count && gr == ncclSuccess; f++) {
                const uint64_t n = sizes[(size_t)r * count + f];
                if (n) gr = g_rccl.Recv(dst + recv_offs[(size_t)r * count + f], (size_t)n, ncclUint8, r, c->comm, c->stream);
            }
    }
    ncclResult_t ge = g_rccl.GroupEnd();
    if (scratch) { (void)hipStreamSynchronize(c->stream); (void)hipFree(scratch); }
    if (gr != ncclSuccess) return cfail(c, J2K_ERR_HIP, std::string("RCCL: ") + g_rccl.GetErrorString(gr) + " while posting the transfers");
    NCCLCHK(c, ge);
    if (status != J2K_OK) { HIPCHKC(c, hipEventRecord(c->done, c->stream)); return cfail(c, status, why); }
    // rank 0's own streams: a device copy to their place (they need no link)
    if (root && !self_loop)
        for (int f = 0; f < count; f++)
            if (send_bytes[f]) HIPCHKC(c, hipMemcpyAsync(d_recv + recv_offs[f], d_send[f], (size_t)send_bytes[f], hipMemcpyDeviceToDevice, c->stream));
    HIPCHKC(c, hipEventRecord(c->done, c->stream));
    return J2K_OK;
}

extern "C" int j2k_comm_wait(j2k_comm *c, j2k_ctx *consumer) {
    if (!c) return J2K_ERR_INVALID_ARG;
    HIPCHKC(c, hipSetDevice(c->device));
    if (consumer) HIPCHKC(c, hipStreamWaitEvent(consumer->stream, c->done, 0));     // device-side: the consumer's next kernels see the bytes
    else HIPCHKC(c, hipStreamSynchronize(c->stream));                               // host-side
    return J2K_OK;
}
