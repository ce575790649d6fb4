// j2k_frame.cpp -- Tier-2 on device buffers and the closed-loop frame codec (packets, tile-parts, decode body) (C ABI of libj2kgfx.so, include/j2kgfx.h; shared declarations: j2k_host.h)
#include "j2k_host.h"

using namespace j2k;

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
void plan_t2_packets(const j2k_plan *P, int layer, std::vector<j2k_t2_dev_packet> &out, std::vector<int> *tile_packet0) {
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
    std::vector<int32_t> ptile(np ? np : 1, 0);                     // packet -> its tile (packet p ends up 14 (ptile[p] + 1) bytes behind its place among the packets)
    for (int t = 0; t < P->tile_count; t++)
        for (int q = tp0[t]; q < tp0[t + 1]; q++) ptile[(size_t)q] = t;
    int r = upload(ctx, &P->d_tile_packet0, tp0);
    if (r == J2K_OK) r = upload(ctx, &P->d_t2_ptile, ptile);
    auto alloc = [&](void **p, size_t bytes) { if (r == J2K_OK) { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 64)); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMalloc (frame codec)"); } };
    alloc((void **)&P->d_t2_cbs, n * sizeof(j2k_t2_dev_cb));
    alloc((void **)&P->d_t2_poffs, (np + 1) * 8);
    alloc(&P->d_t2_ws, ((j2k::t2_dev_workspace((long)np) + 15) & ~size_t(15)) + 64);
    alloc(&P->d_t2_chains, (size_t)P->tile_count * j2k::t2_chain_bytes());
    alloc((void **)&P->d_t2_body_base, np * 8);
    alloc(&P->d_t2_par, j2k::t2_par_workspace((long)np, P->tile_count));
    if (r == J2K_OK) { hipError_t e = hipMemsetAsync(P->d_t2_par, 0, j2k::t2_par_workspace((long)np, P->tile_count), ctx->stream); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMemsetAsync"); }
    alloc((void **)&P->d_frame_status, 64);
    if (r == J2K_OK) { hipError_t e = hipMemsetAsync(P->d_frame_status, 0, 64, ctx->stream); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMemsetAsync"); }
    P->t2_npackets = (int)np;
    int max_ncb = 1;
    for (const j2k_t2_dev_packet &q : pk) max_ncb = std::max(max_ncb, (int)q.ncb);
    P->t2_body_slices = std::min(32, max_ncb);                      // wavefronts that share the bodies of one packet (t2_body_kernel)
    if (r == J2K_OK) r = upload(ctx, &P->d_t2_packets, pk);      // (last: its presence says the tables are complete)
    return r;
}

extern "C" size_t j2k_plan_frame_bound(const j2k_plan *P) {
    if (!P || !P->spec.closed_loop) return 0;
    std::vector<j2k_t2_dev_packet> pk;
    plan_t2_packets(P, 0, pk, nullptr);
    return (size_t)P->bytes_cap + 16 * P->blocks.size() + 16 * pk.size() + 14 * (size_t)P->tile_count + 64;
}

// the packets of a frame written where they end up (t2dev.hip: launch_t2_encode_tile_parts).  d_stream + d_offs: the dense block stream, or
// d_offs == NULL: d_stream is the plan's slot buffer as plan_encode_private_slots left it and every block is gathered from its slot
static int encode_tile_parts_impl(j2k_plan *P, const uint8_t *d_data, const uint64_t *d_offs, const uint32_t *d_lens, const uint8_t *d_numbps,
                                  int sop, int eph, uint8_t *d_out, size_t cap, uint64_t *d_tile_offs) {
    j2k_ctx *ctx = P->ctx;
    const long n = (long)P->blocks.size(), np = P->t2_npackets;
    const int ht = P->spec.coder == J2K_CODER_HT ? 1 : 0;
    uint64_t *d_res = reinterpret_cast<uint64_t *>((uint8_t *)P->d_t2_ws + ((j2k::t2_dev_workspace(np) + 15) & ~size_t(15)));
    HIPCHK(ctx, j2k::launch_t2_fill_cbs(ctx->stream, n, d_offs, d_lens, d_numbps, 31, ht | 2, P->d_t2_cbs, d_res));
    HIPCHK(ctx, j2k::launch_t2_encode_tile_parts(ctx->stream, P->d_t2_packets, np, P->d_t2_cbs, (uint64_t)n, d_data, sop, eph, d_out, (uint64_t)cap, P->d_t2_poffs,
                                                 P->d_t2_ws, d_res, P->d_t2_ptile, P->d_tile_packet0, P->tile_count, P->tile_first, d_tile_offs, P->d_frame_status,
                                                 d_offs ? nullptr : (P->d_bjobs_alias ? P->d_bjobs_alias : P->d_bjobs), d_offs ? nullptr : (ht ? P->d_maglens : nullptr), ht, P->t2_body_slices));
    return J2K_OK;
}

extern "C" int j2k_plan_encode_tile_parts(j2k_plan *P, const uint8_t *d_stream, const uint64_t *d_offs, const uint32_t *d_lens, const uint8_t *d_numbps,
                                          int sop, int eph, uint8_t *d_out, size_t cap, uint64_t *d_tile_offs) {
    if (!P) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (!d_stream || !d_offs || !d_lens || !d_numbps || !d_out || !d_tile_offs) return fail(ctx, J2K_ERR_INVALID_ARG, "null device pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int r = cl_prepare(P);
    if (r != J2K_OK) return r;
    return encode_tile_parts_impl(P, d_stream, d_offs, d_lens, d_numbps, sop, eph, d_out, cap, d_tile_offs);
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
    HIPCHK(ctx, j2k::launch_t2_tile_chains(ctx->stream, d_cs, (uint64_t)len, d_tile_offs, P->tile_count, P->tile_first, P->d_tile_packet0, P->d_t2_chains));
    // (SOP + EPH streams: a tile's packets side by side, checked against the serial rule and redone by it where a guess was off -- t2dec.hip)
    HIPCHK(ctx, j2k::launch_t2_decode_tiles(ctx->stream, P->d_t2_chains, P->tile_count, P->d_tile_packet0, P->d_t2_packets, P->t2_npackets, P->d_t2_cbs, (uint64_t)n, d_cs,
                                            (uint64_t)len, sop, eph, P->d_t2_body_base, P->d_frame_status, ctx->t2_parallel ? P->d_t2_par : nullptr,
                                            P->spec.coder == J2K_CODER_HT ? 1 : 0, 31, d_offs, d_lens, d_numbps));
    return J2K_OK;
}

extern "C" int j2k_plan_frame_parallel_tiles(j2k_plan *P, long *tiles) {
    if (!P || !tiles) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *tiles = 0;
    if (!P->d_frame_status) return J2K_OK;
    int n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, P->d_frame_status + 1, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(P->d_frame_status + 1, 0, sizeof n, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *tiles = n;
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
    if (P->spec.coder != J2K_CODER_HT) alloc((void **)&P->d_cl_decoded, (size_t)P->decoded_elems * 4);     // (HT: decoded in place, below)
    alloc((void **)&P->d_cl_offs, (n + 1) * 8);
    alloc((void **)&P->d_cl_lens, n * 4 + 16);
    alloc((void **)&P->d_cl_numbps, n + 16);
    // HT plans: the decoder's coefficient planes, its own (the encoder's forward transform writes d_cl_coeff) and zeroed once --
    // j2k_plan_decode_frame_pixels relies on it
    if (P->spec.coder == J2K_CODER_HT) {
        alloc((void **)&P->d_cl_coeff_dec, (size_t)P->coeff_elems * 4);
        if (r == J2K_OK) { hipError_t e = hipMemsetAsync(P->d_cl_coeff_dec, 0, std::max<size_t>((size_t)P->coeff_elems * 4, 64), ctx->stream); if (e != hipSuccess) r = fail_hip(ctx, e, "hipMemsetAsync"); }
    }
    alloc((void **)&P->d_cl_coeff, (size_t)P->coeff_elems * 4);     // (last: its presence says the workspaces are complete)
    return r;
}

// coefficients -> tile-parts with one copy of every block's bytes (the closed-loop one-call forms; d_lens / d_numbps: the caller's tables, filled)
int plan_encode_frame_from_coeff(j2k_plan *P, const int32_t *d_coeff, uint32_t *d_lens, uint8_t *d_numbps, int sop, int eph, uint8_t *d_out, size_t cap,
                                 uint64_t *d_tile_offs) {
    int r = cl_prepare(P);
    // block coding into the plan's slots, then every block's bytes ONCE: from its slot to its place in its packet in its tile-part (the stage
    // calls j2k_plan_encode_stream + j2k_plan_encode_tile_parts copy them twice: dense stream, tile-parts)
    if (r == J2K_OK) r = plan_encode_private_slots(P, d_coeff, d_lens, d_numbps);
    if (r == J2K_OK) r = encode_tile_parts_impl(P, (const uint8_t *)P->d_slots, nullptr, d_lens, d_numbps, sop, eph, d_out, cap, d_tile_offs);
    return r;
}

extern "C" int j2k_plan_encode_frame_pixels(j2k_plan *P, int format, const void *d_pix, size_t stride, int sop, int eph, uint8_t *d_out, size_t cap,
                                            uint64_t *d_tile_offs) {
    if (!P) return J2K_ERR_INVALID_ARG;
    if (!d_out || !d_tile_offs) return fail(P->ctx, J2K_ERR_INVALID_ARG, "null device pointer");
    int r = cl_prepare(P);
    if (r == J2K_OK) r = cl_workspaces(P);
    if (r == J2K_OK) r = j2k_plan_forward_pixels(P, format, d_pix, stride, P->d_cl_coeff);
    if (r == J2K_OK) r = plan_encode_frame_from_coeff(P, P->d_cl_coeff, P->d_cl_lens, P->d_cl_numbps, sop, eph, d_out, cap, d_tile_offs);
    return r;
}

extern "C" int j2k_plan_decode_frame_pixels(j2k_plan *P, const uint8_t *d_cs, size_t len, const uint64_t *d_tile_offs, int sop, int eph, void *d_pix,
                                            size_t stride) {
    if (!P) return J2K_ERR_INVALID_ARG;
    int r = cl_prepare(P);
    if (r == J2K_OK) r = cl_workspaces(P);
    if (r == J2K_OK) r = j2k_plan_decode_tile_parts(P, d_cs, len, d_tile_offs, sop, eph, P->d_cl_offs, P->d_cl_lens, P->d_cl_numbps);
    if (r != J2K_OK) return r;
    if (P->spec.coder == J2K_CODER_HT) {
        // The reference's HT decoder writes one row in four and the other rows of a fresh decoder's block are zero.  The
        // coefficient planes here are the plan's own, zeroed when they were made, and nothing else writes them (the inverse transform reads
        // its input only; the encoder's forward transform has planes of its own): the block decoder writes the coded rows straight into each
        // block's window (closed-loop windows partition the plane) -- a quarter of the stores, no dense blocks, no placement copy.
        j2k_ctx *ctx = P->ctx;
        const int n = (int)P->blocks.size();
        r = stage_reserve(ctx, 2, ht_decode_scratch_words(n) * 4 + 256);
        if (r != J2K_OK) return r;
        HIPCHK(ctx, launch_ht_decode(ctx->stream, P->d_djobs, n, d_cs, P->d_cl_offs, P->d_cl_lens, P->d_cl_coeff_dec, (uint32_t *)ctx->stage[2], 1, P->d_djobs_placed));
        return j2k_plan_inverse_pixels(P, P->d_cl_coeff_dec, d_pix, stride);
    } else {
        r = j2k_plan_decode_blocks(P, d_cs, P->d_cl_offs, P->d_cl_lens, P->d_cl_numbps, P->d_cl_decoded);
        if (r == J2K_OK) r = j2k_plan_place_blocks(P, P->d_cl_decoded, P->d_cl_coeff);
    }
    if (r == J2K_OK) r = j2k_plan_inverse_pixels(P, P->d_cl_coeff, d_pix, stride);
    return r;
}

extern "C" int j2k_plan_get_decoded_offsets(const j2k_plan *P, uint64_t *offs, size_t cap) {
    if (!P || !offs) return J2K_ERR_INVALID_ARG;
    if (cap < P->dec_off.size()) return J2K_ERR_CAPACITY;
    if (!P->dec_off.empty()) memcpy(offs, P->dec_off.data(), P->dec_off.size() * sizeof(uint64_t));
    return J2K_OK;
}

