// j2k_hostcalls.cpp -- the host (unit) calls: one per reference function, caller-owned host buffers (C ABI of libj2kgfx.so, include/j2kgfx.h; shared declarations: j2k_host.h)
#include "j2k_host.h"

using namespace j2k;

// ------------------------------------------------------------------------------
// host (unit) calls
// ------------------------------------------------------------------------------
int cached_plan(j2k_ctx *ctx, const PlanSpec &S, j2k_plan **out) {
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


// a plan-owned device buffer that grows with what the call needs (the stream is drained before it is replaced)
static int ensure_sized(j2k_ctx *ctx, void **p, size_t *cur, size_t need) {
    if (*p && *cur >= need) return J2K_OK;
    if (*p) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(*p)); *p = nullptr; *cur = 0; }
    HIPCHK(ctx, hipMalloc(p, std::max<size_t>(need, 64)));
    *cur = std::max<size_t>(need, 64);
    return J2K_OK;
}

// ---- pixels at native width from / to HOST memory: the one-call forms a cgo Encode() / Decode() binds ----------------------------
// image.*.Pix (host) -> H2D at native width (4 x fewer bytes than int32 planes) -> forward transform -> block coder -> tile-parts
// (reference-mode plan: SOT | SOD | the tile's concatenated block bytes, encoder.createTileHeader of encodeTile's output; closed-loop
// plan: SOT | SOD | packets) -> D2H at the exact length.  Synchronous; bench.py --io host shows what pipelining adds on top.
extern "C" int j2k_encode_pixels_host(j2k_plan *P, int format, const void *pix, size_t stride, int sop, int eph, uint8_t *out, size_t cap,
                                      size_t *out_len, uint64_t *tile_offs, uint32_t *lens, uint8_t *numbps) {
    if (!P || !pix || !out_len) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    const PlanSpec &S = P->spec;
    const int pb = format == J2K_PIX_GRAY8 ? 1 : format == J2K_PIX_GRAY16 ? 2 : (format == J2K_PIX_RGBA8 || format == J2K_PIX_NRGBA8) ? 4 :
                   (format == J2K_PIX_RGBA64 || format == J2K_PIX_NRGBA64) ? 8 : 0;
    if (!pb || stride < (size_t)S.W * pb) return fail(ctx, J2K_ERR_INVALID_ARG, "pixel format / stride");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nb = P->blocks.size(), nt = (size_t)P->tile_count;
    const size_t bound = S.closed_loop ? j2k_plan_frame_bound(P) : j2k_plan_tile_parts_bound(P);
    int r;
    if ((r = ensure_sized(ctx, &P->d_host_pix, &P->host_pix_bytes, (size_t)S.H * stride)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_coeff, (size_t)P->coeff_elems * 4)) != J2K_OK) return r;
    if (!S.closed_loop && (r = ensure(ctx, &P->d_stream, (size_t)P->bytes_cap)) != J2K_OK) return r;
    const size_t toff_at = (bound + 64 + 15) & ~size_t(15);                                              // the tile-parts, and behind them their offsets / the length word
    if ((r = ensure_sized(ctx, &P->d_host_io, &P->host_io_bytes, toff_at + (nt + 2) * 8)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_lens, nb * 4 + 16)) != J2K_OK) return r;
    if ((r = ensure(ctx, &P->d_numbps, nb + 16)) != J2K_OK) return r;
    if (!S.closed_loop && (r = ensure(ctx, &P->d_offs, (nb + 1) * 8)) != J2K_OK) return r;
    uint64_t *d_toffs = reinterpret_cast<uint64_t *>((uint8_t *)P->d_host_io + toff_at);
    HIPCHK(ctx, hipMemcpyAsync(P->d_host_pix, pix, (size_t)S.H * stride, hipMemcpyHostToDevice, ctx->stream));
    if ((r = j2k_plan_forward_pixels(P, format, P->d_host_pix, stride, (int32_t *)P->d_coeff)) != J2K_OK) return r;
    std::vector<uint64_t> toffs(nt + 1, 0);
    if (S.closed_loop) {                                     // (blocks gathered from their coding slots straight into the tile-parts: no dense stream)
        if ((r = plan_encode_frame_from_coeff(P, (int32_t *)P->d_coeff, (uint32_t *)P->d_lens, (uint8_t *)P->d_numbps, sop, eph, (uint8_t *)P->d_host_io, bound, d_toffs)) != J2K_OK) return r;
        HIPCHK(ctx, hipMemcpyAsync(toffs.data(), d_toffs, (nt + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        if ((r = j2k_plan_encode_stream(P, (int32_t *)P->d_coeff, (uint8_t *)P->d_stream, (uint64_t *)P->d_offs, (uint32_t *)P->d_lens, (uint8_t *)P->d_numbps)) != J2K_OK) return r;
        if ((r = j2k_plan_assemble_tiles_device(P, (uint8_t *)P->d_stream, (uint64_t *)P->d_offs, (uint8_t *)P->d_host_io, d_toffs + nt)) != J2K_OK) return r;
        HIPCHK(ctx, hipMemcpyAsync(&toffs[nt], d_toffs + nt, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (lens && nb) HIPCHK(ctx, hipMemcpyAsync(lens, P->d_lens, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (numbps && nb) HIPCHK(ctx, hipMemcpyAsync(numbps, P->d_numbps, nb, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint64_t> boffs;
    if (!S.closed_loop && tile_offs) { boffs.resize(nb + 1); if (nb) HIPCHK(ctx, hipMemcpyAsync(boffs.data(), P->d_offs, (nb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream)); }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((r = check_fault(ctx)) != J2K_OK) return r;
    if (S.closed_loop && (r = j2k_plan_frame_status(P)) != J2K_OK) return r;
    const size_t total = (size_t)toffs[nt];
    *out_len = total;
    if (tile_offs) {
        if (S.closed_loop) memcpy(tile_offs, toffs.data(), (nt + 1) * 8);
        else {                                             // tile-part t starts 14 t bytes behind its tile's first block (SOT + SOD per tile before it)
            size_t j = 0;
            for (size_t t = 0; t < nt; t++) {
                while (j < nb && (size_t)P->block_tile[j] < t) j++;
                tile_offs[t] = (j < nb ? boffs[j] : boffs[nb]) + 14 * t;
            }
            tile_offs[nt] = total;
        }
    }
    if (total > cap || (total && !out)) return fail(ctx, J2K_ERR_CAPACITY, "out too small (*out_len says what the tile-parts take)");
    if (total) HIPCHK(ctx, hipMemcpy(out, P->d_host_io, total, hipMemcpyDeviceToHost));
    return J2K_OK;
}

// closed-loop plans: tile-parts (host) -> H2D -> parse -> block decode -> placement -> inverse transform -> image.*.Pix (host), one
// synchronous call.  The pixel format is the plan's: components 1 / 3 / 4, precision <= 8 -> Gray / RGBA, else Gray16 / RGBA64
// (decoder.createImage, decoder.go:417-588).
extern "C" int j2k_decode_pixels_host(j2k_plan *P, const uint8_t *cs, size_t len, int sop, int eph, void *pix, size_t stride) {
    if (!P || !cs || !pix) return J2K_ERR_INVALID_ARG;
    j2k_ctx *ctx = P->ctx;
    if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "a synchronising call while the context captures a graph");
    const PlanSpec &S = P->spec;
    if (!S.closed_loop) return fail(ctx, J2K_ERR_UNSUPPORTED, "the plan was not made with j2k_params.closed_loop: the reference has no decode body to mirror");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int r;
    const size_t pixbytes = (size_t)S.H * stride;
    if ((r = ensure_sized(ctx, &P->d_host_pix, &P->host_pix_bytes, pixbytes)) != J2K_OK) return r;
    if ((r = ensure_sized(ctx, &P->d_host_io, &P->host_io_bytes, len + 64)) != J2K_OK) return r;
    HIPCHK(ctx, hipMemcpyAsync(P->d_host_io, cs, len, hipMemcpyHostToDevice, ctx->stream));
    if ((r = j2k_plan_decode_frame_pixels(P, (const uint8_t *)P->d_host_io, len, nullptr, sop, eph, P->d_host_pix, stride)) != J2K_OK) return r;
    HIPCHK(ctx, hipMemcpyAsync(pix, P->d_host_pix, pixbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return j2k_plan_frame_status(P);
}
