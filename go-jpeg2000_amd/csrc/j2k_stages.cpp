// j2k_stages.cpp -- the plan calls: transform stages, block coders, compaction / transport, pixels, colour conversions; stand-alone coders (C ABI of libj2kgfx.so, include/j2kgfx.h; shared declarations: j2k_host.h)
#include "j2k_host.h"

using namespace j2k;

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
static void pix_launch(LevelLaunch &L, const LevelTab &T, const PixIO &pix, int cls, const PlanSpec &S) {
    L.pix_stride = pix.stride;
    L.pix_src = cls ? (pix.triple == 4 ? 4 : 0) : pix.single;
    L.comp_elems = (long long)S.W * S.H;
    if (cls == 1 && pix.triple == 4) L.pnjobs = T.pnjobs;      // (a table kept for pixel sources only: mk() hides it)
}

int plan_forward_impl(j2k_plan *P, const void *d_frame, void *d_coeff, PixIO pix) {
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

int plan_inverse_impl(j2k_plan *P, const void *d_coeff, void *d_frame, PixIO pix) {
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
int ensure(j2k_ctx *ctx, void **p, size_t bytes) {
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
T1Workspace t1_workspace(const j2k_ctx *ctx, size_t n, size_t wpj) {
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

// Block coding into the PLAN's slot buffer, no compaction (j2k_plan_encode_stream gathers it into the dense stream; the closed-loop frame
// encoder gathers it straight into the tile-parts).  HT: the slot buffer is private, so the MEL zero bytes are not written into it -- whoever
// gathers emits them (gather_job, j2k_internal.h; P->d_maglens says where).
int plan_encode_private_slots(j2k_plan *P, const int32_t *d_coeff, uint32_t *d_lens, uint8_t *d_numbps) {
    j2k_ctx *ctx = P->ctx;
    const int n = (int)P->blocks.size();
    if (!P->d_slots) {
        if (ctx->capturing) return fail(ctx, J2K_ERR_INVALID_ARG, "capture: run the same calls once before j2k_ctx_capture_begin");
        HIPCHK(ctx, hipMalloc(&P->d_slots, (size_t)P->bytes_cap + 64));
    }
    if (P->spec.coder != J2K_CODER_HT) return j2k_plan_encode_blocks(P, d_coeff, (uint8_t *)P->d_slots, d_lens, d_numbps);
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
    int r = plan_encode_private_slots(P, d_coeff, d_lens, d_numbps);
    if (r != J2K_OK) return r;
    if (P->spec.coder == J2K_CODER_HT) {
        for (int rep_ = 0; rep_ < dev_reps(16); rep_++)
        // the transport offsets (a second running sum in the scan, +4 us) only once j2k_plan_pack_stream has asked for them
        HIPCHK(ctx, launch_compact(ctx->stream, P->d_bjobs_alias ? P->d_bjobs_alias : P->d_bjobs, n, (const uint8_t *)P->d_slots, d_lens, d_offs, d_stream, P->d_maglens,
                                   P->want_toffs ? P->d_mels : nullptr, P->want_toffs ? P->d_toffs : nullptr));
        P->toffs_valid = P->want_toffs;
        P->last_stream = d_stream; P->last_lens = d_lens;
        return J2K_OK;
    }
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

