// j2k_planbuild.cpp -- plans: a frame geometry compiled into device job tables (C ABI of libj2kgfx.so, include/j2kgfx.h; shared declarations: j2k_host.h)
#include "j2k_host.h"

using namespace j2k;

// ------------------------------------------------------------------------------
// plan construction
// ------------------------------------------------------------------------------
bool PlanSpec::operator==(const PlanSpec &o) const {
    return W == o.W && H == o.H && C == o.C && tile_w == o.tile_w && tile_h == o.tile_h && levels == o.levels &&
           frame_h == o.frame_h && wavelet == o.wavelet && precision == o.precision && dc_shift == o.dc_shift && dc_shift_inv == o.dc_shift_inv && mct == o.mct && quant == o.quant && quality == o.quality &&
           num_res_jobs == o.num_res_jobs && cb_w == o.cb_w && cb_h == o.cb_h && coder == o.coder &&
           tile_first == o.tile_first && tile_count == o.tile_count && frame_is_f64 == o.frame_is_f64 && closed_loop == o.closed_loop;
}

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

int build_plan(j2k_ctx *ctx, const PlanSpec &S, j2k_plan **out) {
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
        if (r == J2K_OK && P->spec.closed_loop && P->spec.coder == J2K_CODER_HT) {
            // closed-loop HT plans: the frame decoder writes every block straight into its window of the coefficient planes (ht_decode_kernel<true>)
            for (size_t i = 0; i < bj.size(); i++) bj[i].out_off = bj[i].src_off;
            r = upload(ctx, &P->d_djobs_placed, bj);
        }
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
    void *ptrs[] = {P->d_deep_jobs_inv, P->d_mega_fwd_jobs, P->d_mega_inv_jobs, P->d_fwd_top_jobs, P->d_inv_top_jobs, P->d_inv_wg_jobs, P->d_deep_planes, P->d_deep_jobs, P->d_tile_job0, P->d_scrA, P->d_scrB, P->d_tail, P->d_bjobs, P->d_djobs, P->d_djobs_placed, P->d_frame, P->d_coeff, P->d_slots, P->d_stream, P->d_lens, P->d_numbps, P->d_offs, P->d_status, P->d_fwd_pix_jobs, P->d_fwd_wg2_jobs, P->d_fwd_wg_rest_jobs, P->d_fwd_wg_jobs, P->d_fwd97_wg_jobs, P->d_inv97_wg_jobs, P->d_ht_ujobs, P->d_ht_alias_next, P->d_bjobs_alias, P->d_maglens, P->d_mels, P->d_toffs,
                    P->d_t2_packets, P->d_tile_packet0, P->d_t2_cbs, P->d_t2_poffs, P->d_t2_ptile, P->d_t2_ws, P->d_t2_chains, P->d_t2_body_base, P->d_t2_par, P->d_frame_status,
                    P->d_cl_decoded, P->d_cl_coeff, P->d_cl_coeff_dec, P->d_cl_numbps, P->d_cl_offs, P->d_cl_lens, P->d_host_io, P->d_host_pix};
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
