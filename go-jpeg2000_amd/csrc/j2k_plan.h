// j2k_plan.h -- host-side context / plan objects behind the C ABI (include/j2kgfx.h).
#pragma once
#include <string>
#include <vector>

#include "../../include/j2kgfx.h"
#include "j2k_internal.h"

struct j2k_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string last_error;
    int fuse_compact = 0;      // j2k_plan_encode_stream: encode + compact in ONE kernel (decoupled look-back; J2K_FUSE_COMPACT=1).
                               // Measured slower than the three kernels it replaces (61.7 vs 58.6 us: the prefix chain crosses XCDs)
    int fwd_link = 1;          // forward 5-3: bands of one workgroup exchange halo rows through LDS (J2K_FWD_LINK)
    int inv_link = 1;          // same for the inverse kernels (J2K_INV_LINK)
    int pix_fuse = 1;          // J2K_PIX_FUSE: packed pixels read / written by the level-0 kernels (1: every format; 2: RGBA8 and Gray16 only, as before round 4; 0: always the int32 staging frame)
    bool plane_wg3 = false;    // ... also level 0 of RGB triples of int32 planes (J2K_PLANE_WG3; measured equal to the general kernel on 4K frames: 43 vs 41 us)
    int plane_wg = 4;          // single-component planes (every level > 0, gray level 0) in workgroup form: waves per workgroup (J2K_PLANE_WG: 0 off, 4, 8)
    int l0_fuse = 0;           // forward levels 0 + 1 of RGBA8 frames in one launch: waves per workgroup of the fused bands (J2K_L0_FUSE: 0 off, 8, 16)
    int l0_wg = 8;             // packed RGBA8 level-0 forward, workgroup form: wavefronts per workgroup (J2K_L0_WG: 0 off, 4, 8)
    int l0_wg_invw = 4;        // ... of the inverse (J2K_L0_WG_INVW: 0 = same as the forward, 4, 8)
    bool l0_deal = true;       // J2K_L0_DEAL (0 = one contiguous chunk per XCD): the RGBA8 level-0 job table deals each XCD's short bands after its full ones (as the 9-7 tables do)
    int l0_wg97 = 8;           // lossy level-0 forward of an RGB triple, workgroup form: waves per workgroup (J2K_L0_WG97: 0 off, 6..16 even; measured 4K: 8 -> 78 us, 16 -> 88 us, general kernel 160 us)
    int plane_wg97 = 8;        // deeper 9-7 levels (single float64 planes) in workgroup form: waves per workgroup (J2K_PLANE_WG97: 0 = general kernels, 8)
    int l0_wg97_inv = 8;       // lossy level-0 inverse of an RGB triple, workgroup form: waves per workgroup (J2K_L0_WG97_INV: 0 off, 6 8 10 12)
    int l0_xcd_group = 16;     // J2K_L0_XCD_GROUP (inverse level-0 table): > 0 = bands go to the XCDs in groups of this many consecutive ones (0: one contiguous chunk of the table per XCD)
    bool l0_xcd = true;        // XCD-aware order of the workgroup jobs (J2K_L0_XCD=0: plane-major order)
    bool ht_alias = true;      // j2k_plan_encode_stream (HT): code each distinct block window once (J2K_HT_ALIAS=0: every job)
    int l0_inv_wpe = 5;        // its occupancy variant (J2K_L0_INV_WPE: 5 = all in registers, 26.4 us; 6 = odd row parked in LDS for 6 waves per SIMD, measured slower: 33.6 us)
    bool l0_wg_inv = true;     // the inverse level 0 to RGBA8 in workgroup form too (J2K_L0_WG_INV=0: the general kernel)
    int l0_store = 1;          // its final-coefficient store flavour (J2K_L0_STORE: 0 plain, 1 nt, 2 sc1, 4 sc1 nt)
    int band_prows_pix = 3;    // packed-pixel level-0 forward (J2K_BAND_PROWS_PIX)
    int band_prows_97 = 8;     // 9-7 kernels: 7 halo rows per band, so taller bands (J2K_BAND_PROWS_97)
    int band_prows_inv = 0;    // 0: same as band_prows (J2K_BAND_PROWS_INV)
    int fwd_pf = 0;            // forward 5-3 level kernels: software prefetch of the next pair-row (J2K_FWD_PF)
    int band_prows = 5;        // pair-rows per wavefront band (tunable: J2K_BAND_PROWS)
    int use_tail = 1;          // J2K_TAIL=0: every level as its own launch (A/B)
    int mega = 0;              // J2K_MEGA: packed-RGBA8 frames: the deep levels and the level-0 bands independent of them in ONE launch per direction
                               // (dwt53_mega_*_kernel); 0 (default) = level 0 as one launch + the deep launch; 1 / 2 = job order (deep, bands, flat / deep, flat, bands).
                               // Measured (C2, one frame in flight): forward 14.3 + 23.9 us against 22.2 + 15.6, inverse 29.5 + 16.3 against 16.0 + 25.0 -- the launch's
                               // LDS size is that of its largest role (112-144 KB), so the level-0 bands run one 16-wave workgroup per CU and lose what the overlap gains
    int deep_mid_inv = 1;      // J2K_DEEP_MID_INV: the same split in the inverse launch (1 = deep + mid + flat with the compact LDS layout: two workgroups per CU, the default since round 4; 0 = deep + flat; 2 = deep + mid, flat rows inside them)
    int deep_min_planes = 12;  // J2K_DEEP_MIN_PLANES: fewer tile-components than this keep the per-level launches
    int deep_mid = 1;          // J2K_DEEP_MID=0: the deep workgroup keeps the whole top half of its plane (no mid workgroup beside it)
    int use_deep = 1;          // J2K_DEEP=0: level tail_l0 - 1 as its own launch + the LDS tail (round 2) instead of ONE launch for every level below 0 (dwt53_deep.inc)
    int xcd_map = 0;           // J2K_XCD_MAP=0: plain job order (A/B)
    int cpl0 = 0;              // J2K_CPL0: force columns-per-lane of the level-0 5-3 kernels (tuning)
    int force_novec = 0;       // J2K_FORCE_NOVEC=1: always take the scalar-access kernels (testing)
    int t1_split = 1;          // J2K_T1_SPLIT=0: T1 encoder as one kernel (contexts + MQ chain on lane 0) instead of two
    long t1_sym_mb = 8192;     // J2K_T1_SYM_MB: cap of the T1 symbol workspace; blocks with more bit planes than fit take the one-kernel path
    int t1_dec_general = 0;    // J2K_T1_DEC_GENERAL=1: every block on the general T1 decode kernel (A/B against t1_decode64_kernel)
    int t1_dec_lanes = 2;      // J2K_T1_DEC_LANES: 2 = the plane-stepped decoder as one launch per group of 64 blocks (t1_lanes.inc, PERSIST), 1 = its passes as launches per plane, 0 = the plane-stepped decoder's SigProp / Cleanup as round 2's step kernels (one block per wavefront) instead of t1_lanes.inc
    int t1_dec_split = -1;     // J2K_T1_DEC_SPLIT: MQ decode of frames with at least this many blocks runs plane by plane with the MagRef chains
                               // of 64 blocks per wavefront in lock step (t1.hip); 0: always the one-launch kernels; -1 (default): 512 while
                               // at least two contexts of this process code with the MQ coder (frames in flight: throughput), else 0 (one
                               // frame at a time: latency)
    bool capturing = false;    // between j2k_ctx_capture_begin / _end: the plan calls are recorded into a HIP graph, nothing may allocate or synchronise
    bool fault_armed_before_capture = false;
    bool counted_mq = false;   // this context has built an MQ-coder plan (counted in g_mq_ctxs)
    bool t2_parallel = true;   // J2K_T2_PARALLEL: frame decode of SOP + EPH streams parses a tile's packets side by side from their markers (verified; 0 = the tile chain only)
    int t1_lanes = 0;          // J2K_T1_LANES: blocks per wavefront of the lane-parallel MQ kernel (0: blocks / 256, at most 32, while at least two contexts code with the MQ coder, else blocks / 2048)
    // cached single-plane plans for the host (unit) calls
    std::vector<j2k_plan *> cache;
    // host-call staging buffers (device)
    void *stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};    // 0 / 1 staging, 2 T1 workspace, 3 fault word + results, 4 symbol lists of the big-block encoder
    bool fault_armed = false;  // a block-encode launch may have written the sticky fault word (stage[3]) since the last check
    size_t stage_bytes[5] = {0, 0, 0, 0, 0};
    // level-0 kernel timing (j2k_ctx_profile_*)
    int profile = 0;
    std::vector<hipEvent_t> ev;     // pool of event pairs
    std::vector<int> ev_tag;        // per pair: 0 = forward level 0, 1 = forward deeper levels, 2 = inverse level 0, 3 = inverse deeper levels
    size_t ev_used = 0;             // events recorded since the last read
};

namespace j2k {

enum Wavelet { W53 = 0, W97 = 1 };
enum QuantMode {               // how the 9-7 path turns f64 coefficients into int32
    Q_NONE = 0,                // keep f64 (dwt.Forward97... unit calls)
    Q_ENCODER = 1,             // int32(v/step +- 0.5), step = 1/quality (encoder.go:265-276)
    Q_TCD = 2                  // int32(v +- 0.5) (tcd.go:520-532)
};

// Internal, richer form of j2k_params.
struct PlanSpec {
    int W = 0, H = 0, C = 1;
    int tile_w = 0, tile_h = 0;
    int levels = 1;            // DWT levels actually run
    int wavelet = W53;
    int precision = 8;         // component precision of the plan (pixel pack / unpack)
    int dc_shift = 0;          // value subtracted on the way in (encoder.go:218-220)
    int dc_shift_inv = 0;      // value added on the way out: 0 for signed components (decoder.go:344-348), else dc_shift
    int mct = 0;               // fused RCT (W53) / ICT (W97) on comps 0-2 when C>=3
    int quant = Q_NONE;
    int quality = 100;
    int num_res_jobs = 6;      // resolutions for the encodeTile job enumeration
    int cb_w = 64, cb_h = 64;
    int coder = J2K_CODER_MQ;
    int frame_h = 0;           // rows of one frame of a batch (j2k_params.frame_rows; = H for a single frame)
    int tile_first = 0, tile_count = 0;
    bool frame_is_f64 = false; // unit 9-7 calls: source/destination "frame" is f64
    bool closed_loop = false;  // j2k_params.closed_loop: code-block windows = the Mallat rectangles of the plane (they partition it)
    bool operator==(const PlanSpec &o) const;
};

struct Group {                 // one launch unit: a single component or an MCT triple of one tile
    int tile;                  // shard-local tile index
    int comp0, nc;
    int x0, y0, w, h;
    int64_t coef_off[3];       // element offsets into the coefficient buffer
    int64_t scrA_off[3], scrB_off[3];
};

struct LevelTab {
    DwtPlane *d_planes = nullptr;
    DwtJob *d_jobs = nullptr;
    int njobs = 0, nplanes = 0;
    int cpl = 2, vec = 0, ncomp = 1;
    int64_t alg_bytes = 0;
    // single-component planes in workgroup form (dwt53_plane_wg.inc): one entry per (plane, 512-column strip, band of
    // pwaves - 1 pair-rows); empty when a plane of the level breaks the geometry contract
    DwtJob *d_pjobs = nullptr;
    int pnjobs = 0, pwaves = 0, pmulti = 0;
    bool p_pix_only = false;   // the table serves packed-pixel sources only (level 0 of RGB triples: RGBA64); int32 planes keep the general kernels
};

}  // namespace j2k

struct j2k_plan {
    j2k_ctx *ctx = nullptr;
    j2k::PlanSpec spec;
    int tiles_x = 1, tiles_y = 1, tile_first = 0, tile_count = 1;
    std::vector<j2k::Group> groups;
    std::vector<int64_t> plane_desc;        // 7 x int64 per tile-component
    int64_t coeff_elems = 0, scrA_elems = 0, scrB_elems = 0;
    // [cls][level]: cls 0 = single-component groups, cls 1 = MCT triples
    std::vector<j2k::LevelTab> fwd[2], inv[2];
    void *d_scrA = nullptr, *d_scrB = nullptr;   // int32 (5-3) or f64 (9-7) prefixes
    // fused LDS tail (5-3): levels tail_l0 .. levels-1 in one launch per direction; -1 = not used
    int tail_l0 = -1;
    j2k::TailPlane *d_tail = nullptr;
    int ntail = 0;
    size_t tail_lds_fwd = 0, tail_lds_inv = 0;
    // every level below level 0 in one launch per direction (dwt53_deep.inc): levels deep_l0 .. levels-1; -1 = not used
    int deep_l0 = -1;
    j2k::TailPlane *d_deep_planes = nullptr;    // dims at level deep_l0
    j2k::DwtJob *d_deep_jobs = nullptr;         // deep workgroups first, then the flat ones
    int ndeep_jobs = 0;
    size_t deep_lds = 0, deep_lds_fwd = 0;      // dynamic LDS of the inverse / forward launch
    std::vector<j2k::DwtJob> deep_jobs_host, deep_jobs_host_inv, flat_jobs_host, flat_jobs_host_inv;
    j2k::DwtJob *d_deep_jobs_inv = nullptr;     // the inverse launch's table (deep / mid jobs own other level-l0 rows than in the forward one)
    int ndeep_jobs_inv = 0;
    // merged launches for packed RGBA8 frames (dwt53_mega_*_kernel) and the level-0 TOP band tables that go with them
    j2k::DwtJob *d_mega_fwd_jobs = nullptr, *d_mega_inv_jobs = nullptr, *d_fwd_top_jobs = nullptr, *d_inv_top_jobs = nullptr;
    int mega_fwd_njobs = 0, mega_inv_njobs = 0, fwd_top_njobs = 0, inv_top_njobs = 0;
    int64_t fwd_top_bytes = 0, inv_top_bytes = 0;        // algorithmic bytes of the top-band launches (16 B per pixel)
    // code-block jobs
    std::vector<j2k_block> blocks;          // plane = shard-local tile-component index
    std::vector<int32_t> block_tile;        // tile of each job
    std::vector<int32_t> block_res;         // resolution of each job (a packet = the jobs of one tile-component and resolution)
    // closed-loop frame codec (j2k_plan_encode_tile_parts / j2k_plan_decode_tile_parts): tables and workspaces, made at first use
    j2k_t2_dev_packet *d_t2_packets = nullptr;   // one packet per (tile, component, resolution) with blocks, J2K_T2_* flags set
    int *d_tile_packet0 = nullptr;               // first packet of every tile of the shard (+ the packet count)
    int t2_npackets = 0;
    j2k_t2_dev_cb *d_t2_cbs = nullptr;           // code-block table of the packet coder / decoder (one entry per job)
    uint64_t *d_t2_poffs = nullptr;              // encode: where each packet starts among the packets (+ their total)
    int t2_body_slices = 8;
    int32_t *d_t2_ptile = nullptr;               // encode: packet -> tile (its place in the tile-parts is 14 (tile + 1) bytes further)
    void *d_t2_ws = nullptr;                     // encode: the packet coder's workspace + its 3-word result
    void *d_t2_chains = nullptr;                 // decode: one chain per tile
    uint64_t *d_t2_body_base = nullptr;          // decode: where each packet's bodies start
    void *d_t2_par = nullptr;                    // decode: one chain per PACKET, the marker lists and guesses (t2_par_workspace)
    int *d_frame_status = nullptr;               // sticky status word of the asynchronous frame calls (j2k_plan_frame_status)
    int32_t *d_cl_decoded = nullptr, *d_cl_coeff = nullptr;   // j2k_plan_*_frame_pixels: decoded blocks, coefficient planes
    int32_t *d_cl_coeff_dec = nullptr;           // HT plans: the frame DECODER's coefficient planes (zeroed once, only the coded rows ever written)
    uint8_t *d_cl_numbps = nullptr; uint64_t *d_cl_offs = nullptr; uint32_t *d_cl_lens = nullptr;
    int max_block_h = 0;
    void *d_host_io = nullptr, *d_host_pix = nullptr;          // j2k_encode_pixels_host / j2k_decode_pixels_host: tile-parts / pixels on the device
    size_t host_io_bytes = 0, host_pix_bytes = 0;
    int *d_tile_job0 = nullptr;             // first job of each tile of the shard (+ the job count): j2k_plan_assemble_tiles_device
    uint64_t max_tile_bytes = 0;            // slot bytes of the largest tile (an upper bound of its stream bytes)
    std::vector<uint64_t> slot_off;         // byte offset of each job's worst-case slot
    std::vector<uint64_t> dec_off;          // element offset of each job's decoded block
    j2k::BlockJob *d_bjobs = nullptr;       // out_off = slot byte offset (encode)
    j2k::BlockJob *d_djobs = nullptr;       // out_off = decoded element offset (decode)
    j2k::BlockJob *d_djobs_placed = nullptr; // closed-loop HT plans: out_off = the block's window in the coefficient planes, stride = the plane's
    int64_t bytes_cap = 0, decoded_elems = 0, block_samples = 0;
    int64_t dwt_bytes = 0, dwt_level0_bytes = 0;
    // device workspaces owned by the plan for j2k_encode_frame
    void *d_frame = nullptr, *d_coeff = nullptr, *d_slots = nullptr, *d_stream = nullptr;
    void *d_lens = nullptr, *d_numbps = nullptr, *d_offs = nullptr;
    // fused encode + compact (j2k_plan_encode_stream): look-back status words, tagged with the launch epoch
    j2k::DwtJob *d_fwd_pix_jobs = nullptr;  // level-0 forward job table of the packed-pixel path (shorter bands)
    int fwd_pix_njobs = 0;
    j2k::HtUJob *d_ht_ujobs = nullptr; int *d_ht_alias_next = nullptr;   // HT: one entry per distinct window / the job ids sharing each window
    j2k::BlockJob *d_bjobs_alias = nullptr;                  // d_bjobs with every job's slot = the slot of its window's coded job
    int ht_nunique = 0;
    j2k::DwtJob *d_fwd97_wg_jobs = nullptr; // 9-7: one job per workgroup = (plane, component, band) of dwt97_fwd_rgb_wg_kernel
    int fwd97_wg_njobs = 0, fwd97_wg_waves = 0;
    j2k::DwtJob *d_inv97_wg_jobs = nullptr; // 9-7 inverse: one job per workgroup = (plane, band) of dwt97_inv_rgb_wg_kernel
    int inv97_wg_njobs = 0, inv97_wg_waves = 0;
    j2k::DwtJob *d_fwd_wg2_jobs = nullptr, *d_fwd_wg_rest_jobs = nullptr;   // fused levels 0+1: top-half bands / the remaining bands
    int fwd_wg2_njobs = 0, fwd_wg2_waves = 0, fwd_wg_rest_njobs = 0;
    j2k::DwtJob *d_fwd_wg_jobs = nullptr;   // the same as one job per WORKGROUP (dwt53_fwd_rgba8_wg_kernel), when every plane qualifies
    int fwd_wg_njobs = 0, fwd_wg_waves = 0;
    j2k::DwtJob *d_inv_wg_jobs = nullptr;   // the inverse kernel's table (its own waves per workgroup)
    int inv_wg_njobs = 0, inv_wg_waves = 0;
    uint32_t *d_maglens = nullptr;          // j2k_plan_encode_stream: end of each block's MagSgn bytes (the MEL hole starts there)
    uint32_t *d_mels = nullptr;      // per job: bytes of MEL zero run of an HT block (max(64, 2wh) / 4), built with d_maglens
    const void *last_stream = nullptr, *last_lens = nullptr;   // outputs of the last j2k_plan_encode_stream: what d_maglens / d_toffs describe
    bool want_toffs = false, toffs_valid = false;   // pack_stream has been used on this plan / d_toffs belongs to the last encode_stream
    uint64_t *d_toffs = nullptr;     // n + 1: exclusive scan of the transport lengths of the last j2k_plan_encode_stream (pack_stream)
    uint64_t *d_status = nullptr;
    uint32_t epoch = 0;
    bool all_blocks_fast = false;           // every job on the parallel HT path
    uint64_t *d_bigsym_off = nullptr;       // MQ plans with blocks above 64 x 64: where each job's symbol list starts (bytes; n + 1 entries), built at the first encode
    size_t bigsym_total = 0;
    bool dec_coded_rows_only = false;       // j2k_plan_set_decode_coded_rows_only: HT decode leaves the rows the reference's decoder never writes alone
};
