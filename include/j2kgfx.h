/*
 * j2kgfx.h -- C ABI of libj2kgfx.so: the MI355X (gfx950) implementation of the
 * go-jpeg2000 tile-component hot path (DC shift -> MCT -> multi-level DWT ->
 * code-block entropy coding, and the inverse).
 *
 * This is the drop-in boundary.  Every entry point names the
 * reference function (file:line in mrjoshuak/go-jpeg2000) whose body a cgo shim
 * would replace with it; INTEGRATION.md shows the Go side.  Plain pointers and
 * sizes only.  All functions return J2K_OK (0) or a negative status; none abort.
 *
 * Two families:
 *   1. "host" calls -- caller-owned HOST buffers, mutated in place exactly like
 *      the Go slices they stand for; synchronous (H2D, kernels, D2H inside).
 *      One call per reference function, for unit parity and for the cgo shim.
 *   2. "plan" calls -- a frame geometry compiled once into device job tables;
 *      DEVICE pointers, asynchronous on the context's HIP stream.  This is the
 *      batched, tile-component-granular path the encoder/decoder drivers use
 *      (per-call cgo + PCIe cost forbids per-block calls).
 *
 * There is no CPU fallback: without a usable HIP device every compute call
 * returns J2K_ERR_NO_DEVICE.
 */
#ifndef J2KGFX_H
#define J2KGFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define J2K_OK 0
#define J2K_ERR_INVALID_ARG (-1)
#define J2K_ERR_NO_DEVICE (-2)
#define J2K_ERR_HIP (-3)
#define J2K_ERR_CAPACITY (-4)     /* caller's output buffer too small */
#define J2K_ERR_GO_PANIC (-5)     /* input on which the reference itself panics / never returns */
#define J2K_ERR_UNSUPPORTED (-6)

/* entropy.BandLL..BandHH (internal/entropy/t1.go:125-130) */
#define J2K_BAND_LL 0
#define J2K_BAND_HL 1
#define J2K_BAND_LH 2
#define J2K_BAND_HH 3

#define J2K_CODER_MQ 0            /* entropy.T1 (EncodeFast5 / Decode)          */
#define J2K_CODER_HT 1            /* entropy.HTEncoder / HTDecoder (bug for bug) */

typedef struct j2k_ctx j2k_ctx;   /* one HIP device + one stream + workspaces; single-threaded */
typedef struct j2k_plan j2k_plan; /* frame geometry compiled to device job tables */

/* ---- context --------------------------------------------------------------- */
int j2k_ctx_create(int device, j2k_ctx **out);
void j2k_ctx_destroy(j2k_ctx *ctx);
int j2k_ctx_sync(j2k_ctx *ctx);                 /* hipStreamSynchronize on the ctx stream */
void *j2k_ctx_stream(j2k_ctx *ctx);             /* the hipStream_t, for event timing / interop */
const char *j2k_ctx_last_error(j2k_ctx *ctx);   /* text of the last non-OK status */
const char *j2k_status_string(int status);
const char *j2k_version(void);
/* Tuning options: which of its measured kernel forms a context takes (the defaults are what the benchmarks measured best; results
 * are identical under every setting -- tests/test_gpu_knobs.py).  Set BEFORE plans are created on the context.  Names (csrc/j2k_ctx.cpp,
 * ctx_options): "t1_dec_split" (MQ decode plane by plane from this many blocks on; -1 automatic), "t1_lanes", "t1_dec_lanes",
 * "t1_sym_mb" (cap of the MQ encoder's symbol workspace, MiB), "pix_fuse", "l0_wg", "l0_wg_inv", "l0_fuse", "plane_wg", "deep", "mega",
 * ...  An unknown name or a value out of range is J2K_ERR_INVALID_ARG.  The ENVIRONMENT sets none of this unless J2K_TUNING=1 is in
 * it (then J2K_<NAME> is read for every option at j2k_ctx_create: the tests' and tools' A/B switch); J2K_RCCL_LIB, the path of the
 * RCCL library, is the one variable that is always read. */
int j2k_ctx_set_option(j2k_ctx *ctx, const char *name, long value);
/* Kernel timing for bench.py's roofline lines: while enabled, the dispatches of the 5-3 transform stamp a HIP event pair
 * with the kernel's own begin and end (hipExtLaunchKernelGGL start/stop events on the ctx stream).  on = 1: the level-0
 * dispatch of every j2k_plan_forward* only (tag 0; cheapest, used inside bench.py's timed region); on = 2: every
 * dispatch of j2k_plan_forward* and j2k_plan_inverse*, tagged 0 = forward level 0, 1 = forward deeper levels, 2 = inverse
 * level 0, 3 = inverse deeper levels.  j2k_ctx_profile_read_tag synchronises and returns the number of recorded dispatches
 * with that tag and their summed duration in ms; j2k_ctx_profile_read does the same for tag 0 and resets the counters. */
int j2k_ctx_profile_enable(j2k_ctx *ctx, int on);
int j2k_ctx_profile_read(j2k_ctx *ctx, int64_t *launches, double *total_ms);
int j2k_ctx_profile_read_tag(j2k_ctx *ctx, int tag, int64_t *launches, double *total_ms);

/* HIP graphs (no counterpart in the reference: a launch-overhead facility of this boundary).  Between
 * capture_begin and capture_end the asynchronous device-pointer calls of this context (section 2: j2k_plan_forward*,
 * j2k_plan_encode_stream, j2k_plan_decode_blocks, j2k_plan_inverse*, j2k_plan_assemble_tiles_device ...) are recorded
 * instead of run; j2k_graph_launch replays them on the context's stream with the same device pointers (new contents in
 * the same buffers).  The same calls must have run once before the capture (workspaces sized): a call that would have to
 * allocate fails with J2K_ERR_INVALID_ARG and the capture must be ended and discarded.  Synchronous calls (anything in
 * section 1, j2k_ctx_sync, j2k_ctx_profile_read) are not allowed while capturing. */
typedef struct j2k_graph j2k_graph;
int j2k_ctx_capture_begin(j2k_ctx *ctx);
int j2k_ctx_capture_end(j2k_ctx *ctx, j2k_graph **out);
int j2k_graph_launch(j2k_graph *g);
void j2k_graph_destroy(j2k_graph *g);

/* ==== 1. host calls: one per reference function ============================= */

/* mct.DCLevelShiftForward / Inverse   (internal/mct/mct.go:96-101, 113-118) */
int j2k_dc_level_shift_forward(j2k_ctx *ctx, int32_t *data, size_t n, int precision);
int j2k_dc_level_shift_inverse(j2k_ctx *ctx, int32_t *data, size_t n, int precision);
/* mct.ForwardRCT / InverseRCT         (mct.go:28-38, 56-66) */
int j2k_forward_rct(j2k_ctx *ctx, int32_t *r, int32_t *g, int32_t *b, size_t n);
int j2k_inverse_rct(j2k_ctx *ctx, int32_t *y, int32_t *u, int32_t *v, size_t n);
/* mct.ForwardICT / InverseICT         (mct.go:14-24, 43-53) */
int j2k_forward_ict(j2k_ctx *ctx, double *r, double *g, double *b, size_t n);
int j2k_inverse_ict(j2k_ctx *ctx, double *y, double *cb, double *cr, size_t n);

/* dwt.Forward53 / Inverse53 (1-D)     (internal/dwt/dwt.go:73-118, 122-147) */
int j2k_forward53(j2k_ctx *ctx, int32_t *data, int length);
int j2k_inverse53(j2k_ctx *ctx, int32_t *data, int length);
/* dwt.Forward97 / Inverse97 (1-D)     (dwt.go:161-210, 213-262) */
int j2k_forward97(j2k_ctx *ctx, double *data, int length);
int j2k_inverse97(j2k_ctx *ctx, double *data, int length);
/* dwt.Forward2D53 / Inverse2D53 / Forward2D97 / Inverse2D97 (dwt.go:356-473) */
int j2k_forward2d53(j2k_ctx *ctx, int32_t *data, int width, int height);
int j2k_inverse2d53(j2k_ctx *ctx, int32_t *data, int width, int height);
int j2k_forward2d97(j2k_ctx *ctx, double *data, int width, int height);
int j2k_inverse2d97(j2k_ctx *ctx, double *data, int width, int height);
/* dwt.DecomposeMultiLevel53/97, ReconstructMultiLevel53/97 (dwt.go:524-573);
 * the reference's contiguous-prefix level layout is reproduced exactly. */
int j2k_decompose_multilevel53(j2k_ctx *ctx, int32_t *data, int width, int height, int levels);
int j2k_reconstruct_multilevel53(j2k_ctx *ctx, int32_t *data, int width, int height, int levels);
int j2k_decompose_multilevel97(j2k_ctx *ctx, double *data, int width, int height, int levels);
int j2k_reconstruct_multilevel97(j2k_ctx *ctx, double *data, int width, int height, int levels);

/* tcd.TileEncoder.ApplyForwardDWT / TileDecoder.ApplyInverseDWT on one
 * tile-component (internal/tcd/tcd.go:508-534, 416-437): reversible -> 5-3 int32;
 * else int32->f64, 9-7, int32(v +- 0.5) forward / int32(v + 0.5) inverse. */
int j2k_tcd_apply_forward_dwt(j2k_ctx *ctx, int32_t *data, int width, int height, int levels, int reversible);
int j2k_tcd_apply_inverse_dwt(j2k_ctx *ctx, int32_t *data, int width, int height, int levels, int reversible);

/* One code-block job of a batch.  For encode: the window (x0,y0,w,h) of plane
 * `plane` (stride = that plane's width) is the block, exactly what
 * encoder.extractCodeBlockData hands to T1.SetData (encoder.go:763-795).
 * For decode: bytes [in_off, in_off+in_len) of the input stream, numbps, band. */
typedef struct {
    int32_t plane;      /* index into the planes[] array of the call */
    int32_t band;       /* J2K_BAND_* */
    int32_t x0, y0;     /* window origin in the plane */
    int32_t w, h;       /* block size (actualWidth, actualHeight) */
} j2k_block;

/* entropy.(*T1).SetData + Encode(band) (t1.go:292-304, t1_fast5.go:10-899) or
 * entropy.(*HTEncoder).SetData + Encode(band) (ht.go:935-1045) for n blocks.
 * planes[i] is a HOST int32 plane of plane_w[i] x plane_h[i].  Output: block j's
 * bytes at out[offs[j] .. offs[j]+lens[j]) in job order (lens[j]==0 <=> Go nil),
 * numbps[j] = bit length of max |x| (the value T1.Decode must be told). */
int j2k_encode_blocks(j2k_ctx *ctx, int coder, const int32_t *const *planes, const int32_t *plane_w,
                      const int32_t *plane_h, int nplanes, const j2k_block *blocks, size_t nblocks,
                      uint8_t *out, size_t cap, uint64_t *offs, uint32_t *lens, uint8_t *numbps,
                      size_t *total);
/* entropy.NewT1(w,h).Decode(bytes,numBPS,band) (t1.go:1261-1292) or a fresh
 * entropy.NewHTDecoder(w,h).Decode(bytes,numBitplanes,band) (ht.go:93-150) for n
 * blocks.  Block j's coefficients (w*h int32, dense) land at coeffs + coeff_offs[j]. */
int j2k_decode_blocks(j2k_ctx *ctx, int coder, const uint8_t *bytes, const uint64_t *offs,
                      const uint32_t *lens, const uint8_t *numbps, const j2k_block *blocks,
                      size_t nblocks, int32_t *coeffs, const uint64_t *coeff_offs);
/* worst-case bytes one w x h block can produce (sizing of `out`) */
size_t j2k_block_bound(int coder, int w, int h);

/* ==== 2. plan calls: batched frame pipeline on device buffers ================ */

typedef struct {
    int32_t width, height;      /* frame size in samples                                     */
    int32_t ncomp;              /* components; planar int32 [C][H][W] (encoder.componentData) */
    int32_t precision;          /* bits per sample -> DC shift 1<<(p-1)  (encoder.go:218-220) */
    int32_t is_signed;          /* decode: skip DC shift for signed comps (decoder.go:344-348) */
    int32_t lossless;           /* 1: RCT + 5-3 ; 0: ICT + 9-7 + quantise (encoder.go:223-277) */
    int32_t quality;            /* Options.Quality, <=0 -> 100 (encoder.go:265-269)           */
    int32_t num_resolutions;    /* Options.NumResolutions; levels = n-1, <=0 -> 5; jobs: <=0 -> 6 */
    int32_t cb_w, cb_h;         /* REAL code-block size, 1<<(CodeBlockSize+2) (encoder.go:606-613) */
    int32_t tile_w, tile_h;     /* 0 = one tile (the reference); else tile t == the reference
                                   pipeline run on the cropped sub-image          */
    int32_t coder;              /* J2K_CODER_MQ | J2K_CODER_HT                                */
    int32_t tile_first, tile_count; /* shard: tiles [first, first+count); count<=0 -> all     */
    int32_t frame_rows;         /* > 0: a BATCH -- height / frame_rows frames of `width` x frame_rows stacked
                                   vertically (planes [C][height][W], pixels height rows): the tile grid starts
                                   again at every frame, so every frame is coded exactly as it would be alone,
                                   and one call carries all of them in its launches (tiles are numbered frame
                                   after frame).  0: one frame of `height` rows.  height % frame_rows must be 0 */
    int32_t closed_loop;        /* 0 (default): the reference's behaviour, byte for byte -- code-block windows cut from the
                                   top left of the plane for every band (encoder.go:763-795: they overlap, most of the
                                   plane is never coded) and Tier-2 packets that the reference's own decoder cannot read.
                                   1: THIS LIBRARY'S closed-loop mode, outside reference parity (SURVEY 8f rank 3 asks for
                                   "proper sub-band addressing and a real decode body"): band b of resolution r is the
                                   Mallat rectangle of decomposition level l = numRes-1-r of the tile-component
                                   (w_0 = w, w_{l+1} = ceil(w_l/2); LL = [0,w_L) x [0,h_L), HL = [w_{l+1},w_l) x [0,h_{l+1}),
                                   LH = [0,w_{l+1}) x [h_{l+1},h_l), HH the rest), cut into cb_w x cb_h blocks from the
                                   band's origin: the windows PARTITION the plane, job order stays comp -> res -> band ->
                                   row -> column, and j2k_plan_encode_tile_parts / j2k_plan_decode_tile_parts write and
                                   read packets with the J2K_T2_* flags below, so that pixels -> tile-parts -> pixels is
                                   a bit-exact round trip with the MQ coder.  Transform, block coders and every byte of
                                   a code-block are the reference's in both modes */
} j2k_params;

typedef struct {
    int64_t tiles;              /* tiles in this shard                                        */
    int64_t planes;             /* tile-components                                            */
    int64_t blocks;             /* code-block jobs                                            */
    int64_t coeff_elems;        /* int32 elements of the coefficient buffer                   */
    int64_t bytes_cap;          /* bytes of the worst-case slotted block output               */
    int64_t dwt_bytes;          /* algorithmic bytes of the multi-level DWT       */
    int64_t dwt_level0_bytes;   /* algorithmic bytes of the level-0 kernel launch             */
    int64_t block_samples;      /* samples the block coder reads                              */
    int64_t decoded_elems;      /* int32 elements of the dense decoded-block buffer           */
} j2k_plan_info;

int j2k_plan_create(j2k_ctx *ctx, const j2k_params *params, j2k_plan **out);
void j2k_plan_destroy(j2k_plan *plan);
int j2k_plan_get_info(const j2k_plan *plan, j2k_plan_info *info);
/* copy out the job list (reference order: tile -> comp -> res -> band -> cby -> cbx) */
int j2k_plan_get_blocks(const j2k_plan *plan, j2k_block *blocks, size_t cap);
/* per-plane geometry: tile index, component, x0, y0, w, h, coefficient offset (7 x int64 each) */
int j2k_plan_get_planes(const j2k_plan *plan, int64_t *desc7, size_t cap_planes);

/* encoder.preprocess (encoder.go:216-281) for every tile-component of the shard:
 * d_frame = device int32 [C][H][W] pixels (read only), d_coeff = device coefficient
 * buffer (coeff_elems int32; each tile-component dense w_t x h_t at its offset). */
int j2k_plan_forward(j2k_plan *plan, const int32_t *d_frame, int32_t *d_coeff);
/* inverse path: ApplyInverseDWT per tile-component + inverse MCT + DC shift
 * (tcd.go:416-437, decoder.go:321-348): d_coeff -> d_frame (pixels). */
int j2k_plan_inverse(j2k_plan *plan, const int32_t *d_coeff, int32_t *d_frame);
/* encoder.encodeTile job loop (encoder.go:616-688), sequential semantics: every job's
 * bytes into worst-case slots of d_slots (bytes_cap), lengths into d_lens (u32 per job),
 * bit-plane counts into d_numbps (u8 per job). */
int j2k_plan_encode_blocks(j2k_plan *plan, const int32_t *d_coeff, uint8_t *d_slots,
                           uint32_t *d_lens, uint8_t *d_numbps);
/* exclusive scan of d_lens -> d_offs (u64 per job, +1 total at the end) and gather of the
 * slots into the dense stream d_stream (concatenation in job order, encoder.go:684). */
int j2k_plan_compact(j2k_plan *plan, const uint8_t *d_slots, const uint32_t *d_lens,
                     uint64_t *d_offs, uint8_t *d_stream);
/* j2k_plan_encode_blocks + j2k_plan_compact in one call: the dense stream (job order), d_offs (u64
 * per job + total), d_lens, d_numbps; the slot buffer in between is owned by the plan.  (With
 * J2K_FUSE_COMPACT=1, the HT coder and blocks up to 64x64 it is one kernel with a decoupled look-back
 * over the block lengths -- correct, but measured slower than the three kernels, so off by default.) */
int j2k_plan_encode_stream(j2k_plan *plan, const int32_t *d_coeff, uint8_t *d_stream, uint64_t *d_offs,
                           uint32_t *d_lens, uint8_t *d_numbps);
/* Transport form of a stream for the multi-GPU gather (the compressed packets go to rank 0 for codestream
 * assembly, encoder.go:568-579, 746-760).  The reference's HT block carries max(64, 2wh)/4 zero bytes of MEL segment
 * (ht.go:978, 1019) -- two thirds of a 64x64 block -- and every peer has ONE xGMI link to the root, so a peer sends
 *   header | lens u32[n] | maglens u32[n] | numbps u8[n] | offs u64[n+1] | toffs u64[n+1] | blocks without the MEL runs
 * (j2k_plan_pack_bound bytes at most; the first u64 of the pack is its length) and the root rebuilds the dense stream,
 * d_offs (n + 1), d_lens and d_numbps byte for byte with j2k_plan_unpack_stream on a plan of the same geometry (any
 * other pack is reported as J2K_ERR_INVALID_ARG at the next sync).  j2k_plan_pack_stream packs the stream made by the
 * LAST j2k_plan_encode_stream call on this plan (the plan remembers where each block's MagSgn bytes end) and returns
 * J2K_ERR_INVALID_ARG when d_stream / d_lens are not that call's outputs; MQ streams have no zero runs and are packed as
 * they are. */
size_t j2k_plan_pack_bound(const j2k_plan *plan);
int j2k_plan_pack_stream(j2k_plan *plan, const uint8_t *d_stream, const uint64_t *d_offs, const uint32_t *d_lens,
                         const uint8_t *d_numbps, uint8_t *d_pack);
/* pack_bytes = the bytes the caller really holds at d_pack (what it received): a pack is foreign input, and nothing
 * outside [d_pack, d_pack + pack_bytes) is read -- a pack shorter than its header is J2K_ERR_INVALID_ARG at once, one whose
 * fields point outside it or outside the stream is J2K_ERR_INVALID_ARG at the next sync and copies nothing for the
 * offending block (every bound is tested without forming a sum of untrusted 64-bit fields). */
int j2k_plan_unpack_stream(j2k_plan *plan, const uint8_t *d_pack, size_t pack_bytes, uint8_t *d_stream, uint64_t *d_offs,
                           uint32_t *d_lens, uint8_t *d_numbps);
/* the same for `count` packs of this geometry in one launch (host arrays of device pointers / of byte counts): the root
 * of an N-GPU gather rebuilds the N-1 peers' streams of a frame slot at once */
int j2k_plan_unpack_streams(j2k_plan *plan, int count, const uint8_t *const *d_packs, const size_t *pack_bytes,
                            uint8_t *const *d_streams, uint64_t *const *d_offs, uint32_t *const *d_lens, uint8_t *const *d_numbps);
/* tcd.TileDecoder.DecodeCodeBlock (tcd.go:393-413) for every job: dense stream + offsets
 * + lens + numbps -> d_decoded (decoded_elems int32, block j dense at its job offset). */
int j2k_plan_decode_blocks(j2k_plan *plan, const uint8_t *d_stream, const uint64_t *d_offs,
                           const uint32_t *d_lens, const uint8_t *d_numbps, int32_t *d_decoded);
/* HT coder only; default off.  The reference's HT decoder writes ONE row in four of a block (ht.go:589-711: only row y of each
 * 4-row stripe is coded) and returns its internal slice: a fresh decoder's slice is zero elsewhere -- what
 * j2k_plan_decode_blocks reproduces by default, writing all w x h samples -- while a pooled decoder (GetHTDecoder /
 * PutHTDecoder, ht.go:1393-1429) is not cleared on that path and keeps whatever those rows held.  With `on` the call
 * behaves like the pooled decoder: rows y % 4 != 0 of every block in d_decoded are left untouched (the caller zeroed the
 * buffer once, or does not read them), the coded rows are written in full as before.  Three quarters of the decoder's
 * store stream (77 of 103 MB per 4K frame) disappear. */
int j2k_plan_set_decode_coded_rows_only(j2k_plan *plan, int on);
/* job j's offset (in int32 elements) into d_decoded */
int j2k_plan_get_decoded_offsets(const j2k_plan *plan, uint64_t *offs, size_t cap);

/* ---- the multi-GPU exchange -------------------------------------------------------------------------------
 * One process per GPU, each with its own j2k_ctx; a frame's tiles are sharded by j2k_params.tile_first / tile_count and coded
 * independently (no halo).  The ONE exchange step is the gather of the compressed streams to rank 0 for codestream assembly
 * -- it replaces the in-process collection of encoder.encodeTile's job results (encoder.go:690-742) when the jobs ran on other
 * GPUs: ncclAllGather of the byte counts, then inside one ncclGroupStart / ncclGroupEnd every rank != 0 ncclSend()s its
 * streams to rank 0, which posts the matching ncclRecv()s -- direct peer -> root over each peer's own xGMI link, no ring.
 * RCCL is bound at run time (dlopen), so a single-GPU host does not need it: the calls return J2K_ERR_UNSUPPORTED without it.
 *
 *   j2k_comm_get_unique_id   rank 0 makes the 128-byte RCCL id; the HOST carries it to the other ranks (the Go side: any
 *                            channel it has -- a pipe to its worker processes, a file, MPI; the Python tests: torch.distributed)
 *   j2k_comm_create          collective (every rank): ncclCommInitRank on ctx's device; the communicator owns a HIP stream
 *                            of its own for the transfers
 *   j2k_gather_streams       collective.  Rank r passes `count` device buffers d_send[f] of send_bytes[f] bytes (normally the
 *                            packs of j2k_plan_pack_stream, one per frame slot).  all_bytes: the byte counts of every rank,
 *                            rank-major (world x count), when the host already knows them (it usually does: a 16-byte
 *                            message per rank on whatever channel carried the id) -- else NULL and the call gathers them
 *                            with ncclAllGather, which costs a stream synchronisation.  producers: the contexts on whose
 *                            streams the send buffers are being produced; the transfers are ordered behind their work so
 *                            far.  On rank 0, stream f of rank r lands at d_recv + recv_offs[r * count + f] (16-byte
 *                            aligned; recv_offs has world * count + 1 entries, host memory, filled on every rank), its own
 *                            streams by a device copy.  Returns once the transfers are QUEUED on the communicator's
 *                            stream.  FAILURE IS COLLECTIVE where the call can make it so: J2K_ERR_CAPACITY (recv_cap of
 *                            rank 0 too small) is returned by EVERY rank when the counts are gathered by the call (rank 0's
 *                            recv_cap travels with them); with host-provided counts a peer passes rank 0's capacity as its
 *                            own recv_cap (0 = not known: that peer returns J2K_OK).  Either way nobody is left waiting:
 *                            the peers always post their sends, and a rank 0 that cannot take the bytes (too small, NULL
 *                            or misaligned d_recv) receives them into a scratch allocation, drains its stream and then
 *                            returns the status; the communicator stays usable.  Argument errors that only ONE rank can
 *                            see (J2K_ERR_INVALID_ARG: a NULL send pointer, all_bytes that disagrees with send_bytes)
 *                            return before anything is posted on that rank -- the host must agree on them before the call.
 *   j2k_comm_wait            consumer != NULL: that context's stream waits (on the device) for the last gather -- follow with
 *                            j2k_plan_unpack_streams on it; NULL: the host waits.
 * flags: J2K_GATHER_SELF_LOOP (world == 1 only): the lone rank sends to itself through RCCL -- the transfer calls on one GPU. */
typedef struct j2k_comm j2k_comm;
#define J2K_COMM_ID_BYTES 128
#define J2K_GATHER_SELF_LOOP 1
const char *j2k_comm_load_error(void);   /* why RCCL could not be bound (J2K_ERR_UNSUPPORTED); "" if it was.  J2K_RCCL_LIB names the library */
int j2k_comm_get_unique_id(uint8_t *id128);
int j2k_comm_create(j2k_ctx *ctx, const uint8_t *id128, int rank, int world, j2k_comm **out);
void j2k_comm_destroy(j2k_comm *comm);
const char *j2k_comm_last_error(j2k_comm *comm);
void *j2k_comm_stream(j2k_comm *comm);
int j2k_gather_streams(j2k_comm *comm, int count, const uint8_t *const *d_send, const uint64_t *send_bytes,
                       const uint64_t *all_bytes, j2k_ctx *const *producers, int nproducers,
                       uint8_t *d_recv, size_t recv_cap, uint64_t *recv_offs, int flags);
int j2k_comm_wait(j2k_comm *comm, j2k_ctx *consumer);

/* ---- stand-alone arithmetic / bypass coders (internal/entropy/mqc.go), host buffers ------------
 * MQEncoder: NewMQEncoder, n x Encode(ctxs[i], decisions[i]), Flush (mqc.go:169-349); *out_len = 0
 * is Flush's nil.  A context >= 19 is the Go index panic (J2K_ERR_GO_PANIC). */
int j2k_mq_encode(j2k_ctx *ctx, const uint8_t *ctxs, const uint8_t *decisions, size_t n,
                  uint8_t *out, size_t cap, size_t *out_len);
/* MQDecoder: NewMQDecoder(data), n x Decode(ctxs[i]) -> decisions[i] (mqc.go:352-497) */
int j2k_mq_decode(j2k_ctx *ctx, const uint8_t *data, size_t len, const uint8_t *ctxs, size_t n,
                  uint8_t *decisions);
/* RawEncoder: n x EncodeBit(bits[i]), Flush (mqc.go:560-600); RawDecoder: n x DecodeBit (mqc.go:516-557) */
int j2k_raw_encode(j2k_ctx *ctx, const uint8_t *bits, size_t n, uint8_t *out, size_t cap, size_t *out_len);
int j2k_raw_decode(j2k_ctx *ctx, const uint8_t *data, size_t len, size_t n, uint8_t *bits);

/* ---- pixels at native width ------------------------------------------------
 * encoder.extractImageData (encoder.go:79-213) and decoder.createImage (decoder.go:417-588): the
 * host loops on either side of the tile-component path.  Pixel buffers are Go image.* Pix layouts
 * (row-major, `stride` bytes per row, 16-bit samples big-endian). */
enum {
    J2K_PIX_GRAY8 = 0,     /* *image.Gray     1 component,  8 bit   (encoder.go:83-93)   */
    J2K_PIX_GRAY16 = 1,    /* *image.Gray16   1 component, 16 bit   (encoder.go:95-105)  */
    J2K_PIX_RGBA8 = 2,     /* *image.RGBA     3 components (alpha ignored), 8 bit (encoder.go:107-123) */
    J2K_PIX_RGBA64 = 3,    /* *image.RGBA64   3 components, 16 bit  (encoder.go:125-141) */
    J2K_PIX_NRGBA8 = 4,    /* *image.NRGBA    4 components, 8 bit   (encoder.go:143-160) */
    J2K_PIX_NRGBA64 = 5    /* *image.NRGBA64  4 components, 16 bit  (encoder.go:162-179) */
};
/* components / source precision of a pixel format (0 for an unknown format) */
int j2k_pixels_components(int format);
int j2k_pixels_precision(int format);
/* extractImageData on HOST buffers: pix -> planes[c] (ncomp planes of w*h int32, caller-allocated).
 * target_precision 1..16 applies the Options.Precision rescale (encoder.go:196-210); 0 keeps the
 * source precision.  The pixels cross PCIe at native width. */
int j2k_extract_image_data(j2k_ctx *ctx, int format, const void *pix, size_t stride, int w, int h,
                           int target_precision, int32_t *const *planes);
/* createImage on HOST buffers: planes[c] -> pix.  ncomp 1 -> Gray (precision <= 8) / Gray16;
 * 3 or 4 -> RGBA (precision <= 8) / RGBA64; other ncomp -> J2K_ERR_UNSUPPORTED (decoder.go:583-585). */
int j2k_create_image(j2k_ctx *ctx, const int32_t *const *planes, int ncomp, int precision, int w, int h,
                     void *pix, size_t stride);
/* the same on DEVICE buffers (d_planes = ncomp planes of w*h int32, contiguous) */
int j2k_unpack_pixels(j2k_ctx *ctx, int format, const void *d_pix, size_t stride, int w, int h,
                      int target_precision, int32_t *d_planes);
int j2k_pack_pixels(j2k_ctx *ctx, const int32_t *d_planes, int ncomp, int precision, int w, int h,
                    void *d_pix, size_t stride);
/* j2k_plan_forward / j2k_plan_inverse with the frame as packed 8-bit RGBA on the device
 * (extractImageData + preprocess, and the inverse path + createImage, fused): for a 3-component
 * 8-bit plan.  The 5-3 + RCT level-0 kernels read / write the pixels directly when the geometry
 * allows 16-byte accesses (otherwise the pixels pass through an int32 staging frame). */
int j2k_plan_forward_rgba8(j2k_plan *plan, const void *d_pix, size_t stride, int32_t *d_coeff);
int j2k_plan_inverse_rgba8(j2k_plan *plan, const int32_t *d_coeff, void *d_pix, size_t stride);
/* Any pixel format: extractImageData (+ the rescale to the plan's precision) then j2k_plan_forward;
 * j2k_plan_inverse then createImage for the plan's component count and precision.  The plan's
 * component count must equal j2k_pixels_components(format) (forward) / be 1, 3 or 4 (inverse).
 * When the plan's precision is the format's own (8 / 16 bit, unsigned, 5-3) and the geometry allows
 * 16-byte accesses, the level-0 kernels read / write the pixels themselves -- Gray, Gray16, RGBA,
 * RGBA64, NRGBA, NRGBA64 alike (a fourth component is its own plane: encoder.go:152-179), and so do the
 * 9-7 level-0 kernels for image.RGBA and image.Gray on a lossy 8-bit plan (the reference's DefaultOptions); otherwise
 * the pixels pass through an int32 staging frame.  Same results either way. */
int j2k_plan_forward_pixels(j2k_plan *plan, int format, const void *d_pix, size_t stride, int32_t *d_coeff);
int j2k_plan_inverse_pixels(j2k_plan *plan, const int32_t *d_coeff, void *d_pix, size_t stride);
/* 1 when that call (inverse != 0: j2k_plan_inverse_pixels, format ignored) would take the fused
 * kernels for these pixels, 0 when it would stage; < 0 on bad arguments.  Launches nothing. */
int j2k_plan_pixels_fused(const j2k_plan *plan, int format, const void *d_pix, size_t stride, int inverse);

/* ---- decode-side colour conversions to sRGB ---------------------------------
 * getColorConversion(cs)(componentData, precision) (colorspace.go:54-480); the values are the
 * reference's ColorSpace constants (jpeg2000.go:124-197).  Spaces without a conversion (sRGB, gray,
 * bilevel, unknown, unspecified) and component counts below what a conversion needs (3; 4 for
 * CMYK / YCCK) leave the data untouched, as the reference does. */
enum {
    J2K_CS_UNKNOWN = -1, J2K_CS_UNSPECIFIED = 0, J2K_CS_SRGB = 1, J2K_CS_GRAY = 2, J2K_CS_SYCC = 3, J2K_CS_EYCC = 4,
    J2K_CS_CMYK = 5, J2K_CS_BILEVEL = 6, J2K_CS_YCBCR2 = 7, J2K_CS_YCBCR3 = 8, J2K_CS_PHOTOYCC = 9, J2K_CS_CMY = 10,
    J2K_CS_YCCK = 11, J2K_CS_CIELAB = 12, J2K_CS_CIEJAB = 13, J2K_CS_ESRGB = 14, J2K_CS_ROMMRGB = 15,
    J2K_CS_YPBPR60 = 16, J2K_CS_YPBPR50 = 17
};
/* host planes (ncomp pointers to n int32 each), converted in place */
int j2k_convert_colorspace(j2k_ctx *ctx, int colorspace, int32_t *const *planes, int ncomp, size_t n, int precision);
/* device planes, contiguous (ncomp x n int32), converted in place on the ctx stream */
int j2k_convert_colorspace_device(j2k_ctx *ctx, int colorspace, int32_t *d_planes, int ncomp, size_t n, int precision);

/* Whole shard from HOST planes, mirroring encoder.preprocess + encodeTile
 * (encoder.go:216-281, 597-688): planes[c] = host int32 W*H, mutated in place to
 * the coefficients like e.componentData when the frame is a single tile (with tiles
 * the coefficients are returned tile-plane by tile-plane in `coeff` if non-NULL).
 * out receives the concatenated block bytes of all tiles in job order;
 * tile_offs[t] (tiles+1 entries) delimits each tile's data for SOT assembly. */
int j2k_encode_frame(j2k_plan *plan, int32_t *const *planes, int32_t *coeff, uint8_t *out, size_t cap,
                     size_t *out_len, uint64_t *tile_offs, uint32_t *lens, uint8_t *numbps);

/* Pixels at native width from / to HOST memory, one synchronous call each: what a cgo jpeg2000.Encode / Decode binds when the image
 * lives in Go memory (encoder.go:79-213 + 216-281 + 597-688 + 746-760; decoder.go:363-588).  j2k_encode_pixels_host: pix (a Go Pix
 * layout, `stride` bytes per row) crosses PCIe at native width, out receives every tile of the shard as a tile-part -- reference-mode
 * plan: SOT | SOD | the tile's concatenated block bytes (createTileHeader of encodeTile's output); closed-loop plan: SOT | SOD | packets
 * (sop / eph used there only) -- *out_len the bytes (J2K_ERR_CAPACITY with *out_len set when cap is smaller), tile_offs[t] (tiles + 1, may
 * be NULL) where tile-part t starts, lens / numbps per code-block (may be NULL).  j2k_decode_pixels_host (closed-loop plans): the
 * tile-parts back to pixels of the plan's format (1 / 3 / 4 components, precision <= 8: Gray / RGBA, else Gray16 / RGBA64). */
int j2k_encode_pixels_host(j2k_plan *plan, int format, const void *pix, size_t stride, int sop, int eph, uint8_t *out, size_t cap,
                           size_t *out_len, uint64_t *tile_offs, uint32_t *lens, uint8_t *numbps);
int j2k_decode_pixels_host(j2k_plan *plan, const uint8_t *cs, size_t len, int sop, int eph, void *pix, size_t stride);

/* ---- multi-tile codestream assembly: host calls, no device needed ------------------------------
 * j2k_create_tile_header = encoder.createTileHeader(tileIdx, tileData) (encoder.go:746-760): SOT (0xFF90) Lsot = 10,
 * Isot = uint16(tileIdx), Psot = uint32(14 + len), TPsot = 0, TNsot = 1, SOD (0xFF93), then the tile data -- 14 + len bytes.
 * j2k_assemble_tiles applies it to every tile of a (gathered) stream, tile t = stream[tile_offs[t], tile_offs[t+1]) with
 * index tile_first + t, tile-parts laid end to end: what rank 0 hands to the Go side between the main header and EOC
 * (the reference's generateTiles, encoder.go:568-579, emits tile 0 only). */
size_t j2k_tile_part_bound(const uint64_t *tile_offs, int ntiles);
int j2k_create_tile_header(int tile_idx, const uint8_t *tile_data, size_t len, uint8_t *out, size_t cap, size_t *out_len);
int j2k_assemble_tiles(const uint8_t *stream, const uint64_t *tile_offs, int tile_first, int ntiles, uint8_t *out, size_t cap,
                       size_t *out_len);
/* The same on device buffers, for a plan's shard: every tile of d_stream / d_offs (as j2k_plan_encode_stream or
 * j2k_plan_unpack_stream made them) becomes a tile-part with index tile_first + t, laid end to end in d_out
 * (j2k_plan_tile_parts_bound bytes at most); *d_out_len (device) = the bytes written.  Asynchronous on the context's
 * stream: one D2H copy then carries finished tile-parts instead of a stream the host still has to cut up. */
size_t j2k_plan_tile_parts_bound(const j2k_plan *plan);
int j2k_plan_assemble_tiles_device(j2k_plan *plan, const uint8_t *d_stream, const uint64_t *d_offs, uint8_t *d_out, uint64_t *d_out_len);
/* codestream.Parser.ReadTilePartHeader (internal/codestream/parser.go:894-983): the SOT fields of the tile-part whose SOT
 * marker is at cs[pos], its in-header marker segments skipped by length (parser.go:180-190; their contents stay with the
 * Go-side parser: header_off / header_markers say where they are), and where its data lies (Psot = 0: to the end).
 * Errors as the parser's: Lsot != 10, a segment length < 2, truncated input -> J2K_ERR_INVALID_ARG. */
typedef struct j2k_tile_part {
    uint16_t tile_index;          /* Isot  */
    uint8_t tile_part_index;      /* TPsot */
    uint8_t num_tile_parts;       /* TNsot */
    uint32_t tile_part_length;    /* Psot  */
    uint32_t header_markers;      /* marker segments between the SOT segment and SOD */
    uint32_t pad_;
    uint64_t header_off;          /* offset of the first of them (== data_off - 2 when there are none) */
    uint64_t data_off, data_len;  /* the tile-part's data */
} j2k_tile_part;
int j2k_read_tile_part_header(const uint8_t *cs, size_t len, size_t pos, j2k_tile_part *tp);
/* every tile-part of a run of tile-parts (up to EOC or the end); *nparts = how many there are (J2K_ERR_CAPACITY if > cap) */
int j2k_parse_tile_parts(const uint8_t *cs, size_t len, j2k_tile_part *parts, size_t cap, size_t *nparts);

/* ---- Tier-2 packet coding and tile geometry: host calls, no device needed -------------------------
 * The reference's internal/tcd/t2.go and the geometry of tcd.go, reproduced as they are written -- the "tag tree" values
 * are unary (t2.go:368-377), the length-of-length field is 3 bits and wraps for blocks of 128 bytes or more
 * (t2.go:408-437), PCRL runs every component / resolution to the LARGEST precinct count (t2.go:86-114), the packet
 * decoder's Position() moves over markers and bodies only (t2.go:463-503) -- because parity with the reference is the
 * contract; none of this is a conformant Part-1 packet stream (SURVEY 8f calls a conformant mode "explicitly outside
 * reference parity").  In the reference only tests call t2.go; decoder.go:312,373 calls the geometry.
 *
 * j2k_t2_packet_sequence = NewPacketIterator + Next() until exhausted (t2.go:41-238), progression order
 * 0..4 = LRCP, RLCP, RPCL, PCRL, CPRL (codestream/markers.go:177-188; any other order yields no packet, t2.go:86-100).
 * precincts [][][]int is passed flat: prec_ncomp = len(precincts), prec_nres[c] = len(precincts[c]),
 * prec_counts = precincts[c][r][0] for c, r in order (a (c, r) the table does not cover counts as one precinct).
 * *count = packets in the sequence; J2K_ERR_CAPACITY (with *count set) when cap is smaller. */
typedef struct j2k_packet { int32_t layer, resolution, component, precinct; } j2k_packet;
int j2k_t2_packet_sequence(int ncomp, int nres, int nlayers, const int32_t *prec_counts, const int32_t *prec_nres, int prec_ncomp,
                           int order, j2k_packet *out, size_t cap, size_t *count);

/* A precinct as the packet coder sees it (tcd.go:86-128): nbands lists of code-blocks (band_ncb[b] each, flat in cbs) and
 * the widths of its two tag trees (only used as divisors: a width of 0 is the Go divide panic, J2K_ERR_GO_PANIC).
 * Code-block fields: IncludedInLayers, ZeroBitPlanes, len(Passes), Data (data_len bytes at data; data_cap = room the
 * decoder may fill). */
typedef struct j2k_t2_cb {
    int32_t included_in_layers, zero_bit_planes, num_passes;
    uint32_t data_len, data_cap, pad_;
    uint8_t *data;
} j2k_t2_cb;
typedef struct j2k_t2_precinct {
    int32_t nbands, incl_tree_w, imsb_tree_w, pad_;
    const int32_t *band_ncb;
    j2k_t2_cb *cbs;
} j2k_t2_precinct;
/* PacketEncoder.EncodePacket (t2.go:250-290): [SOP FF91 0004 uint16(layer)] header bits through the byte-stuffing
 * writer (bio.go:157-226: after a 0xFF byte the next byte holds 7 bits), flushed; [EPH FF92]; the bodies of the
 * code-blocks with IncludedInLayers <= layer and data.  *bio_delay carries the writer's "last byte was 0xFF" flag from
 * one packet of an encoder to the next (0 for a new encoder).  j2k_t2_packet_bound: an upper bound of *len. */
size_t j2k_t2_packet_bound(const j2k_t2_precinct *p);
int j2k_t2_encode_packet(const j2k_t2_precinct *p, int layer, int sop, int eph, uint8_t *bio_delay, uint8_t *out, size_t cap, size_t *len);
/* PacketDecoder.DecodePacket (t2.go:463-503) on data[0, len): the decoder object's state is the caller's -- pos
 * (Position(): markers and bodies), and the header bit reader's own position / byte / bit count / "saw 0xFF"
 * (a zeroed struct is NewPacketDecoder).  Fills the code-blocks' fields; a block's data is zeroed to its decoded length
 * (J2K_ERR_CAPACITY if data_cap is smaller) and then filled from the body.  Running out of header bits or body bytes is
 * the reference's error return: J2K_ERR_INVALID_ARG. */
typedef struct j2k_t2_dec_state { uint64_t pos, rpos; uint8_t buf, cnt, saw_ff, pad_[5]; } j2k_t2_dec_state;
int j2k_t2_decode_packet(const uint8_t *data, size_t len, j2k_t2_dec_state *st, j2k_t2_precinct *p, int layer, int sop, int eph);
/* PacketEncoder.EncodePacket for a RUN of packets on DEVICE buffers (csrc/t2dev.hip): packet i is coded by the same encoder
 * right after packet i - 1, so the writer's "last byte was 0xFF" flag runs through the run (*bio_delay in and out, as above).
 * A packet = its layer, the widths of its two tag trees (divisors only) and ncb code-blocks from d_cbs[cb0] (a table of ncbs) in coding order
 * (band after band: the band structure changes nothing in the bytes, t2.go:320-364); a code-block's bytes are data_len bytes
 * at d_data + data_off (e.g. the compacted stream of j2k_plan_encode_stream).  Packet i lands at d_out + d_offs[i]
 * (d_offs: npackets + 1 entries, the last = *total).  J2K_ERR_CAPACITY (with *total set, nothing written) when cap is
 * smaller; J2K_ERR_GO_PANIC for a tree width of 0 that the coder divides by.  Synchronises the context's stream. */
typedef struct j2k_t2_dev_cb { int32_t included_in_layers, zero_bit_planes, num_passes; uint32_t data_len; uint64_t data_off; } j2k_t2_dev_cb;
/* flags: 0 = the reference's coder.  The closed-loop mode (j2k_params.closed_loop; NOT the reference) sets
 *   J2K_T2_FRESH     this packet is the first of a new PacketEncoder / PacketDecoder object (one per tile): the byte-stuffing
 *                    writer's / reader's "last byte was 0xFF" flag starts clear
 *   J2K_T2_WIDE_LEN  the length-of-length field has 5 bits (the reference's 3 wrap for blocks of 128 bytes or more, t2.go:408-437)
 *   J2K_T2_SEATED    decoder only: the header is read from Position() and Position() moves past it (the reference's header
 *                    reader runs over the buffer on its own and Position() only moves over markers and bodies, t2.go:463-503) */
#define J2K_T2_FRESH 1
#define J2K_T2_WIDE_LEN 2
#define J2K_T2_SEATED 4
typedef struct j2k_t2_dev_packet { int32_t layer, incl_tree_w, imsb_tree_w, flags; int64_t cb0, ncb; } j2k_t2_dev_packet;
int j2k_t2_encode_packets_device(j2k_ctx *ctx, const j2k_t2_dev_packet *d_packets, size_t npackets, const j2k_t2_dev_cb *d_cbs, size_t ncbs,
                                 const uint8_t *d_data, int sop, int eph, uint8_t *bio_delay, uint8_t *d_out, size_t cap,
                                 uint64_t *d_offs, size_t *total);
/* PacketDecoder.DecodePacket (t2.go:463-652) for a RUN of packets by ONE decoder object on a DEVICE buffer d_data[0, len)
 * (csrc/t2dec.hip): packet i is decoded right after packet i - 1 with the object's state -- *st in and out, as
 * j2k_t2_decode_packet's (a zeroed struct is NewPacketDecoder).  d_cbs is read AND written (IncludedInLayers persists from
 * layer to layer): a block the packet includes gets IncludedInLayers, ZeroBitPlanes, len(Passes), len(Data) = data_len and
 * data_off = where its body lies in d_data (nothing is copied).  *packets_done = packets decoded before the first error;
 * the status is that packet's: J2K_ERR_INVALID_ARG for the reference's error returns (out of header bits, a body past the
 * end), J2K_ERR_GO_PANIC for a tree of width 0 it divides by.  Synchronises the context's stream. */
int j2k_t2_decode_packets_device(j2k_ctx *ctx, const j2k_t2_dev_packet *d_packets, size_t npackets, j2k_t2_dev_cb *d_cbs, size_t ncbs,
                                 const uint8_t *d_data, size_t len, int sop, int eph, j2k_t2_dec_state *st, size_t *packets_done);

/* ---- the closed-loop frame codec (j2k_params.closed_loop = 1): tile-parts of packets out, pixels back -------
 * The reference has the pieces -- PacketEncoder / PacketDecoder (t2.go), createTileHeader (encoder.go:746-760),
 * DecodeCodeBlock and ApplyInverseDWT (tcd.go:393-437) -- but no decode body (decoder.decodeTile is a placeholder,
 * decoder.go:375-411).  These calls are that body on device buffers, asynchronous on the context's stream:
 *   j2k_plan_encode_tile_parts  the block coder's outputs (j2k_plan_encode_stream) -> for every tile of the shard
 *                               SOT | SOD | one packet per (component, resolution) in job order, tile-parts end to end in
 *                               d_out (j2k_plan_frame_bound bytes at most); d_tile_offs[t] (tiles + 1, device) = where
 *                               tile-part t starts, the last entry the total.  Code-block fields as j2k_plan_t2_fill_cbs
 *                               with mb = 31, an empty block IncludedInLayers 1; a new PacketEncoder per tile.
 *                               cap < the bytes needed: nothing is written and d_tile_offs[tiles] holds the need
 *                               (J2K_ERR_CAPACITY from the next j2k_plan_frame_status).
 *   j2k_plan_decode_tile_parts  the inverse: d_cs[0, len) -> d_offs / d_lens / d_numbps as j2k_plan_decode_blocks takes
 *                               them (a block's bytes stay where they are in d_cs: d_offs points into it).  d_tile_offs
 *                               (device, tiles + 1) where the caller knows the tile-parts' positions (the Go-side parser
 *                               has read the SOT segments, parser.go:894-983; or the encode call's own table), else NULL:
 *                               the call walks the SOT segments itself (Psot), one after the other.
 *                               With sop != 0 and eph != 0 a tile's packets are parsed side by side: their starts are guessed from
 *                               the FF91 / FF92 markers, every packet is decoded on its own, and the result is kept only if each
 *                               packet ended in exactly the state the next one was started from -- otherwise (a marker pair inside
 *                               a body, a damaged header) that tile is decoded again one packet after the other.  The output is
 *                               the serial decoder's either way (j2k_ctx_set_option "t2_parallel" = 0: serial only).
 *   j2k_plan_frame_parallel_tiles  diagnostic: synchronises; *tiles = tiles whose packets were parsed side by side in the
 *                               decode calls since the last query
 *   j2k_plan_place_blocks       decoded blocks (j2k_plan_decode_blocks) -> each at its window of the coefficient planes
 *   j2k_plan_frame_status       synchronises and reports what the asynchronous calls above found since the last call:
 *                               J2K_OK, J2K_ERR_CAPACITY, or J2K_ERR_INVALID_ARG for a malformed tile-part / packet
 *   j2k_plan_encode_frame_pixels / j2k_plan_decode_frame_pixels   the whole chain with the plan's own workspaces:
 *                               pixels (any format, as j2k_plan_forward_pixels) -> tile-parts, and tile-parts -> pixels.  Same bytes
 *                               and pixels as the stage calls, with less copying: the encoder gathers every block once from its
 *                               coding slot into its packet in its tile-part (no dense stream in between); on an HT plan the
 *                               block decoder writes only the rows the reference's HT decoder writes (one in four), straight
 *                               into each block's window of coefficient planes the plan zeroed once.
 * All of them return J2K_ERR_UNSUPPORTED on a plan without closed_loop. */
size_t j2k_plan_frame_bound(const j2k_plan *plan);
int j2k_plan_encode_tile_parts(j2k_plan *plan, const uint8_t *d_stream, const uint64_t *d_offs, const uint32_t *d_lens, const uint8_t *d_numbps,
                               int sop, int eph, uint8_t *d_out, size_t cap, uint64_t *d_tile_offs);
int j2k_plan_decode_tile_parts(j2k_plan *plan, const uint8_t *d_cs, size_t len, const uint64_t *d_tile_offs, int sop, int eph,
                               uint64_t *d_offs, uint32_t *d_lens, uint8_t *d_numbps);
int j2k_plan_place_blocks(j2k_plan *plan, const int32_t *d_decoded, int32_t *d_coeff);
int j2k_plan_frame_status(j2k_plan *plan);
int j2k_plan_frame_parallel_tiles(j2k_plan *plan, long *tiles);
int j2k_plan_encode_frame_pixels(j2k_plan *plan, int format, const void *d_pix, size_t stride, int sop, int eph,
                                 uint8_t *d_out, size_t cap, uint64_t *d_tile_offs);
int j2k_plan_decode_frame_pixels(j2k_plan *plan, const uint8_t *d_cs, size_t len, const uint64_t *d_tile_offs, int sop, int eph,
                                 void *d_pix, size_t stride);

/* The block coder's outputs as those tables.  j2k_plan_t2_packets (host table out): one packet per (tile, component,
 * resolution) of the plan in job order (encoder.go:616-673: tile, component, resolution, band, block row, block column), its
 * code-blocks = the plan's jobs of that resolution, tree widths = block columns of its first band; layer as given.
 * j2k_plan_t2_fill_cbs (device, asynchronous): code-block j from d_offs / d_lens / d_numbps of j2k_plan_encode_stream:
 * IncludedInLayers 0, len(Passes) = 3 * numBPS - 2 (the passes EncodeFast5 runs, t1_fast5.go:66-70; HT blocks: 1; empty
 * blocks: 0), ZeroBitPlanes = max(mb - numBPS, 0) -- the reference never fills these fields from real blocks (only its
 * tests build precincts, t2_test.go), so this mapping is this library's. */
int j2k_plan_t2_packets(const j2k_plan *plan, int layer, j2k_t2_dev_packet *packets, size_t cap, size_t *count);
int j2k_plan_t2_fill_cbs(j2k_plan *plan, int mb, const uint64_t *d_offs, const uint32_t *d_lens, const uint8_t *d_numbps,
                         j2k_t2_dev_cb *d_cbs);
/* NewTagTree(width, height) (tcd.go:168-197): number of levels and nodes per level (the coder never walks the tree) */
int j2k_tagtree_shape(int width, int height, int32_t *levels, int64_t *level_sizes, size_t cap);

/* TileDecoder.InitTile (tcd.go:240-390; TileEncoder.InitTile :459-500 computes the same tile and component bounds):
 * tile bounds, component bounds after subsampling (ceilDiv), per resolution r its bounds at scale 2^(NumDecompositions - r)
 * and its bands -- LL for r = 0, else HL, LH, HH with the rectangles initBand writes (HL = upper half, LH = left half,
 * HH = lower right quadrant of the RESOLUTION's rectangle, tcd.go:343-361) -- and every band's code-block grid
 * (2^(exp + 2) squares from the band's origin, clipped).  Flat outputs: comps[ncomp], ress[ncomp * (nd + 1)],
 * bands (component-major, resolution, band order), cbs (band by band, row-major); J2K_ERR_CAPACITY with the counts set
 * when a table is too small.  Field ranges: NumDecompositions <= 32, code-block exponents <= 28, subsampling >= 1
 * (beyond them the reference divides by zero: J2K_ERR_GO_PANIC). */
typedef struct j2k_tcd_header {
    uint32_t image_w, image_h, image_x0, image_y0, tile_w, tile_h, tile_x0, tile_y0, num_tiles_x;
    int32_t ncomp;
    const uint8_t *subsampling;          /* SubsamplingX, SubsamplingY per component */
    uint8_t num_decompositions, cb_w_exp, cb_h_exp, pad_[5];
} j2k_tcd_header;
typedef struct j2k_tcd_rect { int32_t x0, y0, x1, y1; } j2k_tcd_rect;
typedef struct j2k_tcd_band { int32_t comp, res, type, cbx, cby, pad_; j2k_tcd_rect r; uint64_t cb0; } j2k_tcd_band;
int j2k_tcd_init_tile(const j2k_tcd_header *h, int tile_index, j2k_tcd_rect *tile, j2k_tcd_rect *comps, j2k_tcd_rect *ress,
                      j2k_tcd_band *bands, size_t band_cap, size_t *nbands, j2k_tcd_rect *cbs, size_t cb_cap, size_t *ncbs);

#ifdef __cplusplus
}
#endif
#endif
