/*
 * j2k_oracle.h -- CPU restatement (plain C) of the go-jpeg2000 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under go-jpeg2000_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it -- as the checker / reported baseline.
 *
 * Every function cites the reference file:line (relative to the reference
 * repository root) whose arithmetic it restates, bug for bug.
 *
 * Parity pinning: the reference (pure Go) cannot be built here (no Go
 * toolchain) and its tests hold no byte/coefficient-level golden vectors; the
 * oracle is pinned by (a) the reference tests' round-trip identities and spot
 * values (tests/test_oracle_reference_identities.py), (b) the hand-derived
 * values of SURVEY.md section 8c, and (c) bit-for-bit agreement with an independent
 * literal Python transliteration (oracle/pyref) on the committed fixtures in
 * tests/golden/.  HT numerics have no pin in the reference's own tests at all
 * ("parity unpinned" by reference tests; pinned only by (c) and the source).
 */
#ifndef J2K_ORACLE_H
#define J2K_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- internal/mct/mct.go ------------------------------------------------ */
void orc_dc_shift_fwd(int32_t *d, size_t n, int precision);            /* mct.go:96-101  */
void orc_dc_shift_inv(int32_t *d, size_t n, int precision);            /* mct.go:113-118 */
void orc_rct_fwd(int32_t *r, int32_t *g, int32_t *b, size_t n);        /* mct.go:28-38   */
void orc_rct_inv(int32_t *y, int32_t *u, int32_t *v, size_t n);        /* mct.go:56-66   */
void orc_ict_fwd(double *r, double *g, double *b, size_t n);           /* mct.go:14-24   */
void orc_ict_inv(double *y, double *cb, double *cr, size_t n);         /* mct.go:43-53   */

/* ---- internal/dwt/dwt.go ------------------------------------------------ */
void orc_fwd53_1d(int32_t *d, int n);                                  /* dwt.go:73-118  */
void orc_inv53_1d(int32_t *d, int n);                                  /* dwt.go:122-147 */
void orc_fwd97_1d(double *d, int n);                                   /* dwt.go:161-210 */
void orc_inv97_1d(double *d, int n);                                   /* dwt.go:213-262 */
void orc_fwd53_2d(int32_t *d, int w, int h);                           /* dwt.go:356-407 */
void orc_inv53_2d(int32_t *d, int w, int h);                           /* dwt.go:410-429 */
void orc_fwd97_2d(double *d, int w, int h);                            /* dwt.go:432-451 */
void orc_inv97_2d(double *d, int w, int h);                            /* dwt.go:454-473 */
void orc_decompose53(int32_t *d, int w, int h, int levels);            /* dwt.go:524-531 */
void orc_reconstruct53(int32_t *d, int w, int h, int levels);          /* dwt.go:534-548 */
void orc_decompose97(double *d, int w, int h, int levels);             /* dwt.go:551-558 */
void orc_reconstruct97(double *d, int w, int h, int levels);           /* dwt.go:561-573 */

/* ---- caller glue: encoder.go:216-281, tcd.go:416-437,508-534, decoder.go:321-348 */
/* encoder.preprocess on C planes of W*H int32 each (in place).
 * lossless!=0: DC shift, RCT (C>=3), DecomposeMultiLevel53.
 * lossless==0: DC shift, ICT (C>=3) with round-half-away, 9-7, v/(1/quality) +-0.5.
 * num_resolutions<=1 -> 5 levels (encoder.go:249-252); quality<=0 -> 100. */
void orc_preprocess(int32_t **planes, int ncomp, int w, int h, int precision,
                    int lossless, int num_resolutions, int quality);
/* tcd.TileEncoder.ApplyForwardDWT (tcd.go:508-534): reversible!=0 -> 5-3, else 9-7 + int32(v+-0.5) */
void orc_tcd_forward_dwt(int32_t *d, int w, int h, int levels, int reversible);
/* tcd.TileDecoder.ApplyInverseDWT (tcd.go:416-437): 9-7 path rounds int32(v+0.5) */
void orc_tcd_inverse_dwt(int32_t *d, int w, int h, int levels, int reversible);
/* decoder.decodeTiles tail (decoder.go:321-348): inverse MCT (if mct && C>=3) + DC shift (unsigned) */
/* getColorConversion(cs)(componentData, precision), colorspace.go:54-480; cs = the reference's ColorSpace constants
 * (jpeg2000.go:124-197).  In place; spaces / component counts without a conversion leave the data untouched. */
void orc_convert_colorspace(int cs, int32_t **planes, int ncomp, size_t n, int precision);
/* RawEncoder / RawDecoder (mqc.go:516-600): n bits in (one per byte) -> bytes; returns the byte count / fills bits */
long orc_raw_encode(const uint8_t *bits, size_t n, uint8_t *out, size_t cap);
void orc_raw_decode(const uint8_t *data, size_t len, size_t n, uint8_t *bits);
/* encoder.extractImageData (encoder.go:79-213) / decoder.createImage (decoder.go:417-588) on Go image Pix layouts;
 * formats: 0 Gray 1 Gray16 2 RGBA 3 RGBA64 4 NRGBA 5 NRGBA64.  Return the component count (0 = bad argument). */
int orc_extract_image_data(int format, const uint8_t *pix, size_t stride, int w, int h, int target_precision, int32_t **planes);
int orc_create_image(int32_t *const *planes, int ncomp, int precision, int w, int h, uint8_t *pix, size_t stride);
void orc_postprocess(int32_t **planes, int ncomp, size_t n, int precision,
                     int reversible, int mct, int is_signed);

/* ---- internal/entropy: MQ coder (mqc.go:169-349, 352-497) ---------------- */
/* Encode n (ctx,decision) pairs with a fresh MQEncoder and Flush().  Returns the
 * number of bytes written to out (0 == Go nil), or -1 if cap is too small. */
long orc_mq_encode(const uint8_t *ctx, const uint8_t *dec, size_t n, uint8_t *out, size_t cap);
/* Decode n decisions for the given context sequence with a fresh MQDecoder. */
void orc_mq_decode(const uint8_t *bytes, size_t nbytes, const uint8_t *ctx, size_t n, uint8_t *dec_out);
/* copy out the 94-entry state table (qe, nmps, nlps) and the three T1 LUTs */
void orc_mq_table(uint32_t qe[94], uint8_t nmps[94], uint8_t nlps[94]);
void orc_t1_luts(uint8_t zc[1024], uint8_t sc[256], uint8_t sp[256]); /* t1_luts.go:32-231 */

/* ---- internal/entropy: T1 (t1.go, t1_fast5.go, t1_fast.go) ---------------- */
/* T1.SetData + T1.Encode(band) == EncodeFast5 (t1.go:292-304, t1_fast5.go:10-899).
 * data: w*h signed coefficients.  Returns byte count (0 == Go nil: all-zero block)
 * or -1 if cap too small.  *numbps receives the bit-plane count (0 for nil). */
long orc_t1_encode(const int32_t *data, int w, int h, int band,
                   uint8_t *out, size_t cap, int *numbps);
/* NewT1(w,h).Decode(bytes, numBPS, band) (t1.go:1261-1410) -> out[w*h] */
void orc_t1_decode(const uint8_t *bytes, size_t nbytes, int numbps, int band,
                   int w, int h, int32_t *out);

/* ---- internal/entropy: "HT" block coder (ht.go) --------------------------- */
/* NewHTEncoder(w,h).SetData(data).Encode(band) (ht.go:942-1391).
 * Returns byte count (0 == Go nil), -1 cap too small, -2 where the Go code would
 * panic (index out of range in its fixed-size stream buffers) or never return. */
long orc_ht_encode(const int32_t *data, int w, int h, int band, uint8_t *out, size_t cap);
/* NewHTDecoder(w,h).Decode(bytes, numBitplanes, band) (ht.go:93-150, 583-864) on a
 * fresh (zeroed) decoder -> out[w*h].  Returns 0, or -2 where Go would panic. */
int orc_ht_decode(const uint8_t *bytes, size_t nbytes, int num_bitplanes, int band,
                  int w, int h, int32_t *out);
/* worst-case output bytes of orc_ht_encode for a w x h block */
size_t orc_ht_bound(int w, int h);

/* ---- encoder.encodeTile job list (encoder.go:597-688, 763-795) ------------ */
typedef struct {
    int32_t comp, res, band;      /* band: 0 LL, 1 HL, 2 LH, 3 HH (t1.go:125-130) */
    int32_t x0, y0;               /* window origin in the component plane (top-left addressing) */
    int32_t w, h;                 /* actualWidth, actualHeight */
} orc_block;
/* Enumerate jobs in reference order comp -> res -> band -> cby -> cbx.  Returns the
 * number of jobs; writes at most cap of them.  cb_w/cb_h are the REAL block sizes
 * (1 << (CodeBlockSize+2)). num_resolutions<=0 -> 6 (encoder.go:601-604). */
size_t orc_enumerate_blocks(int ncomp, int w, int h, int num_resolutions,
                            int cb_w, int cb_h, orc_block *out, size_t cap);
/* extractCodeBlockData (encoder.go:763-795): copy the job's window out of a plane of
 * stride plane_w (zero outside the plane). */
void orc_extract_block(const int32_t *plane, int plane_w, int plane_h,
                       const orc_block *b, int32_t *dst);
/* Sequential encodeTile body (encoder.go:677-688): concatenated block bytes for one
 * tile.  coder: 0 = MQ T1, 1 = HT (what Options.HighThroughput intends).
 * lens[i]/numbps[i] per job (may be NULL).  Returns total bytes, -1 cap, -2 HT panic. */
long orc_encode_tile_blocks(int32_t *const *planes, int ncomp, int w, int h,
                            int num_resolutions, int cb_w, int cb_h, int coder,
                            uint8_t *out, size_t cap, uint32_t *lens, uint8_t *numbps);

/* The same with a window mode: 0 = the reference's top-left windows (the functions above); 1 = this library's CLOSED-LOOP
 * mode, not the reference: the bands are the Mallat rectangles of the plane, which partition it (see j2k_oracle.c). */
size_t orc_enumerate_blocks2(int ncomp, int w, int h, int num_resolutions,
                             int cb_w, int cb_h, int windows, orc_block *out, size_t cap);
long orc_encode_tile_blocks2(int32_t *const *planes, int ncomp, int w, int h,
                             int num_resolutions, int cb_w, int cb_h, int coder, int windows,
                             uint8_t *out, size_t cap, uint32_t *lens, uint8_t *numbps);
/* DecodeCodeBlock (tcd.go:393-413) for every job + each block put back at its window of the zeroed planes: the decode body
 * decoder.decodeTile leaves out (decoder.go:375-411).  Returns 0, or -2 where the Go HT decoder would panic. */
int orc_decode_tile_blocks(const uint8_t *bytes, const uint32_t *lens, const uint8_t *numbps, int ncomp, int w, int h,
                           int num_resolutions, int cb_w, int cb_h, int coder, int windows, int32_t *const *planes);

/* encoder.createTileHeader(tileIdx, tileData) (encoder.go:746-760): writes 14 + len bytes to out, returns that count */
size_t orc_create_tile_header(int tile_idx, const uint8_t *tile_data, size_t len, uint8_t *out);

/* srgbGamma / srgbInverseGamma (colorspace.go:302-315), probed by colorspace_spec_test.go:396-417 */
double orc_pin_srgb_gamma(double linear);
double orc_pin_srgb_inverse_gamma(double encoded);

/* ---- pins: internals at the granularity of the reference's own unit tests ------------
 * (internal/entropy/coverage_test.go, t1_test.go; tests/test_oracle_reference_pins.py) */
void orc_pin_mq_byte_out(uint8_t *buf, size_t buflen, long bp, uint32_t c,
                         long *bp_out, uint32_t *c_out, uint32_t *ct_out);          /* t1_fast.go:11-34 */
int orc_pin_mq_needs_slow_path(uint8_t buf_byte, uint32_t c);                        /* mq_inline.go:96-98 */
void orc_pin_mq_byte_in(const uint8_t *data, long len, long *bp, uint32_t *C, uint32_t *CT, int *end_counter); /* mqc.go:402-439 */
void orc_pin_sign_context_params(int hc, int vc, int *ctx, int *xorbit);             /* mq_inline.go:26-65 (dead code) */
int orc_pin_sign_contrib(int flag);                                                  /* mq_inline.go:70-78 */
int orc_pin_clamp_contrib(int c);                                                    /* mq_inline.go:83-91 */
int orc_pin_lut_sc_ctx(int hc, int vc);                                              /* t1_luts.go:112-150, 240-258 */
void orc_pin_update_neighbor_flags(uint8_t *flags, int w, int h, int x, int y);      /* t1.go:328-345 */
int orc_pin_has_sig_neighbor(uint8_t *flags, int w, int h, int x, int y);            /* t1.go:1087-1092 */
int orc_pin_mr_context(uint8_t *flags, int w, int h, int x, int y);                  /* t1.go:463-479 */
int orc_pin_zc_context(uint8_t *flags, int w, int h, int x, int y, int band);        /* t1.go:312-384 */
void orc_pin_sc_context(uint8_t *flags, int w, int h, int x, int y, int *ctx, int *pred); /* t1.go:387-460 */
int orc_pin_can_use_run_length(uint8_t *flags, int w, int h, int x, int y);          /* t1.go:1195-1208 */

#ifdef __cplusplus
}
#endif
#endif
