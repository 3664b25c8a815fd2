/*
 * omnitok.h -- C ABI of the MI355X-native OmniTokenizer encode/decode path.
 *
 * The reference (FoundationVision/OmniTokenizer) is pure Python and has no FFI: its boundary is
 * the class OmniTokenizer_VQGAN (reference OmniTokenizer/omnitokenizer.py:63) with
 *     encode(x, is_image, include_embeddings=False)   omnitokenizer.py:247-266
 *     decode(encodings, is_image)                      omnitokenizer.py:268-317
 * and a PyTorch state_dict as the weight format.  This header is what a binding
 * for that path binds instead: plain device pointers, sizes and a hipStream_t, no torch types.
 * omnitokenizer_amd/vqgan.py is the ctypes binding that mirrors the reference class on top of it
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (ids: int64) unless named host_*;
 *  - all tensors are dense row-major in the layout stated per function;
 *  - every launch goes to the given stream, no hidden synchronisation (workspace growth inside
 *    omnitok_encode/decode is the one exception: it may hipMalloc);
 *  - functions return 0 on success, a negative omnitok_status otherwise, and
 *    omnitok_last_error() returns a thread-local description (the Python layer raises it as
 *    RuntimeError/ValueError like the reference's asserts);
 *  - inference only: no autograd, no training-time codebook updates (codebook.py:95-118).
 */
#ifndef OMNITOK_H
#define OMNITOK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *omnitok_stream_t; /* hipStream_t */

enum omnitok_status {
    OMNITOK_OK = 0,
    OMNITOK_ERR_INVALID = -1,     /* bad argument / unsupported shape   */
    OMNITOK_ERR_HIP = -2,         /* a HIP runtime call failed          */
    OMNITOK_ERR_STATE = -3,       /* engine not finalised / missing weight */
    OMNITOK_ERR_UNSUPPORTED = -4  /* configuration outside the built path  */
};

const char *omnitok_last_error(void);
/* "omnitok <version> gfx950 ..." */
const char *omnitok_version(void);
/* Process-wide DEFAULTS of the arithmetic / data-flow modes (an engine follows them unless omnitok_engine_set_option gave
 * it its own value: two engines of one process can run different modes) -- every mode holds the parity bars of tests/:
 *   "gemm_mode"  2 (default) fp32 operands as 2 fp16 planes, 3 fp16-MFMA products (gemm_h2.hip, gemm_pl.h) | 1 three bf16
 *                planes, 6 products (gemm_x3.hip) | 0 fp32-input MFMA (gemm.hip)
 *   "attn_mode"  1 (default) spatial attention on the fp16 matrix cores from split operands (attn_h2.hip) | 0 fp32 MFMA
 *   "attn_vpack" 1 (default) the merged to_q|to_kv launch stores V as packed fp16 planes | 0 attn_pack packs V too
 *   "gemm_pl"    1 (default) plane data flow: attention kernels, the to_out / proj epilogue and the GEGLU epilogue write
 *                the next GEMM's operand as fp16 hi|lo planes (gemm_pl.h) | 0 fp32 activations, split in the K loop
 *   "pl_min_tokens" 0 (default since r06: one data flow at every call size -- a clip gets the same bits alone and in any
 *                batch; the plane GEMM picks thin tiles for small calls, "pl_cfg") | n > 0: calls with fewer tokens
 *                (B * T' * h * w) take the gemm_pl 0 data flow (rounds 4-5 shipped 12288).  Also a per-engine option
 *                (omnitok_engine_set_option); the two flows round differently, both inside the parity bars
 *   "pl_cfg"     0 (default) the plane GEMM's tile configuration by call size (256 x 256 | 128 x 128 | 128 x 64; LayerNorm
 *                epilogue 128 | 64 | 32 rows) | 1, 2, 5, 6 (7) force one.  Results do not depend on it, bit for bit
 *   "temporal_chunk" 0 (default; also per engine) > 0: temporal blocks run GEMM(chunk of clips) -> attention(chunk) through
 *                one chunk-sized q|k|v buffer (bit-identical; measured slower at C3, profiles/r05_temporal_chunk.txt)
 *   "prevq_fuse" 1 (default) omnitok_encode runs pre_vq inside the encoder's last LayerNorm pass (bit-identical) | 0 two passes
 *   "temporal_fused" 1 (default) temporal blocks with T' = 5 (causal, S % 64 == 0) run as two GEMM launches with the attention
 *                in their epilogues (omnitok_pl_gemm epilogues 6 / 7: no fp32 q|k|v in memory) | 0 q|k|v GEMM + the
 *                omnitok_attn_temporal kernel; the two forms agree to rounding (profiles/r05_temporal_fused.txt)
 *   "temporal_kernel" 1 (default, process-wide) the two launches of "temporal_fused" run on gemm_plt_kernel: two workgroups per CU, so
 *                one's epilogue overlaps the other's K loop (-0.9 ms per step, profiles/r05_temporal_plt.txt) | 2 the same kernel with
 *                one workgroup per CU (the reference of the co-residency test) | 0 the first form, gemm_pl_kernel at one wave per SIMD;
 *                all three agree to rounding
 *   "attn_window_mode" 1 (default, process-wide, with gemm_pl / qkv_pl) window attention on the fp16 matrix cores from packed
 *                operands (omnitok_stats_pack_windows -> packing epilogues -> omnitok_attn_window_h2) | 0 fp32 q|k|v and the
 *                fp32-MFMA kernel omnitok_attn_window_planes
 *   "qkv_pl"     1 (default, process-wide) with gemm_pl: the q|k|v projection reads centred planes from omnitok_stats_pack,
 *                has the LayerNorm folded into its weight and writes the attention kernel's packed Q / K / V itself | 0 the
 *                round-2 form (row_stats -> gemm_h2 with in-loop LayerNorm -> attn_pack)
 * Tuning knobs of the stand-alone kernel entry points for A/B measurements (process-wide; results do not depend on them --
 * tests check bitwise independence of the tile shape): "gemm_variant", "gemm_small", "gemm_gn", "gemm_lds_pad_kb",
 * "x3_tile", "h2_tile" (0 auto | 1 256x256 | 3 128x128 | 4 64x64 | 5 256x128 | 6 128x256), "pl_cfg" (above), "pl_tail" (1 default:
 * full rounds of 256x256 tiles + the last round's rows on thin tiles in a second launch | 0 one launch), "sp_small_blocks" (0 default:
 * omnitok_stats_pack in 16-row workgroups at every size | n: only below n 64-row blocks), "attn_h2_variant", "vq_split", "vq_screen" (1 default: omnitok_encode uses the screened search omnitok_vq_argmin_screened | 0 the exact sweep), "vq_screen_split", "vq_variant" (1 default: distance finished on the matrix pipe | 0 VALU epilogue | 2 codebook
 * staged in LDS; all bit-exact, profiles/r03_vq_variants.txt), "pl_stagger" (start delay step of persistent GEMM workgroups in
 * ~1 us units, 0 = off: a measured no-gain knob), "peg_variant" (1 default: LDS-tiled -- the 64-channel kernel of peg_wide.h for
 * 2..8 planes on grids with W % 16 == 0, H % 4 == 0, D % 64 == 0, its one-plane form (9 taps, four workgroups per CU) for images,
 * else the 32-channel time ring | 2 the 64-channel walk kernel whenever the grid allows | 3 time ring only (1, 2, 3 bit-identical)
 * | 0 register-blocked), "lm_wide_u"; LM decode A/B switches (include/omnitok_lm.h; profiles/r05_lm_timeline.txt): "lm_ksliced" 1
 * (default) K-sliced GEMV for B <= 2 and K in {1536, 2048, 6144, 8192} | 0 row GEMV; "lm_balance" 1 (default) 6 waves per workgroup
 * where that makes the row GEMV's grid a whole number of workgroups per CU | 0 four; "lm_attn_waves" 8 (default) | 4 waves per
 * 256-key attention chunk; "lm_attn_short" 1 (default, read by omnitok_lm_alloc_cache) 128-key chunks for caches of up to 4096
 * tokens | 0 always 256; "lm_ks_deep" 0 (default) 8 KiB | 1 16 KiB of weights in flight per wave of the K-sliced GEMV; "lm_mfma" 0 (default) | 1 groups of
 * 4 .. 8 streams on the fp32-MFMA GEMV (measured slower, profiles/r06_lm_mfma.txt); "x3_dbg" / "h2_dbg" select
 * wrong-result ablation builds (tools/x3_ablate.py, tools/h2_bench.py).  Unknown names return OMNITOK_ERR_INVALID. */
int omnitok_set_option(const char *name, int value);
/* Reads the process default of a data-flow option ("gemm_mode", "attn_mode", "gemm_pl", "pl_min_tokens", "temporal_chunk",
 * "prevq_fuse"): what a clip-sharded job consults to pin ONE flow for all its ranks (omnitokenizer_amd/dist.py pin_data_flow). */
int omnitok_get_option(const char *name, int *value);
/* ------------------------------------------------------------------------------------------
 * Per-operator entry points (each is one HIP kernel family; the engine below chains them).
 * They are exported so that every operator is parity-tested against the oracle on its own.
 * ------------------------------------------------------------------------------------------ */

/* y[out_row(m), :] = LayerNorm(x[m, :]) * gamma + beta   (beta may be NULL = 0; eps 1e-5)
 * reference attention.py:73-80 (custom LayerNorm) and nn.LayerNorm (attention.py:163,
 * omnitokenizer.py:809-821).  out_row(m) = (m / rows_per_group) * group_stride + group_offset
 * + m % rows_per_group; pass rows_per_group = 0 for the identity map. dim % 4 == 0, dim <= 1024. */
int omnitok_layernorm(const float *x, const float *gamma, const float *beta, float *y,
                      int64_t rows, int dim, float eps,
                      int64_t rows_per_group, int64_t group_stride, int64_t group_offset,
                      omnitok_stream_t stream);
/* The same LayerNorm with the token transpose '(n a c) d -> (n c a) d' fused into the store (y != x): the final
 * norm_out of a Transformer followed by the rearrange to the other stage's token order
 * (reference omnitokenizer.py:894-903, 1072-1084). */
int omnitok_layernorm_transposed(const float *x, const float *gamma, const float *beta, float *y, int64_t n, int a,
                                 int c, int dim, float eps, omnitok_stream_t stream);
/* The encoder's last norm_out LayerNorm (optionally with the token transpose above: transpose != 0) FUSED with pre_vq:
 * z[row', 0:8] = l2norm(LayerNorm(x[row]) . w[8, dim]^T + b) (reference omnitokenizer.py:143-148 pre_vq_conv, :251-252
 * F.normalize; l2 == 0 skips the normalisation), row' = the LayerNorm's output row.  The normalised tokens are not
 * written.  Bit-identical to omnitok_layernorm[_transposed] followed by omnitok_pre_vq.  dim 256, 512 or 1024. */
int omnitok_layernorm_prevq(const float *x, const float *gamma, const float *beta, const float *w, const float *b,
                            float *z, int64_t n, int a, int c, int dim, float eps, int transpose, int l2,
                            omnitok_stream_t stream);

/* Epilogue flags of omnitok_gemm */
#define OMNITOK_GEMM_BIAS 1      /* + bias[n]                                               */
#define OMNITOK_GEMM_RESIDUAL 2  /* + residual[m, n] (may alias c)                           */
#define OMNITOK_GEMM_GEGLU 4     /* c[m, j] = gelu(acc[m, gate j]) * acc[m, value j]; weight  */
                                 /* rows pre-interleaved by omnitok_pack_geglu_weight         */
#define OMNITOK_GEMM_LEAKY 8     /* leaky_relu(., 0.1) after bias                              */

/* c[M,N] = a[M,K] . w[N,K]^T (+epilogue), fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
 * This is nn.Linear (y = x W^T + b) as used by every projection on the path
 * (attention.py:164,167,271,287,386-393; omnitokenizer.py:146,158,810,819,1007,1013).
 * K % 32 == 0; lda, ldw % 4 == 0; pointers 16-byte aligned. With GEGLU, N is the packed width
 * (2 * padded inner) and c has N/2 columns.
 * a_row(m) = (m / a_rows_per_group) * a_group_stride + a_group_offset + m % a_rows_per_group
 * (0 = identity) lets the A operand be read from a strided group of token rows. */
int omnitok_gemm(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                 const float *residual, int64_t ldr, float *c, int64_t ldc,
                 int64_t M, int N, int K, int flags,
                 int64_t a_rows_per_group, int64_t a_group_stride, int64_t a_group_offset,
                 omnitok_stream_t stream);

/* --- fp32 GEMM on the bf16 matrix cores, operands split in-kernel (csrc/gemm_x3.hip) ----------------
 * Same contract as omnitok_gemm (c = a . w^T + epilogue, fp32 operands in HBM, fp32 result) but each
 * fp32 operand element is split exactly into three bf16 numbers while it is staged to LDS and six
 * v_mfma_f32_32x32x16_bf16 products per element pair are accumulated in fp32 (the three dropped
 * products are < 2^-24 |a b|).  K % 32 == 0, N % 32 == 0.  The per-element arithmetic does not depend
 * on M, N or the tile shape, so results are independent of the batch size.
 * Optional fused LayerNorm of the A operand (reference attention.py:73-80,163: the norm that precedes
 * to_q / qkv / the FeedForward's first Linear): ln_stats[rows][2] = (mean, rstd) per PHYSICAL a row
 * (omnitok_row_stats), ln_gamma[K], ln_beta[K] (NULL = 0), K <= 512; output columns [0, ln_cols) are
 * computed from LN(a), columns [ln_cols, N) from a itself (Q from LN(x), K/V from x in one launch,
 * attention.py:404-412).  ln_stats == NULL: no LayerNorm.
 * split_col > 0 (a multiple of 256, plain / bias epilogues only): output columns [split_col, N) are written
 * to c2 (row stride ldc2) instead of c, so that one launch produces Q and K|V as two dense tensors. */
int omnitok_gemm_x3(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                    const float *residual, int64_t ldr, float *c, int64_t ldc,
                    int64_t M, int N, int K, int flags,
                    int64_t a_rows_per_group, int64_t a_group_stride, int64_t a_group_offset,
                    const float *ln_stats, const float *ln_gamma, const float *ln_beta, int ln_cols,
                    float *c2, int64_t ldc2, int split_col, omnitok_stream_t stream);
/* --- fp32 GEMM on the fp16 matrix cores, 2-way split operands (csrc/gemm_h2.hip) ---------------------
 * c = a . w^T (+epilogue as omnitok_gemm) with each operand element a' = hi + lo (two fp16 numbers, 22
 * significand bits) and three v_mfma_f32_32x32x16_f16 products per element pair, fp32 accumulation.
 * The weight is packed once by omnitok_h2_pack_weight (row-scaled by a power of two, split, in blocks of 64
 * rows x 32 k: planes[ceil(N/64)][K/32][hi|lo][4][64][8] fp16 = 4 bytes per element, scale[N]).  fp16 has 5 exponent bits, so the caller
 * states an UPPER BOUND of |a|: a_bound (> 0), multiplied by a_bound_dev[a_bound_stride * (m /
 * a_rows_per_clip)] for row m when that device pointer is given (bounds produced by an earlier kernel, e.g.
 * omnitok_row_stats; a_rows_per_clip <= 0: one value for all rows); the kernel scales the rows of a clip by
 * one power of two so that |a'| <= 2^15 (tiles never straddle clips: a_rows_per_clip % 64 == 0).  A wrong (too small) bound overflows to inf; elements more than
 * 2^18 below the bound lose relative precision (absolute error <= 2^-40 of the bound).
 * Fused LayerNorm as omnitok_gemm_x3, with ln_bound >= max |LN(a)| (sqrt(K) max|gamma| + max|beta|).
 * K % 32 == 0, N % 32 == 0.  Results are independent of M and of the tile shape. */
int omnitok_h2_pack_weight(const float *w, int64_t ldw, int N, int K, void *planes, float *scale,
                           omnitok_stream_t stream);
int omnitok_gemm_h2(const float *a, int64_t lda, const void *w_planes, const float *w_scale,
                    const float *bias, const float *residual, int64_t ldr, float *c, int64_t ldc,
                    int64_t M, int N, int K, int flags,
                    int64_t a_rows_per_group, int64_t a_group_stride, int64_t a_group_offset,
                    float a_bound, const float *a_bound_dev, int a_bound_stride, int64_t a_rows_per_clip,
                    const float *ln_stats, const float *ln_gamma, const float *ln_beta, int ln_cols,
                    float ln_bound, float *c2, int64_t ldc2, int split_col, omnitok_stream_t stream);
/* omnitok_gemm_h2 whose columns [v_col0, N) -- the V projection of a merged to_q | to_kv launch (reference
 * attention.py:404-412) -- are not stored as fp32 but written straight into the packed fp16 hi|lo V planes of
 * omnitok_attn_spatial_h2 (the 32x32 MFMA accumulator layout is that layout), bit-identical to storing them and
 * running omnitok_attn_pack on the result.  v_planes NULL: plain omnitok_gemm_h2.  Needs flags == 0, the fused
 * LayerNorm, v_col0 % 256 == 0, N - v_col0 == heads * 64, M a whole number of n_tokens-row sequences
 * (n_tokens % 32 == 0) and a_rows_per_clip % 128 == 0; v_bound (times v_bound_dev[v_bound_stride * clip]) >= max|v|. */
int omnitok_gemm_h2_vpack(const float *a, int64_t lda, const void *w_planes, const float *w_scale,
                          const float *bias, const float *residual, int64_t ldr, float *c, int64_t ldc,
                          int64_t M, int N, int K, int flags,
                          int64_t a_rows_per_group, int64_t a_group_stride, int64_t a_group_offset,
                          float a_bound, const float *a_bound_dev, int a_bound_stride, int64_t a_rows_per_clip,
                          const float *ln_stats, const float *ln_gamma, const float *ln_beta, int ln_cols,
                          float ln_bound, float *c2, int64_t ldc2, int split_col, void *v_planes, int v_col0,
                          int n_tokens, int heads, float v_bound, const float *v_bound_dev, int v_bound_stride,
                          omnitok_stream_t stream);
/* --- plane x plane GEMM (csrc/gemm_pl.h, gemm_pl.hip) --------------------------------------------------------
 * The arithmetic of omnitok_gemm_h2 with BOTH operands already stored as fp16 hi|lo planes in blocks of 64 rows
 * x 32 k ([rows/64][K/32][hi|lo][4 k groups][64 rows][8] fp16, 8 KiB per block): the K loop is LDS-DMA + ds_read +
 * MFMA only, the producers of the activations (LayerNorm / GEGLU / attention epilogues, omnitok_pl_pack_rows,
 * omnitok_ln_pack) write the planes.  Every nn.Linear of the Transformer blocks runs through it (reference
 * attention.py:153-168, 386-393, 216-252).
 *   weights: omnitok_pl_pack_weight (row-scaled by a power of two, rows permuted inside groups of 32 so that an
 *            accumulator lane owns runs of 8 consecutive output columns); n_pad % 256 == 0 rows are written.
 *   activations: one power-of-two scale per row (a_scale[m] = the factor that undoes it) and / or one static
 *            factor; the plane buffer must hold ceil(M / 256) * 256 rows (rows >= M are never stored to outputs).
 * c[m, n] = (sum_k a'[m, k] w'[n, k]) * a_scale[m] * a_scale_const * w_scale[n]  (+ bias[n]) (+ residual[m, n]). */
typedef struct omnitok_pl_gemm {
    const void *a;              /* activation planes                                                            */
    const float *a_scale;       /* [M] or NULL                                                                  */
    float a_scale_const;        /* <= 0: 1                                                                      */
    const void *a2;             /* optional second activation operand for output columns >= a_split_n           */
    const float *a2_scale;
    float a2_scale_const;
    int a_split_n;              /* % 256 == 0                                                                   */
    const void *w;              /* weight planes                                                                */
    const float *w_scale;       /* [N]                                                                          */
    const float *bias;          /* [N] or NULL                                                                  */
    const float *residual;      /* [M, ldr] or NULL (may alias c)                                               */
    int64_t ldr;
    float *c;                   /* fp32 output [M, ldc] (epilogue 0)                                            */
    int64_t ldc;
    float *c2;                  /* optional: columns >= c_split_n go to c2[m, n - c_split_n]                    */
    int64_t ldc2;
    int c_split_n;
    void *out_planes;           /* epilogue 1 (GEGLU): hidden planes, K = out_planes_k = N / 2;                 */
    int out_planes_k;           /* epilogue 2: LayerNorm(c) planes, out_planes_k = N                            */
    float out_bound;            /* >= max |plane values|: the consumer uses a_scale_const = omnitok_pl_unscale() */
    const float *ln_gamma;      /* epilogue 2: LayerNorm over the N output columns (two-pass statistics)        */
    const float *ln_beta;       /* or NULL                                                                      */
    float ln_eps;
    int epilogue;               /* 0 fp32 | 1 GEGLU -> planes | 2 fp32 AND LayerNorm -> planes (N == 512) |     */
                                /* 3 packed V planes of omnitok_attn_spatial_h2 | 4 packed Q | K planes |         */
                                /* 5 fp32 pixels (un-patchify store, up_* below) | 6 temporal scores | 7 temporal P.V */
    /* LayerNorm folded into the weight (epilogues 0, 3, 4): the operand a holds the CENTRED rows x - mean(x)
     * (omnitok_stats_pack), the weight rows n < fold_cols were multiplied by gamma before packing, and the epilogue finishes
     *   n <  fold_cols:  LayerNorm(x) . W^T = rstd_m ((x - mean_m) . (W o gamma)^T) + b_n     (b = W beta, or NULL)
     *   n >= fold_cols:  x . W^T           = (x - mean_m) . W^T + mean_m u_n                  (u_n = sum_k W[n][k])   */
    const float *fold_stats;    /* [M][2] (mean, rstd) or NULL                                                  */
    const float *fold_b;
    const float *fold_u;        /* [N]; may be NULL when fold_cols >= N                                          */
    int fold_cols;
    /* epilogues 3 / 4: the attention operands (reference attention.py:417-437).  4: columns [0, qk_k0) are Q, the rest
     * K; per 64-column head RoPE (cos / sin [n_tokens][32] or NULL), l2norm, q_scale / k_scale [64], q_mul (the SDPA
     * scale) and the power-of-two scales of q_bound / k_bound, written as the packed planes omnitok_attn_pack writes.
     * 3: N = heads * 64 columns of V scaled per clip by v_bound (x v_bound_dev[v_bound_stride * clip]).             */
    void *qp;
    void *kp;
    void *vp;
    int qk_k0, n_tokens, heads;
    const float *rope_cos;
    const float *rope_sin;
    const float *q_scale;
    const float *k_scale;
    float q_mul, q_bound, k_bound, v_bound;
    const float *v_bound_dev;
    int v_bound_stride;
    int64_t rows_per_clip;
    int64_t M;
    int N, K;
    int cfg;                    /* 0 auto | 1 256x256 tiles, one workgroup per CU | 2 128x256, two per CU        */
    long long *debug_cycles;    /* measurement: shader-clock span of workgroup 0 (NULL = off)                   */
    /* Operand row map (0 = identity): GEMM row m reads plane row (m / a_rpg) * a_gstride + a_goff + m % a_rpg (a_rpg and
     * a_goff multiples of 256: the frame groups of the token tensor, reference omnitokenizer.py:1006-1017 to_pixels_first_frame /
     * to_pixels take tokens[:, :1] / tokens[:, 1:]).  Needs a_scale == NULL (one static scale, a_scale_const) and no fold. */
    int64_t a_rpg, a_gstride, a_goff;
    /* epilogue 5: fp32 output (+ bias) scattered as pixels -- row m = patch (b, t, gy, gx) of a [B, up_C, up_F, up_H, up_W]
     * video `c`, column n = feature ((ch up_pt + j) up_p + p1) up_p + p2, frame up_f0 + t up_pt + j (the Rearrange
     * 'b t h w (c pt p1 p2) -> b c (t pt) (h p1) (w p2)' of omnitokenizer.py:1006-1017 fused into the store).
     * up_p == 8 and (up_W / up_p) % 32 == 0: a 32-row block of the tile is 32 consecutive patches of one patch row     */
    int up_C, up_F, up_H, up_W, up_f0, up_t, up_pt, up_p;
    /* 0, or the number of leading k that are not zero padding in BOTH operands (planes and weight are laid out for K): the
     * K loop then stops after ceil(k_valid / 16) steps -- FF-out's K = 1408 holds 1365 hidden channels                */
    int k_valid;
    /* epilogues 6 / 7: the temporal stage (reference attention.py:402-486, is_spatial = False, T' = 5, causal) without its
     * q|k|v round trip.  The operand rows are permuted by omnitok_stats_pack_temporal: [tile of 64 sequences][half][time
     * step][32 sequences], M = ceil(t_nseq / 64) * 320.  6: N = heads * 128 columns [q_h | k_h] per head (weight rows
     * permuted accordingly, q rows LayerNorm-folded: fold_b / fold_u / w_scale in the same order), q_scale / k_scale [64],
     * q_mul the SDPA scale, t_alibi [heads] or NULL -> tp[t_nseq][heads][40]: per query step 8 floats (e[0..4], 1 / sum e, 0, 0).
     * 7: N = heads * 64 columns of V -> out_planes (K = N) at the TOKEN rows seq * 5 + t, scaled per clip by v_bound
     * (x v_bound_dev[v_bound_stride * (seq / t_seqs_per_clip)]); t_out_scale[token row] = the factor that undoes it. */
    float *tp;
    int t_nseq, t_heads, t_seqs_per_clip;
    const float *t_alibi;
    float *t_out_scale;
} omnitok_pl_gemm;
int64_t omnitok_pl_planes_bytes(int64_t rows, int K, int row_pad);
/* the factor that undoes the power-of-two scale derived from a static bound (a_scale_const of the consumer) */
float omnitok_pl_unscale(float bound);
int omnitok_pl_pack_weight(const float *w, int64_t ldw, int N, int K, int n_pad, void *planes, float *scale,
                           omnitok_stream_t stream);
/* x[M, ldx] fp32 -> planes (m_pad rows, rows >= M zero); per-row scales into a_scale, or one static scale from
 * static_bound (> 0) */
int omnitok_pl_pack_rows(const float *x, int64_t ldx, int64_t M, int K, int64_t m_pad, void *planes, float *a_scale,
                         float static_bound, omnitok_stream_t stream);
int omnitok_gemm_pl(const omnitok_pl_gemm *g, omnitok_stream_t stream);
/* Row statistics AND planes in one pass over x[rows, dim] (dim % 256 == 0): stats[m] = (mean, 1/sqrt(var + eps)) like
 * omnitok_row_stats, planes = the rows of x (center == 0) or the CENTRED rows x - mean (center != 0) with one power-of-two
 * scale per row (a_scale[m] undoes it), bounds as omnitok_row_stats.  The centred form is the operand of the q|k|v plane
 * GEMM, whose epilogue finishes LayerNorm(x) . Wq^T and x . Wkv^T from it (omnitok_pl_gemm.fold_*; reference
 * attention.py:404-412). */
int omnitok_stats_pack(const float *x, int64_t rows, int dim, float eps, int center, void *planes, int64_t m_pad,
                       float *a_scale, float *stats, float *bounds, int64_t rows_per_clip, omnitok_stream_t stream);
/* The same pass with the OUTPUT rows (planes, a_scale, stats) in window-major order -- (frame, window, position inside the
 * ws x ws window), the row order reference attention.py:170-188 (window_partition) gives the WindowAttention operand --
 * while x [rows, dim] is read in token order (frame, y, x) of a gh x gw grid.  The q|k|v plane GEMM of a 'w' block then sees
 * one window per 64 consecutive rows and its packing epilogues (3 / 4 with n_tokens = ws * ws) write the packed operands
 * of omnitok_attn_window_h2 directly. */
int omnitok_stats_pack_windows(const float *x, int64_t rows, int dim, float eps, int center, void *planes, int64_t m_pad,
                               float *a_scale, float *stats, int gh, int gw, int ws, omnitok_stream_t stream);
/* ... and with the OUTPUT rows in the order the fused temporal stage wants (omnitok_pl_gemm epilogues 6 / 7): x [nseq * 5, dim]
 * in token order (sequence, time step) -> centred planes, a_scale and stats at row ((s / 64) * 2 + (s % 64) / 32) * 160 +
 * t * 32 + s % 32, ceil(nseq / 64) * 320 rows in all; bounds per clip of seqs_per_clip sequences as omnitok_row_stats. */
int omnitok_stats_pack_temporal(const float *x, int64_t nseq, int dim, float eps, void *planes, float *a_scale,
                                float *stats, float *bounds, int64_t seqs_per_clip, omnitok_stream_t stream);
/* LayerNorm(x) [rows, dim] straight into hi|lo planes scaled by the power of two of `bound` (>= max |LayerNorm(x)|; the
 * consumer's a_scale_const = omnitok_pl_unscale(bound)): a standalone LayerNorm in front of a plane GEMM in one pass. */
int omnitok_layernorm_planes(const float *x, int64_t rows, int dim, float eps, const float *gamma, const float *beta,
                             float bound, void *planes, int64_t m_pad, omnitok_stream_t stream);

/* stats[m][2] = (mean, 1/sqrt(var + eps)) of x[m, :dim] (two-pass, like omnitok_layernorm).
 * bounds (optional, [n_clips][2] floats zeroed by the caller, clip of row m = m / rows_per_clip, a single
 * clip if rows_per_clip <= 0) receive, by atomic max over the rows of each clip, rigorous upper bounds of
 * max|x| (|mean| + sqrt(dim)/rstd) and of the row norm ||x||_2: the range information omnitok_gemm_h2
 * needs for operands derived from x, kept per clip so that results never depend on the rest of the batch. */
int omnitok_row_stats(const float *x, int64_t rows, int dim, float eps, float *stats, float *bounds,
                      int64_t rows_per_clip, omnitok_stream_t stream);
/* out2[0] = max_rows ||w_row||_2, out2[1] = max|w| (atomic max into floats zeroed by the caller). */
int omnitok_weight_range(const float *w, int64_t ldw, int rows, int K, float *out2, omnitok_stream_t stream);

/* Packs FeedForward's first Linear weight w1[2*inner, K] (value rows [0,inner), gate rows
 * [inner, 2*inner), reference attention.py:153-156,164) into out[2*inner_pad, K]: 32-row blocks
 * alternate value / gate, zero rows beyond inner.  inner_pad % 64 == 0. */
int omnitok_pack_geglu_weight(const float *w1, int inner, int K, int inner_pad, float *out,
                              omnitok_stream_t stream);

/* Patch gather + first LayerNorm of the patch embedding (reference omnitokenizer.py:806-809 /
 * 814-818): video[B,C,F,H,W] frames [f0, f0 + t*pt) -> out[B*t*(H/p)*(W/p), C*pt*p*p] with
 * feature order (c, pt, p1, p2), each row layer-normalised with gamma/beta.  gamma == beta ==
 * NULL: no normalisation, i.e. the im2col rows of the 'cnn' patch-embed's Conv3d with kernel ==
 * stride (omnitokenizer.py:823-836).  ldo: output row stride in floats (0 = dense); columns
 * [C*pt*p*p, ldo) are zero-filled so that the GEMM's K % 32 rule can be met by padding. */
int omnitok_patchify_ln(const float *video, int B, int C, int F, int H, int W, int f0, int t,
                        int pt, int p, const float *gamma, const float *beta, float eps,
                        float *out, int64_t ldo, omnitok_stream_t stream);

/* Token-grid resampling of the pooling blocks / deferred pools (csrc/resample.hip). mode:
 *  0 avg2d, 1 max2d: x[n,gh,gw,D] -> out[n,gh/2,gw/2,D]   (Pooling 'a'/'m', attention.py:83-106)
 *  2 up2d:           x[n,gh,gw,D] -> out[n,2gh,2gw,D]     (nearest, omnitokenizer.py:1001)
 *  3 avg_t:          x[n,T,gh*gw,D] -> out[n,1+(T-1)/2,gh*gw,D], frame 0 kept (omnitokenizer.py:909-914)
 *  4 up_t:           x[n,T,gh*gw,D] -> out[n,1+(T-1)*2,gh*gw,D], frame 0 kept (omnitokenizer.py:1103-1107) */
int omnitok_token_resample(const float *x, float *out, int mode, int64_t n, int T, int gh, int gw,
                           int D, omnitok_stream_t stream);

/* Inverse rearrange of to_pixels (reference omnitokenizer.py:1008-1009, 1015-1016):
 * tok[B*t*(H/p)*(W/p), C*pt*p*p] -> video[B,C,F,H,W] frames [f0, f0 + t*pt). */
int omnitok_unpatchify(const float *tok, int B, int C, int F, int H, int W, int f0, int t, int pt,
                       int p, float *video, omnitok_stream_t stream);

/* y = depthwise_conv3d_3x3x3(zero-pad(x viewed as [B,T,H,W,D])) + bias + x
 * reference attention.py:298-338 (PEG) with the caller's residual (attention.py:667) fused.
 * x is the raw contiguous token buffer whatever order it was written in.
 * w27 is the conv weight [D,1,3,3,3] repacked to [27, D] by omnitok_pack_peg_weight.
 * causal: frame padding (2,0) else (1,1).  D % 4 == 0.  y must not alias x. */
int omnitok_peg3d(const float *x, const float *w27, const float *bias, float *y, int B, int T,
                  int H, int W, int D, int causal, omnitok_stream_t stream);
int omnitok_pack_peg_weight(const float *w, int D, float *w27, omnitok_stream_t stream);

/* [B, A, C, D] -> [B, C, A, D] token-order transpose ('(b t)(h w) d' <-> '(b h w) t d',
 * reference omnitokenizer.py:891-907, 1072-1084). D % 4 == 0. */
int omnitok_transpose_tokens(const float *x, float *y, int64_t B, int64_t A, int64_t C, int D,
                             omnitok_stream_t stream);

/* cos/sin[N, dim_head/2] of the 2-D rotary table, reference attention.py:28-43. Host-computed
 * into host_cos/host_sin (fp32). */
int omnitok_rope_table(int n_tokens, int dim_head, float theta, float *host_cos, float *host_sin);

/* In-place q/k preparation of Attention (reference attention.py:417-437): optional 2-D RoPE
 * (cos/sin [N, 32], same table for all heads), l2norm over dim_head=64, * q_scale / k_scale[64];
 * q additionally * scale (the SDPA scale 8, a power of two, folded exactly).
 * q: rows x heads*64 (ld ldq), k: rows x heads*64 (ld ldk). cos may be NULL (no RoPE). */
int omnitok_qk_prep(float *q, int64_t ldq, float *k, int64_t ldk, int64_t rows, int n_tokens,
                    int heads, const float *cos, const float *sin, const float *q_scale,
                    const float *k_scale, float scale, omnitok_stream_t stream);

/* Full (non-causal) spatial attention over N tokens per sequence, flash-style on fp32 MFMA:
 * out[b, n, h*64:(h+1)*64] = softmax(q k^T [+ bias]) v  with q,k already prepared by qk_prep.
 * q[Bn*N, heads*64] (ldq), k/v rows with ld ldkv. bias_table: NULL (sdpa mode) or
 * [(2*gh-1)*(2*gw-1), heads] ContinuousPositionBias table (entry (dy+gh-1, dx+gw-1) for query-key
 * offset (dy,dx)) for legacy mode (attention.py:456-466), N == gh*gw.  N % 64 == 0. */
int omnitok_attn_spatial(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv,
                         float *out, int64_t ldo, int Bn, int N, int heads,
                         const float *bias_table, int gh, int gw, omnitok_stream_t stream);

/* fp16-split spatial attention (csrc/attn_h2.hip), the default of the engine ("attn_mode" 1): the same
 * operator as omnitok_qk_prep + omnitok_attn_spatial (reference attention.py:417-483) on the fp16 matrix
 * cores, every operand as an exact-scale fp16 hi|lo pair, three MFMA products per fp32 product, fp32
 * accumulation and softmax.
 *   omnitok_attn_pack: RoPE + l2norm + q/k scales exactly as omnitok_qk_prep, then q, k and v are written as
 *     hi|lo planes in MFMA-fragment order: qp / kp / vp are rows*heads*64*4 bytes each (blocks of 32 tokens of
 *     one head, 8 KiB).  q_bound >= max|q| (= scale * max|q_scale|), k_bound >= max|k| (= max|k_scale|);
 *     v_bound (times v_bound_dev[v_bound_stride * clip] if given, clip = row / rows_per_clip) >= max|v| of the
 *     rows of a clip.  n_tokens % 32 == 0.  v == vp == NULL: only q and k (V written by omnitok_gemm_h2_vpack).
 *   omnitok_attn_spatial_h2: out[Bn*N, heads*64] = softmax(q k^T [+ bias]) v from the packed operands; the same
 *     bounds must be passed (they define the power-of-two operand scales); seq_per_clip = sequences per clip
 *     for v_bound_dev.  N % 64 == 0. */
int omnitok_attn_pack(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv, int64_t rows,
                      int n_tokens, int heads, const float *cos, const float *sin, const float *q_scale,
                      const float *k_scale, float scale, float q_bound, float k_bound, float v_bound,
                      const float *v_bound_dev, int v_bound_stride, int64_t rows_per_clip, void *qp, void *kp,
                      void *vp, omnitok_stream_t stream);
int omnitok_attn_spatial_h2(const void *qp, const void *kp, const void *vp, float *out, int64_t ldo, int Bn, int N,
                            int heads, float q_bound, float k_bound, float v_bound, const float *v_bound_dev,
                            int v_bound_stride, int seq_per_clip, const float *bias_table, int gh, int gw,
                            omnitok_stream_t stream);
/* The same kernel with the output written as fp16 hi|lo planes (out_planes != NULL; the A operand of the to_out GEMM,
 * omnitok_gemm_pl, K = heads * 64) instead of fp32 rows: scaled per clip by the power of two of the V bound (the
 * output is a convex combination of V rows), out_scale[row] receives the factor that undoes it. */
int omnitok_attn_spatial_h2_planes(const void *qp, const void *kp, const void *vp, float *out, int64_t ldo,
                                   void *out_planes, float *out_scale, int Bn, int N, int heads, float q_bound,
                                   float k_bound, float v_bound, const float *v_bound_dev, int v_bound_stride,
                                   int seq_per_clip, const float *bias_table, int gh, int gw, omnitok_stream_t stream);

/* WindowAttention core (reference attention.py:266-286): qkv[Bn*N, 3*heads*64] from LN(x),
 * ws x ws (ws = 8) non-overlapping windows of the gh x gw grid, softmax(0.125 q k^T + bias) v.
 * bias_dense[heads, 64(kv), 64(q)] = relative_position_bias_table gathered by
 * relative_position_index (attention.py:277-281), transposed. out[Bn*N, heads*64]. */
int omnitok_attn_window(const float *qkv, int64_t ldqkv, const float *bias_dense, float *out,
                        int64_t ldo, int Bn, int gh, int gw, int heads, omnitok_stream_t stream);
/* ... with the output as fp16 hi|lo planes (out_planes != NULL) scaled by the power of two of the static bound
 * out_bound >= max |output|; the consuming omnitok_gemm_pl takes a_scale_const = omnitok_pl_unscale(out_bound). */
int omnitok_attn_window_planes(const float *qkv, int64_t ldqkv, const float *bias_dense, float *out, int64_t ldo,
                               void *out_planes, float out_bound, int Bn, int gh, int gw, int heads,
                               omnitok_stream_t stream);
/* Window attention on the fp16 matrix cores from PACKED operands (csrc/attn_h2.hip; reference attention.py:254-293): qp / kp / vp
 * are the fragment-order hi|lo planes the q|k|v plane GEMM's packing epilogues (omnitok_pl_gemm epilogue 4 without scale
 * vectors, q_mul = the head scale; epilogue 3) write when the GEMM's rows are in window-major order
 * (omnitok_stats_pack_windows) with n_tokens = 64; q_bound / k_bound / v_bound the static bounds they were scaled with.
 * bias_dense: [heads][64 keys][64 queries].  Output rows in token order: fp32 `out` [Bn * gh * gw, ldo] or, when out_planes
 * is given, hi|lo planes (K = heads * 64) scaled by the power of two of v_bound. */
int omnitok_attn_window_h2(const void *qp, const void *kp, const void *vp, const float *bias_dense, float *out, int64_t ldo,
                           void *out_planes, float q_bound, float k_bound, float v_bound, int Bn, int gh, int gw, int heads,
                           omnitok_stream_t stream);

/* Temporal attention (reference attention.py:402-486 with is_spatial=False): per (column, head)
 * T tokens; l2norm, q/k scales and scale applied inside. causal: is_causal / causal mask;
 * alibi_slopes: NULL (sdpa) or [heads] (legacy, attention.py:473-474). q[cols*T, heads*64]. */
int omnitok_attn_temporal(const float *q, int64_t ldq, const float *k, const float *v,
                          int64_t ldkv, float *out, int64_t ldo, int64_t cols, int T, int heads,
                          const float *q_scale, const float *k_scale, float scale, int causal,
                          const float *alibi_slopes, omnitok_stream_t stream);
/* ... with the output as fp16 hi|lo planes (out_planes != NULL; T <= 17, cols % 16 == 0), scaled per clip by the power
 * of two of v_bound (x v_bound_dev[v_bound_stride * (col / cols_per_clip)]) >= max |v|; out_scale[row] = the inverse. */
int omnitok_attn_temporal_planes(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv,
                                 float *out, int64_t ldo, void *out_planes, float *out_scale, float v_bound,
                                 const float *v_bound_dev, int v_bound_stride, int64_t cols_per_clip, int64_t cols,
                                 int T, int heads, const float *q_scale, const float *k_scale, float scale,
                                 int causal, const float *alibi_slopes, omnitok_stream_t stream);

/* z[n, 0:cdim] = l2norm(x[n, :] . w[cdim, D]^T + b)   (reference omnitokenizer.py:143-148 +
 * F.normalize :251-252; normalize skipped if l2 == 0).  cdim == 8, D % 64 == 0. */
int omnitok_pre_vq(const float *x, const float *w, const float *b, float *z, int64_t n, int D,
                   int cdim, int l2, omnitok_stream_t stream);

/* Codebook preparation: packed[n_codes*8] = MFMA A-fragment order of E[n_codes, 8];
 * ee[c] = sum_k E[c,k]^2 with sequential separately-rounded adds. */
int omnitok_vq_prepare(const float *codebook, int n_codes, int cdim, float *packed, float *ee,
                       omnitok_stream_t stream);

/* ids[n] = argmin_c ( (sum_k z[n,k]^2 - sum_k (2 z[n,k]) E[c,k]) + ee[c] ), first minimum.
 * Bit-exact restatement of reference modules/codebook.py:82-86 (see oracle/vq_argmin.c): the
 * dot is a k-ordered fp32 FMA chain from 0 on v_mfma_f32_32x32x2_f32, the norms are sequential
 * adds of rounded squares. n_codes % 32 == 0, cdim == 8. ids int64. */
int omnitok_vq_argmin(const float *z, const float *packed, const float *ee, int64_t n, int n_codes,
                      int64_t *ids, omnitok_stream_t stream);
/* The same ids, bit for bit, from a SCREENED search (csrc/vq.hip "vq_screen_kernel"): one fp16 MFMA per 32 codes x 32 rows
 * evaluates a coarse distance with a rigorous error bound, the codes within that bound of the coarse minimum (almost always
 * one tile) are re-evaluated with the exact k-ordered fp32 chain of omnitok_vq_argmin, first minimum kept.  Inputs outside
 * fp16's range, non-finite inputs and codebooks with more than four candidate tiles per row take the exact evaluation for
 * every tile.  screen[n_codes * 8 + 4] floats from omnitok_vq_screen_prepare(E, ee) (ee from omnitok_vq_prepare). */
int omnitok_vq_screen_prepare(const float *codebook, const float *ee, int n_codes, int cdim, float *screen,
                              omnitok_stream_t stream);
int omnitok_vq_argmin_screened(const float *z, const float *packed, const float *ee, const float *screen, int64_t n,
                               int n_codes, int64_t *ids, omnitok_stream_t stream);

/* --use_external_codebook (cosine similarity): ids[n] = first argmax_c sum_k z[n,k] E[c,k]  (reference
 * quantizer/vector_quantize_pytorch.py:646-650: einsum + argmax; z and E rows unit-norm).  Same kernel
 * and the same k-ordered FMA chain as omnitok_vq_argmin; packed from omnitok_vq_prepare(E). */
int omnitok_vq_argmax_cos(const float *z, const float *packed, int64_t n, int n_codes, int64_t *ids,
                          omnitok_stream_t stream);
/* --use_external_codebook without l2_code (EuclideanCodebook, vector_quantize_pytorch.py:29-33, 463):
 * ids[n] = first argmin_c sqrt(clamp((xx[n] + ee[c]) + (-2 dot[n,c]), 0)), i.e. the first argmax of -cdist
 * with the correctly rounded sqrt taking part in the ordering.  packed / ee from omnitok_vq_prepare(E). */
int omnitok_vq_argmin_cdist(const float *z, const float *packed, const float *ee, int64_t n,
                            int n_codes, int64_t *ids, omnitok_stream_t stream);

/* tok[n, :] = E[ids[n], :] . w[D, cdim]^T + b  (F.embedding + post_vq_conv, reference
 * omnitokenizer.py:270, 156-160). Returns OMNITOK_ERR_INVALID through the status word
 * err_flag[0] != 0 if an id is out of range (checked on device, reported at the next sync). */
int omnitok_dequant_post_vq(const int64_t *ids, const float *codebook, int n_codes, int cdim,
                            const float *w, const float *b, float *tok, int64_t n, int D,
                            int *err_flag, omnitok_stream_t stream);

/* The same operator as a table: omnitok_dequant_table builds table[n_codes, D] = E w^T + b with the kernel of
 * omnitok_dequant_post_vq on ids 0..n_codes-1 (rows bit-identical; scratch_ids: n_codes int64 of device scratch),
 * omnitok_gather_rows then is tok[n, :] = table[ids[n], :] with the same out-of-range check.  The engine builds
 * the table once at finalize and decodes by gather. */
int omnitok_dequant_table(const float *codebook, int n_codes, int cdim, const float *w, const float *b,
                          float *table, int D, int64_t *scratch_ids, omnitok_stream_t stream);
int omnitok_gather_rows(const int64_t *ids, const float *table, int n_codes, float *tok, int64_t n, int D,
                        int *err_flag, omnitok_stream_t stream);
/* ... with the token transpose '(b a c) d -> (b c a) d' fused into the store (n = b*a*c rows; a = c = 0: none):
 * the reference's rearranges between its spatial and temporal stages (omnitokenizer.py:1072, 1081). */
int omnitok_gather_rows_transposed(const int64_t *ids, const float *table, int n_codes, float *tok, int64_t n,
                                   int a, int c, int D, int *err_flag, omnitok_stream_t stream);

/* --use_vae posterior sample (reference modules/vae.py:4-17 on top of pre_vq_conv with 2*cdim
 * outputs): h = x[n,:] . w[2*cdim, D]^T + b, n = B*thw rows in (b, thw) order;
 * z[B,cdim,thw] = h[:, :cdim] + exp(0.5*clamp(h[:, cdim:], -30, 20)) * noise[B,cdim,thw]
 * (noise NULL: z = mean).  moments (optional) [B,2*cdim,thw] = h.  cdim == 8, D % 64 == 0. */
int omnitok_vae_sample(const float *x, const float *w, const float *b, const float *noise,
                       float *z, float *moments, int64_t B, int64_t thw, int D, int cdim,
                       omnitok_stream_t stream);

/* tok[n, :] = z[n, :] . w[D, cdim]^T + b for continuous latents (post_vq_conv, reference
 * omnitokenizer.py:156-160, 305/316).  channel_first: z[B,cdim,thw], else z[B*thw, cdim]. */
int omnitok_post_vq(const float *z, int channel_first, int64_t B, int64_t thw, int cdim,
                    const float *w, const float *b, float *tok, int D, omnitok_stream_t stream);

/* emb[b, c, t, h, w] = (E[ids] - z) + z  (straight-through value, codebook.py:120), from
 * z[b,t,h,w,c] / ids[b,t,h,w]; thw = t*h*w. */
int omnitok_vq_embed_st(const int64_t *ids, const float *z, const float *codebook, int cdim,
                        int64_t B, int64_t thw, float *emb, omnitok_stream_t stream);

/* Codebook statistics that Codebook.forward returns next to the ids (reference
 * modules/codebook.py:54-72, 122-140; read by vqgan_eval.py:152,195 through forward(log_image=True)):
 * batch_usage[c] = fraction of the n ids equal to c; out2 = {perplexity, avg_usage};
 * codebook_usage (the module's EMA buffer) is updated in place: = batch_usage on the first call,
 * usage_sigma * old + (1 - usage_sigma) * batch_usage afterwards. counts_scratch: n_codes ints. */
int omnitok_vq_stats(const int64_t *ids, int64_t n, int n_codes, int *counts_scratch,
                     float *batch_usage, float *codebook_usage, int first_call, float usage_sigma,
                     float *out2, omnitok_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Engine: the whole encode()/decode() path behind one handle.
 * ------------------------------------------------------------------------------------------ */

typedef struct omnitok_engine omnitok_engine;

typedef struct omnitok_config {
    int resolution;            /* args.resolution (used for flat video ids, omnitokenizer.py:284) */
    int image_channels;        /* 3 */
    int patch_size;            /* 8 */
    int temporal_patch_size;   /* 4 (stage 2) / 2 (stage 1) */
    int dim;                   /* 512 */
    int heads;                 /* 8 */
    int dim_head;              /* 64 */
    int ff_inner;              /* int(ff_mult * 2/3 * dim) = 1365 */
    int window_size;           /* 8 */
    int n_codes;               /* 8192 */
    int codebook_dim;          /* 8 */
    int l2_code;               /* 1 */
    int spatial_rope;          /* 1: spatial_pos == "rope", 0: "rel" */
    int legacy_attention;      /* 0: SDPA branch, 1: legacy einsum branch (attention.py:439) */
    int causal_temporal;       /* --causal_in_temporal_transformer */
    int causal_peg;            /* --causal_in_peg */
    int temporal_depth;        /* 4 */
    char enc_block[16];        /* "ttww" */
    char dec_block[16];        /* "tttt" */
    int use_vae;               /* --use_vae: pre_vq emits mean|logvar, no quantiser (omnitokenizer.py:143-154) */
    /* config coverage beyond the released checkpoints (all 0 for them) */
    int patch_embed_cnn;       /* --patch_embed cnn: Conv3d / ConvTranspose3d (kernel == stride) + eval-mode
                                * SyncBatchNorm instead of LN-Linear-LN / Linear (omnitokenizer.py:823-836,
                                * 1019-1033); enc_block may also hold 'a' 'm' 'l' pooling blocks
                                * (attention.py:83-113) */
    int defer_temporal_pool;   /* --defer_temporal_pool (omnitokenizer.py:792-797, 985-990), linear only */
    int defer_spatial_pool;    /* --defer_spatial_pool  (omnitokenizer.py:799-804, 992-1003), linear only */
    int gen_upscale;           /* --gen_upscale: decoder patch_size *= gen_upscale (0 / 1 = off) */
    int external_codebook;     /* --use_external_codebook: VectorQuantize with its CosineSimCodebook (l2_code) or
                                * EuclideanCodebook (no l2_code)
                                * (quantizer/vector_quantize_pytorch.py): weights codebook.project_in / project_out /
                                * _codebook.embed replace pre_vq_conv / post_vq_conv / codebook.embeddings;
                                * omnitok_encode's emb_out is then project_out(embed[ids]) as [B,T',h,w,dim] */
} omnitok_config;

int omnitok_engine_create(const omnitok_config *cfg, omnitok_engine **out);
void omnitok_engine_destroy(omnitok_engine *e);

/* Copies one reference state_dict tensor (device fp32, or int64 for relative_position_index)
 * into the engine under its reference key name, e.g.
 * "encoder.enc_spatial_transformer.layers.0.1.to_q.weight". Unknown / off-path keys return 1
 * (ignored), like load_state_dict(strict=False). */
int omnitok_engine_set_weight(omnitok_engine *e, const char *name, const void *dev_ptr,
                              const int64_t *shape, int ndim, int is_int64,
                              omnitok_stream_t stream);
/* Repacks weights (GEGLU interleave, PEG [27,D], codebook fragments, bias tables) and checks
 * that every tensor the configuration needs was set. */
int omnitok_engine_finalize(omnitok_engine *e, omnitok_stream_t stream);
/* number of required-but-missing keys; writes a '\n'-separated list into buf */
int omnitok_engine_missing(omnitok_engine *e, char *buf, int buflen);

/* VQGAN.encode: x[B,C,F,H,W] (is_image: F == 1, i.e. [B,C,H,W]) -> ids[B,T',H/p,W/p] int64,
 * T' = 1 + (F-1)/pt. emb_out (optional, may be NULL): [B,cdim,T',h,w]. z_out (optional):
 * pre-VQ latents [B,T',h,w,cdim]. */
int omnitok_encode(omnitok_engine *e, const float *x, int B, int F, int H, int W,
                   int64_t *ids_out, float *emb_out, float *z_out, omnitok_stream_t stream);

/* VQGAN.decode: ids[B,T',h,w] -> pixels[B,C,F,h*p,w*p], F = 1 + (T'-1)*pt. */
int omnitok_decode(omnitok_engine *e, const int64_t *ids, int B, int T, int h, int w,
                   float *pixels_out, omnitok_stream_t stream);

/* Shapes of the path for this configuration (pooling blocks, deferred pools and gen_upscale change
 * them): encode_shape: pixels F,H,W -> latent T,h,w; decode_shape: latent T,h,w -> pixels F,H,W.
 * Return OMNITOK_ERR_INVALID (with omnitok_last_error) for sizes the path rejects. */
int omnitok_engine_encode_shape(omnitok_engine *e, int F, int H, int W, int *T, int *h, int *w);
int omnitok_engine_decode_shape(omnitok_engine *e, int T, int h, int w, int *F, int *H, int *W);

/* --use_vae variants (reference omnitokenizer.py:260-266 encode, :293-317 decode; the KL-regularised
 * tokenizer DiT/Latte train on, Diffusion/Latte/train.py:216, sample_ddp.py:201-203).  The engine
 * must have been created with cfg.use_vae = 1 (pre_vq_conv.1.weight is [2*cdim, dim]);
 * omnitok_encode/omnitok_decode then return OMNITOK_ERR_STATE and vice versa.
 *
 * encode_vae: x[B,C,F,H,W] -> z_out[B,cdim,T',h,w] = mean + exp(0.5*clamp(logvar,-30,20)) * noise
 * (modules/vae.py:4-17).  noise: [B,cdim,T',h,w] standard normal supplied by the caller (the
 * reference draws it with torch.randn on the host), or NULL for the posterior mode (z = mean).
 * moments_out (optional): the raw pre_vq output [B,2*cdim,T',h,w] (mean | unclamped logvar), what
 * DiagonalGaussianDistribution.kl() consumes. */
int omnitok_encode_vae(omnitok_engine *e, const float *x, int B, int F, int H, int W,
                       const float *noise, float *z_out, float *moments_out,
                       omnitok_stream_t stream);
/* decode_vae: z -> pixels[B,C,F,h*p,w*p].  channel_first = 1: z[B,cdim,T',h,w] (the image
 * branch's 'b c h w', omnitokenizer.py:303-304, and encode_vae's output); channel_first = 0:
 * z[B,T',h,w,cdim] (the video branch's 'b t h w c' and both flat '(t h w) c' forms). */
int omnitok_decode_vae(omnitok_engine *e, const float *z, int channel_first, int B, int T, int h,
                       int w, float *pixels_out, omnitok_stream_t stream);

/* decode() checks ids against [0, n_codes) on the device; this reads the flag back
 * (synchronises the stream) and returns OMNITOK_ERR_INVALID if any id was out of range --
 * the reference raises IndexError from F.embedding (omnitokenizer.py:270). */
int omnitok_engine_check_ids(omnitok_engine *e, omnitok_stream_t stream);

int64_t omnitok_engine_workspace_bytes(omnitok_engine *e);

/* Caller-owned workspace (e.g. a tensor of PyTorch's caching allocator) instead of the engine's own grow-only
 * hipMalloc buffers: omnitok_engine_workspace_need_encode / _decode return the bytes a call of that shape needs
 * (-1: invalid shape); omnitok_engine_set_workspace hands over a 256-byte-aligned device block that must stay
 * alive until it is replaced (NULL, 0: back to internal allocation).  encode / decode fail with OMNITOK_ERR_STATE
 * if the block is too small. */
int64_t omnitok_engine_workspace_need_encode(omnitok_engine *e, int B, int F, int H, int W);
int64_t omnitok_engine_workspace_need_decode(omnitok_engine *e, int B, int T, int h, int w);
int omnitok_engine_set_workspace(omnitok_engine *e, void *dev_ptr, int64_t bytes);

/* Per-kernel timing: when enabled, every launch made by encode/decode is bracketed with HIP
 * events on its stream; omnitok_engine_timing_report fills a '\n'-separated
 * "name calls total_ms" list (synchronises). */
/* Per-engine value of "gemm_mode" | "attn_mode" | "attn_vpack" | "gemm_pl" (value -1: follow the process default again). */
int omnitok_engine_set_option(omnitok_engine *e, const char *name, int value);
int omnitok_engine_set_timing(omnitok_engine *e, int enabled);
int omnitok_engine_timing_report(omnitok_engine *e, char *buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* OMNITOK_H */
