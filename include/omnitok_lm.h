/* omnitok_lm.h -- C ABI of the autoregressive LM consumer of the token path.
 *
 * The reference samples video/image tokens with a minGPT (OmniTokenizer/modules/gpt.py:170-275)
 * through `sample_with_past` (gpt.py:327-359): one token per step, the K/V of every layer kept and
 * re-concatenated on every step (torch.cat(past, dim=-2), gpt.py:244).  This library keeps a
 * preallocated K/V cache in HBM and runs one decode step as a fixed sequence of kernels whose
 * shapes do not depend on the position, so the caller can capture the step in a HIP graph.
 * omnitokenizer_amd/gpt.py is the ctypes binding that mirrors the reference's `GPT` class
 * (forward / forward_with_past) and its sampling loops.
 *
 * Same conventions as omnitok.h: device pointers, fp32 (ids int64, positions int32), row-major,
 * every launch on the given stream, 0 = OK, omnitok_last_error() for the message.
 */
#ifndef OMNITOK_LM_H
#define OMNITOK_LM_H

#include "omnitok.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct omnitok_lm omnitok_lm;

typedef struct omnitok_lm_config {
    int vocab_size; /* gpt_vocab_size = first_stage + cond_stage (+1), lm_transformer.py:60-62 */
    int block_size; /* positions, 5120 + conditioning */
    int n_layer;    /* 24 */
    int n_head;     /* 16 */
    int n_embd;     /* 1536; head_dim = n_embd / n_head must be 64, 96 or 128 */
} omnitok_lm_config;

int omnitok_lm_create(const omnitok_lm_config *cfg, omnitok_lm **out);
void omnitok_lm_destroy(omnitok_lm *lm);
/* reference state_dict keys of modules/gpt.py GPT: pos_emb [1,block,C], tok_emb.weight [V,C],
 * blocks.{i}.ln1|ln2.{weight,bias}, blocks.{i}.attn.{key,query,value,proj}.{weight,bias},
 * blocks.{i}.mlp.{0,2}.{weight,bias}, ln_f.{weight,bias}, head.weight [V,C].  Unknown keys
 * (blocks.{i}.attn.mask) return 1 = ignored. */
int omnitok_lm_set_weight(omnitok_lm *lm, const char *name, const void *dev_ptr,
                          const int64_t *shape, int ndim, omnitok_stream_t stream);
int omnitok_lm_finalize(omnitok_lm *lm, omnitok_stream_t stream);
/* K/V cache [n_layer][2][max_batch][n_head][max_len][head_dim] fp32 + step workspaces. */
int omnitok_lm_alloc_cache(omnitok_lm *lm, int max_batch, int max_len);
int64_t omnitok_lm_cache_bytes(omnitok_lm *lm);

/* One decode step for B independent streams (GPT.forward_with_past with one new token per row,
 * gpt.py:236-275): token idx[b] enters at position embedding pos[b]; its K/V are appended to row
 * b of the cache at index cache_len[b]; it attends to cache entries [0, cache_len[b]] (all of the
 * past plus itself, gpt.py:125).  logits_out[B, vocab] (may be NULL: prefill steps whose logits
 * are not needed skip ln_f + head).  If advance != 0, cache_len[b] and pos[b] are incremented on
 * the device at the end of the step, so that the same captured graph can be replayed.
 * idx int64[B], pos / cache_len int32[B], all on the device. */
int omnitok_lm_step(omnitok_lm *lm, const int64_t *idx, int32_t *pos, int32_t *cache_len, int B,
                    float *logits_out, int advance, omnitok_stream_t stream);

/* The same step with the reference's optional inputs (gpt.py:236-258): emb [B, C] != NULL replaces the token
 * embedding (an explicit `embeddings=` vector; idx may then be NULL), pos_extra [B, C] != NULL is added to the
 * position embedding (the vtokens_pos term: rows of vtokens_pos_emb gathered by the caller from cbox). */
int omnitok_lm_step_ex(omnitok_lm *lm, const int64_t *idx, const float *emb, const float *pos_extra,
                       int32_t *pos, int32_t *cache_len, int B, float *logits_out, int advance,
                       omnitok_stream_t stream);

/* 1 if a decode step since the last call found cache_len[b] >= max_len (a stream stepped past the cache that
 * omnitok_lm_alloc_cache sized; the step then stays inside the stream's own K/V slab and its logits are
 * invalid), 0 otherwise; clears the flag.  Synchronises the stream (call it once after a sampling loop). */
int omnitok_lm_overflowed(omnitok_lm *lm, omnitok_stream_t stream);

/* Batched prefill of a conditioning prefix into EMPTY streams (GPT.forward / the first
 * forward_with_past call of the reference, gpt.py:207-275, positions 0..T-1): idx[B, T] int64.
 * Same arithmetic as T decode steps, executed as [B*T]-row fp32-MFMA GEMMs (omnitok_gemm) + causal
 * flash attention over the cache.  Fills the K/V cache, sets pos[b] = cache_len[b] = T on the
 * device.  logits_out (optional) [B, T, vocab]: teacher-forced logits of every position.
 * B * T <= 65535; may hipMalloc its (grow-only) workspace. */
int omnitok_lm_prefill(omnitok_lm *lm, const int64_t *idx, int32_t *pos, int32_t *cache_len, int B,
                       int T, float *logits_out, omnitok_stream_t stream);

/* ... with explicit embeddings prepended (GPT.forward(idx, embeddings=...), gpt.py:214-216): the sequence is
 * emb[B, T_emb, C] followed by tok_emb[idx[B, T_tok]], T = T_emb + T_tok positions; pos_extra [B, T, C] (optional)
 * is the vtokens_pos term added to the position embeddings (gpt.py:222-226).  logits_out [B, T, vocab]. */
int omnitok_lm_prefill_ex(omnitok_lm *lm, const int64_t *idx, int T_tok, const float *emb, int T_emb,
                          const float *pos_extra, int32_t *pos, int32_t *cache_len, int B,
                          float *logits_out, omnitok_stream_t stream);

/* Token selection of the sampling loops (gpt.py:347-357; CFG blend :428-431) for B streams, one workgroup each:
 *   v = logits / temperature,  or with logits_uncond:  v = cfg_c1 * (logits / T) - cfg_c2 * (logits_uncond / T)
 *   (cfg_c1 = 1 + t, cfg_c2 = t as fp32, the reference's roundings);
 *   top_k < 0: no filtering (the reference's top_k=None);  top_k >= 0: keep v >= k-th largest (0: all), then the
 *   nucleus top_p (1.0: off) as top_k_top_p_filtering (gpt.py:19-51);
 *   sample == 0: out[b] = argmax (lowest index on ties);  else one draw from softmax of the survivors by inverse
 *   CDF with the caller's uniform u[b] in [0, 1).
 * blend_out (optional) [B, V] receives v.  err_flag (optional) is set when more than 16384 values survive top_k
 * while a nucleus cut is requested (the LDS sort buffer; the argmax is returned then). */
int omnitok_lm_select(const float *logits, const float *logits_uncond, int B, int V, float temperature,
                      float cfg_c1, float cfg_c2, int top_k, float top_p, int sample, const float *u,
                      int64_t *out, float *blend_out, int *err_flag, omnitok_stream_t stream);

/* building blocks, exported for the parity tests */
/* y[b, n] = act(sum_k x[b, k] * w[n, k] + bias[n]) (+ residual[b, n]);  act: 0 none, 1 exact-erf
 * GELU (nn.GELU(), gpt.py:152).  B <= 16, K % 256 == 0.  If ln_gamma != NULL, x is layer-normalised
 * first (eps 1e-5, ln_gamma / ln_beta [K]): the LayerNorm of gpt.py:159/162/263 fused in.
 * residual may alias y. HBM-bound: streams w once for up to 8 rows of x. */
int omnitok_lm_gemv(const float *x, const float *w, const float *bias, const float *residual,
                    const float *ln_gamma, const float *ln_beta, float *y, int B, int N, int K,
                    int act, omnitok_stream_t stream);
/* Decode attention for one layer: qkv[B, 3*C] = (query | key | value) of the new token; K/V cache of
 * this layer kc / vc [max_batch][n_head][max_len][head_dim]; out[B, C].  scratch: float
 * [B * n_head * ceil(max_len/256) * (2 + head_dim)]. */
int omnitok_lm_attn_decode(const float *qkv, float *kc, float *vc, const int32_t *cache_len, int B,
                           int n_head, int head_dim, int max_len, float *scratch, float *out,
                           omnitok_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNITOK_LM_H */
