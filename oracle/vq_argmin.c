/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the nearest-codebook lookup.
 *
 * Follows reference OmniTokenizer/modules/codebook.py:82-86
 *     distances = (x ** 2).sum(dim=1, keepdim=True) - 2 * x @ E.t() + (E.t() ** 2).sum(dim=0, keepdim=True)
 *     encoding_indices = torch.argmin(distances, dim=1)
 * with the fp32 evaluation order ATen/MKL use on the CPU path (re-verified
 * against the reference's own Codebook.forward in tests/test_oracle_vs_reference.py and pinned by
 * tests/golden/vq_*.npz):
 *   xx[n]   = sequential k-ordered sum of separately rounded squares  (no FMA)
 *   ee[c]   = same over the code row
 *   dot[n,c]= k-ordered fp32 FMA chain starting from 0 over (2*x[n,k]) * E[c,k]
 *   d[n,c]  = (xx[n] - dot[n,c]) + ee[c]
 *   ids[n]  = first index attaining the minimum
 * Build with -ffp-contract=off so the compiler forms no FMAs of its own (oracle/Makefile).
 * The HIP kernel omnitokenizer_amd/csrc/vq.hip must reproduce these bits exactly.
 */
#include <math.h>
#include <stdint.h>

void oracle_vq_argmin(const float *x, const float *E, int64_t n, int64_t n_codes, int64_t dim,
                      int64_t *ids, float *min_dist /* may be NULL */)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *xi = x + i * dim;
        float xx = 0.0f;
        for (int64_t k = 0; k < dim; ++k) {
            float sq = xi[k] * xi[k];
            xx = xx + sq;
        }
        float best = INFINITY;
        int64_t best_c = 0;
        for (int64_t c = 0; c < n_codes; ++c) {
            const float *e = E + c * dim;
            float ee = 0.0f, dot = 0.0f;
            for (int64_t k = 0; k < dim; ++k) {
                float sq = e[k] * e[k];
                ee = ee + sq;
                dot = fmaf(2.0f * xi[k], e[k], dot);
            }
            float d = (xx - dot) + ee;
            if (d < best || c == 0) { best = d; best_c = c; }
        }
        ids[i] = best_c;
        if (min_dist) min_dist[i] = best;
    }
}

/* --use_external_codebook, cosine similarity: reference quantizer/vector_quantize_pytorch.py:646-650
 *     dist = einsum('h n d, h c d -> h n c', flatten, embed);  ids = dist.argmax(-1)
 * dot[n,c] = k-ordered fp32 FMA chain from 0 (the same sgemm evaluation order as above), first maximum. */
void oracle_vq_argmax_cos(const float *x, const float *E, int64_t n, int64_t n_codes, int64_t dim, int64_t *ids)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *xi = x + i * dim;
        float best = -INFINITY;
        int64_t best_c = 0;
        for (int64_t c = 0; c < n_codes; ++c) {
            const float *e = E + c * dim;
            float dot = 0.0f;
            for (int64_t k = 0; k < dim; ++k) dot = fmaf(xi[k], e[k], dot);
            if (dot > best || c == 0) { best = dot; best_c = c; }
        }
        ids[i] = best_c;
    }
}

/* --use_external_codebook without l2_code, Euclidean: reference quantizer/vector_quantize_pytorch.py:29-33, 463
 *     cdist = sqrt(clamp((x2 + y2) + (-2 * dot), 0));  ids = (-cdist).argmax(-1)
 * x2, y2 sequential sums of rounded squares, dot the k-ordered FMA chain, sqrtf correctly rounded; the
 * first maximum of -cdist is the first minimum of cdist. */
void oracle_vq_argmin_cdist(const float *x, const float *E, int64_t n, int64_t n_codes, int64_t dim, int64_t *ids)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *xi = x + i * dim;
        float xx = 0.0f;
        for (int64_t k = 0; k < dim; ++k) { float sq = xi[k] * xi[k]; xx = xx + sq; }
        float best = INFINITY;
        int64_t best_c = 0;
        for (int64_t c = 0; c < n_codes; ++c) {
            const float *e = E + c * dim;
            float ee = 0.0f, dot = 0.0f;
            for (int64_t k = 0; k < dim; ++k) {
                float sq = e[k] * e[k];
                ee = ee + sq;
                dot = fmaf(xi[k], e[k], dot);
            }
            float d2 = (xx + ee) + (dot * -2.0f);
            float d = sqrtf(d2 > 0.0f ? d2 : 0.0f);
            if (d < best || c == 0) { best = d; best_c = c; }
        }
        ids[i] = best_c;
    }
}
