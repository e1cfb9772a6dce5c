/*
 * purego_oracle.c — CPU restatement (plain C) of nano-vllm-go's purego/tensor
 * prefill + decode forward path.  See purego_oracle.h for the role of this
 * file (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED for arithmetic).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math  (see oracle/Makefile).
 *   -ffp-contract=off because Go on amd64 (GOAMD64=v1) never fuses x*y+z;
 *   every float32 expression below is rounded to float32 after each
 *   operation exactly as the Go spec requires of float32 arithmetic.
 *
 * Citations are file:line under /root/reference/.
 */
#include "purego_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */

static float* fzeros(int64_t n) {
    if (n <= 0) n = 1;
    return (float*)calloc((size_t)n, sizeof(float));
}

/* ------------------------------------------------------------------------ */
/* ops: purego/tensor/tensor.go                                              */
/* ------------------------------------------------------------------------ */

/* Test-time knobs (NOT part of the restated algorithm).
 * po_set_threads(n): rows i of MatMul are independent in the reference's i-p-j loop, so computing different rows on
 *   different threads leaves every output element's arithmetic (p ascending, fp32, unfused) bit-identical; the
 *   reference itself is single-threaded, and the cpu_baseline leg of bench.py keeps n = 1.
 * po_set_lm_head_last_only(1): ForwardWithCache's LM head (generic_model.go:469) runs on the last row only — the row
 *   GetLogitsForLastToken (generic_model.go:595-604) returns; logits_out is then [1, V].  Rows are independent, so the
 *   returned row is bit-identical to the all-rows computation; it only skips work whose result the caller drops. */
static int g_threads = 1;
static int g_last_only = 0;
void po_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
void po_set_lm_head_last_only(int on) { g_last_only = on != 0; }

/* MatMul, tensor.go:62-88: i-p-j loop, C zero-initialised, C += a*b in fp32. */
void po_matmul(const float* a, const float* b, float* c, int m, int k, int n) {
    memset(c, 0, (size_t)m * (size_t)n * sizeof(float));
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1 && m > 1)
    for (int i = 0; i < m; i++) {
        const float* arow = a + (int64_t)i * k;
        float* crow = c + (int64_t)i * n;
        for (int p = 0; p < k; p++) {
            const float av = arow[p];
            const float* brow = b + (int64_t)p * n;
            for (int j = 0; j < n; j++) {
                float prod = av * brow[j];
                crow[j] = crow[j] + prod;
            }
        }
    }
}

/* Transpose, tensor.go:112-125 */
void po_transpose(const float* t, float* out, int m, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++)
            out[(int64_t)j * m + i] = t[(int64_t)i * n + j];
}

/* Softmax (2-D branch), tensor.go:128-160 */
void po_softmax_rows(const float* x, float* y, int rows, int cols) {
    for (int i = 0; i < rows; i++) {
        const float* xr = x + (int64_t)i * cols;
        float* yr = y + (int64_t)i * cols;
        float maxv = xr[0];
        for (int j = 1; j < cols; j++)
            if (xr[j] > maxv) maxv = xr[j];
        float sum = 0.0f;
        for (int j = 0; j < cols; j++) {
            float d = xr[j] - maxv;
            float v = (float)exp((double)d);
            yr[j] = v;
            sum = sum + v;
        }
        for (int j = 0; j < cols; j++) yr[j] = yr[j] / sum;
    }
}

/* GELU (tanh form), tensor.go:181-190 */
void po_gelu(const float* x, float* y, int64_t n) {
    const double c = sqrt(2.0 / M_PI);
    for (int64_t i = 0; i < n; i++) {
        float xv = x[i];
        float x3 = xv * xv;
        x3 = x3 * xv;
        float t = 0.044715f * x3;
        float s = xv + t;
        double inner = c * (double)s;
        float th = (float)tanh(inner);
        float half_x = 0.5f * xv;
        float one_p = 1.0f + th;
        y[i] = half_x * one_p;
    }
}

/* SiLU, mamba2.go:360-367: sigmoid stays in double until the multiply. */
void po_silu(const float* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        double sg = 1.0 / (1.0 + exp(-(double)x[i]));
        y[i] = x[i] * (float)sg;
    }
}

/* LayerNorm, tensor.go:193-250: bias == NULL selects RMSNorm. */
void po_layernorm(const float* x, const float* w, const float* bias, float eps,
                  float* y, int rows, int hidden) {
    for (int i = 0; i < rows; i++) {
        const float* xr = x + (int64_t)i * hidden;
        float* yr = y + (int64_t)i * hidden;
        if (bias == NULL) {
            float rms = 0.0f;
            for (int j = 0; j < hidden; j++) {
                float v = xr[j];
                float sq = v * v;
                rms = rms + sq;
            }
            float ms = rms / (float)hidden;
            ms = ms + eps;
            rms = (float)sqrt((double)ms);
            for (int j = 0; j < hidden; j++) {
                float nrm = xr[j] / rms;
                yr[j] = nrm * w[j];
            }
        } else {
            float mean = 0.0f;
            for (int j = 0; j < hidden; j++) mean = mean + xr[j];
            mean = mean / (float)hidden;
            float var = 0.0f;
            for (int j = 0; j < hidden; j++) {
                float d = xr[j] - mean;
                float sq = d * d;
                var = var + sq;
            }
            var = var / (float)hidden;
            float ve = var + eps;
            float sd = (float)sqrt((double)ve);
            for (int j = 0; j < hidden; j++) {
                float d = xr[j] - mean;
                float nrm = d / sd;
                float sc = nrm * w[j];
                yr[j] = sc + bias[j];
            }
        }
    }
}

/* ConcatenateLastDim, tensor.go:254-281 */
void po_concat_last_dim(const float* a, const float* b, int rows, int c1, int c2, float* out) {
    int tc = c1 + c2;
    for (int i = 0; i < rows; i++) {
        memcpy(out + (int64_t)i * tc, a + (int64_t)i * c1, (size_t)c1 * sizeof(float));
        memcpy(out + (int64_t)i * tc + c1, b + (int64_t)i * c2, (size_t)c2 * sizeof(float));
    }
}

/* Concatenate(dim=2) of [1, heads, s1, hd] and [1, heads, s2, hd], tensor.go:283-321 */
static float* concat_seq(const float* t1, int s1, const float* t2, int s2, int heads, int hd) {
    float* r = fzeros((int64_t)heads * (s1 + s2) * hd);
    for (int h = 0; h < heads; h++) {
        if (s1 > 0)
            memcpy(r + ((int64_t)h * (s1 + s2)) * hd, t1 + ((int64_t)h * s1) * hd,
                   (size_t)s1 * hd * sizeof(float));
        memcpy(r + ((int64_t)h * (s1 + s2) + s1) * hd, t2 + ((int64_t)h * s2) * hd,
               (size_t)s2 * hd * sizeof(float));
    }
    return r;
}

/* ---- sampling.go ---- */
typedef struct { int idx; float prob; } po_iprob;
static int po_iprob_desc(const void* a, const void* b) {           /* sampling.go:143-145,172-174 (ties: by index) */
    const po_iprob *x = (const po_iprob*)a, *y = (const po_iprob*)b;
    if (x->prob > y->prob) return -1;
    if (x->prob < y->prob) return 1;
    return x->idx - y->idx;
}
int po_sample_with_history(const float* logits, int n, const int32_t* prev, int n_prev, float temperature,
                           float top_p, int top_k, float rep_penalty, float u, float* probs_out) {
    float* l = (float*)malloc((size_t)n * sizeof(float));
    float* probs = (float*)malloc((size_t)n * sizeof(float));
    po_iprob* ind = (po_iprob*)malloc((size_t)n * sizeof(po_iprob));
    memcpy(l, logits, (size_t)n * sizeof(float));
    /* repetition penalty (:43-68): counts weighted 3 for the last 10 history tokens */
    if (rep_penalty != 1.0f && n_prev > 0) {
        int* counts = (int*)calloc((size_t)n, sizeof(int));
        for (int i = 0; i < n_prev; i++) {
            const int w = i >= n_prev - 10 ? 3 : 1;
            if (prev[i] >= 0 && prev[i] < n) counts[prev[i]] += w;
        }
        for (int t = 0; t < n; t++) {
            if (!counts[t]) continue;
            const float penalty = rep_penalty * (float)counts[t];
            if (l[t] > 0) l[t] /= penalty; else l[t] *= penalty;
        }
        free(counts);
    }
    if (temperature > 0 && temperature != 1.0f)                     /* :71-75 */
        for (int i = 0; i < n; i++) l[i] /= temperature;
    /* softmax (:105-127) */
    float mx = l[0];
    for (int i = 1; i < n; i++) if (l[i] > mx) mx = l[i];
    float sum = 0.f;
    for (int i = 0; i < n; i++) { probs[i] = (float)exp((double)(l[i] - mx)); sum += probs[i]; }
    for (int i = 0; i < n; i++) probs[i] /= sum;
    if (top_k > 0 && top_k < n) {                                   /* :78-80, :130-156 */
        for (int i = 0; i < n; i++) { ind[i].idx = i; ind[i].prob = probs[i]; }
        qsort(ind, (size_t)n, sizeof(po_iprob), po_iprob_desc);
        memset(probs, 0, (size_t)n * sizeof(float));
        for (int i = 0; i < top_k && i < n; i++) probs[ind[i].idx] = ind[i].prob;
    }
    if (top_p < 1.0f) {                                             /* :83-85, :159-195 */
        for (int i = 0; i < n; i++) { ind[i].idx = i; ind[i].prob = probs[i]; }
        qsort(ind, (size_t)n, sizeof(po_iprob), po_iprob_desc);
        float cum = 0.f;
        int cutoff = n;
        for (int i = 0; i < n; i++) {
            cum += ind[i].prob;
            if (cum >= top_p) { cutoff = i + 1; break; }
        }
        memset(probs, 0, (size_t)n * sizeof(float));
        for (int i = 0; i < cutoff; i++) probs[ind[i].idx] = ind[i].prob;
    }
    sum = 0.f;                                                      /* :88-96 */
    for (int i = 0; i < n; i++) sum += probs[i];
    if (sum > 0) for (int i = 0; i < n; i++) probs[i] /= sum;
    if (probs_out) memcpy(probs_out, probs, (size_t)n * sizeof(float));
    /* sampleMultinomial (:198-217): first index with cum >= r (sort.Search) */
    float c = 0.f, last = 0.f;
    for (int i = 0; i < n; i++) last += probs[i];
    const float r = u * last;
    int idx = n;
    for (int i = 0; i < n; i++) {
        c = (i == 0) ? probs[0] : c + probs[i];
        if (c >= r) { idx = i; break; }
    }
    if (idx >= n) idx = n - 1;
    free(l); free(probs); free(ind);
    return idx;
}

int po_argmax(const float* data, int n) { /* cmd/ask/main.go:389-402 */
    if (n == 0) return 0;
    int mi = 0;
    float mv = data[0];
    for (int i = 0; i < n; i++)
        if (data[i] > mv) { mv = data[i]; mi = i; }
    return mi;
}

/* ------------------------------------------------------------------------ */
/* RoPE: purego/tensor/rope.go                                               */
/* ------------------------------------------------------------------------ */

/* NewRoPECache, rope.go:18-50 */
void po_rope_tables(int head_dim, int max_seq, double base, float* cos_t, float* sin_t) {
    int half = head_dim / 2;
    for (int pos = 0; pos < max_seq; pos++) {
        for (int i = 0; i < half; i++) {
            double freq = 1.0 / pow(base, (double)(2 * i) / (double)head_dim);
            double angle = (double)pos * freq;
            float cv = (float)cos(angle);
            float sv = (float)sin(angle);
            cos_t[(int64_t)pos * head_dim + i] = cv;
            cos_t[(int64_t)pos * head_dim + half + i] = cv;
            sin_t[(int64_t)pos * head_dim + i] = sv;
            sin_t[(int64_t)pos * head_dim + half + i] = sv;
        }
    }
}

/* ApplyRoPESingleTensor, rope.go:153-205 (ApplyRoPE :55-148 is the same
 * per-tensor arithmetic applied to Q then K).  t = [heads, seq, hd]. */
int po_rope_apply(float* t, int heads, int seq, int hd, int start_pos,
                  const float* cos_t, const float* sin_t, int max_seq) {
    int half = hd / 2;
    float orig[1024];
    if (hd > 1024) return -1;
    for (int h = 0; h < heads; h++) {
        for (int s = 0; s < seq; s++) {
            int pos = start_pos + s;
            if (pos >= max_seq) return -1; /* rope.go:176-178 panics */
            float* v = t + ((int64_t)h * seq + s) * hd;
            const float* cr = cos_t + (int64_t)pos * hd;
            const float* sr = sin_t + (int64_t)pos * hd;
            memcpy(orig, v, (size_t)hd * sizeof(float));
            for (int i = 0; i < hd; i++) {
                float rot = (i < half) ? -orig[i + half] : orig[i - half];
                float a = orig[i] * cr[i];
                float b = rot * sr[i];
                v[i] = a + b;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* FFN: purego/tensor/transformer.go:40-96                                   */
/* ------------------------------------------------------------------------ */

void po_ffn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
            int rows, int hidden, int ffn, int swiglu, float* y) {
    if (swiglu) {
        int64_t n2 = (int64_t)rows * 2 * ffn;
        float* h = fzeros(n2);
        po_matmul(x, w1, h, rows, hidden, 2 * ffn);            /* :48 */
        float* gate = fzeros((int64_t)rows * ffn);
        float* up = fzeros((int64_t)rows * ffn);
        for (int i = 0; i < rows; i++) {                       /* SliceLastDim mamba2.go:379 */
            memcpy(gate + (int64_t)i * ffn, h + (int64_t)i * 2 * ffn, (size_t)ffn * sizeof(float));
            memcpy(up + (int64_t)i * ffn, h + (int64_t)i * 2 * ffn + ffn, (size_t)ffn * sizeof(float));
        }
        po_silu(gate, gate, (int64_t)rows * ffn);              /* :61 */
        for (int64_t i = 0; i < (int64_t)rows * ffn; i++) gate[i] = gate[i] * up[i]; /* :63-65 */
        po_matmul(gate, w2, y, rows, ffn, hidden);             /* :81 */
        free(h); free(gate); free(up);
    } else {
        float* h = fzeros((int64_t)rows * ffn);
        po_matmul(x, w1, h, rows, hidden, ffn);
        if (b1) {                                              /* :69-75 */
            for (int i = 0; i < rows; i++)
                for (int j = 0; j < ffn; j++)
                    h[(int64_t)i * ffn + j] = h[(int64_t)i * ffn + j] + b1[j];
        }
        po_gelu(h, h, (int64_t)rows * ffn);                    /* :77 */
        po_matmul(h, w2, y, rows, ffn, hidden);
        free(h);
    }
    if (b2) {                                                  /* :84-90 */
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < hidden; j++)
                y[(int64_t)i * hidden + j] = y[(int64_t)i * hidden + j] + b2[j];
    }
}

/* ------------------------------------------------------------------------ */
/* MoE: purego/tensor/moe.go:43-128 (Forward), :167-226 (separate experts)   */
/* ------------------------------------------------------------------------ */

void po_moe(const float* x, const float* router, const float* w_in, const float* w_out,
            int rows, int hidden, int n_experts, int top_k, int inter, float* y) {
    float* logits = fzeros((int64_t)rows * n_experts);
    float* probs = fzeros((int64_t)rows * n_experts);
    po_matmul(x, router, logits, rows, hidden, n_experts);     /* :63 */
    po_softmax_rows(logits, probs, rows, n_experts);           /* :66 */
    memset(y, 0, (size_t)rows * hidden * sizeof(float));       /* :69 */

    int in_out = 2 * inter;  /* inputShape[1] */
    int* idx = (int*)malloc(sizeof(int) * (size_t)n_experts);
    float* sc = (float*)malloc(sizeof(float) * (size_t)n_experts);
    float* proj1 = fzeros(in_out);
    float* mid = fzeros(inter);
    float* eo = fzeros(hidden);

    for (int i = 0; i < rows; i++) {
        for (int j = 0; j < n_experts; j++) { idx[j] = j; sc[j] = probs[(int64_t)i * n_experts + j]; }
        /* sort.Slice descending by score (moe.go:83-85).  Go's sort is not
         * stable; this restatement fixes the tie rule: equal scores keep the
         * LOWER expert index first (stable insertion sort). */
        for (int a = 1; a < n_experts; a++) {
            int ci = idx[a]; float cs = sc[a];
            int b = a - 1;
            while (b >= 0 && sc[b] < cs) { sc[b + 1] = sc[b]; idx[b + 1] = idx[b]; b--; }
            sc[b + 1] = cs; idx[b + 1] = ci;
        }
        float sum_scores = 0.0f;                               /* :89-92 */
        for (int t = 0; t < top_k; t++) sum_scores = sum_scores + sc[t];
        const float* tok = x + (int64_t)i * hidden;
        for (int t = 0; t < top_k; t++) {
            int e = idx[t];
            float weight = sc[t] / sum_scores;                 /* :103 */
            const float* wi = w_in + (int64_t)e * in_out * hidden;
            const float* wo = w_out + (int64_t)e * hidden * inter;
            for (int r = 0; r < in_out; r++) {                 /* :192-200 */
                float s = 0.0f;
                const float* wr = wi + (int64_t)r * hidden;
                for (int j = 0; j < hidden; j++) { float p = tok[j] * wr[j]; s = s + p; }
                proj1[r] = s;
            }
            for (int r = 0; r < inter; r++) {                  /* :204-212 */
                float gate = proj1[r];
                float up = proj1[r + inter];
                float ex = (float)exp((double)(-gate));        /* exp narrowed first, :208 */
                float den = 1.0f + ex;
                float sg = 1.0f / den;
                float ga = gate * sg;
                mid[r] = ga * up;
            }
            for (int r = 0; r < hidden; r++) {                 /* :216-223 */
                float s = 0.0f;
                const float* wr = wo + (int64_t)r * inter;
                for (int j = 0; j < inter; j++) { float p = mid[j] * wr[j]; s = s + p; }
                eo[r] = s;
            }
            float* yr = y + (int64_t)i * hidden;               /* :117-119 */
            for (int j = 0; j < hidden; j++) { float p = weight * eo[j]; yr[j] = yr[j] + p; }
        }
    }
    free(logits); free(probs); free(idx); free(sc); free(proj1); free(mid); free(eo);
}

/* ------------------------------------------------------------------------ */
/* attention cores                                                           */
/* ------------------------------------------------------------------------ */

/* [S, heads*hd] -> [heads, S, hd]  (splitHeads*, attention.go:88,300,316; reshapeQ mqa.go:142) */
static float* split_heads(const float* x, int S, int heads, int hd) {
    float* r = fzeros((int64_t)heads * S * hd);
    int width = heads * hd;
    for (int s = 0; s < S; s++)
        for (int h = 0; h < heads; h++)
            memcpy(r + ((int64_t)h * S + s) * hd, x + (int64_t)s * width + (int64_t)h * hd,
                   (size_t)hd * sizeof(float));
    return r;
}

/* [heads, S, hd] -> [S, heads*hd]  (combineHeads/mergeHeads/transposeHeadsAndSeq) */
static float* merge_heads(const float* x, int S, int heads, int hd) {
    float* r = fzeros((int64_t)heads * S * hd);
    int width = heads * hd;
    for (int h = 0; h < heads; h++)
        for (int s = 0; s < S; s++)
            memcpy(r + (int64_t)s * width + (int64_t)h * hd, x + ((int64_t)h * S + s) * hd,
                   (size_t)hd * sizeof(float));
    return r;
}

/* repeatKVHeads, attention.go:333-352 / mqa.go:163-182: materialised copy. */
static float* repeat_kv(const float* x, int nKV, int repeat, int T, int hd) {
    float* r = fzeros((int64_t)nKV * repeat * T * hd);
    for (int kvh = 0; kvh < nKV; kvh++)
        for (int rr = 0; rr < repeat; rr++)
            memcpy(r + ((int64_t)(kvh * repeat + rr) * T) * hd, x + ((int64_t)kvh * T) * hd,
                   (size_t)T * hd * sizeof(float));
    return r;
}

/* GQA: computeScores (attention.go:354-397) + softmaxLastDim (:439-470) +
 * applyAttention (:399-421), on K/V already repeated to nH heads. */
static void gqa_sdpa(const float* Q, const float* K, const float* V, int nH, int S, int T,
                     int hd, float scale_cfg, float* out) {
    float scale = scale_cfg;
    if (scale == 0.0f) scale = 1.0f / (float)sqrt((double)hd);   /* :362-364 */
    float* scores = fzeros((int64_t)S * T);
    for (int h = 0; h < nH; h++) {
        const float* q = Q + (int64_t)h * S * hd;
        const float* k = K + (int64_t)h * T * hd;
        const float* v = V + (int64_t)h * T * hd;
        for (int i = 0; i < S; i++) {
            int max_allowed = T - S + i;
            for (int j = 0; j < T; j++) {
                if (j > max_allowed) { scores[(int64_t)i * T + j] = -1e10f; continue; }
                float sum = 0.0f;
                for (int d = 0; d < hd; d++) {
                    float p = q[(int64_t)i * hd + d] * k[(int64_t)j * hd + d];
                    sum = sum + p;
                }
                scores[(int64_t)i * T + j] = sum * scale;
            }
        }
        for (int i = 0; i < S; i++) {                         /* softmaxLastDim */
            float* row = scores + (int64_t)i * T;
            float maxv = -1e10f;
            for (int j = 0; j < T; j++) if (row[j] > maxv) maxv = row[j];
            float sum = 0.0f;
            for (int j = 0; j < T; j++) {
                float d = row[j] - maxv;
                row[j] = (float)exp((double)d);
                sum = sum + row[j];
            }
            for (int j = 0; j < T; j++) row[j] = row[j] / sum;
        }
        for (int i = 0; i < S; i++) {                         /* applyAttention */
            for (int d = 0; d < hd; d++) {
                float sum = 0.0f;
                for (int j = 0; j < T; j++) {
                    float p = scores[(int64_t)i * T + j] * v[(int64_t)j * hd + d];
                    sum = sum + p;
                }
                out[((int64_t)h * S + i) * hd + d] = sum;
            }
        }
    }
    free(scores);
}

void po_gqa_core(const float* q, const float* k, const float* v, int nH, int nKV,
                 int S, int T, int hd, float scale, float* out) {
    int rep = nH / nKV;
    float* kr = repeat_kv(k, nKV, rep, T, hd);
    float* vr = repeat_kv(v, nKV, rep, T, hd);
    gqa_sdpa(q, kr, vr, nH, S, T, hd, scale, out);
    free(kr); free(vr);
}

/* MHA scaledDotProductAttention (attention.go:127-191) and MQA
 * scaledDotProductMQA (mqa.go:184-243): scores for every j, THEN the mask is
 * written over them, then tensor.Softmax (max seeded with element 0). */
static void mha_sdpa(const float* Q, const float* K, const float* V, int nH, int S, int T,
                     int hd, float* out) {
    float scale = (float)(1.0 / sqrt((double)hd));            /* attention.go:136, mqa.go:185 */
    float* scores = fzeros((int64_t)S * T);
    float* probs = fzeros((int64_t)S * T);
    for (int h = 0; h < nH; h++) {
        const float* q = Q + (int64_t)h * S * hd;
        const float* k = K + (int64_t)h * T * hd;
        const float* v = V + (int64_t)h * T * hd;
        for (int i = 0; i < S; i++)
            for (int j = 0; j < T; j++) {
                float sum = 0.0f;
                for (int d = 0; d < hd; d++) {
                    float p = q[(int64_t)i * hd + d] * k[(int64_t)j * hd + d];
                    sum = sum + p;
                }
                scores[(int64_t)i * T + j] = sum * scale;
            }
        for (int i = 0; i < S; i++) {
            int maxpos = T - S + i;
            for (int j = maxpos + 1; j < T; j++) scores[(int64_t)i * T + j] = -1e10f;
        }
        po_softmax_rows(scores, probs, S, T);
        for (int i = 0; i < S; i++)
            for (int d = 0; d < hd; d++) {
                float sum = 0.0f;
                for (int j = 0; j < T; j++) {
                    float p = probs[(int64_t)i * T + j] * v[(int64_t)j * hd + d];
                    sum = sum + p;
                }
                out[((int64_t)h * S + i) * hd + d] = sum;
            }
    }
    free(scores); free(probs);
}

/* ------------------------------------------------------------------------ */
/* model                                                                     */
/* ------------------------------------------------------------------------ */

typedef struct { const float* p; float* owned; int64_t n; } po_slot;

struct po_model {
    po_config cfg;
    po_slot global[PO_T_COUNT];
    po_slot* layer; /* [num_layers][PO_T_COUNT] */
    float* rope_cos; /* one table serves all layers: the reference builds L identical copies
                        (generic_model.go:109-111) */
    float* rope_sin;
    float** ssm_state; /* per layer: Mamba2Layer.SSMState [heads, head_dim, state] (batch 1), NULL = nil (mamba2.go:258) */
};

struct po_kvcache {
    int num_layers;
    int heads, hd;               /* filled by the first forward */
    float** k; float** v; int* t; /* per layer: [heads, T, hd] */
};

po_model* po_model_new(const po_config* cfg) {
    po_model* m = (po_model*)calloc(1, sizeof(po_model));
    m->cfg = *cfg;
    m->layer = (po_slot*)calloc((size_t)cfg->num_layers * PO_T_COUNT, sizeof(po_slot));
    m->ssm_state = (float**)calloc((size_t)cfg->num_layers, sizeof(float*));
    int uses_rope = 0;
    double base = cfg->rope_base;
    if (cfg->attention_type == PO_ATTN_MQA) { uses_rope = 1; base = 10000.0; } /* mqa.go:35 ignores rope_theta */
    else if (cfg->attention_type == PO_ATTN_GQA && cfg->position_type == PO_POS_ROPE) uses_rope = 1;
    if (uses_rope) {
        int64_t n = (int64_t)cfg->max_seq_len * cfg->head_dim;
        m->rope_cos = fzeros(n);
        m->rope_sin = fzeros(n);
        po_rope_tables(cfg->head_dim, cfg->max_seq_len, base, m->rope_cos, m->rope_sin);
    }
    return m;
}

void po_model_free(po_model* m) {
    if (!m) return;
    for (int s = 0; s < PO_T_COUNT; s++) free(m->global[s].owned);
    for (int64_t i = 0; i < (int64_t)m->cfg.num_layers * PO_T_COUNT; i++) free(m->layer[i].owned);
    po_mamba2_reset(m);
    free(m->ssm_state);
    free(m->layer); free(m->rope_cos); free(m->rope_sin); free(m);
}

static po_slot* slot_of(po_model* m, int slot, int layer) {
    if (slot < 0 || slot >= PO_T_COUNT) return NULL;
    if (slot < PO_T_ATTN_NORM_W) return &m->global[slot];
    if (layer < 0 || layer >= m->cfg.num_layers) return NULL;
    return &m->layer[(int64_t)layer * PO_T_COUNT + slot];
}

int po_model_set(po_model* m, int slot, int layer, const float* data, int64_t n) {
    po_slot* s = slot_of(m, slot, layer);
    if (!s) return -1;
    free(s->owned);
    s->owned = (float*)malloc((size_t)n * sizeof(float));
    memcpy(s->owned, data, (size_t)n * sizeof(float));
    s->p = s->owned; s->n = n;
    return 0;
}

int po_model_set_borrowed(po_model* m, int slot, int layer, const float* data, int64_t n) {
    po_slot* s = slot_of(m, slot, layer);
    if (!s) return -1;
    free(s->owned); s->owned = NULL;
    s->p = data; s->n = n;
    return 0;
}

po_kvcache* po_kvcache_new(int num_layers) {
    po_kvcache* kv = (po_kvcache*)calloc(1, sizeof(po_kvcache));
    kv->num_layers = num_layers;
    kv->k = (float**)calloc((size_t)num_layers, sizeof(float*));
    kv->v = (float**)calloc((size_t)num_layers, sizeof(float*));
    kv->t = (int*)calloc((size_t)num_layers, sizeof(int));
    return kv;
}

void po_kvcache_free(po_kvcache* kv) {
    if (!kv) return;
    for (int i = 0; i < kv->num_layers; i++) { free(kv->k[i]); free(kv->v[i]); }
    free(kv->k); free(kv->v); free(kv->t); free(kv);
}

int po_kvcache_len(const po_kvcache* kv) { return kv->num_layers > 0 ? kv->t[0] : 0; }

static int kv_heads_of(const po_config* c) {
    if (c->attention_type == PO_ATTN_MHA) return c->num_heads;
    if (c->attention_type == PO_ATTN_MQA) return 1;
    return c->num_kv_heads;
}

/* copies layer K (which=0) or V (which=1), [kv_heads, T, hd], into out (may be NULL); returns T */
int po_kvcache_get(const po_kvcache* kv, int layer, int which, float* out) {
    if (layer < 0 || layer >= kv->num_layers) return -1;
    const float* src = which ? kv->v[layer] : kv->k[layer];
    if (out && src)
        memcpy(out, src, (size_t)kv->heads * kv->t[layer] * kv->hd * sizeof(float));
    return kv->t[layer];
}

#define L_(slot) (m->layer[(int64_t)li * PO_T_COUNT + (slot)].p)

static void add_bias_rows(float* x, const float* b, int rows, int cols) { /* attention.go:77-83 */
    if (!b) return;
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++)
            x[(int64_t)i * cols + j] = x[(int64_t)i * cols + j] + b[j];
}

/* one attention layer: returns output [S, H]; updates cache for layer li */
static float* attention_layer(po_model* m, int li, const float* x, int S, po_kvcache* kv,
                              int pos_offset, int* err) {
    const po_config* c = &m->cfg;
    int H = c->hidden, nH = c->num_heads, hd = c->head_dim;
    int qw = nH * hd;
    int nKV = kv_heads_of(c);
    float *Q, *K, *V;

    float* qf = fzeros((int64_t)S * qw);
    po_matmul(x, L_(PO_T_WQ), qf, S, H, qw);
    float* kf; float* vf;
    if (c->attention_type == PO_ATTN_MQA) {
        /* projectKV, mqa.go:116-140: fused [H, 2hd] then split */
        float* kvf = fzeros((int64_t)S * 2 * hd);
        po_matmul(x, L_(PO_T_WKV), kvf, S, H, 2 * hd);
        kf = fzeros((int64_t)S * hd); vf = fzeros((int64_t)S * hd);
        for (int i = 0; i < S; i++) {
            memcpy(kf + (int64_t)i * hd, kvf + (int64_t)i * 2 * hd, (size_t)hd * sizeof(float));
            memcpy(vf + (int64_t)i * hd, kvf + (int64_t)i * 2 * hd + hd, (size_t)hd * sizeof(float));
        }
        free(kvf);
    } else {
        kf = fzeros((int64_t)S * nKV * hd); vf = fzeros((int64_t)S * nKV * hd);
        po_matmul(x, L_(PO_T_WK), kf, S, H, nKV * hd);
        po_matmul(x, L_(PO_T_WV), vf, S, H, nKV * hd);
    }
    if (c->attention_type == PO_ATTN_MHA) {                   /* project(), attention.go:66-86 */
        add_bias_rows(qf, L_(PO_T_BQ), S, qw);
        add_bias_rows(kf, L_(PO_T_BK), S, nKV * hd);
        add_bias_rows(vf, L_(PO_T_BV), S, nKV * hd);
    }
    Q = split_heads(qf, S, nH, hd);
    K = split_heads(kf, S, nKV, hd);
    V = split_heads(vf, S, nKV, hd);
    free(qf); free(kf); free(vf);

    if (m->rope_cos && c->attention_type != PO_ATTN_MHA) {    /* attention.go:234-237, mqa.go:65-67 */
        if (po_rope_apply(Q, nH, S, hd, pos_offset, m->rope_cos, m->rope_sin, c->max_seq_len) != 0 ||
            po_rope_apply(K, nKV, S, hd, pos_offset, m->rope_cos, m->rope_sin, c->max_seq_len) != 0) {
            *err = -1;
        }
    }

    int T0 = kv->t[li];
    float* Kall = concat_seq(kv->k[li], T0, K, S, nKV, hd);   /* attention.go:241-244 */
    float* Vall = concat_seq(kv->v[li], T0, V, S, nKV, hd);
    free(K); free(V);
    free(kv->k[li]); free(kv->v[li]);
    kv->k[li] = Kall; kv->v[li] = Vall; kv->t[li] = T0 + S;
    kv->heads = nKV; kv->hd = hd;
    int T = T0 + S;

    float* ctx = fzeros((int64_t)nH * S * hd);
    if (c->attention_type == PO_ATTN_GQA) {
        int rep = nH / nKV;
        float* kr = repeat_kv(Kall, nKV, rep, T, hd);
        float* vr = repeat_kv(Vall, nKV, rep, T, hd);
        gqa_sdpa(Q, kr, vr, nH, S, T, hd, c->attention_multiplier, ctx);
        free(kr); free(vr);
    } else if (c->attention_type == PO_ATTN_MQA) {
        float* kr = repeat_kv(Kall, 1, nH, T, hd);
        float* vr = repeat_kv(Vall, 1, nH, T, hd);
        mha_sdpa(Q, kr, vr, nH, S, T, hd, ctx);
        free(kr); free(vr);
    } else {
        mha_sdpa(Q, Kall, Vall, nH, S, T, hd, ctx);
    }
    free(Q);
    float* merged = merge_heads(ctx, S, nH, hd);
    free(ctx);
    float* out = fzeros((int64_t)S * H);
    po_matmul(merged, L_(PO_T_WO), out, S, qw, H);
    if (c->attention_type == PO_ATTN_MHA) add_bias_rows(out, L_(PO_T_BO), S, H);
    free(merged);
    return out;
}

static void residual_add(float* x /* new */, const float* residual, float mult, int64_t n) {
    /* generic_model.go:320-326 etc. */
    if (mult != 0.0f) {
        for (int64_t j = 0; j < n; j++) { float p = mult * x[j]; x[j] = residual[j] + p; }
    } else {
        for (int64_t j = 0; j < n; j++) x[j] = x[j] + residual[j]; /* Add(x, residual) */
    }
}

/* ------------------------------------------------------------------------ */
/* Mamba2: purego/tensor/mamba2.go                                           */
/* ------------------------------------------------------------------------ */

/* ResetState, mamba2.go:353-357 (called by ForwardWithCache for a brand-new sequence, generic_model.go:285-292) */
void po_mamba2_reset(po_model* m) {
    for (int li = 0; li < m->cfg.num_layers; li++) { free(m->ssm_state[li]); m->ssm_state[li] = NULL; }
}
int po_mamba2_get_state(const po_model* m, int layer, float* out) {
    const po_config* c = &m->cfg;
    int hd = c->mamba_head_dim ? c->mamba_head_dim : (c->mamba_num_heads ? c->mamba_expand * c->hidden / c->mamba_num_heads : 0);
    int n = c->mamba_num_heads * hd * c->mamba_state_size;
    if (layer < 0 || layer >= c->num_layers || !m->ssm_state[layer]) return 0;
    memcpy(out, m->ssm_state[layer], (size_t)n * sizeof(float));
    return n;
}

/* Softplus, mamba2.go:370-376 */
static float softplus_f(float x) { return (float)log(1.0 + exp((double)x)); }

/* Mamba2Layer.Forward, mamba2.go:74-181 (batch = 1).  x [S, H] -> out [S, H] (caller frees). */
static float* mamba2_forward(po_model* m, int li, const float* x, int S) {
    const po_config* c = &m->cfg;
    const int H = c->hidden, EH = c->mamba_expand * c->hidden, nh = c->mamba_num_heads, ss = c->mamba_state_size;
    const int hd = c->mamba_head_dim ? c->mamba_head_dim : EH / nh;           /* mamba2.go:58-61 */
    const int ng = c->mamba_n_groups, K = c->mamba_conv_kernel;
    const int gate_size = EH, conv_dim = EH + 2 * ng * ss, dt_size = nh;      /* :92-95 (ConvWeight.Shape[0]) */
    const int P = gate_size + conv_dim + dt_size;
    const float* in_proj = L_(PO_T_MAMBA_IN_PROJ);

    /* 1. projected = MatMul(xFlat, Transpose(InProj)) :89-90 — Transpose materialises [H, P], then the i-p-j MatMul */
    float* in_t = fzeros((int64_t)H * P);
    po_transpose(in_proj, in_t, P, H);
    float* proj = fzeros((int64_t)S * P);
    po_matmul(x, in_t, proj, S, H, P);
    free(in_t);
    /* split [gate | xBC | dt] :98-100 */
    /* 2. causal conv on xBC :107, mamba2.go:183-254: left zero padding of K-1, "forward" kernel, first S positions.
     *    NOTE (reference behaviour): no convolution state is carried across calls (ConvCache is never used), so a
     *    one-token decode call convolves the token with zeros. */
    const int pad = K - 1, PS = S + pad;
    const float* cw = L_(PO_T_MAMBA_CONV_W);
    const float* cb = L_(PO_T_MAMBA_CONV_B);
    float* padded = fzeros((int64_t)PS * conv_dim);
    for (int t = 0; t < S; t++)
        memcpy(padded + (int64_t)(t + pad) * conv_dim, proj + (int64_t)t * P + gate_size, (size_t)conv_dim * sizeof(float));
    float* xbc = fzeros((int64_t)S * conv_dim);
    for (int t = 0; t < S; t++)               /* only the first S of the PS computed positions are kept (:241-251) */
        for (int ch = 0; ch < conv_dim; ch++) {
            float sum = 0.0f;
            for (int k = 0; k < K; k++) {
                int pos = t + k;
                if (pos >= 0 && pos < PS) {
                    float prod = padded[(int64_t)pos * conv_dim + ch] * cw[(int64_t)ch * K + k];
                    sum = sum + prod;
                }
            }
            if (cb) sum = sum + cb[ch];
            xbc[(int64_t)t * conv_dim + ch] = sum;
        }
    free(padded);
    /* 3. SiLU :110 */
    po_silu(xbc, xbc, (int64_t)S * conv_dim);
    /* 4. split x | B | C :115-117; 5. delta = Softplus(dt + bias) :124-131 */
    float* delta = fzeros((int64_t)S * nh);
    const float* dtb = L_(PO_T_MAMBA_DT_BIAS);
    for (int t = 0; t < S; t++)
        for (int h = 0; h < nh; h++) {
            float v = proj[(int64_t)t * P + gate_size + conv_dim + h];
            if (dtb) v = v + dtb[h];
            delta[(int64_t)t * nh + h] = softplus_f(v);
        }
    /* 6. selectiveScan :256-351 */
    if (!m->ssm_state[li]) m->ssm_state[li] = fzeros((int64_t)nh * hd * ss);
    float* state = m->ssm_state[li];
    const float* alog = L_(PO_T_MAMBA_A_LOG);
    const float* Dp = L_(PO_T_MAMBA_D);
    float* y = fzeros((int64_t)S * EH);
    for (int t = 0; t < S; t++)
        for (int h = 0; h < nh; h++) {
            const float dt = delta[(int64_t)t * nh + h];
            float abar = 1.0f;
            if (alog) {
                float A = -(float)exp((double)alog[h]);           /* :286 */
                abar = (float)exp((double)(A * dt));              /* :287 */
            }
            int g = h * ng / nh;                                  /* :301-304 */
            if (g >= ng) g = ng - 1;
            const float* u = xbc + (int64_t)t * conv_dim + (int64_t)h * hd;
            const float* Bt = xbc + (int64_t)t * conv_dim + EH + (int64_t)g * ss;
            const float* Ct = xbc + (int64_t)t * conv_dim + EH + (int64_t)ng * ss + (int64_t)g * ss;
            for (int d = 0; d < hd; d++)
                for (int s2 = 0; s2 < ss; s2++) {
                    float* st = state + ((int64_t)h * hd + d) * ss + s2;
                    float a = abar * *st;                         /* :325: ABar*oldState + dt*Bt[s]*u[d] (left to right) */
                    float b = dt * Bt[s2];
                    b = b * u[d];
                    *st = a + b;
                }
            for (int d = 0; d < hd; d++) {
                float sum = 0.0f;
                for (int s2 = 0; s2 < ss; s2++) {
                    float prod = Ct[s2] * state[((int64_t)h * hd + d) * ss + s2];
                    sum = sum + prod;
                }
                if (Dp) { float prod = Dp[h] * u[d]; sum = sum + prod; }      /* :339-341 */
                y[(int64_t)t * EH + (int64_t)h * hd + d] = sum;
            }
        }
    /* 7. gated RMSNorm :135-170: y *= SiLU(gate); rms over EH with eps 1e-5; *= Norm */
    {
        float* gact = fzeros((int64_t)S * gate_size);
        for (int t = 0; t < S; t++) memcpy(gact + (int64_t)t * gate_size, proj + (int64_t)t * P, (size_t)gate_size * sizeof(float));
        po_silu(gact, gact, (int64_t)S * gate_size);
        for (int64_t i = 0; i < (int64_t)S * EH; i++) y[i] = y[i] * gact[i];
        free(gact);
        const float eps = 1e-5f;
        for (int t = 0; t < S; t++) {
            float var = 0.0f;
            float* yr = y + (int64_t)t * EH;
            for (int d = 0; d < EH; d++) { float prod = yr[d] * yr[d]; var = var + prod; }
            var = var / (float)EH;
            float rms = 1.0f / (float)sqrt((double)(var + eps));
            for (int d = 0; d < EH; d++) yr[d] = yr[d] * rms;
        }
        const float* nw = L_(PO_T_MAMBA_NORM);
        if (nw) for (int64_t i = 0; i < (int64_t)S * EH; i++) y[i] = y[i] * nw[i % EH];
    }
    /* 9. output = MatMul(yFlat, Transpose(OutProj)) :173-176 */
    const float* out_proj = L_(PO_T_MAMBA_OUT_PROJ);
    float* out_t = fzeros((int64_t)EH * H);
    po_transpose(out_proj, out_t, H, EH);
    float* out = fzeros((int64_t)S * H);
    po_matmul(y, out_t, out, S, EH, H);
    free(out_t); free(y); free(delta); free(xbc); free(proj);
    return out;
}

int po_forward_with_cache(po_model* m, const int32_t* tokens, int n_tokens,
                          po_kvcache* kv, int pos_offset, float* logits_out, float* hidden_out) {
    const po_config* c = &m->cfg;
    int H = c->hidden, S = n_tokens, V = c->vocab_size;
    int err = 0;
    int64_t n = (int64_t)S * H;

    /* Mamba2 state is reset only for a brand-new sequence, generic_model.go:285-292 (posOffset == 0 && seqLen > 1) */
    if (pos_offset == 0 && S > 1) po_mamba2_reset(m);

    /* embedWithOffset, generic_model.go:567-592 */
    float* x = fzeros(n);
    const float* emb = m->global[PO_T_TOK_EMB].p;
    const float* pemb = m->global[PO_T_POS_EMB].p;
    for (int i = 0; i < S; i++) {
        int tok = tokens[i];
        if (tok < 0 || tok >= V) { free(x); return -1; }
        memcpy(x + (int64_t)i * H, emb + (int64_t)tok * H, (size_t)H * sizeof(float));
        if (c->position_type == PO_POS_LEARNED && pemb) {
            int ap = pos_offset + i;
            if (ap < c->max_seq_len)
                for (int j = 0; j < H; j++)
                    x[(int64_t)i * H + j] = x[(int64_t)i * H + j] + pemb[(int64_t)ap * H + j];
        }
    }
    if (c->embedding_multiplier != 0.0f)                      /* :298-302 */
        for (int64_t j = 0; j < n; j++) x[j] = x[j] * c->embedding_multiplier;

    float* normed = fzeros(n);
    for (int li = 0; li < c->num_layers; li++) {
        if (li < 128 && c->layer_is_mamba[li]) {
            /* block.Forward -> forwardMamba2, generic_model.go:456-459, 160-202: norm, Mamba2, residual; norm, FFN, residual */
            po_layernorm(x, L_(PO_T_ATTN_NORM_W), L_(PO_T_ATTN_NORM_B), c->norm_eps, normed, S, H);
            float* mo = mamba2_forward(m, li, normed, S);
            residual_add(mo, x, c->residual_multiplier, n);
            memcpy(x, mo, (size_t)n * sizeof(float));
            free(mo);
            if (L_(PO_T_W1)) {
                po_layernorm(x, L_(PO_T_FFN_NORM_W), L_(PO_T_FFN_NORM_B), c->norm_eps, normed, S, H);
                float* f = fzeros(n);
                po_ffn(normed, L_(PO_T_W1), L_(PO_T_B1), L_(PO_T_W2), L_(PO_T_B2), S, H, c->ffn_dim,
                       c->activation_type == PO_ACT_SWIGLU, f);
                residual_add(f, x, c->residual_multiplier, n);
                memcpy(x, f, (size_t)n * sizeof(float));
                free(f);
            }
        } else if (c->attention_type == PO_ATTN_MQA && c->block_style == PO_BLOCK_PARALLEL) {
            /* generic_model.go:395-418 */
            po_layernorm(x, L_(PO_T_ATTN_NORM_W), L_(PO_T_ATTN_NORM_B), c->norm_eps, normed, S, H);
            float* attn = attention_layer(m, li, normed, S, kv, pos_offset, &err);
            float* ffn = fzeros(n);
            po_ffn(normed, L_(PO_T_W1), L_(PO_T_B1), L_(PO_T_W2), L_(PO_T_B2), S, H, c->ffn_dim,
                   c->activation_type == PO_ACT_SWIGLU, ffn);
            if (c->attention_multiplier != 0.0f && c->residual_multiplier != 0.0f) {
                for (int64_t j = 0; j < n; j++) {
                    float a = c->attention_multiplier * attn[j];
                    float r = x[j] + a;
                    float f = c->residual_multiplier * ffn[j];
                    x[j] = r + f;
                }
            } else {
                for (int64_t j = 0; j < n; j++) { float r = x[j] + attn[j]; x[j] = r + ffn[j]; }
            }
            free(attn); free(ffn);
        } else {
            /* sequential: MHA :308-346, GQA :347-390, MQA sequential :419-456 */
            po_layernorm(x, L_(PO_T_ATTN_NORM_W), L_(PO_T_ATTN_NORM_B), c->norm_eps, normed, S, H);
            float* attn = attention_layer(m, li, normed, S, kv, pos_offset, &err);
            residual_add(attn, x, c->residual_multiplier, n);
            memcpy(x, attn, (size_t)n * sizeof(float));
            free(attn);

            po_layernorm(x, L_(PO_T_FFN_NORM_W), L_(PO_T_FFN_NORM_B), c->norm_eps, normed, S, H);
            float* f = fzeros(n);
            if (c->use_moe && c->attention_type == PO_ATTN_GQA) {
                po_moe(normed, L_(PO_T_ROUTER), L_(PO_T_MOE_IN), L_(PO_T_MOE_OUT), S, H,
                       c->num_experts, c->num_experts_per_tok, c->ffn_dim, f);
            } else {
                po_ffn(normed, L_(PO_T_W1), L_(PO_T_B1), L_(PO_T_W2), L_(PO_T_B2), S, H,
                       c->ffn_dim, c->activation_type == PO_ACT_SWIGLU, f);
            }
            residual_add(f, x, c->residual_multiplier, n);
            memcpy(x, f, (size_t)n * sizeof(float));
            free(f);
        }
        if (hidden_out) memcpy(hidden_out + (int64_t)li * n, x, (size_t)n * sizeof(float));
    }

    /* final norm + LM head on ALL rows, generic_model.go:464-477 */
    po_layernorm(x, m->global[PO_T_FINAL_NORM_W].p, m->global[PO_T_FINAL_NORM_B].p, c->norm_eps,
                 normed, S, H);
    const int lrows = g_last_only ? 1 : S;      /* test-time knob, see po_set_lm_head_last_only */
    po_matmul(normed + (int64_t)(S - lrows) * H, m->global[PO_T_LM_HEAD].p, logits_out, lrows, H, V);
    if (c->logits_scaling != 0.0f)
        for (int64_t j = 0; j < (int64_t)lrows * V; j++) logits_out[j] = logits_out[j] / c->logits_scaling;

    free(x); free(normed);
    return err;
}

/* ------------------------------------------------------------------------ */
/* load-time layout contract: purego/tensor/generic_loader.go                */
/* ------------------------------------------------------------------------ */

/* splitGPT2QKV, generic_loader.go:674-702: c_attn [H, 3H] split by COLUMNS */
void po_split_gpt2_qkv(const float* qkv, int hidden, float* q, float* k, float* v) {
    for (int row = 0; row < hidden; row++) {
        const float* src = qkv + (int64_t)row * 3 * hidden;
        memcpy(q + (int64_t)row * hidden, src, (size_t)hidden * sizeof(float));
        memcpy(k + (int64_t)row * hidden, src + hidden, (size_t)hidden * sizeof(float));
        memcpy(v + (int64_t)row * hidden, src + 2 * hidden, (size_t)hidden * sizeof(float));
    }
}

/* splitFalconQKV, generic_loader.go:705-748: per row [Q_0..Q_{nH-1}, K, V] chunks of hd */
void po_split_falcon_qkv(const float* qkv, int hidden, int num_heads, int head_dim,
                         float* q, float* k, float* v) {
    int row_w = (num_heads + 2) * head_dim;
    for (int row = 0; row < hidden; row++) {
        const float* src = qkv + (int64_t)row * row_w;
        for (int h = 0; h < num_heads; h++)
            memcpy(q + (int64_t)row * num_heads * head_dim + (int64_t)h * head_dim,
                   src + (int64_t)h * head_dim, (size_t)head_dim * sizeof(float));
        memcpy(k + (int64_t)row * head_dim, src + (int64_t)num_heads * head_dim,
               (size_t)head_dim * sizeof(float));
        memcpy(v + (int64_t)row * head_dim, src + (int64_t)(num_heads + 1) * head_dim,
               (size_t)head_dim * sizeof(float));
    }
}

/* combineMQAKV, generic_loader.go:751-765 */
void po_combine_mqa_kv(const float* k, const float* v, int hidden, int head_dim, float* kv) {
    for (int i = 0; i < hidden; i++) {
        memcpy(kv + (int64_t)i * 2 * head_dim, k + (int64_t)i * head_dim, (size_t)head_dim * sizeof(float));
        memcpy(kv + (int64_t)i * 2 * head_dim + head_dim, v + (int64_t)i * head_dim,
               (size_t)head_dim * sizeof(float));
    }
}

float po_f32_from_bf16(uint16_t bits) {
    uint32_t u = ((uint32_t)bits) << 16;
    float f; memcpy(&f, &u, 4); return f;
}

float po_f32_from_f16(uint16_t bits) { /* generic_loader.go:774-800 */
    uint32_t sign = (bits >> 15) & 1u, ex = (bits >> 10) & 0x1Fu, frac = bits & 0x3FFu;
    uint32_t r;
    if (ex == 0) {
        if (frac == 0) { r = sign << 31; float f; memcpy(&f, &r, 4); return f; }
        ex = 127 - 14;
        while ((frac & 0x400u) == 0) { frac <<= 1; ex--; }
        frac &= 0x3FFu;
    } else if (ex == 0x1F) {
        ex = 0xFF;
    } else {
        ex += 127 - 15;
    }
    r = (sign << 31) | (ex << 23) | (frac << 13);
    float f; memcpy(&f, &r, 4); return f;
}
