/*
 * ring_oracle.c -- CPU restatement in plain C of the CACHED (ring-buffer, "fast
 * WaveNet") form of the reference's autoregressive generation.  TEST INFRASTRUCTURE
 * ONLY: built by oracle/Makefile into oracle/_build/libring_oracle.so and loaded only
 * by tests/ and by bench.py's cpu baselines (oracle/ring_c.py); nothing under
 * movenet_amd/ links, loads or calls it.
 *
 * What it restates (paths relative to /root/reference):
 *   movenet/wavenet.py:217-237   the per-sample loop of WaveNet.generate, greedy branch
 *                                (:189-191 softmax, :233 softmax again -> argmax)
 *   movenet/modules.py:19-30     causal conv on a one-hot input = two weight columns
 *   movenet/modules.py:36-46,:73 dilated k=2 convs: tap 0 on x[t-d] (the queue), tap 1 on x[t]
 *   movenet/modules.py:80        z = tanh(f) * sigmoid(g)
 *   movenet/modules.py:83-91     residual 1x1 + bias + input; skip 1x1 + bias, summed
 *   movenet/modules.py:139-142   head: conv2(lrelu(conv1(lrelu(skip))))
 * The reference re-runs the whole network on an RF-long window per sample (Q4); this
 * file keeps, per layer, the last d_l inputs instead.  Equivalence of the two forms is
 * shown on CPU by tests/test_oracle_golden.py (numpy twin: oracle/wavenet_oracle.py
 * RingState) and this file is pinned against the reference's own greedy output (fixture
 * G3) and logits (G2) by tests/test_ring_c_oracle.py.
 *
 * Parity status: PINNED (audio-only path) through those fixtures.
 *
 * Layout of the weight blob `w` (floats, built by oracle/ring_c.py from the state_dict;
 * every matrix is stored [in][out] so that the inner loop runs over contiguous outputs):
 *   E0[Q][C] E1[Q][C]                                  causal conv taps 0 / 1
 *   per layer: F0[C][C] F1[C][C] G0[C][C] G1[C][C]     filter / gate, tap 0 | tap 1
 *              R[C][C] br[C] S[C][K] bs[K]
 *   W1[K][Q] b1[Q] W2[Q][Q] b2[Q]
 */
#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LEAKY 0.01f

typedef struct {
  int layer_size, stack_size, Q, C, K, L, batch;
  const float *w;
  float *rings; /* [batch][sum_l d_l][C] */
  long ring_per_seq;
  int t;
  int *prev; /* [batch], -1 = none (zero padding of the causal conv) */
  int half_operands; /* fp16-operand / fp32-accumulate form (BASELINE configs[4]): the vector
                        operand of every product is rounded to fp16 (nearest even); the caller
                        packs weight matrices already rounded (oracle/ring_c.py) */
} ring_t;

void ro_set_half_operands(void *h, int on) { ((ring_t *)h)->half_operands = on; }

/* x -> fp16 -> fp32 (F16C, round to nearest even): the operand rounding of the fp16 form */
static inline void round_operand(const ring_t *r, const float *x, float *y, int n) {
  if (!r->half_operands) {
    memcpy(y, x, sizeof(float) * n);
    return;
  }
  for (int i = 0; i < n; ++i) y[i] = _cvtsh_ss(_cvtss_sh(x[i], _MM_FROUND_TO_NEAREST_INT));
}

static long layer_floats(int C, int K) { return 5L * C * C + C + (long)C * K + K; }

long ro_weight_floats(int layer_size, int stack_size, int Q, int C, int K) {
  return 2L * Q * C + (long)layer_size * stack_size * layer_floats(C, K) + (long)K * Q + Q +
         (long)Q * Q + Q;
}

void *ro_create(int layer_size, int stack_size, int Q, int C, int K, int batch, const float *w) {
  ring_t *r = (ring_t *)calloc(1, sizeof(ring_t));
  if (!r) return NULL;
  r->layer_size = layer_size; r->stack_size = stack_size; r->Q = Q; r->C = C; r->K = K;
  r->L = layer_size * stack_size; r->batch = batch; r->w = w;
  r->ring_per_seq = (long)stack_size * ((1L << layer_size) - 1) * C;
  r->rings = (float *)calloc((size_t)batch * r->ring_per_seq, sizeof(float));
  r->prev = (int *)malloc(sizeof(int) * batch);
  if (!r->rings || !r->prev) return NULL;
  for (int b = 0; b < batch; ++b) r->prev[b] = -1;
  return r;
}

void ro_destroy(void *h) {
  ring_t *r = (ring_t *)h;
  if (!r) return;
  free(r->rings);
  free(r->prev);
  free(r);
}

/* y[o] += sum_k x[k] * W[k][o]  (W stored [in][out]) */
static inline void matvec_acc(const float *restrict W, const float *restrict x, float *restrict y,
                              int nin, int nout) {
  for (int k = 0; k < nin; ++k) {
    const float xv = x[k];
    const float *restrict wr = W + (long)k * nout;
    for (int o = 0; o < nout; ++o) y[o] += xv * wr[o];
  }
}

/* one sequence, one time step: consume class `idx` at time t, write Q logits for t+1 */
static void step_one(ring_t *r, int b, int idx, float *logits) {
  const int C = r->C, K = r->K, Q = r->Q;
  float h[512], past[512], f[512], g[512], z[512], skip[512], a0[512], a1[2048], ho[512];
  const float *E0 = r->w, *E1 = r->w + (long)Q * C;
  for (int c = 0; c < C; ++c) h[c] = E1[(long)idx * C + c];
  if (r->prev[b] >= 0)
    for (int c = 0; c < C; ++c) h[c] += E0[(long)r->prev[b] * C + c];
  memset(skip, 0, sizeof(float) * K);
  const float *lw = r->w + 2L * Q * C;
  float *ring = r->rings + (long)b * r->ring_per_seq;
  long off = 0;
  for (int l = 0; l < r->L; ++l) {
    const int d = 1 << (l % r->layer_size);
    const float *F0 = lw, *F1 = F0 + (long)C * C, *G0 = F1 + (long)C * C, *G1 = G0 + (long)C * C;
    const float *R = G1 + (long)C * C, *br = R + (long)C * C, *S = br + C, *bs = S + (long)C * K;
    float *slot = ring + off + (long)(r->t % d) * C;
    round_operand(r, slot, past, C);
    memcpy(slot, h, sizeof(float) * C);
    round_operand(r, h, ho, C);
    memset(f, 0, sizeof(float) * C);
    memset(g, 0, sizeof(float) * C);
    matvec_acc(F0, past, f, C, C);
    matvec_acc(F1, ho, f, C, C);
    matvec_acc(G0, past, g, C, C);
    matvec_acc(G1, ho, g, C, C);
    for (int c = 0; c < C; ++c) z[c] = tanhf(f[c]) * (1.0f / (1.0f + expf(-g[c])));
    round_operand(r, z, z, C);
    for (int k = 0; k < K; ++k) skip[k] += bs[k];
    matvec_acc(S, z, skip, C, K);
    for (int c = 0; c < C; ++c) h[c] += br[c];
    matvec_acc(R, z, h, C, C);
    lw += layer_floats(C, K);
    off += (long)d * C;
  }
  const float *W1 = lw, *b1 = W1 + (long)K * Q, *W2 = b1 + Q, *b2 = W2 + (long)Q * Q;
  for (int k = 0; k < K; ++k) a0[k] = skip[k] > 0.f ? skip[k] : LEAKY * skip[k];
  round_operand(r, a0, a0, K);
  for (int q = 0; q < Q; ++q) a1[q] = b1[q];
  matvec_acc(W1, a0, a1, K, Q);
  for (int q = 0; q < Q; ++q) a1[q] = a1[q] > 0.f ? a1[q] : LEAKY * a1[q];
  round_operand(r, a1, a1, Q);
  for (int q = 0; q < Q; ++q) logits[q] = b2[q];
  matvec_acc(W2, a1, logits, Q, Q);
  r->prev[b] = idx;
}

/* Greedy generation.  samples (batch, n_total) int32: columns < n_given are the prompt /
 * teacher-forced history, the rest is written.  choices_out (batch, n_total) optional: what
 * the model picks for time u (also where teacher-forced).  logits_out optional
 * (batch, n_total - logits_t0, Q): raw head output predicting time u >= logits_t0.
 * Steps t in [t_begin, t_end) (call with contiguous ranges starting at 0).  Returns 0. */
int ro_generate(void *h, int *samples, int n_total, int n_given, int t_begin, int t_end,
                int *choices_out, float *logits_out, int logits_t0, int nthreads) {
  ring_t *r = (ring_t *)h;
  if (!r || r->C > 512 || r->K > 512 || r->Q > 2048 || r->t != t_begin) return -1;
  const int Q = r->Q;
  for (int t = t_begin; t < t_end; ++t) {
    const int u = t + 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < r->batch; ++b) {
      float lg[2048];
      int idx = samples[(long)b * n_total + t];
      if (idx < 0) idx = 0;
      if (idx >= Q) idx = Q - 1;
      step_one(r, b, idx, lg);
      if (u < n_total) {
        /* wavenet.py:189-191 softmax, then :233 softmax again and argmax (first maximum:
         * torch.argmax's tie rule) */
        float p[2048], m = lg[0], sum = 0.f;
        for (int q = 1; q < Q; ++q) m = lg[q] > m ? lg[q] : m;
        for (int q = 0; q < Q; ++q) { p[q] = expf(lg[q] - m); sum += p[q]; }
        float m2 = 0.f, sum2 = 0.f;
        for (int q = 0; q < Q; ++q) { p[q] /= sum; m2 = p[q] > m2 ? p[q] : m2; }
        for (int q = 0; q < Q; ++q) { p[q] = expf(p[q] - m2); sum2 += p[q]; }
        int best = 0;
        for (int q = 0; q < Q; ++q) p[q] /= sum2;
        for (int q = 1; q < Q; ++q)
          if (p[q] > p[best]) best = q;
        if (choices_out && u >= logits_t0) choices_out[(long)b * n_total + u] = best;
        if (logits_out && u >= logits_t0)
          memcpy(logits_out + ((long)b * (n_total - logits_t0) + (u - logits_t0)) * Q, lg,
                 sizeof(float) * Q);
        if (u >= n_given) samples[(long)b * n_total + u] = best;
      }
    }
    r->t = t + 1;
  }
  return 0;
}
