/*
 * movenet_hip.h -- C ABI of the MI355X-native WaveNet decoder path.
 *
 * The reference (cosmicBboy/movenet) is pure Python/PyTorch and has NO native
 * interface; the boundary a drop-in must honour is the Python API of
 * movenet/wavenet.py (WaveNet.forward :158-191, WaveNet.generate :193-239,
 * receptive_fields :125-134).  This header is the C-ABI layer UNDER that API:
 * every entry point below names the reference lines whose arithmetic it
 * replaces.  movenet_amd/_native.py binds it with ctypes; INTEGRATION.md shows
 * the stub a movenet maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (e.g. the PyTorch
 *    caching allocator) unless the parameter is documented as a host array;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *    work is enqueued on it and nothing synchronises the host;
 *  - functions return MVN_OK (0) or a negative MVN_ERR_* code and never throw;
 *    mvn_last_error() returns a host string for the calling thread;
 *  - one host thread per GPU; no internal threads; no hidden allocations (the
 *    sizing helpers (mvn_*_floats) size the caller-provided workspaces).
 *  - tensors are fp32, layouts are the reference's: audio (B, Q, T) one-hot is
 *    represented by its class indices (B, T) int32; weights arrive in the
 *    reference's state_dict layouts.
 */
#ifndef MOVENET_HIP_H
#define MOVENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVN_ABI_VERSION 2  /* r4: mvn_upsample_video_backward takes a scratch buffer; new entry points */

#define MVN_OK 0
#define MVN_ERR_BAD_DIMS (-1)      /* unsupported / inconsistent dimensions   */
#define MVN_ERR_BAD_ARG (-2)       /* NULL pointer, negative size, bad range  */
#define MVN_ERR_TOO_SHORT (-3)     /* T < receptive_fields (ValueError in the
                                      reference, movenet/wavenet.py:141-146)  */
#define MVN_ERR_LAUNCH (-4)        /* HIP launch/runtime error                */
#define MVN_ERR_UNSUPPORTED (-5)   /* variant not available for these dims    */

/* movenet/wavenet.py:75-91 constructor arguments (context_in_channels is used
 * by the video encoder only and is not part of the decoder kernels). */
typedef struct mvn_dims {
  int32_t layer_size;        /* dilations 2^0 .. 2^(layer_size-1) per stack  */
  int32_t stack_size;
  int32_t input_channels;    /* Q: mu-law classes                            */
  int32_t residual_channels; /* C                                            */
  int32_t skip_channels;     /* K                                            */
} mvn_dims;

/* Device pointers to the decoder parameters in the reference's own layouts
 * (movenet/modules.py:19-26, :36-43, :52-65, :136-137).  Per-layer members are
 * HOST arrays of n_layers device pointers.  ctx_* may be NULL (audio-only). */
typedef struct mvn_params {
  const float *causal_w;              /* causal_conv.conv.weight (C,Q,2)     */
  const float *const *filter_w;       /* conv_filter.conv.weight (C,C,2)     */
  const float *const *gate_w;         /* conv_gate.conv.weight   (C,C,2)     */
  const float *const *residual_w;     /* conv_residual.weight    (C,C,1)     */
  const float *const *residual_b;     /* conv_residual.bias      (C)         */
  const float *const *skip_w;         /* conv_skip.weight        (K,C,1)     */
  const float *const *skip_b;         /* conv_skip.bias          (K)         */
  const float *const *ctx_filter_w;   /* context_conv_filter.weight (C,C,1)  */
  const float *const *ctx_filter_b;
  const float *const *ctx_gate_w;     /* context_conv_gate.weight (C,C,1)    */
  const float *const *ctx_gate_b;
  const float *head1_w;               /* dense_conv.conv1.weight (Q,K,1)     */
  const float *head1_b;               /* dense_conv.conv1.bias   (Q)         */
  const float *head2_w;               /* dense_conv.conv2.weight (Q,Q,1)     */
  const float *head2_b;               /* dense_conv.conv2.bias   (Q)         */
} mvn_params;

int mvn_abi_version(void);
/* The library's A/B switches (MOVENET_HIP_* environment variables selecting kernel forms: cross-checks in the
 * tests, same-box comparisons) are parsed once per process at first use; this parses them again. */
int mvn_reload_switches(void);
const char *mvn_last_error(void);

/* movenet/wavenet.py:125-134 (receptive_fields) and :136-147
 * (compute_output_size; returns MVN_ERR_TOO_SHORT where the reference raises
 * ValueError). */
int mvn_receptive_fields(const mvn_dims *dims);
int mvn_output_size(const mvn_dims *dims, int t_len);

/* ------------------------------------------------------------------------
 * Autoregressive generation (replaces the loop of movenet/wavenet.py:217-237:
 * window forward -> [probs/T] -> softmax -> multinomial|argmax -> one-hot
 * scatter) by a ring-buffer ("fast WaveNet") formulation that is
 * result-equivalent: each layer keeps the last `dilation`
 * inputs it saw, one generated sample costs one pass over the weights.
 * ------------------------------------------------------------------------ */

/* Kernel variants: GENERIC handles any dims (C,K <= 256, Q <= 1024); STREAM
 * (C=K=64, Q=256) runs one workgroup per sequence and streams the weights from
 * L2 through double-buffered registers; PIPE see below.                        */
#define MVN_GEN_AUTO 0
#define MVN_GEN_GENERIC 1
#define MVN_GEN_STREAM 2
#define MVN_GEN_PIPE 3 /* C=K in {64,128}, Q=256: layer pipeline over ceil(L/4)+1 (C=64)
                          or L+1 (C=128) CUs, weights resident in registers/LDS,
                          activations handed on as 8-byte granules; needs all stages
                          co-resident, at most 32 per XCD: 24 pipelines at config 2, 4 at
                          config 5 (61 stages, two XCDs each); a pipeline serves up to 8
                          (C=64) / 16 (C=128) sequences in turn within one launch: 192 /
                          64 sequences.  Config 5: 73 us per step for 1 .. 64 sequences.   */

#define MVN_GEN_PIPE_F16 4 /* C=K=128, Q=256: the PIPE structure with FP16 OPERANDS and FP32
                              ACCUMULATION (BASELINE configs[4]; precedent: torch.autocast,
                              movenet/trainer.py:124): weights and every product's vector
                              operand rounded to fp16, sums in fp32 (v_mfma_f32_16x16x32_f16 in
                              the layer stages, v_dot2c_f32_f16 in the head); two layers
                              per stage, ceil(L/2)+1 stages (31 for 60 layers: one XCD): 8
                              pipelines per launch, each serving up to 8 sequences in turn (64
                              per launch).  Never chosen by MVN_GEN_AUTO: fp32 is the default.   */

#define MVN_GEN_FOLD 5 /* C=K=64, Q=256: the PIPE structure with the residual 1x1 of layer j folded
                          into the filter/gate matrix of layer j+1 (products formed at pack
                          time): ONE dependent mat-vec + gate per layer instead of two; three
                          layers per stage, ceil(L/3)+1 stages (11 for 30 layers): 16 pipelines
                          inside XCDs + 7 across them, each serving up to 8 sequences in turn
                          within one launch (184 sequences).  Config 2: 14.7 us per step for 16
                          sequences (PIPE: 17.5), 15.4 for 64, 15.9 for 128, 21.1 for 184 (STREAM:
                          79): what MVN_GEN_AUTO runs whenever it holds the batch.              */

/* Resolve MVN_GEN_AUTO for `dims` and `batch` sequences per launch; returns the
 * variant or a negative error.  The packed weight layout depends on the variant:
 * pack and generate must be given the same resolved value. */
int mvn_gen_variant(const mvn_dims *dims, int requested, int batch);

/* Pipelines (workgroup chains with resident weights) a launch of `batch` sequences of the
 * pipelined `variant` (PIPE, PIPE_F16, FOLD) runs on -- its sequences take turns on them,
 * ceil(batch / pipelines) each; 0 for the other variants, negative on bad dims.  FOLD at
 * config 2: `batch` up to 16, 16 up to 80 sequences (whole pipelines inside one XCD: the
 * pipeline's latency bounds the step), 23 beyond (plus seven across XCDs: the step is
 * rounds x the stages' service time there).  For cost models (generation.auto_plan). */
int mvn_gen_launch_pipelines(const mvn_dims *dims, int variant, int batch);

/* 1 when the pipelined generators (MVN_GEN_PIPE / _PIPE_F16 / _FOLD) of this process are launched with
 * hipLaunchCooperativeKernel -- the default: the runtime guarantees the co-residency their hand-offs
 * need -- 0 when with an ordinary launch: a profiler's tool library is attached (rocprofv3: the
 * cooperative queue's tear-down crashes the process at exit under rocprofiler-sdk) or
 * MOVENET_PIPE_COOPERATIVE_LAUNCH=0 asks for it.  Decided once per process. */
int mvn_gen_launch_is_cooperative(void);

/* Size in floats of the packed weight blob / of the generator state (dilation
 * queues, plus the PIPE variant's hand-off area when the dims allow PIPE). */
size_t mvn_gen_weights_floats(const mvn_dims *dims, int variant);
size_t mvn_gen_state_floats(const mvn_dims *dims, int batch);

/* Float offset, inside the state, of the PIPE variant's status word (int32, zero =
 * no hand-off timed out since the state was last zeroed; read it after synchronising
 * the stream); (size_t)-1 when the dims have no PIPE hand-off area. */
size_t mvn_gen_status_offset(const mvn_dims *dims, int batch);

/* state_dict layouts -> the variant's streaming layout (done once per weight
 * update; DESIGN.md "Data layout in HBM"). */
int mvn_gen_pack_weights(const mvn_dims *dims, int variant, const mvn_params *params,
                         float *packed, void *stream);

/* Advance every sequence over the time steps t in [t_begin, t_end).
 *
 *  samples      (batch, sample_stride) int32 class indices.  Step t consumes
 *               samples[b][t] and predicts time t+1; when t+1 >= n_given and
 *               t+1 < n_total the chosen class is written to samples[b][t+1].
 *               Columns < n_given are never written (prompt / teacher forcing).
 *  state        mvn_gen_state_floats() floats; must be zero before t = 0 and is
 *               carried between calls (calls must cover t contiguously from 0).
 *  temperature  > 0: sample from softmax(softmax(logits)/T) (the reference's
 *               double softmax, movenet/wavenet.py:227-231); <= 0: first argmax
 *               of softmax(softmax(logits)) (:233).
 *  seed         Philox4x32-10 key; the draw for (b, t) does not depend on the
 *               launch partition.
 *  logits_out   optional (batch, n_total - logits_t0, Q): raw head output
 *               predicting time u is stored at [b][u - logits_t0] for u >= logits_t0.
 *  choices_out  optional (batch, n_total) int32: the class the model picks for
 *               time u, also where samples[] is teacher-forced (u >= logits_t0).
 *  context_tm   optional local conditioning, see mvn_transpose_context below.
 */
int mvn_generate(const mvn_dims *dims, int variant, const float *packed, float *state,
                 int32_t *samples, int batch, int sample_stride, int n_total, int n_given,
                 int t_begin, int t_end, float temperature, uint64_t seed,
                 float *logits_out, int32_t *choices_out, int logits_t0,
                 const float *context_tm, void *stream);

/* Local conditioning in generation (BUILD DEFINITION, the reference raises: SURVEY.md
 * Q7): step t adds the context column of time t to every layer's filter/gate sums.
 * context_tm is (batch, n_total, C) TIME-major (one coalesced 4C-byte read per step);
 * mvn_transpose_context builds it from the (batch, C, ctx_ld) output of
 * mvn_upsample_video.  GENERIC and PIPE variants only. */
int mvn_transpose_context(const float *ctx, int ctx_ld, int batch, int channels, int t_len,
                          float *context_tm, void *stream);

/* ------------------------------------------------------------------------
 * Full-sequence forward (replaces movenet/wavenet.py:166-191: causal conv ->
 * L gated residual layers -> sum of skips -> dense head -> [drop last] ->
 * [softmax]) and its backward.  Used for training, for forward() and to prime
 * the generator's queues from a prompt.
 *
 * Activations live on an ABSOLUTE time axis: every per-layer tensor is
 * (batch, channels, Tp) with Tp = mvn_padded_len(T) and column t = input time
 * t; layer l's input is valid for t >= A_l (A_0 = 0, A_{l+1} = A_l + d_l), so
 * the reference's right-aligned slices (modules.py:84, :91) become "same t".
 * Skip/head tensors are (batch, channels, Sp), Sp = mvn_padded_len(S + 31),
 * S = T - RF + 1; the column of time t is t - ((RF-1) & ~31), i.e. the S valid
 * columns start at column (RF-1) & 31, so that column == t (mod 32): 16-byte
 * accesses and the 128-byte cache lines line up on both time axes.
 * ------------------------------------------------------------------------ */
int mvn_padded_len(int n); /* n rounded up to a multiple of 64 */

/* Caller-provided device buffers (floats).  n_act = save ? L+1 : 2. */
typedef struct mvn_fwd_buffers {
  float *acts;  /* n_act x (B, C, Tp): layer inputs; acts[0] = causal conv out  */
  float *th;    /* save: L x (B, C, Tp) tanh(f);      else NULL                 */
  float *sg;    /* save: L x (B, C, Tp) sigmoid(g);   else NULL                 */
  float *z;     /* (B, C, Tp) scratch: the gated activation of the unfused layer
                   kernels; the fused paths keep z on chip and write the layers'
                   packed weight images here (fused_layer.h, fused_fwd_bf3.h)        */
  float *skip;  /* (B, K, Sp) sum of skips                                      */
  float *a1;    /* (B, Q, Sp) head hidden activation lrelu(conv1(lrelu(skip)))  */
  const float *ctx; /* optional local conditioning (B, C, ctx_ld), column t = time t
                       (output of mvn_upsample_video); NULL = audio only.  BUILD
                       DEFINITION of the alignment: the reference raises at
                       modules.py:75-77                          */
  int32_t ctx_ld;
  const float *dense_audio; /* optional (B, Q, dense_ld) fp32 input that is NOT one-hot: the
                               causal conv then runs as a dense product and `index` is
                               ignored (may be NULL); NULL = use `index`                */
  int32_t dense_ld;
} mvn_fwd_buffers;

/* out: (B, Q, S_out) contiguous, S_out = S - (remove_last ? 1 : 0); softmax over
 * Q when `normalize` (the reference's inverted flag output_unnormalized=True,
 * wavenet.py:189-191).  save != 0 keeps what mvn_backward needs in `buf`. */
int mvn_forward(const mvn_dims *dims, const mvn_params *params, const int32_t *index,
                int index_stride, int batch, int t_len, const mvn_fwd_buffers *buf, float *out,
                int normalize, int remove_last, int save, void *stream);

/* mvn_forward with FP16 OPERANDS and FP32 ACCUMULATION in every product of the layers and
 * the head (BASELINE configs[4] "fp16 MFMA 1x1 convs"; reference precedent for reduced
 * precision: torch.autocast, movenet/trainer.py:124): weights and the activations entering a
 * product are rounded to fp16 as they are staged, multiplied by v_mfma_f32_32x32x16_f16 and
 * summed in fp32; the causal conv's gather, biases, gating, residual adds, the skip sum, the
 * softmax and every tensor in HBM stay fp32 -- the rounding points of MVN_GEN_PIPE_F16.  Same
 * arguments and buffers as mvn_forward; `save` keeps th/sg/acts, but mvn_backward
 * differentiates the fp32 forward: train in fp32 (the default). */
int mvn_forward_f16(const mvn_dims *dims, const mvn_params *params, const int32_t *index,
                    int index_stride, int batch, int t_len, const mvn_fwd_buffers *buf, float *out,
                    int normalize, int remove_last, int save, void *stream);

/* Gradients, same layouts as mvn_params (members may not be NULL except ctx_*);
 * mvn_backward ACCUMULATES into them (zero them first for a fresh gradient). */
typedef struct mvn_param_grads {
  float *causal_w;
  float *const *filter_w;
  float *const *gate_w;
  float *const *residual_w;
  float *const *residual_b;
  float *const *skip_w;
  float *const *skip_b;
  float *head1_w;
  float *head1_b;
  float *head2_w;
  float *head2_b;
  float *const *ctx_filter_w; /* only used (and required) when fwd->ctx != NULL */
  float *const *ctx_filter_b;
  float *const *ctx_gate_w;
  float *const *ctx_gate_b;
} mvn_param_grads;

/* Scratch for the backward pass (floats). */
typedef struct mvn_bwd_buffers {
  float *dx_a;   /* (B, C, Tp) */
  float *dx_b;   /* (B, C, Tp) */
  float *dfg;    /* (B, 2C, Tp) */
  float *dskip;  /* (B, K, Sp) */
  float *da1;    /* (B, Q, Sp) */
  float *dlogit; /* (B, Q, Sp) */
  float *dctx;   /* (B, C, Tp) gradient w.r.t. fwd->ctx (written); NULL when audio only */
} mvn_bwd_buffers;

/* dout: gradient w.r.t. mvn_forward's `out` (same shape); `out` itself is needed
 * when normalize != 0 (softmax backward).  dout == NULL: the caller has already written the
 * gradient w.r.t. the LOGITS into bwd->dlogit (layout (B, Q, Sp), the column of output
 * position s is s + ((RF-1) & 31), columns of positions >= S_out zero) -- what
 * mvn_softmax_ce_backward does.  Requires the forward ran with save.
 * Part of the work is enqueued on a stream the library owns (one per device) and joined
 * back into `stream` with events before the call returns: to the caller it is ordinary
 * stream-ordered work.  da1 / dlogit / dfg double as scratch of the weight gradients. */
int mvn_backward(const mvn_dims *dims, const mvn_params *params, const mvn_param_grads *grads,
                 const int32_t *index, int index_stride, int batch, int t_len,
                 const mvn_fwd_buffers *fwd, const mvn_bwd_buffers *bwd, const float *out,
                 const float *dout, int normalize, int remove_last, void *stream);

/* Diagnostic: which form the layer loop of the process's last mvn_backward took -- 0 none yet,
 * 1 the generic two-kernel forms (any dims), 2 the two fused halves per layer (C = K = 64,
 * csrc/fused_bwd.h), 3 ONE kernel per layer with the input gradient in scatter form
 * (csrc/fused_bwd_l.h, the default at C = K = 64).  Tests assert on it so that a silent
 * fall-back to a slower form cannot pass for the fast one. */
#define MVN_BWD_FORM_GENERIC 1
#define MVN_BWD_FORM_HALVES 2
#define MVN_BWD_FORM_ONE 3
int mvn_last_backward_form(void);

/* ------------------------------------------------------------------------
 * Local conditioning: video encoder + learned upsampler (movenet/wavenet.py:94-118,
 * :149-156).  video (B, F, 64, 64, Cin) fp32 -> Conv3d(k=(1,64,64)) -> (B, C, F)
 * -> 3 x ConvTranspose1d(k=10, stride=10) -> ctx (B, C, 1000 F).
 * Intermediates are caller-provided (B, C, mvn_padded_len(n)) buffers with
 * n = F, 10 F, 100 F; ctx has row stride ctx_ld >= 1000 F.
 * ------------------------------------------------------------------------ */
typedef struct mvn_video_params {
  const float *conv_w;   /* video_conv.weight (C, Cin, 1, 64, 64) */
  const float *conv_b;   /* video_conv.bias   (C)                 */
  const float *up_w[3];  /* video_transpose.{0,1,2}.weight (C, C, 10) = (in, out, tap) */
  const float *up_b[3];  /* video_transpose.{0,1,2}.bias   (C)    */
} mvn_video_params;

typedef struct mvn_video_grads { /* accumulated into */
  float *conv_w;
  float *conv_b;
  float *up_w[3];
  float *up_b[3];
} mvn_video_grads;

int mvn_upsample_video(const mvn_dims *dims, const mvn_video_params *vp, const float *video,
                       int batch, int frames, int cin, float *enc, float *u1, float *u2, float *ctx,
                       int ctx_ld, void *stream);

/* d_u2, d_u1, d_enc: scratch of the same shapes as u2, u1, enc.  `scratch` (r4): room for the per-workgroup
 * weight-gradient slabs of the up-sampler's backward kernel (C = 64), mvn_upsample_video_scratch_floats(dims, batch,
 * frames) floats; NULL or too small: the gradients are added with atomics instead (correct, ~5x slower, and the
 * summation order then varies from run to run). */
size_t mvn_upsample_video_scratch_floats(const mvn_dims *dims, int batch, int frames);
int mvn_upsample_video_backward(const mvn_dims *dims, const mvn_video_params *vp,
                                const mvn_video_grads *vg, const float *video, int batch, int frames,
                                int cin, const float *enc, const float *u1, const float *u2,
                                const float *dctx, int dctx_ld, float *d_u2, float *d_u1,
                                float *d_enc, float *scratch, size_t scratch_floats, void *stream);

/* Fill the generator's dilation queues from a saved forward (acts of a prompt
 * of t_len >= RF samples): equivalent to mvn_generate over t in [0, t_len-1). */
int mvn_gen_prime_from_forward(const mvn_dims *dims, const mvn_fwd_buffers *fwd, int batch,
                               int t_len, float *state, void *stream);

/* (B,Q,T) one-hot fp32 <-> (B,T) int32 indices: movenet/dataset.py:285-288 and
 * the scatter at movenet/wavenet.py:235-237.  onehot_to_index writes -1 where a
 * column is not exactly one-hot (the host wrapper then refuses the input). */
int mvn_onehot_to_index(const float *onehot, int32_t *index, int batch, int classes, int t_len,
                        void *stream);
int mvn_index_to_onehot(const int32_t *index, int index_stride, float *onehot, int batch,
                        int classes, int t_len, void *stream);

/* Plumbing for the host wrappers (no reference counterpart): publish up to 64 device 32-bit words
 * to the HOST without a copy engine or a stream synchronisation -- the one-hot validation's
 * minimum index (the reference feeds one-hot tensors to a dense conv and never looks) and the
 * trainer's per-step log scalars (loss, accuracy, gradient norm).  `host_words` points to n + 1
 * words in pinned, device-visible host memory (hipHostMalloc / torch pin_memory()): one wave on
 * `stream` stores words[0..n) to host_words[0..n), then `seq` to host_words[n] (system-scope
 * release); the host polls host_words[n] == seq and so waits for exactly the kernels queued in
 * front of this one -- a stream-synchronising read (tensor.item()) waits for everything the
 * caller has enqueued since. */
int mvn_publish_words(const uint32_t *words, int n, int32_t seq, uint32_t *host_words, void *stream);

/* The trainer's loss and accuracy on the model output (row F3 of SURVEY.md section 8;
 * movenet/pytorch_lightning_trainer.py:64-66): cross_entropy applied to PROBABILITIES
 * (SURVEY Q2), i.e.  loss = mean over (b,s) of  logsumexp_q(probs[b,:,s]) - probs[b,target,s],
 * accuracy = mean of [first argmax_q probs[b,:,s] == target[b,s]].
 *   probs (B, Q, S) fp32, target (B, S) int64 class indices.
 * forward: per-workgroup partial sums, loss_part / correct_part have mvn_ce_parts(B,S)
 *   entries each, ZEROED by the caller (not every slot is written for every shape) and
 *   summed by it in a fixed order: deterministic.
 * backward: dprobs = scale * upstream * (softmax_q(probs) - onehot(target)); scale = 1/(B*S) for
 *   the mean, `upstream` = device pointer to the scalar gradient of the loss (NULL: 1), read
 *   by the kernel so that the host never has to synchronise for it. */
int mvn_ce_parts(int batch, int s_len);
int mvn_ce_on_probs_forward(const float *probs, const long long *target, int batch, int classes,
                            int s_len, float *loss_part, int32_t *correct_part, void *stream);
int mvn_ce_on_probs_backward(const float *probs, const long long *target, int batch, int classes,
                             int s_len, float scale, const float *upstream, float *dprobs,
                             void *stream);

/* The same loss and accuracy FUSED with the model's final softmax (wavenet.py:189-191):
 * forward turns the head's logits (B, Q, S) into probabilities IN PLACE and accumulates the
 * loss / accuracy partial sums of mvn_ce_on_probs_forward in the same pass (same slots, same
 * bits); backward writes d loss / d logits -- through cross_entropy's log-softmax AND the
 * model's softmax -- into a (B, Q, dlogit_ld) tensor at columns dlogit_col0 .. +dlogit_cols
 * (zero for the dlogit_cols - s_len trailing columns), i.e. straight into mvn_backward's
 * dlogit buffer (call mvn_backward with dout = NULL). */
int mvn_softmax_ce_forward(float *logits_probs, const long long *target, int batch, int classes,
                           int s_len, float *loss_part, int32_t *correct_part, void *stream);
int mvn_softmax_ce_backward(const float *probs, const long long *target, int batch, int classes,
                            int s_len, float scale, const float *upstream, float *dlogit,
                            long long dlogit_batch_stride, int dlogit_ld, int dlogit_col0,
                            int dlogit_cols, void *stream);

/* One optimizer step of torch.optim.AdamW (decoupled != 0) or torch.optim.Adam (decoupled == 0,
 * weight decay added to the gradient) over flat fp32 buffers of n elements
 * (movenet/pytorch_lightning_trainer.py:186-189; arithmetic of torch's _single_tensor_adam,
 * amsgrad off).  step counts from 1.  skip_ranges: HOST array of n_skip <= 4 [lo, hi) element
 * ranges left untouched (parameters without a gradient this step). */
int mvn_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   int decoupled, const size_t *skip_ranges, int n_skip, void *stream);

/* mu-law companding either side of the model (movenet/dataset.py:278-289 encode ->
 * one-hot; movenet/callbacks.py:66-76 argmax -> decode).  The reference calls torchaudio,
 * which is absent offline and pinned by no fixture: formula of RESEARCH.md:156-163,
 * PARITY UNPINNED.  Shipping indices instead of (B,Q,T) one-hot floats removes a
 * 256x larger host->device copy. */
int mvn_mu_law_encode(const float *x, int32_t *index, size_t n, int classes, void *stream);
int mvn_mu_law_decode(const int32_t *index, float *x, size_t n, int classes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MOVENET_HIP_H */
