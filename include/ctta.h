/*
 * ctta.h -- C ABI of the MI355X-native ConsistencyTTA hot path (libctta_hip.so).
 *
 * The reference (Bai-YT/ConsistencyTTA) has no FFI: its "plugin boundary" for this path is
 * the PyTorch nn.Module protocol.  Each entry point below replaces one of
 * those module calls and cites it.  All pointers are raw DEVICE pointers unless marked
 * host; all calls are asynchronous on `stream` (a hipStream_t passed as void*), borrow
 * their arguments for the duration of the call only, and return a ctta_status (no
 * exceptions cross the boundary; ctta_last_error() gives the message for this thread).
 * Handles own their pre-packed bf16 weights and activation arena and are freed by
 * *_destroy.  One handle per (device, model); calls on one handle must not overlap.
 *
 * Layout contract at the boundary: fp32, the reference's own layouts (NCHW latents/mel,
 * (B,L,X) text states, (B,L) uint8 mask).  Internally activations are NHWC bf16 with fp32
 * accumulation (the reference's own GPU recipe is bf16 autocast, inference.py:190).
 */
#ifndef CTTA_H
#define CTTA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  CTTA_OK = 0,
  CTTA_ERR_INVALID = 1,     /* bad argument / unsupported shape (reference: ValueError/assert) */
  CTTA_ERR_HIP = 2,         /* a HIP runtime call failed */
  CTTA_ERR_MISSING_KEY = 3, /* weight table lacks a state-dict key (reference: load_state_dict) */
  CTTA_ERR_NOMEM = 4
} ctta_status;

const char* ctta_last_error(void);
int ctta_version(void);

/* Library options -- the only switches of the library (nothing in it reads the environment; there is no other global state
 * than this table and the per-thread bindings documented below).  Process-wide plain ints, read by the call that uses
 * them; set them before the handles they concern are created or between calls, not while another host thread is inside
 * the library.  Names, defaults and what each one trades:
 *   "xcd"          1  blockIdx -> tile order that keeps the tiles sharing operands on one XCD's L2 (0: plain 2-D grids)
 *   "splitk"       1  deep thin GEMMs (few output tiles, long K) are cut over K: two-pass split-K or stream-K (0: never)
 *   "streamk"      1  ... as ONE persistent launch that folds its partial tiles itself, where the tile rules choose it
 *                     (0: two-pass split-K only); needs a workspace with a zeroed header (ctta_conv_bind_workspace_ex)
 *   "streamk_grid" 0  tuning: workgroups of a stream-K launch (0: one per CU slot)
 *   "wgrad_stream" 1  weight-gradient launches of ctta_unet_backward* go to the handle's side stream (0: everything on the
 *                     caller's stream -- what a per-launch profile needs; read at every backward call)
 *   "gn_fuse"      1  GroupNorm statistics come from the producing convolution's epilogue (= ctta_set_gn_fuse)
 *   "fused_res"    1  HiFi-GAN ResBlock units run as fused pair kernels (0: one conv_gemm launch per convolution)
 *   "ffn_fuse"     1  the inference forward runs the 256-wide transformer feed-forward as one row-tile kernel
 *                     (ctta_ffn_geglu; 0: ff1 + GEGLU epilogue and ff2 as two conv_gemm launches); read when a handle is built
 * Every option is exercised in its non-default position by tests/test_options_gpu.py. */
ctta_status ctta_set_option(const char* name, int value);
ctta_status ctta_get_option(const char* name, int* value);
int ctta_num_options(void);
const char* ctta_option_name(int i);
int ctta_option_default(int i);

/* One entry of a state dict: reference key name, fp32 device data, shape. */
typedef struct {
  const char* name;
  const float* data;
  int ndim;
  int64_t shape[4];
} ctta_tensor;

/* ------------------------------------------------------------------------------------ *
 * U-Net.  Replaces UNet2DConditionGuidedModel.forward
 * (diffusers/models/unet_2d_condition_guided.py:716-945) and, with guided=0, the teacher
 * UNet2DConditionModel.forward (unet_2d_condition.py:668-907).
 * ------------------------------------------------------------------------------------ */
#define CTTA_MAX_LEVELS 4
typedef struct {
  int in_channels, out_channels;
  int n_levels;                             /* len(block_out_channels), <= 4 */
  int block_out_channels[CTTA_MAX_LEVELS];
  int heads[CTTA_MAX_LEVELS];               /* config "attention_head_dim" = head COUNT */
  int layers_per_block[CTTA_MAX_LEVELS];
  int down_cross[CTTA_MAX_LEVELS];          /* 1: CrossAttnDownBlock2D, 0: DownBlock2D */
  int up_cross[CTTA_MAX_LEVELS];            /* 1: CrossAttnUpBlock2D,   0: UpBlock2D   */
  int cross_attention_dim;
  int norm_num_groups;
  float norm_eps;
  int flip_sin_to_cos;
  float freq_shift;
  int guided;                               /* 1: guidance Fourier branch present */
  int max_batch, height, width, max_text_len; /* arena sizing */
  int debug_taps;                           /* 1: keep every named intermediate (tests) */
  int enable_training;                      /* 1: also build the data-gradient packs and size the arena for
                                               ctta_unet_forward_train + ctta_unet_backward */
} ctta_unet_config;

typedef struct ctta_unet ctta_unet;

ctta_status ctta_unet_create(const ctta_unet_config* cfg, const ctta_tensor* weights,
                             int n_weights, void* stream, ctta_unet** out);
void ctta_unet_destroy(ctta_unet* h);
/* Re-reads (re-packs) all weights from a new table, e.g. after an optimizer/EMA step. */
ctta_status ctta_unet_load_weights(ctta_unet* h, const ctta_tensor* weights, int n_weights,
                                   void* stream);
/* sample (B,C,H,W) f32; timesteps (B) f32; guidance (B) f64 or NULL when !guided;
 * enc (B,L,X) f32; mask (B,L) u8 (1 = keep) or NULL; out (B,Cout,H,W) f32. */
/* One-shot hint for the NEXT ctta_unet_forward / _forward_train on this handle: reuse != 0 promises that its
 * encoder_hidden_states and encoder_attention_mask (and batch / text_len) are those of the previous forward on the handle,
 * so the cross-attention K / V projections of the text states (attention_processor.py:1107-1111: `to_k` / `to_v` of
 * encoder_hidden_states, 32 small GEMMs per forward) are taken from the handle's text cache instead of recomputed -- the
 * second CFG teacher query of a distillation step, every query of a Heun teacher loop.  Ignored when nothing valid is cached
 * (first forward, other batch / length, after ctta_unet_load_weights).  Results are bit-identical either way. */
ctta_status ctta_unet_reuse_text(ctta_unet* h, int reuse);
ctta_status ctta_unet_forward(ctta_unet* h, const float* sample, const float* timesteps,
                              const double* guidance, const float* enc, const uint8_t* mask,
                              int batch, int text_len, float* out, void* stream);
size_t ctta_unet_arena_bytes(const ctta_unet* h);
/* Distillation step (train.py:332-346: loss = model(...); accelerator.backward(loss)).
 * forward_train = forward that keeps every activation the backward pass needs; backward takes
 * dL/d(out) as NHWC bf16 [B][H*W][8] (ctta_snr_mse_grad writes exactly that) and ACCUMULATES
 * dL/d(param) into `grads`: fp32 tensors with the names / shapes of the state dict (`.grad`
 * semantics; guidance_proj.weight has requires_grad=False in the reference and gets none). */
ctta_status ctta_unet_forward_train(ctta_unet* h, const float* sample, const float* timesteps,
                                    const double* guidance, const float* enc, const uint8_t* mask,
                                    int batch, int text_len, float* out, void* stream);
ctta_status ctta_unet_backward(ctta_unet* h, const void* dout_nhwc, const ctta_tensor* grads,
                               int n_grads, void* stream);
/* The same backward pass, resumable block by block, so that the data-parallel all-reduce of a finished
 * block's gradients (accelerate/DDP's bucketed all-reduce, train.py:377-379) overlaps with the remaining
 * backward work: `begin` finishes block 2n+2 (conv_norm_out, conv_out); every `next` finishes one more
 * block and reports its id -- n+2+i = up_blocks.i, n+1 = mid_block, 1+i = down_blocks.i, 0 = conv_in +
 * time/guidance embedding MLPs (last, *finished = 1). */
ctta_status ctta_unet_backward_begin(ctta_unet* h, const void* dout_nhwc, const ctta_tensor* grads,
                                     int n_grads, void* stream);
ctta_status ctta_unet_backward_next(ctta_unet* h, const ctta_tensor* grads, int n_grads, void* stream,
                                    int* block_done, int* finished);
/* Debug taps (only when cfg.debug_taps): named NCHW fp32 copies of intermediates. */
int ctta_unet_num_taps(const ctta_unet* h);
ctta_status ctta_unet_tap_info(const ctta_unet* h, int i, const char** name, int dims[4]);
ctta_status ctta_unet_tap_read(ctta_unet* h, int i, float* dst_nchw, void* stream);

/* ------------------------------------------------------------------------------------ *
 * VAE decoder + vocoder.  Replaces AutoencoderKL.decode_first_stage
 * (audioldm/variational_autoencoder/autoencoder.py:103-106 -> Decoder.forward
 * modules.py:650-683) and AutoencoderKL.decode_to_waveform (autoencoder.py:108-111 ->
 * vocoder_infer hifigan/utilities.py:76-91 -> Generator.forward hifigan/models.py:101-117).
 * ------------------------------------------------------------------------------------ */
typedef struct {
  int z_channels, embed_dim, ch, out_ch, n_levels, num_res_blocks;
  int ch_mult[CTTA_MAX_LEVELS];
  float scale_factor;
  int max_batch, latent_h, latent_w;
  int debug_taps;
  int enable_grad;   /* 1: also build the data-gradient packs and size the arena for
                        ctta_vae_decode_with_grad + ctta_vae_decode_backward (decoder only) */
} ctta_vae_config;

typedef struct ctta_vae ctta_vae;
ctta_status ctta_vae_create(const ctta_vae_config* cfg, const ctta_tensor* weights,
                            int n_weights, void* stream, ctta_vae** out);
void ctta_vae_destroy(ctta_vae* h);
/* z (B,zc,T,F) f32 -> mel (B,out_ch,4T,4F) f32 */
ctta_status ctta_vae_decode(ctta_vae* h, const float* z, int batch, float* mel, void* stream);
/* decode_first_stage(z, allow_grad=True) (autoencoder.py:103-106 as called by CLAPLoss, tools/losses.py:294-296):
 * the same decode, keeping what the input gradient needs; ctta_vae_decode_backward then turns d loss / d mel
 * (B,1,4T,4F) f32 into d loss / d z (B,zc,T,F) f32 -- the decoder is frozen, no parameter gradients.  One forward
 * may be pending per handle; the backward consumes it.  Requires cfg.enable_grad. */
ctta_status ctta_vae_decode_with_grad(ctta_vae* h, const float* z, int batch, float* mel, void* stream);
ctta_status ctta_vae_decode_backward(ctta_vae* h, const float* grad_mel, int batch, float* grad_z, void* stream);
size_t ctta_vae_arena_bytes(const ctta_vae* h);
int ctta_vae_num_taps(const ctta_vae* h);
ctta_status ctta_vae_tap_info(const ctta_vae* h, int i, const char** name, int dims[4]);
ctta_status ctta_vae_tap_read(ctta_vae* h, int i, float* dst_nchw, void* stream);

#define CTTA_MAX_UPS 8
/* VAE encoder: AutoencoderKL.encode_first_stage (autoencoder.py:80-89 -> Encoder.forward modules.py:519-543 +
 * quant_conv), the training-side latent encoder of tools/train_utils.py:155-162.  Same ctta_vae_config as the decoder
 * (out_ch = the mel's channel count 1; latent_h/_w = the LATENT extent, the mel is 2^(n_levels-1) times larger).
 * mel (B,1,T,F) f32 -> moments (B, 2*embed_dim, T/s, F/s) f32 = [mean | logvar] of the diagonal Gaussian posterior;
 * sampling (DiagonalGaussianDistribution.sample, distributions.py:37-41) stays with the caller's RNG. */
typedef struct ctta_vae_encoder ctta_vae_encoder;
ctta_status ctta_vae_encoder_create(const ctta_vae_config* cfg, const ctta_tensor* weights, int n_weights,
                                    void* stream, ctta_vae_encoder** out);
void ctta_vae_encoder_destroy(ctta_vae_encoder* h);
ctta_status ctta_vae_encoder_load_weights(ctta_vae_encoder* h, const ctta_tensor* weights, int n_weights,
                                          void* stream);
ctta_status ctta_vae_encode(ctta_vae_encoder* h, const float* mel, int batch, float* moments, void* stream);
int ctta_vae_encoder_num_taps(const ctta_vae_encoder* h);
ctta_status ctta_vae_encoder_tap_info(const ctta_vae_encoder* h, int i, const char** name, int dims[4]);
ctta_status ctta_vae_encoder_tap_read(ctta_vae_encoder* h, int i, float* dst_nchw, void* stream);

/* Waveform -> log-mel front-end of the training step: tools/torch_tools.py:126-135 (wav_to_fbank) with
 * audioldm/audio/stft.py:132-186 (TacotronSTFT: reflect pad, conv1d STFT at hop `hop_length`, magnitude, mel
 * filterbank (librosa.filters.mel, Slaney scale / area norm), log(clamp(x, 1e-5))) and torch_tools.py:38-51 (_pad_spec).
 * wav (B, n_samples) f32 (clipped to [-1, 1], NaN -> 0 like get_mel_from_wav) -> fbank (B, target_length, n_mels) f32
 * and, when logmag != NULL, the log-magnitude spectrogram (B, target_length, filter_length/2) f32.  Frames past
 * n_samples/hop + 1 are zero (not log(1e-5)), exactly as the reference pads. */
typedef struct ctta_mel_frontend ctta_mel_frontend;
ctta_status ctta_mel_frontend_create(int filter_length, int hop_length, int win_length, int n_mels,
                                     int sampling_rate, float mel_fmin, float mel_fmax, int max_batch,
                                     int max_samples, ctta_mel_frontend** out);
void ctta_mel_frontend_destroy(ctta_mel_frontend* h);
ctta_status ctta_wav_to_fbank(ctta_mel_frontend* h, const float* wav, int batch, int n_samples,
                              int target_length, float* fbank, float* logmag, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Text encoder.  Replaces `self.text_encoder(input_ids=, attention_mask=)[0]` of
 * models/audio_distilled_model.py:208-214 (transformers T5EncoderModel, FLAN-T5-large; weights in the
 * T5EncoderModel state-dict layout: shared.weight, encoder.block.N.layer.{0.SelfAttention.{q,k,v,o},
 * 0.layer_norm, 1.DenseReluDense.{wi_0,wi_1,wo}, 1.layer_norm}.weight, block 0's relative_attention_bias,
 * encoder.final_layer_norm.weight).  gated-gelu feed-forward, d_kv = 64.
 * ------------------------------------------------------------------------------------ */
typedef struct {
  int vocab_size, d_model, d_kv, d_ff, num_layers, num_heads;
  int rel_buckets, rel_max_distance;   /* relative_attention_num_buckets / _max_distance */
  float eps;                           /* layer_norm_epsilon */
  int max_batch, max_len;              /* arena sizing */
} ctta_t5_config;
typedef struct ctta_t5 ctta_t5;
ctta_status ctta_t5_create(const ctta_t5_config* cfg, const ctta_tensor* weights, int n_weights, void* stream,
                           ctta_t5** out);
void ctta_t5_destroy(ctta_t5* h);
/* input_ids (B, L) int64, attention_mask (B, L) uint8 (1 = token, 0 = padding; every row needs one token)
 * -> last_hidden_state (B, L, d_model) f32 */
ctta_status ctta_t5_encode(ctta_t5* h, const int64_t* input_ids, const uint8_t* attention_mask, int batch, int len,
                           float* last_hidden_state, void* stream);
size_t ctta_t5_arena_bytes(const ctta_t5* h);

typedef struct {
  int num_mels, upsample_initial_channel, n_ups, n_kernels;
  int upsample_rates[CTTA_MAX_UPS], upsample_kernel_sizes[CTTA_MAX_UPS];
  int resblock_kernel_sizes[4];
  int resblock_dilations[4][3];
  int max_batch, max_frames;
  int debug_taps;
  int enable_grad;   /* 1: ctta_hifigan_forward_with_grad + ctta_hifigan_backward are available */
} ctta_hifigan_config;

typedef struct ctta_hifigan ctta_hifigan;
ctta_status ctta_hifigan_create(const ctta_hifigan_config* cfg, const ctta_tensor* weights,
                                int n_weights, void* stream, ctta_hifigan** out);
void ctta_hifigan_destroy(ctta_hifigan* h);
int64_t ctta_hifigan_out_len(const ctta_hifigan* h, int frames);
/* mel (B,frames,num_mels) f32 (== (B,1,T,F) NCHW mel) -> wav (B,out_len) f32 in (-1,1),
 * BEFORE the reference's batch-global centring. */
ctta_status ctta_hifigan_forward(ctta_hifigan* h, const float* mel, int batch, int frames,
                                 float* wav, void* stream);
/* vocoder_infer(..., allow_grad=True) (hifigan/utilities.py:79-81): the same forward keeping every LeakyReLU mask;
 * ctta_hifigan_backward turns d loss / d wav (B,out_len) f32 (`wav` = the forward's output, for tanh') into
 * d loss / d mel (B,frames,num_mels) f32.  The generator is frozen.  Requires cfg.enable_grad. */
ctta_status ctta_hifigan_forward_with_grad(ctta_hifigan* h, const float* mel, int batch, int frames, float* wav,
                                           void* stream);
ctta_status ctta_hifigan_backward(ctta_hifigan* h, const float* grad_wav, const float* wav, int batch, int frames,
                                  float* grad_mel, void* stream);
/* vocoder_infer post-processing (utilities.py:83-86): wav -= (max+min)/2 over the WHOLE
 * batch, *32768, truncation to int16.  `scratch` >= 2 floats of device memory.
 * centred (float, may be NULL) and pcm (int16, may be NULL) receive the results. */
ctta_status ctta_wav_finalize(const float* wav, int64_t n, float* scratch, float* centred,
                              int16_t* pcm, void* stream);
/* The same in two halves for a batch SHARDED over ranks: the centring uses the extrema of the whole batch
 * (hifigan/utilities.py:85 `(wavs.max() + wavs.min()) / 2` on the full tensor), so every rank computes its local
 * (max, min) -> max_min[2] (device floats), the caller MAX-reduces (max, -min) over ranks (dist_util.global_wav_extrema_),
 * and ctta_wav_center applies the reduced pair.  scratch >= 2 floats. */
ctta_status ctta_wav_extrema(const float* wav, int64_t n, float* scratch, float* max_min, void* stream);
ctta_status ctta_wav_center(const float* wav, int64_t n, const float* max_min, float* scratch, float* centred,
                            int16_t* pcm, void* stream);
size_t ctta_hifigan_arena_bytes(const ctta_hifigan* h);
int ctta_hifigan_num_taps(const ctta_hifigan* h);
ctta_status ctta_hifigan_tap_info(const ctta_hifigan* h, int i, const char** name, int dims[4]);
ctta_status ctta_hifigan_tap_read(ctta_hifigan* h, int i, float* dst_ncl, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Heun solver elementwise steps with PER-SAMPLE sigmas.  Replace
 * HeunDiscreteScheduler.{scale_model_input,add_noise,step}
 * (diffusers/schedulers/scheduling_heun_discrete.py:151-172,364-385,273-362), v-prediction.
 * x etc. are (B, n_per_sample) f32; sigma arrays are (B) f32 on device.
 * ------------------------------------------------------------------------------------ */
ctta_status ctta_heun_scale_model_input(const float* x, const float* sigma, float* out,
                                        int batch, int64_t n_per_sample, void* stream);
ctta_status ctta_heun_add_noise(const float* x0, const float* noise, const float* sigma,
                                float* out, int batch, int64_t n_per_sample, void* stream);
/* 1st-order half: prev = x + d*(sigma_next-sigma), d = (x - x0_hat)/sigma; writes d. */
ctta_status ctta_heun_step_first(const float* v, const float* x, const float* sigma,
                                 const float* sigma_next, float* prev, float* deriv,
                                 int batch, int64_t n_per_sample, void* stream);
/* 2nd-order half: d2 at (x_hat, sigma_next); prev = x_stored + (d_prev+d2)/2*(sigma_next-sigma) */
ctta_status ctta_heun_step_second(const float* v, const float* x_hat, const float* x_stored,
                                  const float* deriv_prev, const float* sigma,
                                  const float* sigma_next, float* prev, int batch,
                                  int64_t n_per_sample, void* stream);
/* AudioDistilledModel._query_teacher CFG combine (models/audio_distilled_model.py:313-319):
 * out = (1-w)*uncond + w*cond, per-sample w (B) f32. */
ctta_status ctta_cfg_combine(const float* uncond, const float* cond, const float* w, float* out,
                             int batch, int64_t n_per_sample, void* stream);
/* get_loss with MSELoss('instance') and SNR clamp (models/audio_consistency_model.py:250-266,
 * tools/losses.py:28-33): loss = mean_b( mean((pred-target)^2) * min(sigma^-2, gamma) ).
 * gamma <= 0 disables the weighting.  loss: 1 float on device. */
ctta_status ctta_snr_mse_loss(const float* pred, const float* target, const float* sigma,
                              float gamma, float* per_instance, float* loss, int batch,
                              int64_t n_per_sample, void* stream);
/* Stage-1 guided distillation pieces: per-sample linear combination out = a[b]*x + b[b]*y
 * (optionally clamped to [-clamp, clamp]) -- DDPMScheduler/DDIMScheduler.add_noise (scheduling_ddpm.py:420-443) and the
 * v-prediction terms of DDIMScheduler.step (scheduling_ddim.py:243-370) -- and the MSE with explicit per-instance weights
 * of AudioGDM.forward's get_loss (models/audio_guided_model.py:92-117). */
ctta_status ctta_lincomb2_rows(const float* x, const float* y, const float* a, const float* b, float* out,
                               int batch, int64_t n_per_sample, float clamp, void* stream);
ctta_status ctta_weighted_mse_loss(const float* pred, const float* target, const float* weights,
                                   float* per_instance, float* loss, int batch, int64_t n_per_sample,
                                   void* stream);
ctta_status ctta_weighted_mse_grad(const float* pred, const float* target, const float* weights, float loss_scale,
                                   int batch, int c, int hw, int c_pad, void* dpred_nhwc, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Fused two-shadow EMA.  Replaces do_ema_update (tools/train_utils.py:255-282) as called by
 * AudioLCM.update_ema (models/audio_consistency_model.py:221-227): for each shadow s_k:
 * s_k += (1-decay_k)*(p - s_k), fp32, in place.  shadow_b/decay_b may be NULL/0.
 * ------------------------------------------------------------------------------------ */
ctta_status ctta_ema_update2(const float* param, float* shadow_a, double decay_a, float* shadow_b,
                             double decay_b, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Operator-level entry points (used by the engines; exported for per-op parity tests and
 * micro-benchmarks).  bf16 tensors are uint16_t bit patterns, NHWC.
 * ------------------------------------------------------------------------------------ */
typedef struct {
  const void* x0; int c0;     /* NHWC bf16 source, channel count (== row stride unless x_stride) */
  const void* x1; int c1;     /* optional second source, concatenated after x0's channels */
  int batch, hi, wi;          /* LOGICAL input extent (after the optional x2 upsample) */
  int upsample;               /* 1: sources are (hi/2, wi/2), nearest-upsampled on the fly */
  int ho, wo;
  int kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
  const void* w;              /* packed bf16 [n_rows >= n][k_pad], k = (kh,kw,c), zero padded */
  int k_pad;                  /* row stride of w, multiple of 64 */
  int n;                      /* valid output channels; multiple of 4 unless the epilogue is
                                 plain (no bias/rowvec/res/accumulate) and round_up(n,4) <= ldc */
  const float* bias;          /* [n] or NULL */
  const float* bias_m;        /* per output ROW (pixel) bias [m] or NULL (transposed products) */
  const float* rowvec; int rowvec_ld;   /* + rowvec[b*ld + n] (time-embedding shift) or NULL */
  const void* res; int res_ld;          /* + res[m*ld + n] bf16 or NULL */
  int in_act;                 /* on the A operand: 0 none, 1 leaky_relu(in_slope) */
  float in_slope;
  int out_act;                /* 0 none, 1 silu, 2 tanh, 3 leaky_relu(out_slope) */
  float out_slope;
  float alpha;                /* v = alpha*(acc + bias + rowvec + res [+ old out]) */
  int accumulate;             /* 1: add the existing output before scaling */
  void* out; int ldc;         /* bf16 (or f32 when out_f32) [m][ldc]; ldc %% 4 != 0 selects
                                 element-wise stores (tiny Cout, bias-only epilogue) */
  int out_f32;
  void* out2; float out2_slope; /* optional 2nd bf16 output = leaky_relu(out, out2_slope), same indexing */
  int64_t out_batch_stride;   /* 0 -> ho*wo*ldc */
  int64_t out_offset;         /* element offset added inside a batch (ConvTranspose remap) */
  int64_t out_limit;          /* >0: store only if 0 <= idx_in_batch < out_limit */
  /* batched GEMM (grid.z): extra strides in elements; 0 = shared */
  int groups; int64_t x_group_stride, w_group_stride, out_group_stride;
  int tile;                   /* 0 = auto; else force a kernel variant id (tuning) */
  int x_stride;               /* 0 -> c0; else row stride of x0 in elements (column slice of a wider matrix) */
  /* GroupNorm statistics of the output from the epilogue (saves GroupNorm's own read pass over the tensor): when
   * gn_part != NULL, gn_groups divides n and gn_hw (rows per sample) divides the row count, every workgroup writes
   * (sum, sum of squares) per channel group of its tile to gn_part [batch][chunks][gn_groups][2] fp32, chunks = gn_hw /
   * tile rows.  Only some launches can do it (plain bf16 output through the wide-store epilogue, whole tiles per sample,
   * no split-K): ctta_conv_last_gn_chunks() tells the caller whether -- and with how many chunks -- it happened. */
  void* gn_part; int gn_groups; int gn_hw;
  int64_t gn_part_floats;     /* capacity of gn_part; launches whose partials would not fit write none */
} ctta_conv_desc;

ctta_status ctta_conv_gemm(const ctta_conv_desc* d, void* stream);
/* chunks per sample of the GroupNorm partials the calling thread's LAST ctta_conv_gemm wrote (0: none were written) */
int ctta_conv_last_gn_chunks(void);
/* Split-K partial sums (deep, narrow problems: < 192 output tiles and K >= 1024) go through a device workspace.
 * Threading contract: every engine handle (ctta_unet / ctta_vae / ctta_hifigan / ctta_t5 ...) owns its own workspace
 * and binds it to the calling host thread inside each of its entry points, so different handles may run on different
 * streams or host threads concurrently.  A RAW ctta_conv_gemm caller is served from one lazily allocated workspace per
 * device: launches that share it must be ordered on ONE stream -- or the caller binds its own buffer of
 * ctta_conv_workspace_bytes() bytes for the calling thread (NULL unbinds). */
void ctta_conv_bind_workspace(void* ws, size_t bytes);
/* The same with a promise about the workspace's first ctta_conv_workspace_header_bytes() bytes: header_zeroed != 0 says they were
 * zero when the buffer was created and have only been touched by this library since.  Stream-K launches (one persistent
 * launch that folds its partial tiles itself: start tickets, an epoch and one flag word per workgroup live there; the launches
 * leave the header consistent for the next one) are only taken with such a workspace; the engine handles bind theirs this
 * way, ctta_conv_bind_workspace() promises nothing and keeps the two-pass split-K.  The per-device default workspace is
 * created with a zero header. */
void ctta_conv_bind_workspace_ex(void* ws, size_t bytes, int header_zeroed);
int ctta_conv_bound_workspace_header(void);
size_t ctta_conv_workspace_header_bytes(void);
/* The calling thread's current binding (NULL / 0 when none).  Engine / STFT / mel / T5 entry points bind their own
 * handle's workspace while they enqueue work and RESTORE the caller's binding on return (round 3; before, they left the
 * thread unbound and later raw launches silently fell back to the shared per-device workspace). */
void ctta_conv_bound_workspace(void** ws, size_t* bytes);
/* on != 0: launches issued by the calling host thread take no split-K path until switched off again (used for GEMMs that
 * an engine enqueues on a second stream of the same handle, which must not share the handle's partial-sum slabs) */
void ctta_conv_suppress_splitk(int on);
/* Diagnostic: while `buf` is non-NULL every conv_gemm workgroup launched from this host thread writes six 64-bit words
 * {HW_ID register, s_memtime at entry, after the first K-tile landed, after the main loop, after the epilogue's last
 * store was issued, before the first K-tile request} to buf[6 * linear workgroup index] (tools/tile_timeline.py).  buf: >= 48 bytes per workgroup. */
void ctta_conv_debug_stamps(void* buf);
/* Diagnostic: while `buf` is non-NULL the self-attention forward launches of this host thread (no bias, nk % 64 == 0) run an
 * instrumented twin that writes, per workgroup, 4 waves x 64 key tiles x 6 s_memtime stamps (low 32 bits: tile top, after
 * the barrier, after the LDS-DMA issue, after the next tile's scores, after the softmax, after P V) followed by the four
 * waves' HW_ID and XCC_ID registers: (4 * 64 * 6 + 8) 32-bit words per workgroup (tools/attn_timeline.py). */
void ctta_attention_debug_stamps(void* buf);
size_t ctta_conv_workspace_bytes(void);
int ctta_conv_gemm_num_variants(void);
const char* ctta_conv_gemm_variant_name(int id);

/* Fused HiFi-GAN ResBlock unit (hifigan/models.py:56-63) for C = 32 / 64 / 128 / 256 channels, odd k <= 11 (C = 512: k = 3 / 7 / 11):
 *   out = act( alpha * ( [old out +] x + conv2(leaky_relu(conv1(leaky_relu(x, slope)) + b1, slope)) + b2 ) )
 * conv1: k taps, dilation `dil`; conv2: k taps, dilation 1; "same" zero padding; x / out bf16 [batch][len][channels].
 * The intermediate never leaves LDS.  Weights are FRAGMENT-MAJOR copies ([n/16][k*channels/32][64][8] bf16) of the
 * packed [n][k_pad] operands, made by ctta_frag_pack (k_valid = k * channels).  out_slope > 0 applies a final
 * leaky_relu.  ctta_resunit_supported reports whether a (channels, k, dil) combination fits the kernel. */
int ctta_resunit_supported(int channels, int k, int dil);
ctta_status ctta_frag_pack(const void* packed, int n, int k_pad, int k_valid, void* dst, void* stream);
ctta_status ctta_resunit_conv1d(const void* x, int batch, int len, int channels, int k, int dil,
                                const void* w1_frag, const float* b1, const void* w2_frag, const float* b2,
                                float slope, void* out, int accumulate, float alpha, float out_slope, void* stream);
/* The three units of one ResBlock chained in ONE launch (C = 32 / 64, k in {3, 5, 7}): x_{u+1} = x_u + unit_u(x_u) with the
 * residual stream kept in LDS between units (rounded to bf16 after every unit, as ctta_resunit_conv1d stores it), the last
 * unit finished by the same epilogue as ctta_resunit_conv1d.  Bit-identical to three ctta_resunit_conv1d calls.
 * dils / w1_frag / b1 / w2_frag / b2: HOST arrays of three entries (device pointers inside).  out must not alias x. */
int ctta_reschain_supported(int channels, int k, const int* dils);
ctta_status ctta_reschain_conv1d(const void* x, int batch, int len, int channels, int k, const int* dils,
                                 const void* const* w1_frag, const float* const* b1, const void* const* w2_frag,
                                 const float* const* b2, float slope, void* out, int accumulate, float alpha,
                                 float out_slope, void* stream);

/* Fused GEGLU feed-forward of a transformer block (diffusers/models/attention.py:276-334, 383-386, 430-432: ff.net.0.proj ->
 * chunk -> value * gelu(gate) -> ff.net.2, plus the residual) for the 256- and 512-wide levels, ONE launch:
 *   out[m][0..n_valid) = res[m][..] + b2 + W2 . ( (W1v . x[m] + b1v) * gelu(W1g . x[m] + b1g) )
 * x: bf16 [M][ld_x] (the LayerNorm output, cp = 256 or 512 columns used); the hidden activations (ffp columns) never leave the
 * CU.  With ln_gamma != NULL x is the LayerNorm's INPUT and the rows are normalised over their first ln_d columns while they
 * are staged (norm3 of the block, attention.py:329-334; same arithmetic and summation order as ctta_layernorm: bit-identical
 * to running it first) -- the normalised tensor never exists in HBM either.
 * `packed` = ctta_ffn_pack's per-wave weight streams (ctta_ffn_pack_bytes bytes) made from the conv_gemm operands of the two
 * linears: ff1 [2*ffp][k_pad1] with its rows in 16-blocks [16 value][16 gate] (what the out_act = 4 epilogue takes), ff2
 * [cp][k_pad2].  b1 follows ff1's row order.  Bit-identical to ctta_conv_gemm(out_act = 4) + ctta_conv_gemm(res).
 * ctta_ffn_geglu_supported: 1 when (cp, ffp) fits the kernel and option "ffn_fuse" is on (asked when a handle is built: it then
 * keeps the weight streams); ctta_ffn_geglu_wanted: 1 when M rows of a fitting (cp, ffp) fill the kernel's row tiles (one
 * workgroup per CU) well enough to beat the two launches (asked per forward; a pure function of its arguments). */
int ctta_ffn_geglu_supported(int cp, int ffp);
int ctta_ffn_geglu_wanted(int cp, int ffp, int64_t M);
size_t ctta_ffn_pack_bytes(int cp, int ffp);
ctta_status ctta_ffn_pack(const void* w1_packed, int k_pad1, const void* w2_packed, int k_pad2, int cp, int ffp, void* dst,
                          void* stream);
ctta_status ctta_ffn_geglu(const void* x, int ld_x, int64_t M, int cp, int ffp, const void* packed, const float* b1,
                           const float* b2, const void* res, int res_ld, void* out, int ldc, int n_valid,
                           const float* ln_gamma, const float* ln_beta, int ln_d, float ln_eps, void* stream);
/* The same launch by descriptor, with two optional neighbours of the feed-forward inside the workgroup:
 * TAIL PROJECTION (proj_packed != NULL; proj_out of the Transformer2DModel + the block's input as residual, transformer_2d.py):
 *   out[m][0..n_valid) = proj_res[m][..] + proj_bias + Wp . bf16( res[m] + b2 + W2 . (...) )
 * the feed-forward result is rounded to bf16 (what the separate launch would store), becomes the operand of one more cp x cp
 * GEMM and never reaches HBM; `out` / `ldc` / `n_valid` then describe the projection's output.
 * FRONT PROJECTION (front_packed != NULL; attn2.to_out + residual and norm3, attention.py:318-334):
 *   s2[m] = front_res[m] + front_bias + W0 . att[m][0..front_k)      (bf16, stored to s2_out: the feed-forward's residual)
 *   x[m]  = LayerNorm(s2[m])   (ln_gamma / ln_beta / ln_d / ln_eps; the normalised rows exist in LDS only)
 * `x` / `ld_x` / `res` / `res_ld` are ignored.  The streams of both projections are made by ctta_ffn_proj_pack from their
 * conv_gemm operands [cp][k_pad] (k = cp for the tail, k = the head-padded attention width for the front).
 * Bit-identical to the separate launches (ctta_conv_gemm x 4 + ctta_layernorm). */
typedef struct {
  const void* x; int ld_x; int64_t M; int cp, ffp;
  const void* packed; const float* b1; const float* b2;
  const void* res; int res_ld;
  void* out; int ldc; int n_valid;
  const float* ln_gamma; const float* ln_beta; int ln_d; float ln_eps;
  const void* proj_packed; const float* proj_bias; const void* proj_res; int proj_res_ld;
  const void* front_packed; const float* front_bias; const void* att; int att_ld; int front_k;
  const void* front_res; int front_res_ld; void* s2_out; int s2_ld;
} ctta_ffn_desc;
void ctta_ffn_desc_init(ctta_ffn_desc* d);
size_t ctta_ffn_proj_pack_bytes(int cp, int k);
ctta_status ctta_ffn_proj_pack(const void* w_packed, int k_pad, int k, int cp, void* dst, void* stream);
ctta_status ctta_ffn_block(const ctta_ffn_desc* d, void* stream);

/* Direct convolution for tiny Cout (<= 8): bf16 NHWC in, fp32 weights [n][kh][kw][c],
 * fp32 planar (NCHW) out, optional input leaky-relu and output tanh. */
ctta_status ctta_conv_small_n(const void* x, int c, int batch, int hi, int wi, int kh, int kw,
                              int pad_h, int pad_w, const float* w, const float* bias, int n,
                              int in_act, float in_slope, int out_act, float* out_nchw,
                              void* out_bf16_nhwc, void* stream);

/* fp32 weight matrix -> packed bf16 [n_rows][k_pad]: dst[r][k] = src[row_off[r]+col_off[k]]
 * (zero when either offset < 0, or when aux_limit > 0 and row_aux[r]+col_aux[k] >= aux_limit).
 * All index arrays are DEVICE int32. */
ctta_status ctta_pack_weight(const float* src, const int32_t* row_off, const int32_t* col_off,
                             const int32_t* row_aux, const int32_t* col_aux, int aux_limit,
                             int n_rows, int k_pad, void* dst, void* stream);

/* Table-driven variants used by the engines' (re)load path: all pack jobs / all fp32 copies of a
 * state dict in ONE launch each.  `jobs` / `segs` are DEVICE arrays; jobs must be sorted by block0
 * (block0 = first thread block of the job, CTTA_PACK_ELEMS_PER_BLOCK outputs per block). */
#define CTTA_PACK_ELEMS_PER_BLOCK 2048
#define CTTA_COPY_ELEMS_PER_BLOCK 16384
typedef struct {
  const float* src;
  const int32_t *row_off, *col_off, *row_aux, *col_aux;
  int aux_limit, n_rows, k_pad, block0;
  void* dst;
  int src_row_len;     /* > 0: every source row is the contiguous range [row_off[r], row_off[r] + src_row_len) and
                          all col_off < src_row_len: eligible for ctta_pack_weight_rows_multi */
  int rows_per_block;  /* ctta_pack_weight_rows_multi only: rows a block stages (<= 64; block0 counts such blocks) */
} ctta_pack_job;
typedef struct {
  const float* src;
  float* dst;
  int count;
} ctta_copy_seg;
ctta_status ctta_pack_weight_multi(const ctta_pack_job* jobs, int n_jobs, int total_blocks, void* stream);
/* Same result for jobs with src_row_len > 0.  A block of `threads` (256 / 512 / 1024) threads stages
 * rows_per_block source rows (rows_per_block * src_row_len <= lds_floats floats of dynamic shared memory), read once
 * and coalesced, and permutes them from LDS -- the strided (cin, kh, kw) -> (kh, kw, cin) gather of a conv weight
 * otherwise touches every cache line kh*kw times.  Callers launch one table per LDS class so that short rows keep
 * several blocks per CU. */
ctta_status ctta_pack_weight_rows_multi(const ctta_pack_job* jobs, int n_jobs, int total_blocks, int lds_floats,
                                        int threads, void* stream);
/* Table-driven bf16 transposes (one 64x64 tile per block): dst[c][r] = src[r][c] for r < rows, c < cols, zero for
 * rows <= r < wcols; only the first wcols columns of a dst row are written.  Used to derive the data-gradient
 * operands (W^T, rotated conv taps) from the freshly packed forward operands without re-reading fp32 weights. */
typedef struct {
  const void* src;
  void* dst;
  int rows, cols, src_ld, dst_ld, wcols, block0, tiles_r;
} ctta_tpose_job;
ctta_status ctta_transpose_multi(const ctta_tpose_job* jobs, int n_jobs, int total_blocks, void* stream);
ctta_status ctta_copy_segments_multi(const ctta_copy_seg* segs, int n_segs, void* stream);

ctta_status ctta_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int batch, int c, int h,
                                       int w, int c_pad, float scale, void* stream);
/* (B, HW, c_stride) fp32 rows -> (B, C, HW) fp32: the last hop of UNet2DCondition*Model.forward's `sample` (conv_out's
 * [pixel][channel] result of ctta_conv_gemm to the reference's NCHW, unet_2d_condition_guided.py:940-945) */
ctta_status ctta_nhwc_f32_to_nchw_f32(const float* src, float* dst, int batch, int c, int hw, int c_stride, void* stream);
ctta_status ctta_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int batch, int c, int h,
                                       int w, int c_stride, void* stream);
ctta_status ctta_rows_f32_to_bf16(const float* src, void* dst, int64_t rows, int cols,
                                  int cols_pad, void* stream);
ctta_status ctta_concat_channels(const void* a, int ca, const void* b, int cb, void* dst,
                                 int64_t pixels, void* stream);
/* The same concatenation, also leaving the GroupNorm partial sums of the result ([batch][*nchunk][groups][2], what
 * ctta_groupnorm_from_partials reads; chunking and summation order of ctta_groupnorm's own first pass, so the normalised tensor
 * is bit-identical to concat + ctta_groupnorm): unet_2d_blocks.py:2053 + resnet.py:553 in one pass over the two sources. */
ctta_status ctta_concat_channels_gn(const void* a, int ca, const void* b, int cb, void* dst, int batch, int hw, int groups,
                                    float* partials, int64_t partials_floats, int* nchunk_out, void* stream);

/* GroupNorm over NHWC bf16 (+ optional SiLU): F.group_norm semantics (biased variance).
 * scratch: fp32 device buffer of ctta_groupnorm_scratch_floats(...) floats. */
size_t ctta_groupnorm_scratch_floats(int batch, int hw, int c, int groups);
ctta_status ctta_groupnorm(const void* x, void* y, int batch, int hw, int c, int groups,
                           const float* gamma, const float* beta, float eps, int silu,
                           float* scratch, void* stream);
/* Same, and also writes (mean, rstd) per (sample, group) to stats [B][G][2] when stats != NULL: what
 * ctta_groupnorm_bwd needs, for free from the forward's own fp64 fold. */
ctta_status ctta_groupnorm_stats_out(const void* x, void* y, int batch, int hw, int c, int groups,
                                     const float* gamma, const float* beta, float eps, int silu,
                                     float* scratch, float* stats, void* stream);
/* Same normalisation from partial sums that a convolution's epilogue already wrote (ctta_conv_desc.gn_part, `nchunk` =
 * ctta_conv_last_gn_chunks()): finalize + apply, no statistics pass over x.  scratch: >= batch * 2 * c floats. */
ctta_status ctta_groupnorm_from_partials(const void* x, void* y, int batch, int hw, int c, int groups,
                                         const float* gamma, const float* beta, float eps, int silu,
                                         const float* partials, int nchunk, float* scratch, float* stats, void* stream);
/* LayerNorm over rows of a padded bf16 matrix: dims [rows][ld], true width d (pad -> 0). */
ctta_status ctta_layernorm(const void* x, void* y, int64_t rows, int d, int ld,
                           const float* gamma, const float* beta, float eps, void* stream);
/* GEGLU: in [rows][2*hp] = [value | gate] -> out [rows][hp] = value * gelu_erf(gate) */
ctta_status ctta_geglu(const void* in, void* out, int64_t rows, int hp, int interleaved, void* stream);
/* Row softmax: fp32 scores [rows][cols] * scale -> bf16 probabilities [rows][cols] */
ctta_status ctta_softmax_rows(const float* s, void* p, int64_t rows, int cols, float scale,
                              void* stream);

/* Multi-head attention, head dim padded to 64 (pad lanes must be zero):
 * q [B][nq][..] row stride q_ld, head h at column h*64; k [B][k_rows >= nk][..] likewise; vt is V
 * TRANSPOSED: [B][heads*64][vt_ld] (keys contiguous).  bias (B, nk) f32 additive per key or NULL.
 * softmax(q k^T * scale + bias) v  ->  out [B][nq][out_ld], head h at column h*64. */
/* Same with a relative-position table added to the scores: rel_bias_log2 [heads][nq+nk-1], entry
 * (key - query + nq - 1), ALREADY multiplied by log2(e); q / out batches are q_rows rows apart (T5 self-attention). */
ctta_status ctta_attention_rel(const void* q, int q_ld, int q_rows, const void* k, int k_ld, int k_rows,
                               const void* vt, int vt_ld, const float* key_bias, const float* rel_bias_log2,
                               void* out, int out_ld, int batch, int heads, int nq, int nk, float scale, void* stream);
ctta_status ctta_attention(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                           int vt_ld, const float* bias, void* out, int out_ld, int batch,
                           int heads, int nq, int nk, float scale, void* stream);
/* Same, and (when lse != NULL) also writes the log2-domain log-sum-exp of every query row,
 * lse[b][head][q] = log2(sum_k exp2(s*scale*log2e + bias*log2e)), which ctta_attention_bwd needs. */
ctta_status ctta_attention_lse(const void* q, int q_ld, const void* k, int k_ld, int k_rows,
                               const void* vt, int vt_ld, const float* bias, void* out, int out_ld,
                               int batch, int heads, int nq, int nk, float scale, float* lse,
                               void* stream);
/* Backward of the attention above (autograd of F.scaled_dot_product_attention,
 * attention_processor.py:1127-1129) without materialising the scores.  Operands:
 *   q, k            as in the forward;  out / dout: forward output and its gradient [B*nq][*_ld]
 *   vn [B][vn_rows][vn_ld]   V in natural layout (head h at columns h*64)
 *   kt [B][heads*64][kt_ld]  K^T (keys contiguous, zero beyond nk; kt_ld multiple of 64)
 *   qt, dot [B][heads*64][qt_ld]  Q^T and dout^T (queries contiguous, zero beyond nq)
 *   lse  from ctta_attention_lse;  dsum [B][heads][nq] fp32 scratch (D = rowsum(dout*out))
 * Writes dq [B*nq][dq_ld], dk / dv [B*k_rows][*_ld] for keys < nk (head h at columns h*64).
 * partial (optional fp32 scratch of partial_floats): with few keys (cross-attention) the query walk of
 * the dk/dv kernel is split over up to 32 workgroups whose fp32 partial sums are folded afterwards;
 * needs 2 * splits * batch * k_rows * heads*64 floats, fewer splits are used if it is smaller. */
ctta_status ctta_attention_bwd(const void* q, int q_ld, const void* k, int k_ld, int k_rows,
                               const void* vn, int vn_ld, int vn_rows, const void* kt, int kt_ld,
                               const void* qt, const void* dot, int qt_ld, const float* bias,
                               const void* out, int out_ld, const void* dout, int do_ld,
                               const float* lse, float* dsum, void* dq, int dq_ld, void* dk, int dk_ld,
                               void* dv, int dv_ld, int batch, int heads, int nq, int nk, float scale,
                               float* partial, int64_t partial_floats, void* stream);
/* The same backward with every operand read where it lies (replaces the four whole-tensor transposes per call that fed
 * ctta_attention_bwd): q / k / dout as above, vt = V TRANSPOSED exactly as ctta_attention[_lse] took it
 * ([B][heads*64][vt_ld], vt_ld >= nk, a multiple of 8).  Same results (autograd of attention_processor.py:1127-1129). */
ctta_status ctta_attention_bwd_inplace(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                       int vt_ld, const float* bias, const void* out, int out_ld, const void* dout,
                                       int do_ld, const float* lse, float* dsum, void* dq, int dq_ld, void* dk, int dk_ld,
                                       void* dv, int dv_ld, int batch, int heads, int nq, int nk, float scale,
                                       float* partial, int64_t partial_floats, void* stream);

/* Window attention of the CLAP audio tower (Swin, laion_clap/clap_module/htsat.py:336-361): the flash kernels above with a
 * FULL additive bias table full_bias_log2 [n_bias_batches][heads][nq][nk] (already multiplied by log2 e; relative-position
 * bias per head plus the shifted-window mask, one table per window position); batch item b reads table b % n_bias_batches.
 * Heads are padded to 64 lanes like everywhere else.  lse may be NULL in the forward. */
ctta_status ctta_attention_fullbias(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                    int vt_ld, const float* full_bias_log2, int n_bias_batches, void* out, int out_ld,
                                    int batch, int heads, int nq, int nk, float scale, float* lse, void* stream);
ctta_status ctta_attention_fullbias_bwd(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vn,
                                        int vn_ld, int vn_rows, const void* kt, int kt_ld, const void* qt,
                                        const void* dot, int qt_ld, const float* full_bias_log2, int n_bias_batches,
                                        const void* out, int out_ld, const void* dout, int do_ld, const float* lse,
                                        float* dsum, void* dq, int dq_ld, void* dk, int dk_ld, void* dv, int dv_ld,
                                        int batch, int heads, int nq, int nk, float scale, void* stream);

/* ------------------------------------------------------------------------------------ *
 * CLAP fine-tuning stage (tools/losses.py:259-316): glue kernels of the audio tower.
 * ------------------------------------------------------------------------------------ */
/* torchaudio.functional.resample(sinc_interp_kaiser) as a polyphase FIR (tools/losses.py:299-303):
 * y[b][n*up + p] = sum_j kernels[p][j] * x[b][n*down + j - width], zero outside [0, len); kernels fp32 [up][taps] built by
 * the host from the published window formula; out_len = ceil(up * len / down).  _bwd is the adjoint (d y -> d x). */
ctta_status ctta_resample_poly(const float* x, int batch, int len, const float* kernels, int up, int down, int width,
                               int taps, float* y, int64_t out_len, void* stream);
ctta_status ctta_resample_poly_bwd(const float* dy, int batch, int64_t out_len, const float* kernels, int up, int down,
                                   int width, int taps, float* dx, int len, void* stream);
/* |STFT| with input gradient for MultiResolutionSTFTLoss (tools/losses.py:146-169 `stft`, :187-256): torch.stft(center,
 * reflect padding, periodic Hann window of win_length centred in fft_size) -> sqrt(clamp(re^2 + im^2, 1e-8)), laid out
 * (batch, frames, fft_size / 2 + 1) with frames = n_samples / hop_size + 1.  The reference computes in float64; here both
 * GEMM operands are split into three bf16 parts (fp32-grade).  ctta_stft_magnitude_bwd maps d|STFT| of the LAST
 * ctta_stft_magnitude call on the handle to d wav (batch, n_samples).  One handle per resolution; not thread-safe. */
typedef struct ctta_stft ctta_stft;
ctta_status ctta_stft_create(int fft_size, int hop_size, int win_length, int max_batch, int max_samples, ctta_stft** out);
void ctta_stft_destroy(ctta_stft* h);
int ctta_stft_frames(const ctta_stft* h, int n_samples);
ctta_status ctta_stft_magnitude(ctta_stft* h, const float* wav, int batch, int n_samples, float* mag, void* stream);
ctta_status ctta_stft_magnitude_bwd(ctta_stft* h, const float* dmag, int batch, int n_samples, float* dwav, void* stream);

/* Spectrogram(power 2) + LogmelFilterBank (dB, amin) of htsat.py:684-697 on a ctta_mel_frontend handle: (batch, n_samples)
 * fp32 -> logmel [batch][n_samples / hop + 1][n_mels] fp32; _bwd maps d logmel to d wav for the last forward call. */
ctta_status ctta_wav_to_logmel_db(ctta_mel_frontend* h, const float* wav, int batch, int n_samples, float amin,
                                  float* logmel, void* stream);
ctta_status ctta_wav_to_logmel_db_bwd(ctta_mel_frontend* h, const float* dlogmel, int batch, int n_samples, float amin,
                                      float* dwav, void* stream);
/* bn0 (eval: per-mel-bin scale / shift) + bicubic stretch of the frame axis (tap tables [out_frames][4] from the host) +
 * fold into the (spec_size x spec_size) image, htsat.py:913-925,856-878 -> NHWC bf16 [batch][S][S][cpad] (channel 0).
 * _bwd: gradient of the patch-embedding conv in token layout [batch][(S/patch)^2][ld >= patch^2] fp32 -> d logmel, with the
 * transposed tap list in CSR form (rt_ptr [frames + 1], rt_tt, rt_w). */
ctta_status ctta_htsat_image(const float* logmel, int batch, int frames, int mel_bins, const float* bn_scale,
                             const float* bn_shift, const int32_t* tap_idx, const float* tap_w, int out_frames,
                             int spec_size, int cpad, void* image_nhwc, void* stream);
ctta_status ctta_htsat_image_bwd(const float* dtokens, int dtokens_ld, int patch, int batch, int frames, int mel_bins,
                                 const float* bn_scale, const int32_t* rt_ptr, const int32_t* rt_tt, const float* rt_w,
                                 int spec_size, float* dlogmel, void* stream);
/* dst[r][:] = src[idx[r]][:] over bf16 rows (idx < 0: zeros): window partition / cyclic shift / patch merging and their
 * inverses are all row permutations of the token matrix (htsat.py:259-287,471-492,517-537). */
ctta_status ctta_gather_rows(const void* src, int src_ld, const int32_t* idx, void* dst, int dst_ld, int64_t n_rows,
                             int row_elems, void* stream);
/* nn.GELU (erf) on bf16 and its backward from the pre-activation (htsat.py:156-174) */
ctta_status ctta_gelu(const void* x, void* y, int64_t n, void* stream);
ctta_status ctta_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
/* mean over the token axis: bf16 [batch][tokens][ld] -> fp32 [batch][channels], and its backward (htsat.py:818-819) */
ctta_status ctta_mean_tokens(const void* x, int batch, int tokens, int channels, int ld, float* y, void* stream);
ctta_status ctta_mean_tokens_bwd(const float* dy, int batch, int tokens, int channels, int ld, void* dx, void* stream);

/* Evaluation-suite classifier glue (PANNs Cnn14, audioldm_eval/feature_extractors/panns/models.py:168-323; the twelve 3x3
 * convolutions run on ctta_conv_gemm with BatchNorm folded in, the front end on ctta_wav_to_logmel_db):
 *   ctta_logmel_to_image  bn0 in eval mode (models.py:276-278: scale / shift per mel bin) of logmel fp32 [batch][frames][mel_bins]
 *                         -> image bf16 [batch][frames][mel_bins][8], channel 0 live;
 *   ctta_avgpool2         F.avg_pool2d(2) (models.py:71-72) on NHWC bf16 [batch][hi][wi][c] -> [batch][hi/2][wi/2][c], c % 8 == 0,
 *                         a trailing odd row / column dropped as torch does;
 *   ctta_cnn14_head       mean over frequency, then max over time + mean over time (models.py:305-309):
 *                         bf16 [batch][frames][freq][c] -> fp32 [batch][c]. */
ctta_status ctta_logmel_to_image(const float* logmel, int batch, int frames, int mel_bins, const float* scale,
                                 const float* shift, void* image, void* stream);
ctta_status ctta_avgpool2(const void* x, void* y, int batch, int hi, int wi, int c, void* stream);
ctta_status ctta_cnn14_head(const void* x, int batch, int frames, int freq, int c, float* y, void* stream);

/* Small fp32 linear: y[m][n] = act_out(sum_k act_in(x[m][k]) * w[n][k] + b[n]); m <= 1024.
 * act: 0 none, 1 silu. */
ctta_status ctta_linear_f32(const float* x, const float* w, const float* b, float* y, int m,
                            int n, int k, int act_in, int act_out, void* stream);
/* Sinusoidal timestep features (embeddings.py:25-65) and Gaussian Fourier features
 * (embeddings.py:239-249) in one launch.  freqs: [dim/2] f32 table. */
ctta_status ctta_time_features(const float* t, const float* freqs, int dim, int flip, float* out,
                               int batch, void* stream);
ctta_status ctta_fourier_features(const double* w, const float* weight, int half, int flip,
                                  float* out, int batch, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Backward-pass operators of the distillation step (student U-Net); torch-autograd semantics of
 * the reference modules.  All contractions run on ctta_conv_gemm; these are the glue kernels.
 * ------------------------------------------------------------------------------------ */
/* dst[g][c][r] = src[g][r][col0 + c] (r < rows), zero for rows <= r < dst_ld */
ctta_status ctta_transpose_bf16(const void* src, int64_t src_group_stride, int rows, int cols, int src_ld,
                                int col0, void* dst, int64_t dst_group_stride, int dst_ld, int groups,
                                void* stream);
/* Q[(ch*taps + tap)][m] = X[pixel(m, tap)][ch] (transposed im2col, channel-major rows = the (cin,kh,kw) order
 * of a conv weight row; rows padded to m_pad); when
 * indicator_batches >= 0 appends 1 + indicator_batches rows: all-ones, then per-sample indicators. */
ctta_status ctta_im2col_t(const void* x, int c, int batch, int hi, int wi, int upsample, int ho, int wo,
                          int kh, int kw, int stride, int pad_h, int pad_w, int dil_w, void* dst, int m_pad,
                          int indicator_batches, void* stream);
/* grad_w[row_off[n] + col_off[k]] (+)= sum_s slabs[s][k][n]: inverse of ctta_pack_weight
 * (col_off NULL = identity; negative offsets are skipped) */
ctta_status ctta_wgrad_scatter(const float* slabs, int n_slabs, int64_t slab_stride, int ldn, int k_rows,
                               int n_cols, const int32_t* row_off, const int32_t* col_off,
                               const int32_t* row_aux, const int32_t* col_aux, int aux_limit, float* grad,
                               int accumulate, void* stream);
/* same contraction for row-major slabs [S][n_rows][ldk] (k contiguous):
 * grad_w[row_off[n] + col_off[k]] (+)= sum_s slabs[s][n][k] */
/* Implicit weight-gradient GEMM (csrc/wgrad_gemm.hip): slabs[s][n][c * taps + t] = sum over split s's positions m of
 * dY^T[n][m] * X[pixel(m, tap t)][c] -- the product ctta_im2col_t + ctta_conv_gemm compute, without materialising im2col(X)^T.
 * dyt: bf16 [n][mp] (ctta_transpose_bf16 of dY, zero beyond m_valid); x: bf16 NHWC (batch, h, w) with x_ld elements per
 * pixel, c channels used; taps = 9: 3x3, stride 1, pad 1, columns in (cin, kh, kw) order; taps = 1: a linear layer
 * (batch * h * w = rows).  bias_col >= 0 also writes the row sums of dY^T into column bias_col (bias gradient) and, for
 * sample_cols > 0, per-sample sums into the columns behind it (d temb).  mp % (64 * splits) == 0.
 * ctta_wgrad_implicit_supported: whether a geometry is inside the kernel's range (CTTA_WGRAD_IMPLICIT=0 turns it off). */
int ctta_wgrad_implicit_supported(int taps, int c, int h, int w, int x_ld, int n);
/* ctta_wgrad_implicit with dY [m_valid][ldy] read where it lies (taps = 9 only): no transposed copy, bias / per-sample
 * columns summed inside the kernel. */
ctta_status ctta_wgrad_implicit_inplace(const void* dy, int ldy, int n, int mp, const void* x, int x_ld, int c, int batch, int h,
                                        int w, int taps, int m_valid, int splits, int bias_col, int sample_cols, float* slabs,
                                        int64_t slab_stride, int ld, void* stream);
/* Weight gradient of a linear layer with BOTH operands read in place (no transposed copies):
 * slabs[s][r][k] = sum over split s's rows m of dY[m][r] * X[m][k]; dy [m_valid][ldy] (n columns used), x [m_valid][ldx]
 * (c columns used), row-major bf16; rows beyond m_valid count as zero up to mp (mp %% (64 * splits) == 0).  bias_col >= c
 * also writes the column sums of dY into slabs[s][r][bias_col].  Replaces autograd's grad_weight / grad_bias of F.linear
 * (diffusers/models/attention.py:276-334, attention_processor.py:1107-1136) in the student's backward pass. */
ctta_status ctta_wgrad_tn(const void* dy, int ldy, int n, const void* x, int ldx, int c, int m_valid, int mp, int splits,
                          int bias_col, float* slabs, int64_t slab_stride, int ld, void* stream);
/* The same two products with ONE split and the tile added STRAIGHT into the layer's gradient tensors through the pack map
 * (autograd's accumulate into .grad, tools/train_utils.py:166): grad_w[row_off[n] + col(k)] += dW[n][k] for k < k_cols, rows with
 * row_off[n] < 0 skipped (padding), col = col_off[k] (< 0 skipped) or the identity when col_off is NULL;
 * grad_b[bias_idx[n] or n] += column sum of dY for n < n_bias (grad_b NULL: no bias).  No fp32 slab, no scatter launch; the
 * same bits as slab + ctta_wgrad_scatter_rows_bias(accumulate = 1) with one slab. */
ctta_status ctta_wgrad_tn_direct(const void* dy, int ldy, int n, const void* x, int ldx, int c, int m_valid, int mp,
                                 int k_cols, int n_rows, const int32_t* row_off, const int32_t* col_off, float* grad_w,
                                 int n_bias, const int32_t* bias_idx, float* grad_b, void* stream);
ctta_status ctta_wgrad_implicit_direct(const void* dy, int ldy, int n, int mp, const void* x, int x_ld, int c, int batch,
                                       int h, int w, int m_valid, int k_cols, int n_rows, const int32_t* row_off,
                                       float* grad_w, int n_bias, const int32_t* bias_idx, float* grad_b, void* stream);
/* the row sums alone: slabs[s][r][bias_col] = sum over split s of dY^T[r][m] (+ per-sample sums in the sample_cols columns
 * behind it): bias gradient / d temb of the layers whose weight-gradient product runs on ctta_conv_gemm */
ctta_status ctta_wgrad_rowsum(const void* dyt, int n, int mp, int m_valid, int splits, int hw, int sample_cols, float* slabs,
                              int64_t slab_stride, int ld, int bias_col, void* stream);
ctta_status ctta_wgrad_implicit(const void* dyt, int n, int mp, const void* x, int x_ld, int c, int batch, int h, int w,
                                int taps, int m_valid, int splits, int bias_col, int sample_cols, float* slabs,
                                int64_t slab_stride, int ld, void* stream);
ctta_status ctta_wgrad_scatter_rows(const float* slabs, int n_slabs, int64_t slab_stride, int ldk, int k_cols,
                                    int n_rows, const int32_t* row_off, const int32_t* col_off, float* grad,
                                    int accumulate, void* stream);
/* dst[j*dst_stride + idx[n]] (+)= sum_s slabs[s][n][col + j], j < n_cols (bias / per-sample columns) */
/* ctta_wgrad_scatter_rows that also folds column `bias_col` of the slabs into grad_bias[bias_idx[n]] for n < n_bias (the
 * layer's bias gradient in the same launch; bias_col < 0: none) */
ctta_status ctta_wgrad_scatter_rows_bias(const float* slabs, int n_slabs, int64_t slab_stride, int ldk, int k_cols, int n_rows,
                                         const int32_t* row_off, const int32_t* col_off, float* grad, int bias_col, int n_bias,
                                         const int32_t* bias_idx, float* grad_bias, int accumulate, void* stream);
ctta_status ctta_col_scatter(const float* slabs, int n_slabs, int64_t slab_stride, int ldk, int col, int n_cols,
                             int n_rows, const int32_t* idx, float* dst, int64_t dst_stride, int accumulate,
                             void* stream);
ctta_status ctta_row_scatter(const float* slabs, int n_slabs, int64_t slab_stride, int ldn, int row,
                             int n_cols, const int32_t* idx, float* dst, int accumulate, void* stream);
/* out (+)= res + alpha * g * (act > 0 ? 1 : slope): LeakyReLU backward on the saved activation, residual folded in */
ctta_status ctta_lrelu_bwd(const void* g, const void* act, float slope, float alpha, const void* res, void* out,
                           int64_t n, int accumulate, void* stream);
/* data gradient of a 1-output-channel conv (w fp32 [tap][c]); gy is scaled by (1 - y_tanh^2) when y_tanh != NULL and
 * the result by leaky_relu'(act) when act != NULL; dx NHWC bf16 */
ctta_status ctta_conv_cout1_dgrad(const float* gy, const float* y_tanh, const float* w, int batch, int h, int wd,
                                  int kh, int kw, int pad_h, int pad_w, int c, const void* act, float slope,
                                  void* dx, void* stream);
/* GroupNorm(+SiLU) backward; stats [B][G][2] = (mean, rstd) from ctta_groupnorm_stats */
size_t ctta_groupnorm_bwd_scratch_floats(int batch, int hw, int c, int groups);
ctta_status ctta_groupnorm_stats(const void* x, int batch, int hw, int c, int groups, float eps, float* stats,
                                 void* stream);
ctta_status ctta_groupnorm_bwd(const void* x, const void* dy, void* dx, int batch, int hw, int c, int groups,
                               const float* stats, const float* gamma, const float* beta, int silu,
                               int accumulate_dx, float* dgamma, float* dbeta, int accumulate_param,
                               float* scratch, void* stream);
/* LayerNorm backward (statistics recomputed); dgamma/dbeta are ACCUMULATED into (atomics) */
ctta_status ctta_layernorm_bwd(const void* x, const void* dy, void* dx, int64_t rows, int d, int ld,
                               const float* gamma, float eps, int accumulate_dx, float* dgamma, float* dbeta,
                               void* stream);
/* dx = dx_add + dL/dx with a separate output (dx_add may be NULL, or equal dx = the in-place accumulate above) */
ctta_status ctta_layernorm_bwd_add(const void* x, const void* dy, const void* dx_add, void* dx, int64_t rows, int d, int ld,
                                   const float* gamma, float eps, float* dgamma, float* dbeta, void* stream);
/* The same with a CALLER-OWNED partial table for d gamma / d beta (per-block sums + a sliced fold instead of one atomic per
   block and column); scratch may be NULL (atomics path).  The library holds no workspace of its own for this call. */
size_t ctta_layernorm_bwd_scratch_floats(int64_t rows, int ld);
ctta_status ctta_layernorm_bwd_ws(const void* x, const void* dy, const void* dx_add, void* dx, int64_t rows, int d, int ld,
                                  const float* gamma, float eps, float* dgamma, float* dbeta, float* scratch,
                                  size_t scratch_floats, void* stream);
ctta_status ctta_geglu_bwd(const void* f, const void* dout, void* df, int64_t rows, int hp, int interleaved,
                           void* stream);
ctta_status ctta_add_slices(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows,
                            int cols, void* stream);
ctta_status ctta_zero_insert2(const void* dy, void* dz, int batch, int ho, int wo, int hz, int wz, int c,
                              void* stream);
ctta_status ctta_pool2_sum(const void* dup, void* dx, int batch, int h, int w, int c, int accumulate,
                           void* stream);
ctta_status ctta_softmax_bias_rows(const float* s, int lds, const float* bias, int rows_per_bias, void* p,
                                   int64_t rows, int cols, int ldp, float scale, void* stream);
ctta_status ctta_softmax_bwd_rows(const void* p, const float* dp, int lddp, void* ds, int64_t rows, int cols,
                                  int ldp, float scale, void* stream);
ctta_status ctta_linear_f32_bwd(const float* x, const float* w, const float* dy, int dy_ld,
                                const float* xpre_silu, float* dx, float* dw, float* db, int m, int n, int k,
                                int accumulate_dx, int accumulate_param, void* stream);
/* d/dpred of get_loss (MSE 'instance' x SNR clamp, models/audio_consistency_model.py:250-266) as NHWC bf16 */
ctta_status ctta_snr_mse_grad(const float* pred, const float* target, const float* sigma, float gamma,
                              float loss_scale, int batch, int c, int hw, int c_pad, void* dpred_nhwc,
                              void* stream);
/* torch.optim.AdamW step (tools/train_utils.py:59-63): decoupled weight decay, bias correction by `step` */
ctta_status ctta_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                            float grad_scale, void* stream);
/* The optimizer tail of one training step as ONE pass over the training state: optimizer.step() -> optimizer.zero_grad() ->
 * update_ema() (tools/train_utils.py:177-183, 255-282).  param / grad / both shadows are the flat fp32 buffers of n_all
 * elements (trainable prefix of n_train elements first, see AudioLCM.prepare_training); exp_avg / exp_avg_sq cover the
 * prefix.  Reads p, g, m, v, shadow_a, shadow_b; writes p, m, v, both shadows and g = 0.  do_step = 0 leaves p, m, v alone
 * (the reference skips the update on a NaN loss, train_utils.py:167-172, but still zeroes the gradients and moves the
 * shadows).  shadow_b may be NULL.  Per element the same fp32 operations in the same order as ctta_adamw_step followed by
 * ctta_ema_update2: bit-identical state.  n_train and n_all must be multiples of 4, all buffers 16-byte aligned. */
ctta_status ctta_adamw_ema2_zero(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n_train, int64_t n_all,
                                 float* shadow_a, double decay_a, float* shadow_b, double decay_b, int do_step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                 void* stream);

/* Opt-in launch profiler (bench.py's live roofline leg): when enabled, every conv_gemm (kind 0)
 * and attention (kind 1) launch is bracketed by hipEvents on its own stream.  collect() waits
 * for the recorded launches, returns their summed duration / executed FLOPs / count, optionally
 * appends one CSV line per launch (kind,variant,m,n,k,groups,ms,tflops) and clears the log. */
/* GroupNorm statistics from the producing convolution's epilogue (default on; option "gn_fuse" = 0 /
 * ctta_set_gn_fuse(0) turn it off, process-wide, taking effect at the next engine call).  Off: a sample's result is
 * bit-identical at every batch size; on: at a fixed batch size (LABNOTES.md 4). */
void ctta_set_gn_fuse(int on);
int ctta_get_gn_fuse(void);
void ctta_prof_enable(int on);
ctta_status ctta_prof_collect(int kind, double* total_ms, double* total_flops, int64_t* launches,
                              const char* csv_path);

#ifdef __cplusplus
}
#endif
#endif /* CTTA_H */
