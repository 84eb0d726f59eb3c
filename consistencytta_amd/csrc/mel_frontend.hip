// Waveform -> log-mel front-end of the training step: tools/torch_tools.py:126-135 (wav_to_fbank) ->
// audioldm/audio/stft.py:165-186 (TacotronSTFT.mel_spectrogram) -> :52-84 (STFT.transform) ->
// audio_processing.py:85-91 (log(clamp(x, 1e-5))) -> torch_tools.py:38-51 (_pad_spec).
//
// The STFT is the reference's strided conv1d against the windowed DFT basis, i.e. a GEMM of
// frames x (2 * 513) x 1024 -- run on conv_gemm over the waveform viewed as (L/8) "pixels" of 8 samples (hop 160 =
// 20 pixels, window 1024 = 128 taps).  bf16 operands alone would bury quiet bins under quantisation noise (8-bit
// mantissa ~ -48 dB; two-way splits still leave a -96 dB floor, visible in the log-mel of near-silent bins), so both
// operands are split into THREE bf16 parts (x = x0 + x1 + x2, 24 mantissa bits) and the six products x_i * b_j with
// i + j <= 2 are accumulated in fp32: fp32-grade magnitudes at MFMA cost (13 GF per 10 s clip).
#include "engine_common.h"

#include <math.h>

struct ctta_mel_frontend {
  int n_fft = 0, hop = 0, n_mels = 0, cutoff = 0, n_rows = 0;   // n_rows = 2*cutoff padded to a multiple of 4
  int max_batch = 0, max_samples = 0;
  bf16_t* b[3] = {nullptr, nullptr, nullptr};   // basis parts [n_rows][n_fft]
  float* mel_w = nullptr;                       // [n_mels][cutoff]
  bf16_t* x[3] = {nullptr, nullptr, nullptr};   // waveform parts [B][lp]
  float* ft = nullptr;                       // [B * frames][n_rows]
  size_t lp_max = 0, frames_max = 0;
  SplitWs splitws;
  // power / dB mode with input gradient (CLAP audio tower): d(re, im) in bf16, the basis as a [n_fft][kpad] operand
  // (re / im rows contiguous along K), and the per-frame sample gradients; allocated on first use
  bf16_t* dft = nullptr;
  bf16_t* basis_t = nullptr;
  float* dframes = nullptr;
  int kpad = 0;
};

// clip to [-1, 1], nan_to_num, reflect-pad n_fft/2 on both sides, split into bf16 hi + lo
__global__ void mel_prepare_kernel(const float* __restrict__ wav, int T, int lp, int half, bf16_t* __restrict__ p0,
                                   bf16_t* __restrict__ p1, bf16_t* __restrict__ p2, int clip) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < lp; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < T + 2 * half) {
      int j = i - half;
      if (j < 0) j = -j;
      if (j >= T) j = 2 * (T - 1) - j;
      v = wav[(size_t)b * T + j];
      if (clip) {
        if (v != v) v = 0.f;                                     // nan_to_num after clip: NaN -> 0
        v = fminf(fmaxf(v, -1.0f), 1.0f);                        // also maps +-inf to +-1 (clip happens first)
      }
    }
    const bf16_t h0 = f2bf(v);
    const float r1 = v - bf2f(h0);
    const bf16_t h1 = f2bf(r1);
    p0[(size_t)b * lp + i] = h0;
    p1[(size_t)b * lp + i] = h1;
    p2[(size_t)b * lp + i] = f2bf(r1 - bf2f(h1));
  }
}

// one block per output frame: magnitude -> mel filterbank -> log(clamp) ; frames beyond `frames` are zero padding
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ ft, int frames, int n_rows, int cutoff,
                                                  const float* __restrict__ mel_w, int n_mels, int target,
                                                  float* __restrict__ fbank, float* __restrict__ logmag) {
  extern __shared__ float mag[];
  const int f = blockIdx.x, b = blockIdx.y;
  const int lm_cols = (cutoff % 2) ? cutoff - 1 : cutoff;        // _pad_spec drops the last bin of an odd channel count
  if (f >= frames) {
    for (int j = threadIdx.x; j < n_mels; j += 256) fbank[((size_t)b * target + f) * n_mels + j] = 0.f;
    if (logmag)
      for (int k = threadIdx.x; k < lm_cols; k += 256) logmag[((size_t)b * target + f) * lm_cols + k] = 0.f;
    return;
  }
  const float* row = ft + ((size_t)b * frames + f) * n_rows;
  for (int k = threadIdx.x; k < cutoff; k += 256) {
    const float re = row[k], im = row[cutoff + k];
    const float m = sqrtf(re * re + im * im);
    mag[k] = m;
    if (logmag && k < lm_cols) logmag[((size_t)b * target + f) * lm_cols + k] = logf(fmaxf(m, 1e-5f));
  }
  __syncthreads();
  // 4 lanes per mel band
  const int j = threadIdx.x >> 2, part = threadIdx.x & 3;
  if (j < n_mels) {
    const float* w = mel_w + (size_t)j * cutoff;
    float acc = 0.f;
    for (int k = part; k < cutoff; k += 4) acc += w[k] * mag[k];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) fbank[((size_t)b * target + f) * n_mels + j] = logf(fmaxf(acc, 1e-5f));
  }
}

// librosa.filters.mel (htk=False, norm="slaney"), restated: Slaney mel scale, triangles on the rfft bin centres,
// area normalisation (see oracle/mel.py for the pinned reference restatement)
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_hz / f_sp + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

extern "C" void ctta_mel_frontend_destroy(ctta_mel_frontend* M) {
  if (!M) return;
  for (void* p : {(void*)M->b[0], (void*)M->b[1], (void*)M->b[2], (void*)M->mel_w, (void*)M->x[0], (void*)M->x[1],
                  (void*)M->x[2], (void*)M->ft})
    if (p) (void)hipFree(p);
  for (void* p : {(void*)M->dft, (void*)M->basis_t, (void*)M->dframes})
    if (p) (void)hipFree(p);
  M->splitws.destroy();
  delete M;
}

extern "C" ctta_status ctta_mel_frontend_create(int filter_length, int hop_length, int win_length, int n_mels,
                                                int sampling_rate, float mel_fmin, float mel_fmax, int max_batch,
                                                int max_samples, ctta_mel_frontend** out) {
  CTTA_REQUIRE(out && filter_length >= 64 && filter_length % 64 == 0 && hop_length % 8 == 0 && hop_length > 0,
               "mel_frontend_create: filter_length must be a multiple of 64 and hop_length a multiple of 8");
  CTTA_REQUIRE(win_length == filter_length, "mel_frontend_create: win_length must equal filter_length (the reference's config)");
  CTTA_REQUIRE(n_mels >= 1 && n_mels <= 64 && max_batch >= 1 && max_samples > filter_length, "mel_frontend_create: bad sizes");
  ctta_mel_frontend* M = new ctta_mel_frontend();
  const int N = filter_length, cutoff = N / 2 + 1;
  M->n_fft = N; M->hop = hop_length; M->n_mels = n_mels; M->cutoff = cutoff; M->n_rows = round_up(2 * cutoff, 4);
  M->max_batch = max_batch; M->max_samples = max_samples;
  // windowed DFT basis (stft.py:17-48): rows [Re F[:cutoff] ; Im F[:cutoff]] * periodic Hann, float32 like the reference
  std::vector<bf16_t> parts[3];
  for (auto& v : parts) v.assign((size_t)M->n_rows * N, 0);
  auto to_bf16 = [](float v) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    return (bf16_t)((bits + 0x7fffu + ((bits >> 16) & 1u)) >> 16);
  };
  auto from_bf16 = [](bf16_t h) {
    const uint32_t hb = (uint32_t)h << 16;
    float f;
    memcpy(&f, &hb, 4);
    return f;
  };
  for (int r = 0; r < 2 * cutoff; ++r) {
    const int k = r < cutoff ? r : r - cutoff;
    for (int n = 0; n < N; ++n) {
      const double ang = 2.0 * M_PI * (double)(((long long)k * n) % N) / N;
      const float base = (float)(r < cutoff ? cos(ang) : -sin(ang));     // np.fft.fft: exp(-i ang)
      const float win = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / N));
      float rem = base * win;
      for (int part = 0; part < 3; ++part) {
        const bf16_t h = to_bf16(rem);
        parts[part][(size_t)r * N + n] = h;
        rem -= from_bf16(h);
      }
    }
  }
  std::vector<float> w((size_t)n_mels * cutoff, 0.f);
  {
    std::vector<double> mel_f(n_mels + 2);
    const double m0 = hz_to_mel(mel_fmin), m1 = hz_to_mel(mel_fmax);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
    for (int i = 0; i < n_mels; ++i) {
      const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
      for (int k = 0; k < cutoff; ++k) {
        const double fk = (double)k * sampling_rate / N;
        const double lower = (fk - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
        const double upper = (mel_f[i + 2] - fk) / (mel_f[i + 2] - mel_f[i + 1]);
        const double v = fmax(0.0, fmin(lower, upper));
        w[(size_t)i * cutoff + k] = (float)(v * enorm);
      }
    }
  }
  const size_t lp = (size_t)round_up(max_samples + N, 8);
  const size_t frames = (size_t)max_samples / hop_length + 1;
  M->lp_max = lp; M->frames_max = frames;
  bool ok = hipMalloc((void**)&M->mel_w, w.size() * 4) == hipSuccess &&
            hipMalloc((void**)&M->ft, (size_t)max_batch * frames * M->n_rows * 4) == hipSuccess;
  for (int part = 0; part < 3 && ok; ++part)
    ok = hipMalloc((void**)&M->b[part], parts[part].size() * 2) == hipSuccess &&
         hipMalloc((void**)&M->x[part], (size_t)max_batch * lp * 2) == hipSuccess &&
         hipMemcpy(M->b[part], parts[part].data(), parts[part].size() * 2, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) ok = hipMemcpy(M->mel_w, w.data(), w.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) ok = M->splitws.init() == CTTA_OK;
  if (!ok) {
    ctta_set_error("mel_frontend_create: device allocation / upload failed");
    ctta_mel_frontend_destroy(M);
    return CTTA_ERR_NOMEM;
  }
  *out = M;
  return CTTA_OK;
}

extern "C" ctta_status ctta_wav_to_fbank(ctta_mel_frontend* M, const float* wav, int batch, int n_samples,
                                         int target_length, float* fbank, float* logmag, void* stream) {
  CTTA_REQUIRE(M && wav && fbank, "wav_to_fbank: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= M->max_batch && n_samples > M->n_fft / 2 && n_samples <= M->max_samples &&
                   target_length >= 1,
               "wav_to_fbank: batch %d / samples %d outside the handle's limits (%d, %d)", batch, n_samples, M->max_batch,
               M->max_samples);
  hipStream_t s = (hipStream_t)stream;
  WsBind bind(M->splitws);
  const int N = M->n_fft, half = N / 2;
  const int lp = round_up(n_samples + N, 8);
  const int frames_all = n_samples / M->hop + 1;                  // (T + 2*half - N) / hop + 1
  const int frames = frames_all < target_length ? frames_all : target_length;   // later frames are cut by _pad_spec
  hipLaunchKernelGGL(mel_prepare_kernel, dim3((lp + 255) / 256 > 1024 ? 1024 : (lp + 255) / 256, batch), dim3(256), 0, s, wav,
                     n_samples, lp, half, M->x[0], M->x[1], M->x[2], 1);
  CTTA_LAUNCH_CHECK();
  // x_i * b_j for i + j <= 2, smallest terms first so that they are not absorbed by the large one
  const int xi[6] = {2, 1, 0, 1, 0, 0}, bj[6] = {0, 1, 2, 0, 1, 0};
  for (int pass = 0; pass < 6; ++pass) {
    ctta_conv_desc d;
    desc_init(&d);
    d.x0 = M->x[xi[pass]]; d.c0 = 8;
    d.batch = batch; d.hi = 1; d.wi = lp / 8; d.ho = 1; d.wo = frames;
    d.kh = 1; d.kw = N / 8; d.stride_w = M->hop / 8;
    d.w = M->b[bj[pass]]; d.k_pad = N; d.n = M->n_rows;
    d.out = M->ft; d.ldc = M->n_rows; d.out_f32 = 1; d.accumulate = pass > 0 ? 1 : 0;
    CTTA_TRY(ctta_conv_gemm(&d, s));
  }
  hipLaunchKernelGGL(mel_kernel, dim3(target_length, batch), dim3(256), (size_t)M->cutoff * sizeof(float), s, M->ft, frames,
                     M->n_rows, M->cutoff, M->mel_w, M->n_mels, target_length, fbank, logmag);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Power / dB variant for the CLAP audio tower (torchlibrosa Spectrogram(power=2) + LogmelFilterBank as configured at
// laion_clap/clap_module/htsat.py:684-697): out[b][f][m] = 10 log10(max(sum_k W[m][k] (re^2 + im^2), amin)), no
// clipping of the waveform, every frame of the centred STFT (n_samples / hop + 1).  The STFT itself is the same
// split-bf16 GEMM as above; `ft` stays in the handle for the backward call.
__global__ __launch_bounds__(256) void mel_power_db_kernel(const float* __restrict__ ft, int frames, int n_rows, int cutoff,
                                                           const float* __restrict__ mel_w, int n_mels, float amin,
                                                           float* __restrict__ out) {
  extern __shared__ float pw[];
  const int f = blockIdx.x, b = blockIdx.y;
  const float* row = ft + ((size_t)b * frames + f) * n_rows;
  for (int k = threadIdx.x; k < cutoff; k += 256) {
    const float re = row[k], im = row[cutoff + k];
    pw[k] = re * re + im * im;
  }
  __syncthreads();
  const int j = threadIdx.x >> 2, part = threadIdx.x & 3;
  if (j < n_mels) {
    const float* w = mel_w + (size_t)j * cutoff;
    float acc = 0.f;
    for (int k = part; k < cutoff; k += 4) acc += w[k] * pw[k];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) out[((size_t)b * frames + f) * n_mels + j] = 10.0f * log10f(fmaxf(acc, amin));
  }
}
// one block per frame: g_mel[m] = dout[m] * (10 / ln 10) / mel[m] (0 where mel <= amin), dP[k] = sum_m W[m][k] g_mel[m],
// d re = 2 re dP, d im = 2 im dP  ->  bf16 [B*frames][kpad] (re block, then im block, zero padded)
__global__ __launch_bounds__(256) void mel_power_db_bwd_kernel(const float* __restrict__ ft, int frames, int n_rows, int cutoff,
                                                               const float* __restrict__ mel_w, int n_mels, float amin,
                                                               const float* __restrict__ dout, bf16_t* __restrict__ dft,
                                                               int kpad) {
  extern __shared__ float sm[];
  float* pw = sm;               // [cutoff]
  float* gm = sm + cutoff;      // [n_mels]
  const int f = blockIdx.x, b = blockIdx.y;
  const float* row = ft + ((size_t)b * frames + f) * n_rows;
  for (int k = threadIdx.x; k < cutoff; k += 256) {
    const float re = row[k], im = row[cutoff + k];
    pw[k] = re * re + im * im;
  }
  __syncthreads();
  const int j = threadIdx.x >> 2, part = threadIdx.x & 3;
  if (j < n_mels) {
    const float* w = mel_w + (size_t)j * cutoff;
    float acc = 0.f;
    for (int k = part; k < cutoff; k += 4) acc += w[k] * pw[k];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0)
      gm[j] = acc > amin ? dout[((size_t)b * frames + f) * n_mels + j] * 4.3429448190325175f / acc : 0.f;
  }
  __syncthreads();
  bf16_t* o = dft + ((size_t)b * frames + f) * kpad;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    float v = 0.f;
    if (k < 2 * cutoff) {
      const int kk = k < cutoff ? k : k - cutoff;
      float dp = 0.f;
      for (int m = 0; m < n_mels; ++m) dp += mel_w[(size_t)m * cutoff + kk] * gm[m];
      v = 2.0f * row[k] * dp;
    }
    o[k] = f2bf(v);
  }
}
// overlap-add of the per-frame sample gradients plus the adjoint of the reflect padding:
// val[i] = sum_f dframes[f][i - f*hop] over the padded axis, dwav[j] = val[j + half] (+ the mirrored halo entries)
__global__ __launch_bounds__(256) void stft_overlap_add_kernel(const float* __restrict__ dframes, int frames, int N, int hop,
                                                               int T, float* __restrict__ dwav) {
  const int b = blockIdx.y, half = N / 2;
  const float* df = dframes + (size_t)b * frames * N;
  auto val = [&](int i) {
    float acc = 0.f;
    int f_hi = i / hop;
    if (f_hi > frames - 1) f_hi = frames - 1;
    for (int f = f_hi; f >= 0 && i - f * hop < N; --f) acc += df[(size_t)f * N + (i - f * hop)];
    return acc;
  };
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    float acc = val(j + half);
    if (j >= 1 && j <= half) acc += val(half - j);                          // left halo: padded i = half - j
    const int r = half + 2 * (T - 1) - j;                                   // right halo: padded i = half + 2(T-1) - j
    if (j <= T - 2 && r >= half + T && r < T + 2 * half) acc += val(r);
    dwav[(size_t)b * T + j] = acc;
  }
}

extern "C" ctta_status ctta_wav_to_logmel_db(ctta_mel_frontend* M, const float* wav, int batch, int n_samples, float amin,
                                             float* logmel, void* stream) {
  CTTA_REQUIRE(M && wav && logmel, "wav_to_logmel_db: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= M->max_batch && n_samples > M->n_fft / 2 && n_samples <= M->max_samples,
               "wav_to_logmel_db: batch %d / samples %d outside the handle's limits (%d, %d)", batch, n_samples, M->max_batch,
               M->max_samples);
  hipStream_t s = (hipStream_t)stream;
  WsBind bind(M->splitws);
  const int N = M->n_fft, half = N / 2;
  const int lp = round_up(n_samples + N, 8);
  const int frames = n_samples / M->hop + 1;
  hipLaunchKernelGGL(mel_prepare_kernel, dim3((lp + 255) / 256 > 1024 ? 1024 : (lp + 255) / 256, batch), dim3(256), 0, s, wav,
                     n_samples, lp, half, M->x[0], M->x[1], M->x[2], 0);
  CTTA_LAUNCH_CHECK();
  const int xi[6] = {2, 1, 0, 1, 0, 0}, bj[6] = {0, 1, 2, 0, 1, 0};
  for (int pass = 0; pass < 6; ++pass) {
    ctta_conv_desc d;
    desc_init(&d);
    d.x0 = M->x[xi[pass]]; d.c0 = 8;
    d.batch = batch; d.hi = 1; d.wi = lp / 8; d.ho = 1; d.wo = frames;
    d.kh = 1; d.kw = N / 8; d.stride_w = M->hop / 8;
    d.w = M->b[bj[pass]]; d.k_pad = N; d.n = M->n_rows;
    d.out = M->ft; d.ldc = M->n_rows; d.out_f32 = 1; d.accumulate = pass > 0 ? 1 : 0;
    CTTA_TRY(ctta_conv_gemm(&d, s));
  }
  hipLaunchKernelGGL(mel_power_db_kernel, dim3(frames, batch), dim3(256), (size_t)M->cutoff * sizeof(float), s, M->ft, frames,
                     M->n_rows, M->cutoff, M->mel_w, M->n_mels, amin, logmel);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// d logmel -> d wav for the LAST ctta_wav_to_logmel_db call on this handle (same batch / n_samples / amin)
extern "C" ctta_status ctta_wav_to_logmel_db_bwd(ctta_mel_frontend* M, const float* dlogmel, int batch, int n_samples,
                                                 float amin, float* dwav, void* stream) {
  CTTA_REQUIRE(M && dlogmel && dwav, "wav_to_logmel_db_bwd: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= M->max_batch && n_samples > M->n_fft / 2 && n_samples <= M->max_samples,
               "wav_to_logmel_db_bwd: batch %d / samples %d outside the handle's limits", batch, n_samples);
  hipStream_t s = (hipStream_t)stream;
  WsBind bind(M->splitws);
  const int N = M->n_fft, cutoff = M->cutoff;
  const int frames = n_samples / M->hop + 1;
  if (!M->dft) {   // first backward: the gradient-side buffers and the transposed basis
    M->kpad = round_up(2 * cutoff, 64);
    const size_t rows = (size_t)M->max_batch * M->frames_max;
    CTTA_CHECK_HIP(hipMalloc((void**)&M->dft, rows * M->kpad * 2));
    CTTA_CHECK_HIP(hipMalloc((void**)&M->dframes, rows * N * 4));
    CTTA_CHECK_HIP(hipMalloc((void**)&M->basis_t, (size_t)N * M->kpad * 2));
    CTTA_CHECK_HIP(ctta_zero_async(M->basis_t, (size_t)N * M->kpad * 2, s));
    // basis_t[n][r] = basis[r][n] (leading bf16 part), r < 2*cutoff
    CTTA_TRY(ctta_transpose_bf16(M->b[0], 0, 2 * cutoff, N, N, 0, M->basis_t, 0, M->kpad, 1, s));
  }
  hipLaunchKernelGGL(mel_power_db_bwd_kernel, dim3(frames, batch), dim3(256), (size_t)(cutoff + M->n_mels) * sizeof(float), s,
                     M->ft, frames, M->n_rows, cutoff, M->mel_w, M->n_mels, amin, dlogmel, M->dft, M->kpad);
  CTTA_LAUNCH_CHECK();
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = M->dft; d.c0 = M->kpad;
  d.batch = 1; d.hi = batch * frames; d.wi = 1; d.ho = batch * frames; d.wo = 1;
  d.w = M->basis_t; d.k_pad = M->kpad; d.n = N;
  d.out = M->dframes; d.ldc = N; d.out_f32 = 1;
  CTTA_TRY(ctta_conv_gemm(&d, s));
  int blocks = (n_samples + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stft_overlap_add_kernel, dim3(blocks, batch), dim3(256), 0, s, M->dframes, frames, N, M->hop, n_samples,
                     dwav);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
