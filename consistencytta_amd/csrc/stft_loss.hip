// |STFT| with input gradient for the multi-resolution STFT loss of the CLAP fine-tuning stage:
// tools/losses.py:146-169 (`stft`: torch.stft(x.double(), fft_size, hop_size, win_length, window) -> sqrt(clamp(re^2 +
// im^2, 1e-8)) transposed to (B, frames, fft_size / 2 + 1)), used by SpectralConvergengeLoss / LogSTFTMagnitudeLoss at
// the three resolutions (1024, 120, 600), (2048, 240, 1200), (512, 50, 240) of tools/losses.py:190-194.
//
// torch.stft(center=True, pad_mode="reflect", onesided) restated: reflect-pad n_fft / 2 samples on both sides, frame f
// starts at f * hop, frames = 1 + T / hop, the window (periodic Hann of win_length) is zero-padded to n_fft centred,
//     X[f][k] = sum_n w[n] x_pad[f * hop + n] exp(-2 pi i k n / n_fft).
// Like the mel front end this is a GEMM against the windowed DFT basis with both operands split into three bf16 parts
// (24 mantissa bits: the reference computes in float64 and rounds the magnitudes to float32).  The hops are not
// multiples of 8 samples (50), so the frames are gathered explicitly ([B * frames][n_fft], ~10x the waveform at the
// finest resolution: 3.3 M elements per 10 s clip) instead of being a strided view of the waveform.
// Backward: d|X| -> d(re, im) = d|X| * (re, im) / |X| (zero where the power was clamped), a bf16 GEMM against the
// transposed basis gives the per-frame sample gradients, overlap-add + the adjoint of the reflect padding gives d x.
#include "engine_common.h"

#include <math.h>

struct ctta_stft {
  int n_fft = 0, hop = 0, win = 0, cutoff = 0, n_rows = 0, kpad = 0;
  int max_batch = 0, max_samples = 0;
  size_t frames_max = 0;
  bf16_t* b[3] = {nullptr, nullptr, nullptr};    // basis parts [n_rows][n_fft]
  bf16_t* fr[3] = {nullptr, nullptr, nullptr};   // frame parts [B * frames][n_fft]
  float* ft = nullptr;                           // [B * frames][n_rows]  (re block, im block)
  bf16_t* dft = nullptr;                         // [B * frames][kpad]
  bf16_t* basis_t = nullptr;                     // [n_fft][kpad]
  float* dframes = nullptr;                      // [B * frames][n_fft]
  SplitWs splitws;
};

// frames of the reflect-padded waveform, split into three bf16 parts
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ wav, int T, int frames, int N, int hop,
                                                          bf16_t* __restrict__ p0, bf16_t* __restrict__ p1,
                                                          bf16_t* __restrict__ p2) {
  const int f = blockIdx.x, b = blockIdx.y, half = N / 2;
  const size_t row = ((size_t)b * frames + f) * N;
  for (int n = threadIdx.x; n < N; n += 256) {
    int j = f * hop + n - half;
    if (j < 0) j = -j;
    if (j >= T) j = 2 * (T - 1) - j;
    const float v = wav[(size_t)b * T + j];
    const bf16_t h0 = f2bf(v);
    const float r1 = v - bf2f(h0);
    const bf16_t h1 = f2bf(r1);
    p0[row + n] = h0;
    p1[row + n] = h1;
    p2[row + n] = f2bf(r1 - bf2f(h1));
  }
}

__global__ __launch_bounds__(256) void stft_mag_kernel(const float* __restrict__ ft, long long rows, int n_rows, int cutoff,
                                                       float* __restrict__ mag) {
  const long long total = rows * cutoff;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cutoff;
    const int k = (int)(i - r * cutoff);
    const float re = ft[r * n_rows + k], im = ft[r * n_rows + cutoff + k];
    mag[i] = sqrtf(fmaxf(re * re + im * im, 1e-8f));
  }
}

// d|X| -> d(re, im), bf16 [rows][kpad] (re block, im block, zero padding)
__global__ __launch_bounds__(256) void stft_mag_bwd_kernel(const float* __restrict__ ft, const float* __restrict__ dmag,
                                                           long long rows, int n_rows, int cutoff, int kpad,
                                                           bf16_t* __restrict__ dft) {
  const long long total = rows * kpad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / kpad;
    const int k = (int)(i - r * kpad);
    float v = 0.f;
    if (k < 2 * cutoff) {
      const int kk = k < cutoff ? k : k - cutoff;
      const float re = ft[r * n_rows + kk], im = ft[r * n_rows + cutoff + kk];
      const float pw = re * re + im * im;
      if (pw > 1e-8f) v = dmag[r * cutoff + kk] * (k < cutoff ? re : im) * rsqrtf(pw);
    }
    dft[i] = f2bf(v);
  }
}

// overlap-add of the per-frame sample gradients plus the adjoint of the reflect padding (same walk as the mel front
// end's: val[i] = sum_f dframes[f][i - f * hop] on the padded axis, d x[j] = val[j + half] + the mirrored halo entries)
__global__ __launch_bounds__(256) void stft_loss_overlap_add_kernel(const float* __restrict__ dframes, int frames, int N,
                                                                    int hop, int T, float* __restrict__ dwav) {
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
    if (j >= 1 && j <= half) acc += val(half - j);
    const int r = half + 2 * (T - 1) - j;
    if (j <= T - 2 && r >= half + T && r < T + 2 * half) acc += val(r);
    dwav[(size_t)b * T + j] = acc;
  }
}

extern "C" void ctta_stft_destroy(ctta_stft* S) {
  if (!S) return;
  for (void* p : {(void*)S->b[0], (void*)S->b[1], (void*)S->b[2], (void*)S->fr[0], (void*)S->fr[1], (void*)S->fr[2],
                  (void*)S->ft, (void*)S->dft, (void*)S->basis_t, (void*)S->dframes})
    if (p) (void)hipFree(p);
  S->splitws.destroy();
  delete S;
}

extern "C" ctta_status ctta_stft_create(int fft_size, int hop_size, int win_length, int max_batch, int max_samples,
                                        ctta_stft** out) {
  CTTA_REQUIRE(out && fft_size >= 64 && fft_size % 64 == 0 && hop_size >= 1 && win_length >= 1 && win_length <= fft_size,
               "stft_create: fft_size=%d must be a multiple of 64, 1 <= win_length=%d <= fft_size, hop_size=%d >= 1", fft_size,
               win_length, hop_size);
  CTTA_REQUIRE(max_batch >= 1 && max_samples > fft_size / 2, "stft_create: bad sizes");
  ctta_stft* S = new ctta_stft();
  const int N = fft_size, cutoff = N / 2 + 1;
  S->n_fft = N; S->hop = hop_size; S->win = win_length; S->cutoff = cutoff; S->n_rows = round_up(2 * cutoff, 4);
  S->kpad = round_up(2 * cutoff, 64);
  S->max_batch = max_batch; S->max_samples = max_samples;
  S->frames_max = (size_t)max_samples / hop_size + 1;
  std::vector<bf16_t> parts[3];
  for (auto& v : parts) v.assign((size_t)S->n_rows * N, 0);
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
  const int left = (N - win_length) / 2;   // torch.stft centres a short window inside n_fft
  for (int r = 0; r < 2 * cutoff; ++r) {
    const int k = r < cutoff ? r : r - cutoff;
    for (int n = 0; n < N; ++n) {
      const int wn = n - left;
      if (wn < 0 || wn >= win_length) continue;
      const double win = 0.5 - 0.5 * cos(2.0 * M_PI * wn / win_length);   // torch.hann_window (periodic)
      const double ang = 2.0 * M_PI * (double)(((long long)k * n) % N) / N;
      float rem = (float)((r < cutoff ? cos(ang) : -sin(ang)) * win);
      for (int part = 0; part < 3; ++part) {
        const bf16_t h = to_bf16(rem);
        parts[part][(size_t)r * N + n] = h;
        rem -= from_bf16(h);
      }
    }
  }
  const size_t rows = (size_t)max_batch * S->frames_max;
  bool ok = hipMalloc((void**)&S->ft, rows * S->n_rows * 4) == hipSuccess;
  for (int part = 0; part < 3 && ok; ++part)
    ok = hipMalloc((void**)&S->b[part], parts[part].size() * 2) == hipSuccess &&
         hipMalloc((void**)&S->fr[part], rows * N * 2) == hipSuccess &&
         hipMemcpy(S->b[part], parts[part].data(), parts[part].size() * 2, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) ok = S->splitws.init() == CTTA_OK;
  if (!ok) {
    ctta_set_error("stft_create: device allocation / upload failed");
    ctta_stft_destroy(S);
    return CTTA_ERR_NOMEM;
  }
  *out = S;
  return CTTA_OK;
}

static ctta_status stft_check(const ctta_stft* S, int batch, int n_samples, const char* who) {
  CTTA_REQUIRE(batch >= 1 && batch <= S->max_batch && n_samples > S->n_fft / 2 && n_samples <= S->max_samples,
               "%s: batch %d / samples %d outside the handle's limits (%d, %d; at least n_fft / 2 + 1 samples for the reflect "
               "padding)", who, batch, n_samples, S->max_batch, S->max_samples);
  return CTTA_OK;
}

extern "C" int ctta_stft_frames(const ctta_stft* S, int n_samples) { return S ? n_samples / S->hop + 1 : 0; }

extern "C" ctta_status ctta_stft_magnitude(ctta_stft* S, const float* wav, int batch, int n_samples, float* mag,
                                           void* stream) {
  CTTA_REQUIRE(S && wav && mag, "stft_magnitude: null pointer");
  CTTA_TRY(stft_check(S, batch, n_samples, "stft_magnitude"));
  hipStream_t s = (hipStream_t)stream;
  WsBind bind(S->splitws);
  const int N = S->n_fft, frames = n_samples / S->hop + 1;
  const long long rows = (long long)batch * frames;
  hipLaunchKernelGGL(stft_frames_kernel, dim3(frames, batch), dim3(256), 0, s, wav, n_samples, frames, N, S->hop, S->fr[0],
                     S->fr[1], S->fr[2]);
  CTTA_LAUNCH_CHECK();
  const int xi[6] = {2, 1, 0, 1, 0, 0}, bj[6] = {0, 1, 2, 0, 1, 0};   // smallest products first
  for (int pass = 0; pass < 6; ++pass) {
    ctta_conv_desc d;
    desc_init(&d);
    d.x0 = S->fr[xi[pass]]; d.c0 = N;
    d.batch = 1; d.hi = (int)rows; d.wi = 1; d.ho = (int)rows; d.wo = 1;
    d.w = S->b[bj[pass]]; d.k_pad = N; d.n = S->n_rows;
    d.out = S->ft; d.ldc = S->n_rows; d.out_f32 = 1; d.accumulate = pass > 0 ? 1 : 0;
    CTTA_TRY(ctta_conv_gemm(&d, s));
  }
  const long long total = rows * S->cutoff;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(stft_mag_kernel, dim3(blocks), dim3(256), 0, s, S->ft, rows, S->n_rows, S->cutoff, mag);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// d|X| (batch, frames, cutoff) -> d wav (batch, n_samples) for the LAST ctta_stft_magnitude call on this handle
extern "C" ctta_status ctta_stft_magnitude_bwd(ctta_stft* S, const float* dmag, int batch, int n_samples, float* dwav,
                                               void* stream) {
  CTTA_REQUIRE(S && dmag && dwav, "stft_magnitude_bwd: null pointer");
  CTTA_TRY(stft_check(S, batch, n_samples, "stft_magnitude_bwd"));
  hipStream_t s = (hipStream_t)stream;
  WsBind bind(S->splitws);
  const int N = S->n_fft, cutoff = S->cutoff, frames = n_samples / S->hop + 1;
  const long long rows = (long long)batch * frames;
  if (!S->dft) {   // first backward: gradient-side buffers and the transposed basis (leading bf16 part)
    const size_t rmax = (size_t)S->max_batch * S->frames_max;
    CTTA_CHECK_HIP(hipMalloc((void**)&S->dft, rmax * S->kpad * 2));
    CTTA_CHECK_HIP(hipMalloc((void**)&S->dframes, rmax * N * 4));
    CTTA_CHECK_HIP(hipMalloc((void**)&S->basis_t, (size_t)N * S->kpad * 2));
    CTTA_CHECK_HIP(ctta_zero_async(S->basis_t, (size_t)N * S->kpad * 2, s));
    CTTA_TRY(ctta_transpose_bf16(S->b[0], 0, 2 * cutoff, N, N, 0, S->basis_t, 0, S->kpad, 1, s));
  }
  const long long total = rows * S->kpad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(stft_mag_bwd_kernel, dim3(blocks), dim3(256), 0, s, S->ft, dmag, rows, S->n_rows, cutoff, S->kpad, S->dft);
  CTTA_LAUNCH_CHECK();
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = S->dft; d.c0 = S->kpad;
  d.batch = 1; d.hi = (int)rows; d.wi = 1; d.ho = (int)rows; d.wo = 1;
  d.w = S->basis_t; d.k_pad = S->kpad; d.n = N;
  d.out = S->dframes; d.ldc = N; d.out_f32 = 1;
  CTTA_TRY(ctta_conv_gemm(&d, s));
  int ob = (n_samples + 255) / 256;
  if (ob > 2048) ob = 2048;
  hipLaunchKernelGGL(stft_loss_overlap_add_kernel, dim3(ob, batch), dim3(256), 0, s, S->dframes, frames, N, S->hop, n_samples,
                     dwav);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
