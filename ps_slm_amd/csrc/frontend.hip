// Audio front end of the SenseVoice path on the GPU: Kaldi-compatible log-mel filterbank + low-frame-rate stacking + CMVN,
// i.e. what funasr's WavFrontend does on the host inside the reference's dataset (Multitask/dataset/
// speech_dataset_large.py:133-146 -> funasr frontends/wav_frontend.py -> torchaudio.compliance.kaldi.fbank; third-party,
// restated from the published algorithm: oracle/fbank_oracle.py, PARITY UNPINNED).
//   fbank_kernel: one 256-thread block per frame (25 ms = 400 samples at 16 kHz, hop 160): DC removal, pre-emphasis,
//     window, zero-pad to 512, radix-2 FFT in LDS (9 stages, one butterfly per thread and stage), power spectrum, dense
//     [n_mels, 257] mel matrix, log.  48 k frames per 16-utterance batch of 30-s audio: microseconds of work, kept on the
//     device so that features never cross PCIe.
//   lfr_cmvn_kernel: out[i, m*D + k] = (fb[frame(i, m), k] + mean) * scale, frame = clamp(i*n + m - (M-1)/2, 0, T-1).
#include "common.h"
#include "../../include/tasu_hip.h"

namespace {

constexpr int NFFT = 512, LOG2_NFFT = 9;

__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wave, int win, int shift, float scale,
                                                    const float* __restrict__ window, const float* __restrict__ mel, int n_mels,
                                                    float preemph, float* __restrict__ out) {
  __shared__ float re[NFFT], im[NFFT], red[4];
  const int t = threadIdx.x;
  const float* x = wave + (size_t)blockIdx.x * shift;
  // raw samples (two per thread), frame mean
  const float a0 = t < win ? x[t] * scale : 0.f;
  const float a1 = t + 256 < win ? x[t + 256] * scale : 0.f;
  const float mean = block_sum<4>(a0 + a1, red) / (float)win;
  re[t] = a0;
  re[t + 256] = a1;
  __syncthreads();
  // y[i] = (x[i] - mean) - p * (x[max(i-1, 0)] - mean), windowed, stored bit-reversed for the in-place DIT FFT
  float y[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = t + h * 256;
    float v = 0.f;
    if (i < win) {
      const float cur = re[i] - mean, prev = re[i > 0 ? i - 1 : 0] - mean;
      v = (cur - preemph * prev) * window[i];
    }
    y[h] = v;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = t + h * 256;
    const int r = (int)(__brev((unsigned)i) >> (32 - LOG2_NFFT));
    re[r] = y[h];
    im[r] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < LOG2_NFFT; ++s) {
    const int half = 1 << s;
    const int pos = t & (half - 1);
    const int i0 = ((t >> s) << (s + 1)) + pos, i1 = i0 + half;
    float sn, cs;
    sincospif(-(float)pos / (float)half, &sn, &cs);       // exp(-2 pi i pos / (2 half))
    const float br = re[i1] * cs - im[i1] * sn, bi = re[i1] * sn + im[i1] * cs;
    const float ar = re[i0], ai = im[i0];
    re[i0] = ar + br;
    im[i0] = ai + bi;
    re[i1] = ar - br;
    im[i1] = ai - bi;
    __syncthreads();
  }
  // power spectrum, bins 0..256 -> re[0..256]
  const float p0 = re[t] * re[t] + im[t] * im[t];
  const float p256 = t == 0 ? re[256] * re[256] + im[256] * im[256] : 0.f;
  __syncthreads();
  re[t] = p0;
  if (t == 0) re[256] = p256;
  __syncthreads();
  for (int m = t; m < n_mels; m += 256) {
    const float* w = mel + (size_t)m * (NFFT / 2 + 1);
    float e = 0.f;
    for (int k = 0; k <= NFFT / 2; ++k) e += re[k] * w[k];
    out[(size_t)blockIdx.x * n_mels + m] = logf(fmaxf(e, 1.1920928955078125e-07f));
  }
}

__global__ __launch_bounds__(256) void lfr_cmvn_kernel(const float* __restrict__ fb, int T, int D, int lfr_m, int lfr_n,
                                                       const float* __restrict__ means, const float* __restrict__ scales,
                                                       float* __restrict__ out) {
  const int i = blockIdx.x, W = lfr_m * D, pad = (lfr_m - 1) / 2;
  for (int c = threadIdx.x; c < W; c += 256) {
    const int mi = c / D, k = c - mi * D;
    const int f = min(max(i * lfr_n + mi - pad, 0), T - 1);
    float v = fb[(size_t)f * D + k];
    if (means) v = (v + means[c]) * scales[c];
    out[(size_t)i * W + c] = v;
  }
}

}  // namespace

extern "C" int tasu_fbank(const float* wave, int64_t n_samples, float scale, int win, int shift, const float* window,
                          const float* mel, int n_mels, float preemph, float* out, void* stream) {
  if (!wave || !window || !mel || !out || win <= 0 || win > NFFT || shift <= 0 || n_mels <= 0) return TASU_ERR_ARG;
  if (n_samples < win) return TASU_OK;                                        // no frame (snip_edges)
  const int64_t frames = 1 + (n_samples - win) / shift;
  if (frames > 0x7fffffff) return TASU_ERR_ARG;
  TASU_LAUNCH(fbank_kernel, dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream, wave, win, shift, scale, window, mel,
              n_mels, preemph, out);
  return TASU_OK;
}

extern "C" int tasu_lfr_cmvn(const float* fb, int T, int D, int lfr_m, int lfr_n, const float* means, const float* scales,
                             float* out, void* stream) {
  if (!fb || !out || T <= 0 || D <= 0 || lfr_m <= 0 || lfr_n <= 0 || (means && !scales)) return TASU_ERR_ARG;
  const int T_lfr = (T + lfr_n - 1) / lfr_n;
  TASU_LAUNCH(lfr_cmvn_kernel, dim3(T_lfr), dim3(256), 0, (hipStream_t)stream, fb, T, D, lfr_m, lfr_n, means, scales, out);
  return TASU_OK;
}
