// GPU resampler + down-mix: the step in front of the fingerprinter.  The reference does this with FFmpeg's
// swresample (needle/src/audio/analyzer.rs:180-187,231-282), a third-party library that cannot be reproduced
// bit for bit; this front-end is this project's own specification (oracle/ora_resample.h states it):
// integer down-mix, rational polyphase Kaiser-windowed-sinc FIR to 11025 Hz with f32 coefficients and f32 fused
// multiply-adds in tap order, round-to-nearest-even, clamp.  Integer/f32 work with a fixed operation order, so
// the GPU output equals the oracle's bit for bit.
//
// HBM-stream-bound: each input sample is read once from HBM (a workgroup stages the span of input its 256
// outputs need in LDS, neighbouring workgroups overlap by the filter length only) and each output written once.
#include "hipctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

namespace needle {

namespace {

constexpr int kTarget = 11025;
constexpr int kZeroCrossings = 16;
constexpr double kRolloff = 0.94;
constexpr double kKaiserBeta = 9.0;
constexpr int kOutPerBlock = 256;

struct Design {
  int L = 1, M = 1, T = 0;
  std::vector<float> coef;  // [L][T]
  float *d_coef = nullptr;
};

double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  for (int k = 1; k <= 40; k++) {
    term *= (x / (2.0 * k)) * (x / (2.0 * k));
    sum += term;
  }
  return sum;
}

int gcd_i(int a, int b) {
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

void design_filter(int rate, Design *d) {
  if (rate == kTarget) {  // nothing to resample: identity (a single unit tap), only the down-mix applies
    d->L = d->M = 1;
    d->T = 2;
    d->coef = {1.0f, 0.0f};
    return;
  }
  const int g = gcd_i(kTarget, rate);
  d->L = kTarget / g;
  d->M = rate / g;
  const double ratio = (double)d->L / (double)d->M;
  const double scale = kRolloff * (ratio < 1.0 ? ratio : 1.0);
  const int half = (int)std::ceil(kZeroCrossings / (ratio < 1.0 ? ratio : 1.0));
  d->T = 2 * half;
  d->coef.assign((size_t)d->L * d->T, 0.f);
  const double pi = 3.14159265358979323846;
  const double i0b = bessel_i0(kKaiserBeta);
  std::vector<double> tmp(d->T);
  for (int p = 0; p < d->L; p++) {
    double sum = 0.0;
    for (int k = 0; k < d->T; k++) {
      const double tau = (double)(k - half + 1) - (double)p / (double)d->L;
      const double x = tau * scale;
      const double sinc = x == 0.0 ? 1.0 : std::sin(pi * x) / (pi * x);
      const double u = tau / (double)half;
      const double win = std::fabs(u) >= 1.0 ? 0.0 : bessel_i0(kKaiserBeta * std::sqrt(1.0 - u * u)) / i0b;
      tmp[k] = scale * sinc * win;
      sum += tmp[k];
    }
    for (int k = 0; k < d->T; k++) d->coef[(size_t)p * d->T + k] = (float)(tmp[k] / sum);
  }
}

std::mutex g_mu;
std::map<std::pair<int, int>, Design> g_designs;  // (device, rate)

struct RsStream {
  uint64_t in_off;   // s16 values into the input arena
  uint64_t n_in;     // samples per channel
  uint64_t out_off;  // samples into the output arena
  uint64_t n_out;
  uint32_t block_base;
  uint32_t pad;
};

template <int CH>
__global__ __launch_bounds__(256) void resample_kernel(const int16_t *__restrict__ in, const RsStream *__restrict__ streams,
                                                       int num_streams, const float *__restrict__ coef, int L, int M,
                                                       int T, int span, int16_t *__restrict__ out) {
  extern __shared__ float stage[];  // `span` mono samples this workgroup's outputs read
  int lo = 0, hi = num_streams - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RsStream st = streams[lo];
  const uint64_t m0 = (uint64_t)(blockIdx.x - st.block_base) * kOutPerBlock;
  const int half = T / 2;
  const long long first0 = (long long)((m0 * (uint64_t)M) / (uint64_t)L) - half + 1;
  const int16_t *src = in + st.in_off;
  for (int i = threadIdx.x; i < span; i += blockDim.x) {
    const long long idx = first0 + i;
    int s = 0;
    if (idx >= 0 && (uint64_t)idx < st.n_in) {
      if (CH == 1) s = src[idx];
      else s = ((int)src[2 * idx] + (int)src[2 * idx + 1]) / 2;  // integer down-mix, C truncation
    }
    stage[i] = (float)s;
  }
  __syncthreads();
  const uint64_t m = m0 + threadIdx.x;
  if (m >= st.n_out) return;
  const uint64_t pos = m * (uint64_t)M;
  const long long center = (long long)(pos / (uint64_t)L);
  const int phase = (int)(pos % (uint64_t)L);
  const int rel = (int)(center - half + 1 - first0);
  const float *c = coef + (size_t)phase * T;
  float acc = 0.0f;
  for (int k = 0; k < T; k++) acc = fmaf(c[k], stage[rel + k], acc);
  float r = rintf(acc);
  r = fminf(fmaxf(r, -32768.0f), 32767.0f);
  out[st.out_off + m] = (int16_t)r;
}

}  // namespace

size_t resample_out_len(size_t n_in, int rate) {
  if (rate == kTarget) return n_in;
  const int g = gcd_i(kTarget, rate);
  const unsigned long long L = kTarget / g, M = rate / g;
  return (size_t)(((unsigned long long)n_in * L + M - 1) / M);
}

// streams: (in_off in s16 values, n_in samples per channel, out_off samples); d_out receives mono s16 @ 11025 Hz
Status gpu_resample_device(const int16_t *d_in, const std::vector<ResampleSpan> &spans, int channels, int rate,
                           int16_t *d_out, bool sync) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  if (channels != 1 && channels != 2) return Status::Make(NeedleError_InvalidArgument, "resample: channels must be 1 or 2");
  if (rate < 2000 || rate > 768000) return Status::Make(NeedleError_InvalidArgument, "resample: unsupported sample rate");
  Status s = ensure_device();
  if (!s.ok()) return s;
  int dev = 0;
  NEEDLE_HIP_TRY(hipGetDevice(&dev));
  Design *d;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    d = &g_designs[{dev, rate}];
    if (d->T == 0) {
      design_filter(rate, d);
      NEEDLE_HIP_TRY(hipMalloc((void **)&d->d_coef, d->coef.size() * sizeof(float)));
      NEEDLE_HIP_TRY(hipMemcpy(d->d_coef, d->coef.data(), d->coef.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  std::vector<RsStream> meta;
  uint64_t blocks = 0;
  for (const ResampleSpan &sp : spans) {
    RsStream m;
    m.in_off = sp.in_off;
    m.n_in = sp.n_in;
    m.out_off = sp.out_off;
    m.n_out = resample_out_len(sp.n_in, rate);
    m.block_base = (uint32_t)blocks;
    m.pad = 0;
    blocks += (m.n_out + kOutPerBlock - 1) / kOutPerBlock;
    if (m.n_out) meta.push_back(m);
  }
  if (blocks > 0x7FFFFFFFull) return Status::Make(NeedleError_InvalidArgument, "resample: batch too large for one launch");
  if (!meta.empty()) {
    hipStream_t stream = library_stream();
    static std::map<int, std::pair<DeviceBuffer<RsStream> *, PinnedStage *>> ws;
    auto &w = ws[dev];
    if (!w.first) {
      w.first = new DeviceBuffer<RsStream>();
      w.second = new PinnedStage();
    }
    if (!(s = w.first->reserve(meta.size())).ok()) return s;
    if (!(s = w.second->acquire(meta.size() * sizeof(RsStream))).ok()) return s;
    std::memcpy(w.second->ptr, meta.data(), meta.size() * sizeof(RsStream));
    NEEDLE_HIP_TRY(hipMemcpyAsync(w.first->ptr, w.second->ptr, meta.size() * sizeof(RsStream), hipMemcpyHostToDevice, stream));
    w.second->mark(stream);
    const int span = (int)(((uint64_t)(kOutPerBlock - 1) * d->M) / d->L) + d->T + 2;
    KernelTimer timer("resample");
    if (channels == 1)
      hipLaunchKernelGGL(resample_kernel<1>, dim3((uint32_t)blocks), dim3(256), span * sizeof(float), stream, d_in,
                         w.first->ptr, (int)meta.size(), d->d_coef, d->L, d->M, d->T, span, d_out);
    else
      hipLaunchKernelGGL(resample_kernel<2>, dim3((uint32_t)blocks), dim3(256), span * sizeof(float), stream, d_in,
                         w.first->ptr, (int)meta.size(), d->d_coef, d->L, d->M, d->T, span, d_out);
    NEEDLE_HIP_TRY(hipGetLastError());
  }
  if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(library_stream()));
  return Status::Ok();
}

Status gpu_resample_host(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values, int channels,
                         int rate, std::vector<std::vector<int16_t>> *out) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  if (channels != 1 && channels != 2) return Status::Make(NeedleError_InvalidArgument, "resample: channels must be 1 or 2");
  const size_t n = pcm.size();
  out->assign(n, {});
  std::vector<ResampleSpan> spans(n);
  uint64_t in_total = 0, out_total = 0;
  for (size_t i = 0; i < n; i++) {
    spans[i].in_off = in_total;
    spans[i].n_in = num_values[i] / (size_t)channels;
    spans[i].out_off = out_total;
    in_total += (num_values[i] + 1) & ~(uint64_t)1;
    out_total += (resample_out_len(spans[i].n_in, rate) + 1) & ~(uint64_t)1;
  }
  DeviceBuffer<int16_t> d_in, d_out;
  if (!(s = d_in.reserve(std::max<uint64_t>(in_total, 1))).ok()) return s;
  if (!(s = d_out.reserve(std::max<uint64_t>(out_total, 1))).ok()) return s;
  hipStream_t stream = library_stream();
  for (size_t i = 0; i < n; i++)
    if (num_values[i])
      NEEDLE_HIP_TRY(hipMemcpyAsync(d_in.ptr + spans[i].in_off, pcm[i], num_values[i] * sizeof(int16_t),
                                    hipMemcpyHostToDevice, stream));
  s = gpu_resample_device(d_in.ptr, spans, channels, rate, d_out.ptr, false);
  if (!s.ok()) return s;
  std::vector<int16_t> host(std::max<uint64_t>(out_total, 1));
  NEEDLE_HIP_TRY(hipMemcpyAsync(host.data(), d_out.ptr, out_total * sizeof(int16_t), hipMemcpyDeviceToHost, stream));
  NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  for (size_t i = 0; i < n; i++) {
    const size_t k = resample_out_len(spans[i].n_in, rate);
    (*out)[i].assign(host.begin() + spans[i].out_off, host.begin() + spans[i].out_off + k);
  }
  return Status::Ok();
}

}  // namespace needle
