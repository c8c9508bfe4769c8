// HARNESS, not product: synthetic episode libraries generated directly in HBM, for the tests and measurements at
// BASELINE.json configs[4]'s full size (2000 episodes x 45 min = 59.5 GB of opening-window PCM; synthesising that on
// 16 host CPUs takes 13 minutes, on the device a few hundred milliseconds).  Built into libneedle_synth_hip.so; the
// product library does not link it.
//
// Content follows SURVEY.md §8(d) like csrc/synth.c does on the host -- tonal (notes of three partials from the
// equal-tempered scale 110 .. 1760 Hz, 20 ms raised-cosine attack and release, peak about half of full scale, noise
// 40 dB below), a unique body per episode, one shared intro at an episode-specific offset -- but it is its own
// generator (f32 arithmetic, a sine table): the two need not agree sample for sample, the checker reads the PCM back.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint32_t mix(uint64_t x) {  // splitmix64 finaliser, folded to 32 bits
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return (uint32_t)(x ^ (x >> 32));
}

constexpr int kNote = 6007;   // samples per note (0.545 s; prime, so note boundaries drift against the 1365-sample hop)
constexpr int kRamp = 220;    // 20 ms

// sample n of the stream with this seed (notes only)
__device__ __forceinline__ float tone(uint64_t seed, uint32_t n, const float *sine, const uint32_t *inc) {
  const uint32_t q = n / kNote, r = n - q * kNote;
  const uint32_t h = mix(seed * 0x9e3779b97f4a7c15ull + q);
  float acc = 0.0f, wsum = 0.0f;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const uint32_t semitone = ((h >> (8 * p)) & 0xffu) % 49u;       // 110 Hz * 2^(s / 12), s = 0 .. 48
    const float a = 1.0f / (float)(1 + p) + 0.25f * (float)((h >> (24 + 2 * p)) & 3u);
    const uint32_t phase = inc[semitone] * r + (h << (3 * p));      // wraps mod 2^32: one period of the table
    acc += a * sine[phase >> 20];
    wsum += a;
  }
  float env = 1.0f;
  if (r < kRamp) env = 0.5f - 0.5f * sine[(1024u + (r * 2048u) / kRamp) & 4095u];                    // 0.5 (1 - cos)
  else if (r >= (uint32_t)(kNote - kRamp)) env = 0.5f - 0.5f * sine[(1024u + ((kNote - r) * 2048u) / kRamp) & 4095u];
  return 0.5f * env * acc / wsum;
}

__global__ __launch_bounds__(256) void synth_library_kernel(int16_t *out, uint64_t stride, uint32_t first_episode,
                                                            uint32_t samples, const uint32_t *intro_off, uint32_t intro_len,
                                                            uint64_t seed_base, uint64_t intro_seed) {
  __shared__ float sine[4096];
  __shared__ uint32_t inc[49];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sine[i] = __sinf(6.283185307179586f * (float)i / 4096.0f);
  if (threadIdx.x < 49) inc[threadIdx.x] = (uint32_t)(110.0 * exp2((double)threadIdx.x / 12.0) / 11025.0 * 4294967296.0);
  __syncthreads();
  const uint32_t k = blockIdx.y;
  const uint32_t episode = first_episode + k;
  const uint64_t seed = seed_base ^ (uint64_t)episode;
  const uint32_t off = intro_off[k];
  int16_t *dst = out + (uint64_t)k * stride;
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < samples; n += gridDim.x * blockDim.x) {
    const bool in_intro = n >= off && n - off < intro_len;
    const float v = in_intro ? tone(intro_seed, n - off, sine, inc) : tone(seed, n, sine, inc);
    const float noise = ((float)(mix(seed * 0xd1342543de82ef95ull + n) & 0xffffu) - 32767.5f) * (1.0f / 32768.0f);  // [-1, 1)
    const float s = 32767.0f * (v + 0.005f * noise);   // episode-specific noise on the shared segment too
    dst[n] = (int16_t)__float2int_rn(fminf(fmaxf(s, -32768.0f), 32767.0f));
  }
}

// ---- the HOSTILE corpus (round 6): what the tonal corpus above never shows the first pass's certification radius and the
// scan's aligned-window filter -- broadband, speech-like material (a noise carrier plus three drifting formant partials under a
// syllabic envelope, ~4 syllables per second), noise 20 dB under the programme everywhere, and inside the search window of
// some episodes a stretch of DIGITAL SILENCE (exact zeros: every chroma row is cut at the 0.01 norm, the hashes are one
// constant) and a stretch of one SUSTAINED CHORD (the same three partials in every episode that has it: constant hashes
// within an episode and equal ones across episodes -- an S x S block of matching cells per pair, ~2 S runs).  The shared
// intro stays tonal (it has to be found).  Segment descriptor per episode: {silence offset, silence length, chord offset,
// chord length} in samples, 0 length = none.
__device__ __forceinline__ float hostile_body(uint64_t seed, uint32_t n, const float *sine) {
  constexpr uint32_t kSyllable = 2756;  // 0.25 s
  const uint32_t q = n / kSyllable, r = n - q * kSyllable;
  const uint32_t h = mix(seed * 0x9e3779b97f4a7c15ull + 0x5151ull + q);
  const float height = 0.15f + 0.85f * (float)(h & 0xffu) * (1.0f / 255.0f);         // some syllables nearly vanish
  const float env = height * (0.5f - 0.5f * sine[(1024u + (r * 4096u) / kSyllable) & 4095u]);
  float formants = 0.0f;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const uint32_t hz = 250u + ((h >> (8 + 7 * p)) & 0x7fu) * 22u + 700u * (uint32_t)p;  // 250 .. 4450 Hz, not on any scale
    const uint32_t phase = (uint32_t)(((uint64_t)hz << 32) / 11025u) * r + (h << (5 * p));
    formants += sine[phase >> 20];
  }
  const float noise = ((float)(mix(seed * 0xa0761d6478bd642full + n) & 0xffffu) - 32767.5f) * (1.0f / 32768.0f);
  return 0.45f * env * (0.6f * noise + 0.4f * (1.0f / 3.0f) * formants);
}

__global__ __launch_bounds__(256) void synth_hostile_kernel(int16_t *out, uint64_t stride, uint32_t first_episode, uint32_t samples,
                                                            const uint32_t *intro_off, uint32_t intro_len, const uint32_t *segments,
                                                            uint64_t seed_base, uint64_t intro_seed) {
  __shared__ float sine[4096];
  __shared__ uint32_t inc[49];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sine[i] = __sinf(6.283185307179586f * (float)i / 4096.0f);
  if (threadIdx.x < 49) inc[threadIdx.x] = (uint32_t)(110.0 * exp2((double)threadIdx.x / 12.0) / 11025.0 * 4294967296.0);
  __syncthreads();
  const uint32_t k = blockIdx.y;
  const uint64_t seed = seed_base ^ (uint64_t)(first_episode + k);
  const uint32_t off = intro_off[k];
  const uint32_t sil_off = segments[4 * k], sil_len = segments[4 * k + 1], chord_off = segments[4 * k + 2], chord_len = segments[4 * k + 3];
  int16_t *dst = out + (uint64_t)k * stride;
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < samples; n += gridDim.x * blockDim.x) {
    if (n >= sil_off && n - sil_off < sil_len) {  // digital silence: no noise floor either
      dst[n] = 0;
      continue;
    }
    float v;
    float floor_level = 0.05f;  // -20 dB under the programme's 0.5
    if (n >= off && n - off < intro_len) {
      v = tone(intro_seed, n - off, sine, inc);
    } else if (n >= chord_off && n - chord_off < chord_len) {  // A3, C#4, E4 from the scale table, constant amplitude
      const uint32_t r = n - chord_off;
      v = 0.4f * (1.0f / 3.0f) * (sine[(inc[12] * r) >> 20] + sine[(inc[16] * r) >> 20] + sine[(inc[19] * r) >> 20]);
      floor_level = 0.0005f;  // (a sustained chord over -60 dB: the hashes must stay within the threshold of each other)
    } else {
      v = hostile_body(seed, n, sine);
    }
    const float noise = ((float)(mix(seed * 0xd1342543de82ef95ull + n) & 0xffffu) - 32767.5f) * (1.0f / 32768.0f);
    const float s = 32767.0f * (v + floor_level * noise);
    dst[n] = (int16_t)__float2int_rn(fminf(fmaxf(s, -32768.0f), 32767.0f));
  }
}

}  // namespace

extern "C" {

// The hostile corpus: as needle_synth_hip_library, plus d_segments[4 * k ..] = {silence offset, silence length, chord offset,
// chord length} of episode k (device, samples; a length of 0 = the episode has none).
int needle_synth_hip_library_hostile(int16_t *d_out, uint64_t stride, uint32_t n_eps, uint32_t first_episode, uint32_t samples,
                                     const uint32_t *d_intro_off, uint32_t intro_len, const uint32_t *d_segments, uint64_t seed_base,
                                     uint64_t intro_seed, void *stream) {
  if (!n_eps || !samples) return 0;
  const uint32_t bx = (uint32_t)((samples + 256u * 16u - 1) / (256u * 16u));
  hipLaunchKernelGGL(synth_hostile_kernel, dim3(bx, n_eps), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, stride,
                     first_episode, samples, d_intro_off, intro_len, d_segments, seed_base, intro_seed);
  return (int)hipGetLastError();
}

// Episodes first_episode .. first_episode + n_eps - 1, `samples` mono s16 values each, written to d_out + k * stride.
// d_intro_off[k] (device): where the shared intro (intro_len samples) sits in episode k.  Enqueued on `stream`
// (a hipStream_t; NULL = the default stream); returns a hipError_t.
int needle_synth_hip_library(int16_t *d_out, uint64_t stride, uint32_t n_eps, uint32_t first_episode, uint32_t samples,
                             const uint32_t *d_intro_off, uint32_t intro_len, uint64_t seed_base, uint64_t intro_seed,
                             void *stream) {
  if (!n_eps || !samples) return 0;
  const uint32_t bx = (uint32_t)((samples + 256u * 16u - 1) / (256u * 16u));
  hipLaunchKernelGGL(synth_library_kernel, dim3(bx, n_eps), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, stride,
                     first_episode, samples, d_intro_off, intro_len, seed_base, intro_seed);
  return (int)hipGetLastError();
}

}  // extern "C"
