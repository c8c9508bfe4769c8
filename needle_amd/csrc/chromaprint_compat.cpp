// libneedle_chromaprint.so: libchromaprint's streaming C API (the part needle uses) over the batched GPU
// fingerprinter.  See include/needle_chromaprint.h.
#include "../../include/needle_chromaprint.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/needle_hip.h"

struct ChromaprintContextPrivate {
  int algorithm = CHROMAPRINT_ALGORITHM_DEFAULT;
  int channels = 1;
  int rate = 11025;
  bool started = false, finished = false;
  std::vector<int16_t> pcm;        // everything fed since start()
  std::vector<uint32_t> raw;       // raw fingerprint after finish()
};

extern "C" {

const char *chromaprint_get_version(void) { return "1.5.1-needle-hip"; }

ChromaprintContext *chromaprint_new(int algorithm) {
  if (algorithm != CHROMAPRINT_ALGORITHM_TEST2) return nullptr;  // the only algorithm needle uses (Context::default)
  ChromaprintContext *ctx = new (std::nothrow) ChromaprintContextPrivate();
  if (ctx) ctx->algorithm = algorithm;
  return ctx;
}

void chromaprint_free(ChromaprintContext *ctx) { delete ctx; }
int chromaprint_get_algorithm(ChromaprintContext *ctx) { return ctx ? ctx->algorithm : -1; }
int chromaprint_get_num_channels(ChromaprintContext *) { return 1; }  // channels of the internal (mono) signal
int chromaprint_get_sample_rate(ChromaprintContext *) { return needle_hip_fingerprint_sample_rate(); }
int chromaprint_get_item_duration(ChromaprintContext *) { return 1365; }
int chromaprint_get_item_duration_ms(ChromaprintContext *) { return needle_hip_fingerprint_item_duration_ms(); }
int chromaprint_get_delay(ChromaprintContext *) { return (4 + 15) * 1365 + (4096 - 1365); }
int chromaprint_get_delay_ms(ChromaprintContext *) { return needle_hip_fingerprint_delay_ms(); }

int chromaprint_start(ChromaprintContext *ctx, int sample_rate, int num_channels) {
  if (!ctx) return 0;
  // needle feeds chromaprint's own 11025 Hz (analyzer.rs:179-187); any other rate goes through the device
  // resampler first, as libchromaprint's internal resampler would
  if (sample_rate < 2000 || sample_rate > 768000) return 0;
  if (num_channels != 1 && num_channels != 2) return 0;
  ctx->channels = num_channels;
  ctx->rate = sample_rate;
  ctx->pcm.clear();
  ctx->raw.clear();
  ctx->started = true;
  ctx->finished = false;
  return 1;
}

int chromaprint_feed(ChromaprintContext *ctx, const int16_t *data, int size) {
  if (!ctx || !ctx->started || ctx->finished || size < 0 || (size && !data)) return 0;
  if (size % ctx->channels != 0) return 0;
  try {
    ctx->pcm.insert(ctx->pcm.end(), data, data + size);
  } catch (...) {
    return 0;
  }
  return 1;
}

int chromaprint_finish(ChromaprintContext *ctx) {
  if (!ctx || !ctx->started) return 0;
  if (ctx->finished) return 1;
  try {
    std::vector<int16_t> mono;
    int channels = ctx->channels;
    if (ctx->rate != needle_hip_fingerprint_sample_rate()) {  // down-mix + resample to mono 11025 Hz on the device
      mono.assign(needle_hip_resample_out_len(ctx->pcm.size() / (size_t)ctx->channels, ctx->rate) + 1, 0);
      const int16_t *in[1] = {ctx->pcm.data()};
      const size_t in_len[1] = {ctx->pcm.size()};
      int16_t *out[1] = {mono.data()};
      if (needle_hip_resample_host(in, in_len, 1, ctx->channels, ctx->rate, out) != NeedleError_Ok) return 0;
      mono.pop_back();
      channels = 1;
    }
    const std::vector<int16_t> &pcm = ctx->rate == needle_hip_fingerprint_sample_rate() ? ctx->pcm : mono;
    const size_t n = needle_hip_fingerprint_num_items(pcm.size() / (size_t)channels);
    ctx->raw.assign(n ? n : 1, 0);
    const int16_t *ptrs[1] = {pcm.data()};
    const size_t lens[1] = {pcm.size()};
    uint32_t *outs[1] = {ctx->raw.data()};
    if (needle_hip_fingerprint_host(ptrs, lens, 1, channels, 1, outs) != NeedleError_Ok) return 0;
    ctx->raw.resize(n);
    ctx->pcm.clear();
    ctx->pcm.shrink_to_fit();
  } catch (...) {
    return 0;
  }
  ctx->finished = true;
  return 1;
}

int chromaprint_get_raw_fingerprint(ChromaprintContext *ctx, uint32_t **fingerprint, int *size) {
  if (!ctx || !fingerprint || !size || !ctx->finished) return 0;
  uint32_t *out = static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * (ctx->raw.size() ? ctx->raw.size() : 1)));
  if (!out) return 0;
  if (!ctx->raw.empty()) std::memcpy(out, ctx->raw.data(), sizeof(uint32_t) * ctx->raw.size());
  *fingerprint = out;
  *size = (int)ctx->raw.size();
  return 1;
}

int chromaprint_get_raw_fingerprint_size(ChromaprintContext *ctx, int *size) {
  if (!ctx || !size || !ctx->finished) return 0;
  *size = (int)ctx->raw.size();
  return 1;
}

int chromaprint_clear_fingerprint(ChromaprintContext *ctx) {
  if (!ctx) return 0;
  ctx->raw.clear();
  return 1;
}

void chromaprint_dealloc(void *ptr) { std::free(ptr); }

}  // extern "C"
