// The f32 first pass of the STFT in a translation unit of its own: it is compiled with -fno-slp-vectorize.  Left to
// itself the SLP vectoriser turns this kernel's complex arithmetic into v_pk_*_f32 (no faster per flop on gfx950 than
// two plain instructions) at the price of 135 register moves per frame pair and, because packed operands need aligned
// register pairs, 74 spilled VGPRs at three waves per SIMD; without it the kernel takes 163 VGPRs and no scratch.
#include "fingerprint32.h"

#include <algorithm>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "hipctx.h"
#include "stft32_kernel.h"

namespace needle {

Status launch_stft_chroma32(int channels, const Stft32Schedule &schedule, hipStream_t stream, const int16_t *d_pcm,
                            const stft::FpStream *streams, int num_streams, const core::cf *tw32, const float *win32,
                            const uint16_t *bin_slot, const uint32_t *fold_tab, double *chroma, float *energy,
                            uint32_t total_pairs, uint32_t *zero_words, uint32_t num_zero_words, hipEvent_t start,
                            hipEvent_t stop) {
  const uint32_t grid = 8u * schedule.blocks_per_xcd;
  if (grid == 0) return Status::Ok();
  size_t lds_bytes = core::kLds2Slots * sizeof(core::cf);  // 34 832 B: under the 64 KiB that needs no opt-in
  // experiment (NEEDLE_HIP_STFT_SHARE): a larger request caps the workgroups per CU (160 KiB of LDS) and leaves registers
  // and LDS for the previous job's tail kernels running beside this launch; <= 64 KiB so that no opt-in is needed
  static const size_t lds_request = [] {
    const char *e = getenv("NEEDLE_HIP_STFT_LDS_BYTES");
    return e ? (size_t)std::min(65536l, std::max(0l, atol(e))) : (size_t)0;
  }();
  lds_bytes = std::max(lds_bytes, lds_request);
  if (channels == 1)
    hipExtLaunchKernelGGL((stft::stft_chroma32_kernel<1, kStft32WavesPerSimd>), dim3(grid), dim3(256), lds_bytes, stream, start, stop, 0,
                          d_pcm, streams, num_streams, tw32, win32, bin_slot, fold_tab, chroma, energy, total_pairs, schedule,
                          zero_words, num_zero_words);
  else
    hipExtLaunchKernelGGL((stft::stft_chroma32_kernel<2, kStft32WavesPerSimd>), dim3(grid), dim3(256), lds_bytes, stream, start, stop, 0,
                          d_pcm, streams, num_streams, tw32, win32, bin_slot, fold_tab, chroma, energy, total_pairs, schedule,
                          zero_words, num_zero_words);
  NEEDLE_HIP_TRY(hipGetLastError());
  return Status::Ok();
}

}  // namespace needle
