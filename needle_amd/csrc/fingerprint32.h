// Launcher of stft_chroma32_kernel (stft32_kernel.h), which lives in fingerprint32.hip: see there for why.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "common.h"
#include "fp_core.h"
#include "stft32_schedule.h"

namespace needle {
namespace stft {
struct FpStream;
}

constexpr int kStft32WavesPerSimd = 3;  // = workgroups per CU (a workgroup puts one wave on each SIMD): 163 VGPRs

Status launch_stft_chroma32(int channels, const Stft32Schedule &schedule, hipStream_t stream, const int16_t *d_pcm,
                            const stft::FpStream *streams, int num_streams, const core::cf *tw32, const float *win32,
                            const uint16_t *bin_slot, const uint32_t *fold_tab, double *chroma, float *energy,
                            uint32_t total_pairs, uint32_t *zero_words, uint32_t num_zero_words,
                            hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// (start / stop: events bound to the dispatch itself -- hipExtLaunchKernelGGL -- instead of marker packets around it)

}  // namespace needle
