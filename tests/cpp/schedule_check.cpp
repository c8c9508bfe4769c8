// Host-side check of needle_amd/csrc/stft32_schedule.h: every frame pair of a launch belongs to exactly one workgroup,
// workgroups of an XCD walk its part front to back, nothing lies beyond the launch.  Exit code 0 = all shapes pass.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../needle_amd/csrc/stft32_schedule.h"

using needle::Stft32Schedule;

static int check(uint64_t total, uint32_t ppb, uint64_t slots, bool guided, uint32_t tenths = 10) {
  const Stft32Schedule s = needle::stft32_schedule(total, ppb, slots, guided, tenths);
  std::vector<uint8_t> seen(total, 0);
  std::vector<uint32_t> prev_last(8, 0);
  uint64_t smaller = 0;
  for (uint32_t b = 0; b < 8u * s.blocks_per_xcd; b++) {
    uint32_t f, l;
    needle::stft32_block_range(s, b, (uint32_t)total, &f, &l);
    if (f > l || l > total) return std::printf("range [%u,%u) of block %u beyond %llu\n", f, l, b, (unsigned long long)total), 1;
    if (f < l) {
      const uint32_t x = b & 7u;
      if (prev_last[x] != 0 && f != prev_last[x]) return std::printf("block %u does not continue its XCD's part\n", b), 1;
      prev_last[x] = l;
      if (l - f < ppb) smaller++;
    }
    for (uint32_t g = f; g < l; g++)
      if (seen[g]++) return std::printf("pair %u twice (total %llu ppb %u)\n", g, (unsigned long long)total, ppb), 1;
  }
  for (uint64_t g = 0; g < total; g++)
    if (!seen[g]) return std::printf("pair %llu missing (total %llu ppb %u slots %llu guided %d)\n", (unsigned long long)g,
                                     (unsigned long long)total, ppb, (unsigned long long)slots, (int)guided), 1;
  if (guided && tenths == 10 && ppb >= 8 && (total + ppb - 1) / ppb >= 8 * 3 * slots + 64 && smaller < slots)
    return std::printf("guided schedule without a fine tail (total %llu)\n", (unsigned long long)total), 1;
  return 0;
}

int main() {
  int bad = 0;
  for (uint64_t total : {0ull, 1ull, 7ull, 8ull, 9ull, 100ull, 2907ull, 11626ull, 81396ull, 81397ull, 500001ull, 5888000ull})
    for (uint32_t ppb : {1u, 4u, 6u, 8u, 16u, 23u, 24u, 32u, 40u})
      for (uint64_t slots : {0ull, 1ull, 12ull, 96ull})
        for (bool guided : {false, true}) bad += check(total, ppb, slots, guided);
  for (uint64_t total = 20000; total < 60000; total += 997) bad += check(total, 24, 96, true);
  // every tail the tuning switch accepts (NEEDLE_HIP_STFT_GUIDED: tenths of a round, 1 .. 20), on launches barely past the
  // guard of 1.25 rounds: a tail longer than an XCD's part used to wrap around
  for (uint32_t tenths = 1; tenths <= 20; tenths++)
    for (uint64_t total : {8ull * 24 * 96 * 5 / 4, 8ull * 24 * 96 * 5 / 4 + 191, 8ull * 24 * 96 * 3 / 2, 8ull * 24 * 96 * 2, 500001ull})
      bad += check(total, 24, 96, true, tenths);
  std::printf("%s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}
