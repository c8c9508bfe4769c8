// Which frame pairs a workgroup of stft_chroma32_kernel takes (stft32_kernel.h; plain C++, shared with the launchers).
//
// The timeline of a launch is cut into eight contiguous parts, one per XCD (workgroup b runs on XCD b & 7: neighbouring
// frames overlap by two thirds and meet in that XCD's L2), and every XCD walks its part front to back in workgroups of
// size[0] pairs.  GUIDED: the last round's worth of pairs of every part goes in workgroups of a half, a quarter and an
// eighth of that size instead.  The device holds three workgroups per CU and a workgroup of 16 pairs runs ~70 us: the end
// of a launch is a ramp of that length on which slots run dry one by one, and the next launch of the stream does not start
// before the last workgroup has retired.  Smaller workgroups at the END shorten the ramp to an eighth at the price of a few
// hundred more workgroup prologues (46 loop-invariant loads each), not of thousands as a smaller size throughout would.
#pragma once

#include <stdint.h>

namespace needle {

struct Stft32Schedule {
  uint32_t pairs_per_xcd = 0;
  uint32_t count[3] = {0, 0, 0};  // workgroups per XCD at size[0], size[1], size[2]; the rest: size[3]
  uint32_t size[4] = {1, 1, 1, 1};
  uint32_t blocks_per_xcd = 0;    // grid = 8 x this
};

inline Stft32Schedule stft32_schedule(uint64_t total_pairs, uint32_t pairs_per_block, uint64_t slots_per_xcd, bool guided,
                                      uint32_t tail_tenths = 10) {
  Stft32Schedule s;
  const uint32_t ppb = pairs_per_block < 1 ? 1 : pairs_per_block;
  const uint64_t blocks = ((total_pairs + ppb - 1) / ppb + 7) / 8;  // per XCD, all at full size
  for (int k = 0; k < 4; k++) s.size[k] = ppb;
  s.pairs_per_xcd = (uint32_t)(blocks * ppb);
  s.count[0] = s.blocks_per_xcd = (uint32_t)blocks;
  if (!guided || ppb < 8 || 4 * blocks < 5 * slots_per_xcd || tail_tenths < 1 || tail_tenths > 20) return s;  // short launches: every slot gets one workgroup anyway
  const uint64_t part = (total_pairs + 7) / 8;
  uint64_t tail = slots_per_xcd * ppb * tail_tenths / 10;           // one round's worth of pairs (tail_tenths: tuning)
  if (tail > part) tail = part;                                     // (tenths > 12: the guard above only promises 1.25 rounds)
  s.pairs_per_xcd = (uint32_t)part;
  s.count[0] = (uint32_t)((part - tail) / ppb);
  const uint64_t rest = part - (uint64_t)s.count[0] * ppb;
  s.size[1] = ppb / 2;
  s.size[2] = ppb / 4;
  s.size[3] = ppb / 8;
  s.count[1] = (uint32_t)((rest / 2 + s.size[1] - 1) / s.size[1]);
  const uint64_t used1 = (uint64_t)s.count[1] * s.size[1];
  const uint64_t rest2 = used1 > rest ? 0 : rest - used1;
  s.count[2] = (uint32_t)((rest2 / 2 + s.size[2] - 1) / s.size[2]);
  const uint64_t used = (uint64_t)s.count[2] * s.size[2];
  const uint64_t rest3 = used > rest2 ? 0 : rest2 - used;
  s.blocks_per_xcd = s.count[0] + s.count[1] + s.count[2] + (uint32_t)((rest3 + s.size[3] - 1) / s.size[3]);
  return s;
}

// [first, last) of workgroup `block` (the kernel's own arithmetic, also used by the host-side test of the schedule)
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    void stft32_block_range(const Stft32Schedule &s, uint32_t block, uint32_t total_pairs, uint32_t *first, uint32_t *last) {
  const uint32_t x = block & 7u;
  uint32_t q = block >> 3, base = 0, size = s.size[0];
  if (q >= s.count[0]) {
    q -= s.count[0];
    base += s.count[0] * s.size[0];
    size = s.size[1];
    if (q >= s.count[1]) {
      q -= s.count[1];
      base += s.count[1] * s.size[1];
      size = s.size[2];
      if (q >= s.count[2]) {
        q -= s.count[2];
        base += s.count[2] * s.size[2];
        size = s.size[3];
      }
    }
  }
  const uint64_t xfirst = (uint64_t)x * s.pairs_per_xcd;
  uint64_t xend = xfirst + s.pairs_per_xcd;
  if (xend > total_pairs) xend = total_pairs;
  uint64_t f = xfirst + base + (uint64_t)q * size;
  if (f > xend) f = xend;  // (also when the part itself lies beyond the end: xend < xfirst, an empty range)
  uint64_t l = f + size;
  if (l > xend) l = xend;
  if (l < f) l = f;
  *first = (uint32_t)f;
  *last = (uint32_t)l;
}

}  // namespace needle
