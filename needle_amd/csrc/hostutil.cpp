// Host-side helpers of the needle path: std::time::Duration semantics, the bincode image of
// FrameHashes (needle/src/audio/data.rs), header MD5 (needle/src/util.rs:99-105), serde_json-style f32
// text, the step/timestamp rule of analyzer.rs:288-323 and a minimal RIFF/WAVE reader.
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "common.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace needle {

// ---- Duration ------------------------------------------------------------------------------------------
// Duration::from_secs_f{32,64}: the exact binary value, rounded to the nearest nanosecond, ties to
// even.  v = m * 2^e with an integer 53-bit m, so m * 1e9 (< 2^83) is exact in 128 bits.
static bool exact_ns(double v, ns_t *out) {
  if (!(v >= 0.0) || std::isinf(v)) return false;
  if (v == 0.0) {
    *out = 0;
    return true;
  }
  int e = 0;
  const double fr = std::frexp(v, &e);
  const uint64_t m = (uint64_t)std::ldexp(fr, 53);
  e -= 53;
  unsigned __int128 num = (unsigned __int128)m * kNanosPerSec;
  if (e >= 0) {
    if (e > 10) return false;  // beyond u64 seconds
    num <<= e;
  } else {
    const int k = -e;
    if (k >= 126) {
      *out = 0;
      return true;
    }
    const unsigned __int128 q = num >> k;
    const unsigned __int128 rem = num & ((((unsigned __int128)1) << k) - 1);
    const unsigned __int128 half = ((unsigned __int128)1) << (k - 1);
    num = q + ((rem > half || (rem == half && (q & 1))) ? 1 : 0);
  }
  if (num > (unsigned __int128)0xFFFFFFFFFFFFFFFFull) return false;
  *out = (ns_t)num;
  return true;
}

ns_t duration_from_secs_f32(float s, bool *ok) {
  ns_t r = 0;
  const bool good = exact_ns((double)s, &r);
  if (ok) *ok = good;
  return r;
}

ns_t duration_from_secs_f64(double s, bool *ok) {
  ns_t r = 0;
  const bool good = exact_ns(s, &r);
  if (ok) *ok = good;
  return r;
}

float duration_as_secs_f32(ns_t d) {
  const float secs = (float)(d / kNanosPerSec);
  const float frac = (float)(uint32_t)(d % kNanosPerSec) / 1000000000.0f;
  return secs + frac;
}

double duration_as_secs_f64(ns_t d) {
  return (double)(d / kNanosPerSec) + (double)(uint32_t)(d % kNanosPerSec) / 1000000000.0;
}

ns_t duration_mul_f32(ns_t d, float rhs, bool *ok) {
  const float prod = rhs * duration_as_secs_f32(d);
  return duration_from_secs_f32(prod, ok);
}

// ---- analyzer.rs:293-318 ---------------------------------------------------------------------------------
bool step_for_hash_duration(ns_t hash_duration, uint32_t *step) {
  const uint64_t hd_ms = hash_duration / 1000000ull;  // Duration::as_millis
  const uint64_t st = hd_ms / (uint64_t)kItemDurationMs;
  if (st == 0 || st > 0xFFFFFFFFull) return false;  // Iterator::step_by(0) panics upstream
  *step = (uint32_t)st;
  return true;
}

void attach_timestamps(const uint32_t *kept, size_t n_kept, uint32_t step, bool has_seek, ns_t seek_to,
                       std::vector<HashTs> *out) {
  const ns_t delay = (ns_t)kDelayMs * 1000000ull;
  const ns_t item = (ns_t)kItemDurationMs * 1000000ull;
  out->resize(n_kept);
  for (size_t k = 0; k < n_kept; k++) {
    const size_t i = k * (size_t)step;                       // raw item index
    ns_t ts = delay + duration_mul_f32(item, (float)i);      // :309
    if (has_seek) ts += seek_to;                             // :314-318
    (*out)[k] = HashTs{kept[k], ts};
  }
}

// ---- MD5 (RFC 1321) ---------------------------------------------------------------------------------------
namespace {
struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  static uint32_t rotl(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t *p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
        0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
        0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
        0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
        0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
        0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int R[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
    uint32_t w[16];
    for (int i = 0; i < 16; i++) std::memcpy(&w[i], p + 4 * i, 4);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; i++) {
      uint32_t f;
      int g;
      switch (i >> 4) {
        case 0: f = (B & C) | (~B & D); g = i; break;
        case 1: f = (D & B) | (~D & C); g = (5 * i + 1) & 15; break;
        case 2: f = B ^ C ^ D; g = (3 * i + 5) & 15; break;
        default: f = C ^ (B | ~D); g = (7 * i) & 15; break;
      }
      const uint32_t tmp = D;
      D = C;
      C = B;
      B = B + rotl(A + f + K[i] + w[g], R[i >> 4][i & 3]);
      A = tmp;
    }
    a += A;
    b += B;
    c += C;
    d += D;
  }
};
}  // namespace

std::string md5_hex(const uint8_t *data, size_t n) {
  Md5 h;
  size_t off = 0;
  for (; off + 64 <= n; off += 64) h.block(data + off);
  uint8_t tail[128] = {0};
  const size_t rem = n - off;
  std::memcpy(tail, data + off, rem);
  tail[rem] = 0x80;
  const size_t total = rem + 9 <= 64 ? 64 : 128;
  const uint64_t bits = (uint64_t)n * 8;
  std::memcpy(tail + total - 8, &bits, 8);
  h.block(tail);
  if (total == 128) h.block(tail + 64);
  const uint32_t v[4] = {h.a, h.b, h.c, h.d};
  char out[33];
  for (int i = 0; i < 16; i++) std::snprintf(out + 2 * i, 3, "%02x", (v[i >> 2] >> (8 * (i & 3))) & 0xffu);
  return std::string(out, 32);
}

Status header_md5(const std::string &path, std::string *out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return Status::Make(NeedleError_IOError, "IO error: cannot open " + path);
  uint8_t buf[8 * 1024];
  f.read(reinterpret_cast<char *>(buf), sizeof(buf));
  if ((size_t)f.gcount() != sizeof(buf))  // read_exact -> UnexpectedEof
    return Status::Make(NeedleError_IOError, "IO error: failed to fill whole buffer: " + path);
  *out = md5_hex(buf, sizeof(buf));
  return Status::Ok();
}

// ---- paths / text ---------------------------------------------------------------------------------------------
std::string with_extension(const std::string &path, const std::string &ext) {
  // std::path::Path::with_extension: replace what follows the last '.' of the file name (a leading
  // dot does not start an extension), or append ".ext" if there is none.
  const size_t slash = path.find_last_of('/');
  const size_t name = slash == std::string::npos ? 0 : slash + 1;
  const size_t dot = path.find_last_of('.');
  std::string stem = path;
  if (dot != std::string::npos && dot > name) stem = path.substr(0, dot);
  return stem + "." + ext;
}

std::string format_time(ns_t t) {
  const uint64_t secs = t / kNanosPerSec;
  char buf[48];
  std::snprintf(buf, sizeof(buf), "%02llu:%02llus", (unsigned long long)(secs / 60), (unsigned long long)(secs % 60));
  return buf;
}

std::string format_f32_json(float v) {
  // serde_json writes finite f32 with ryu: shortest digits that round-trip, printed plainly while
  // the decimal point falls within 13 digits / 5 leading zeros, else d.ddde<exp>.
  if (!std::isfinite(v)) return "null";
  if (v == 0.0f) return std::signbit(v) ? "-0.0" : "0.0";
  char sci[64];
  auto res = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
  std::string s(sci, res.ptr);
  std::string out;
  size_t pos = 0;
  if (s[0] == '-') {
    out = "-";
    pos = 1;
  }
  const size_t epos = s.find('e');
  std::string digits;
  for (size_t i = pos; i < epos; i++)
    if (s[i] != '.') digits.push_back(s[i]);
  const int exp10 = std::atoi(s.c_str() + epos + 1);
  const int nd = (int)digits.size();
  const int kk = exp10 + 1;   // position of the decimal point relative to the first digit
  const int k = kk - nd;      // value = digits * 10^k
  if (k >= 0 && kk <= 13) {
    out += digits + std::string((size_t)k, '0') + ".0";
  } else if (kk > 0 && kk <= 13) {
    out += digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
  } else if (kk > -6 && kk <= 0) {
    out += "0." + std::string((size_t)(-kk), '0') + digits;
  } else if (nd == 1) {
    out += digits + "e" + std::to_string(kk - 1);
  } else {
    out += digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
  }
  return out;
}

// ---- bincode image of FrameHashes -------------------------------------------------------
namespace {
struct Writer {
  std::string buf;
  template <typename T>
  void put(T v) {
    buf.append(reinterpret_cast<const char *>(&v), sizeof(T));
  }
  void duration(ns_t d) {
    put<uint64_t>(d / kNanosPerSec);
    put<uint32_t>((uint32_t)(d % kNanosPerSec));
  }
  void hashes(const std::vector<HashTs> &v) {
    put<uint64_t>(v.size());
    for (const HashTs &h : v) {
      put<uint32_t>(h.hash);
      duration(h.ts);
    }
  }
};

struct Reader {
  const std::string &buf;
  size_t off = 0;
  bool bad = false;
  explicit Reader(const std::string &b) : buf(b) {}
  template <typename T>
  T get() {
    T v{};
    if (bad || buf.size() - off < sizeof(T)) {
      bad = true;
      return v;
    }
    std::memcpy(&v, buf.data() + off, sizeof(T));
    off += sizeof(T);
    return v;
  }
  ns_t duration() {
    const uint64_t secs = get<uint64_t>();
    const uint32_t nanos = get<uint32_t>();
    // serde's Duration visitor checks secs + nanos/1e9 for overflow, then Duration::new carries
    if (secs > (0xFFFFFFFFFFFFFFFFull - nanos) / kNanosPerSec) {
      bad = true;
      return 0;
    }
    return secs * kNanosPerSec + nanos;
  }
  void hashes(std::vector<HashTs> *v) {
    const uint64_t len = get<uint64_t>();
    if (bad || len > (buf.size() - off) / 16) {
      bad = true;
      return;
    }
    // len entries of 16 bytes are there (checked above): one pass without a bounds check per field -- a library's search-only
    // call parses hundreds of these files, and field by field this loop was half of what a file cost
    v->resize((size_t)len);
    const char *p = buf.data() + off;
    HashTs *o = v->data();
    for (uint64_t i = 0; i < len; i++, p += 16) {
      uint64_t secs;
      uint32_t hash, nanos;
      std::memcpy(&hash, p, 4);
      std::memcpy(&secs, p + 4, 8);
      std::memcpy(&nanos, p + 12, 4);
      if (secs > (0xFFFFFFFFFFFFFFFFull - nanos) / kNanosPerSec) {  // (duration(): serde's overflow check)
        bad = true;
        return;
      }
      o[i].hash = hash;
      o[i].ts = secs * kNanosPerSec + nanos;
    }
    off += (size_t)len * 16;
  }
};
}  // namespace

Status frame_hashes_write(const std::string &path, const FrameHashesData &fh) {
  Writer w;
  w.put<uint32_t>(0);  // FrameHashesVersion::V1 — bincode writes the variant INDEX, data.rs:16-18
  w.put<uint32_t>(0);  // FrameHashesData::V1, data.rs:61-66
  w.hashes(fh.opening);
  w.hashes(fh.ending);
  w.duration(fh.hash_duration);
  w.put<uint64_t>(fh.md5.size());
  w.buf.append(fh.md5);
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) return Status::Make(NeedleError_IOError, "IO error: cannot create " + path);
  f.write(w.buf.data(), (std::streamsize)w.buf.size());
  if (!f) return Status::Make(NeedleError_IOError, "IO error: short write to " + path);
  return Status::Ok();
}

Status frame_hashes_read(const std::string &path, FrameHashesData *out) {
  // the whole file with one read (a library has one of these per video: an iostream iterator costs more than the parse)
  const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0)  // data.rs:106-108
    return Status::Make(NeedleError_FrameHashDataNotFound, "frame hash data not found at: \"" + path + "\"");
  std::string buf;
  struct stat st;
  // (one byte more than the file holds: the read that finds the end of the file then needs no larger buffer)
  if (::fstat(fd, &st) == 0 && st.st_size > 0) buf.resize((size_t)st.st_size + 1);
  size_t got = 0;
  for (;;) {
    if (got == buf.size()) buf.resize(buf.size() + 65536);  // not a regular file, or it grew: keep reading
    const ssize_t k = ::read(fd, &buf[got], buf.size() - got);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) break;
    got += (size_t)k;
  }
  ::close(fd);
  buf.resize(got);
  Reader r(buf);
  const uint32_t version = r.get<uint32_t>();
  if (!r.bad && version != 0) r.bad = true;  // unknown variant index is a bincode error
  const uint32_t tag = r.get<uint32_t>();
  if (!r.bad && tag != 0) r.bad = true;
  // parsed INTO *out (whose vectors keep what they had allocated: the comparator hands the same objects in call after call);
  // on failure *out holds rubbish and the callers drop it
  FrameHashesData &fh = *out;
  fh.opening.clear();
  fh.ending.clear();
  if (!r.bad) r.hashes(&fh.opening);
  if (!r.bad) r.hashes(&fh.ending);
  fh.hash_duration = r.duration();
  const uint64_t len = r.get<uint64_t>();
  if (!r.bad && len > buf.size() - r.off) r.bad = true;
  if (r.bad)  // needle::Error::BincodeError -> InvalidFrameHashData, needle-capi/src/lib.rs:128
    return Status::Make(NeedleError_InvalidFrameHashData, "bincode error: malformed frame hash data in " + path);
  fh.md5.assign(buf.data() + r.off, (size_t)len);
  // is_version_valid (data.rs:96-101): with one variant on each side, index 0/0 always agrees;
  // kept as an explicit check for future versions.
  if (version != tag) return Status::Make(NeedleError_FrameHashDataInvalidVersion, "invalid frame hash data version");
  return Status::Ok();
}

// ---- RIFF/WAVE -----------------------------------------------------------------------------------------------------
// The header walk reads chunk headers only, and samples are read by byte range, so a library of long files costs
// the bytes of its search windows and nothing else (the reference likewise stops decoding at the end of the opening
// window and seeks to the ending window, analyzer.rs:231-282,384-402).
namespace {

struct Fd {
  int fd = -1;
  explicit Fd(const std::string &path) : fd(::open(path.c_str(), O_RDONLY | O_CLOEXEC)) {}
  ~Fd() { if (fd >= 0) ::close(fd); }
  Fd(const Fd &) = delete;
  Fd &operator=(const Fd &) = delete;
  // all of [off, off+len) or as much as the file holds
  size_t read_at(void *dst, size_t len, uint64_t off) const {
    size_t done = 0;
    while (done < len) {
      const ssize_t r = ::pread(fd, (char *)dst + done, len - done, (off_t)(off + done));
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) break;
      done += (size_t)r;
    }
    return done;
  }
};

// to s16 the way the reference's resampler does when it is asked for AV_SAMPLE_FMT_S16 (analyzer.rs:180-187;
// swresample's sample-format conversions): integers keep their top 16 bits, u8 is re-centred, floats are
// scaled by 2^15, rounded to nearest and clipped
void wav_convert(const unsigned char *p, int format, int bits, size_t values, int16_t *dst) {
  const size_t width = (size_t)bits / 8;
  if (format == 1 && bits == 16) {
    std::memcpy(dst, p, values * 2);
  } else if (format == 1 && bits == 8) {
    for (size_t i = 0; i < values; i++) dst[i] = (int16_t)(((int)p[i] - 0x80) << 8);
  } else if (format == 1) {  // 24 / 32 bit: the two most significant bytes
    p += width - 2;
    for (size_t i = 0; i < values; i++, p += width) dst[i] = (int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8));
  } else {
    for (size_t i = 0; i < values; i++, p += width) {
      double x;
      if (bits == 32) {
        float f32;
        std::memcpy(&f32, p, 4);
        x = (double)std::nearbyintf(f32 * 32768.0f);
      } else {
        double f64;
        std::memcpy(&f64, p, 8);
        x = std::nearbyint(f64 * 32768.0);
      }
      dst[i] = (int16_t)(x != x ? 0.0 : std::min(32767.0, std::max(-32768.0, x)));
    }
  }
}

}  // namespace

Status wav_probe(const std::string &path, WavInfo *out) {
  const Fd f(path);
  if (f.fd < 0) return Status::Make(NeedleError_IOError, "IO error: cannot open " + path);
  struct stat st;
  if (::fstat(f.fd, &st) != 0) return Status::Make(NeedleError_IOError, "IO error: cannot stat " + path);
  const uint64_t size = (uint64_t)st.st_size;
  unsigned char head[12];
  if (f.read_at(head, 12, 0) != 12 || std::memcmp(head, "RIFF", 4) != 0 || std::memcmp(head + 8, "WAVE", 4) != 0)
    return Status::Make(NeedleError_Unknown, "unsupported media (only RIFF/WAVE PCM is read here; FFmpeg decode is "
                                             "outside this build): " + path);
  auto u32 = [](const unsigned char *p) { uint32_t v; std::memcpy(&v, p, 4); return v; };
  auto u16 = [](const unsigned char *p) { uint16_t v; std::memcpy(&v, p, 2); return v; };
  uint64_t off = 12;
  bool have_fmt = false;
  *out = WavInfo{};
  while (off + 8 <= size) {
    unsigned char ch[8];
    if (f.read_at(ch, 8, off) != 8) break;
    const uint64_t len = u32(ch + 4), body = off + 8;
    if (std::memcmp(ch, "fmt ", 4) == 0 && body + 16 <= size) {
      unsigned char fmt[40] = {0};
      f.read_at(fmt, (size_t)std::min<uint64_t>({len, sizeof(fmt), size - body}), body);
      out->format = u16(fmt);
      out->channels = u16(fmt + 2);
      out->sample_rate = (int)u32(fmt + 4);
      out->bits = u16(fmt + 14);
      if (out->format == 0xFFFE && len >= 26) out->format = u16(fmt + 24);  // WAVE_FORMAT_EXTENSIBLE sub-format
      have_fmt = true;
    } else if (std::memcmp(ch, "data", 4) == 0) {
      if (!have_fmt) break;
      const bool integer = out->format == 1 && (out->bits == 8 || out->bits == 16 || out->bits == 24 || out->bits == 32);
      const bool floating = out->format == 3 && (out->bits == 32 || out->bits == 64);
      if ((!integer && !floating) || out->channels < 1 || out->channels > 2)
        return Status::Make(NeedleError_Unknown,
                            "unsupported WAV encoding (need PCM 8/16/24/32-bit or IEEE float, 1-2 channels): " + path);
      const uint64_t avail = std::min(len, size - body);  // a truncated file ends the data chunk early
      out->data_offset = body;
      out->frames = avail / ((uint64_t)out->bits / 8) / (uint64_t)out->channels;
      return Status::Ok();
    }
    off = body + len + (len & 1);
  }
  return Status::Make(NeedleError_Unknown, "malformed WAV (no fmt/data chunk): " + path);
}

Status wav_read_frames(const std::string &path, const WavInfo &info, uint64_t first, uint64_t count, int16_t *dst) {
  if (first > info.frames || count > info.frames - first)
    return Status::Make(NeedleError_InvalidArgument, "WAV frame range outside the data chunk: " + path);
  if (count == 0) return Status::Ok();
  const Fd f(path);
  if (f.fd < 0) return Status::Make(NeedleError_IOError, "IO error: cannot open " + path);
  const size_t width = (size_t)info.bits / 8, frame_bytes = width * (size_t)info.channels;
  const uint64_t at = info.data_offset + first * frame_bytes;
  if (info.format == 1 && info.bits == 16) {  // already s16: straight into the destination
    if (f.read_at(dst, count * frame_bytes, at) != count * frame_bytes)
      return Status::Make(NeedleError_IOError, "IO error: short read from " + path);
    return Status::Ok();
  }
  const uint64_t block = (1u << 20) / frame_bytes * frame_bytes;  // ~1 MiB of source per conversion pass
  std::vector<unsigned char> buf((size_t)std::min<uint64_t>(block, count * frame_bytes));
  for (uint64_t done = 0; done < count * frame_bytes;) {
    const size_t n = (size_t)std::min<uint64_t>(block, count * frame_bytes - done);
    if (f.read_at(buf.data(), n, at + done) != n)
      return Status::Make(NeedleError_IOError, "IO error: short read from " + path);
    wav_convert(buf.data(), info.format, info.bits, n / width, dst + done / width);
    done += n;
  }
  return Status::Ok();
}

Status wav_read(const std::string &path, WavData *out) {
  WavInfo info;
  Status s = wav_probe(path, &info);
  if (!s.ok()) return s;
  out->channels = info.channels;
  out->sample_rate = info.sample_rate;
  out->pcm.resize((size_t)info.frames * (size_t)info.channels);
  return wav_read_frames(path, info, 0, info.frames, out->pcm.data());
}

unsigned usable_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, (unsigned)std::max(1, CPU_COUNT(&set)));
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char quota[32] = {0};
    long period = 0;
    if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
      n = std::min(n, (unsigned)std::max(1L, std::atol(quota) / period));
    std::fclose(f);
  } else if (FILE *q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    long quota = -1, period = 0;
    if (std::fscanf(q, "%ld", &quota) != 1) quota = -1;
    std::fclose(q);
    if (FILE *p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(p, "%ld", &period) != 1) period = 0;
      std::fclose(p);
    }
    if (quota > 0 && period > 0) n = std::min(n, (unsigned)std::max(1L, quota / period));
  }
  return n;
}

namespace {
std::atomic<unsigned> g_node_ranks{1};
}
void set_node_ranks(unsigned ranks) { g_node_ranks.store(std::max(1u, ranks)); }
unsigned host_threads() {
  if (const char *e = std::getenv("NEEDLE_HOST_THREADS")) return (unsigned)std::max(1, std::atoi(e));
  static const unsigned usable = usable_cpus();  // the cgroup files do not change under a running process
  return std::max(1u, usable / g_node_ranks.load());
}

}  // namespace needle
