// The closed-form pair index -> (i, j) map of comparator.cpp (comparator.rs:534-545 enumerates pairs i-major)
// against the enumeration itself, exhaustively for small libraries and at the row boundaries of a huge one.
// Built and run by tests/test_capi_cpu.py against libneedle_capi.so (host code only, no GPU).
#include <cstddef>
#include <cstdio>
#include <initializer_list>
namespace needle { void pair_at(size_t, size_t, size_t *, size_t *); size_t pair_count(size_t); }
int main() {
  for (size_t n : {2, 3, 4, 5, 17, 100, 1000, 2001, 4097}) {
    size_t idx = 0;
    for (size_t i = 0; i + 1 < n; i++)
      for (size_t j = i + 1; j < n; j++, idx++) {
        size_t a, b;
        needle::pair_at(n, idx, &a, &b);
        if (a != i || b != j) { std::printf("MISMATCH n=%zu idx=%zu got (%zu,%zu) want (%zu,%zu)\n", n, idx, a, b, i, j); return 1; }
      }
    if (idx != needle::pair_count(n)) { std::printf("count mismatch\n"); return 1; }
  }
  // very large n: spot checks at row boundaries
  const size_t n = 3000000;
  for (size_t i : {(size_t)0, (size_t)1, (size_t)12345, n / 2, n - 3, n - 2}) {
    const size_t start = i * (2 * n - i - 1) / 2;
    size_t a, b;
    needle::pair_at(n, start, &a, &b);
    if (a != i || b != i + 1) { std::printf("big mismatch at row %zu\n", i); return 1; }
    if (start) { needle::pair_at(n, start - 1, &a, &b); if (a != i - 1 || b != n - 1) { std::printf("big mismatch before row %zu\n", i); return 1; } }
  }
  std::printf("pair_at ok\n");
  return 0;
}
