// kvz_arena.hpp -- per-thread staging arena shared by the backends of kvz_api_impl.hpp.
//
// One call lays its buffers out in a single region:
//     [ inputs ............ | zeroed accumulators | outputs ........ | scratch ... ]
//       ^ uploaded (H2D) .................^          ^ downloaded (D2H, from the first zeroed/out) ^
// Host addresses (h + off) and device addresses (d + off) share offsets, so a device pointer handed to an op
// maps back to its staging bytes with host().
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace kvz {

struct ArenaBase {
  uint8_t *h = nullptr, *d = nullptr;
  size_t cap = 0, cur = 0, up_end = 0, dl_begin = 0, dl_end = 0;
  int phase = 0;  // 0 inputs, 1 zeroed, 2 out, 3 scratch

  size_t take(size_t bytes)
  {
    const size_t off = (cur + 63) & ~(size_t)63;
    if (off + bytes > cap) {
      fprintf(stderr, "kvz_hip: staging arena overflow (%zu + %zu > %zu)\n", off, bytes, cap);
      abort();
    }
    cur = off + bytes;
    return off;
  }
  void reset() { cur = up_end = dl_begin = dl_end = 0; phase = 0; }

  template <class T> T *in_raw(size_t n)
  {
    const size_t off = take(n * sizeof(T));
    up_end = cur;
    return (T *)(d + off);
  }
  template <class T> T *in(const T *host_src, size_t n)
  {
    T *p = in_raw<T>(n);
    memcpy(h + ((uint8_t *)p - d), host_src, n * sizeof(T));
    return p;
  }
  template <class T> T *in_rows(const T *host_src, int w, int hgt, long stride)
  {
    T *p = in_raw<T>((size_t)w * hgt);
    T *dst = (T *)(h + ((uint8_t *)p - d));
    for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * w, host_src + (long)y * stride, w * sizeof(T));
    return p;
  }
  template <class T> T *host_rw(T *dev) { return (T *)(h + ((uint8_t *)dev - d)); }
  template <class T> const T *host(const T *dev) const { return (const T *)(h + ((const uint8_t *)dev - d)); }
  template <class T> void mark_download_from(T *dev)
  {
    if (phase < 1) { phase = 1; dl_begin = (size_t)((uint8_t *)dev - d); }
  }
  template <class T> T *zeroed(size_t n)
  {
    const size_t off = take(n * sizeof(T));
    if (phase < 1) { phase = 1; dl_begin = off; }
    memset(h + off, 0, n * sizeof(T));
    up_end = cur;
    dl_end = cur;
    return (T *)(d + off);
  }
  template <class T> T *out(size_t n)
  {
    const size_t off = take(n * sizeof(T));
    if (phase < 1) dl_begin = off;
    phase = 2;
    dl_end = cur;
    return (T *)(d + off);
  }
  template <class T> T *scratch(size_t n) { return (T *)(d + take(n * sizeof(T))); }
};

}  // namespace kvz
