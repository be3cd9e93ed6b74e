// amg_x.h — the 16-byte slot of the exact-key tables (amg_build_x.hip), shared with the
// multi-GPU merge (amg_dist.hip), which reads local keys back out of it.
#pragma once
#include "amg_device.h"

struct __attribute__((aligned(16))) Slot16 {
  unsigned long long w1;
  unsigned long long w2;
};
static_assert(sizeof(Slot16) == 16, "slot16");

// token j of a packed canonical tuple: w1 = (low 63 bits << 1) | 1, tag = (high 31 bits << 1) | 1
__device__ __forceinline__ int x_unpack(unsigned long long w1, unsigned int tag, int bits, int j) {
  const unsigned long long lo = w1 >> 1, hi = (unsigned long long)(tag >> 1);
  const int sh = j * bits;
  unsigned long long v = sh < 63 ? ((lo >> sh) | (hi << (63 - sh))) : (hi >> (sh - 63));
  return (int)(v & ((1ull << bits) - 1ull));
}


// First-seen of a claim lives in TWO adjacent words, both holding the complement (so that larger =
// earlier) and both zero-initialised: [2c] is raised with atomicMax by every window that is not
// the creator, [2c + 1] is the creator's own plain store; the larger one wins.  Adjacent, so
// that the per-window check is one 8-byte load.
__device__ __forceinline__ unsigned int x_first_inv(const unsigned int* first2, long long c) {
  const uint2 f = reinterpret_cast<const uint2*>(first2)[c];
  return f.x > f.y ? f.x : f.y;
}
