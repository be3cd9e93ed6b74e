// amg_tile.h — token tiles shared by the build kernels: one block stages TILE consecutive
// tokens (plus a k-token halo) and the read-end flags that fall in the tile in LDS, so that
// every sliding window of the tile (construct_read.py get_geneMers) is cut from LDS.
#pragma once
#include "amg_device.h"

#define TILE_ITEMS 4
#define TILE_THREADS 256
#define TILE (TILE_THREADS * TILE_ITEMS)

// per-read window / short-read counts (construct_graph.py:53-55) and the read-end bitmap:
// bit t of bnd_bits is set when a read ends (exclusively) at token t
static __global__ void k_read_stats(const long long* __restrict__ read_off, long long n_reads, long long n_tokens,
                                    int k, unsigned long long* status, unsigned int* __restrict__ bnd_bits) {
  __shared__ unsigned long long s_w[4], s_s[4];
  // grid-stride: a few hundred workgroups whatever the number of reads — each ends with one atomicAdd per counter,
  // and a single counter word takes only ~90 atomics per microsecond (one workgroup per 256 reads spent 40 of the
  // kernel's 50 us queueing there)
  unsigned long long w = 0, sh = 0;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (long long)gridDim.x * blockDim.x) {
    const long long beg = read_off[r], end = read_off[r + 1];
    // the CSR may be caller-owned device memory nobody has looked at yet: offsets must start at 0,
    // never decrease and end at the token count before anything is indexed with them
    const bool bad = beg < 0 || end < beg || end > n_tokens || (r == 0 && beg != 0) ||
                     (r == n_reads - 1 && end != n_tokens);
    if (bad) {
      status[ST_BADINPUT] = 1;  // benign race: every writer stores 1
    } else {
      const long long len = end - beg;
      if (len >= k)
        w += (unsigned long long)(len - k + 1);
      else
        sh += 1;
      atomicOr(&bnd_bits[end >> 5], 1u << (end & 31));
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    w += __shfl_down(w, d, 64);
    sh += __shfl_down(sh, d, 64);
  }
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_w[wave] = w;
    s_s[wave] = sh;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tw = 0, ts = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
      tw += s_w[i];
      ts += s_s[i];
    }
    if (tw) atomicAdd(&status[ST_N_WINDOWS], tw);
    if (ts) atomicAdd(&status[ST_N_SHORT], ts);
  }
}

struct LdsView {
  const int* p;
  __device__ __forceinline__ int operator[](int j) const { return p[j]; }
};

// words of the read-end bitmap one tile needs: bits t0 .. t0 + TILE + 63 (t0 is a multiple of 32)
#define TILE_BIT_WORDS (TILE / 32 + 2)
#define BND_PAD_WORDS (TILE_BIT_WORDS + 8)  // words allocated beyond (n_tokens >> 5)

// stage the tile's tokens and its slice of the read-end bitmap in LDS: two independent
// coalesced loads and one barrier (no per-tile search, no dependent loads)
__device__ __forceinline__ void stage_tile(const int* __restrict__ tokens,
                                           const unsigned int* __restrict__ bnd_bits,
                                           long long n_tokens, int k, long long t0, int* s_tok,
                                           unsigned int* s_bits, int two_v, unsigned long long* status) {
  const int tid = threadIdx.x;
  const int span = TILE + k;  // tokens t0 .. t0 + TILE + k - 1
  bool bad = false;
  for (int i = tid; i < span; i += TILE_THREADS) {
    long long t = t0 + i;
    // streamed once: a non-temporal load leaves the L2 to the table's hot lines (measured: tools/ubench/pass_bench)
    const int v = t < n_tokens ? __builtin_nontemporal_load(tokens + t) : 0;
    // a token outside [0, two_v) would alias another tuple in the packed key (and index the
    // vocabulary out of range on the way back): refuse the build
    bad = bad || (unsigned int)v >= (unsigned int)two_v;
    s_tok[i] = v;
  }
  if (bad) status[ST_BADINPUT] = 2;
  if (tid < TILE_BIT_WORDS) s_bits[tid] = bnd_bits[(t0 >> 5) + tid];
  __syncthreads();
}

// bits [pos, pos + n) of the tile's bitmap slice, n <= 32
__device__ __forceinline__ unsigned int tile_bits(const unsigned int* s_bits, int pos, int n) {
  const unsigned long long w =
      (unsigned long long)s_bits[pos >> 5] | ((unsigned long long)s_bits[(pos >> 5) + 1] << 32);
  return (unsigned int)(w >> (pos & 31)) & (unsigned int)((1ull << n) - 1ull);
}
// window starting at tile position i: valid when no read ends inside it (bits i+1 .. i+k-1),
// last of its read when one ends right after it (bit i+k)
__device__ __forceinline__ void tile_window(const unsigned int* s_bits, int i, int k, bool& inside,
                                            bool& last) {
  const unsigned int b = tile_bits(s_bits, i + 1, k);  // bits i+1 .. i+k
  inside = (b & ((1u << (k - 1)) - 1u)) == 0u;
  last = (b >> (k - 1)) & 1u;
}
