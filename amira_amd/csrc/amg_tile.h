// amg_tile.h — token tiles shared by the build kernels: one block stages TILE consecutive
// tokens (plus a k-token halo) and the read-end flags that fall in the tile in LDS, so that
// every sliding window of the tile (construct_read.py get_geneMers) is cut from LDS.
#pragma once
#include "amg_device.h"

#define TILE_ITEMS 4
#define TILE_THREADS 256
#define TILE (TILE_THREADS * TILE_ITEMS)

static __global__ void k_read_stats(const long long* __restrict__ read_off, long long n_reads, int k,
                             unsigned long long* status) {
  __shared__ unsigned long long s_w[4], s_s[4];
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long w = 0, sh = 0;
  if (r < n_reads) {
    long long len = read_off[r + 1] - read_off[r];
    if (len >= k)
      w = (unsigned long long)(len - k + 1);
    else
      sh = 1;
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

// tile_lo[b] = first j in [1, n_reads] with read_off[j] > b * TILE  (n_reads + 1 if none)
static __global__ void k_tile_reads(const long long* __restrict__ read_off, long long n_reads,
                             long long n_tiles_plus2, long long* __restrict__ tile_lo) {
  long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_tiles_plus2) return;
  long long target = b * (long long)TILE;
  long long lo = 1, hi = n_reads + 1;  // search in [1, n_reads + 1)
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    if (read_off[mid] > target)
      hi = mid;
    else
      lo = mid + 1;
  }
  tile_lo[b] = lo;
}

struct LdsView {
  const int* p;
  __device__ __forceinline__ int operator[](int j) const { return p[j]; }
};

// shared by k_node_upsert and k_edges: stage the tile's tokens and read-end flags in LDS
__device__ __forceinline__ void stage_tile(const int* __restrict__ tokens,
                                           const long long* __restrict__ read_off,
                                           const long long* __restrict__ tile_lo,
                                           long long n_reads, long long n_tokens, int k,
                                           long long t0, int* s_tok, unsigned char* s_bnd) {
  const int tid = threadIdx.x;
  const int span = TILE + k;  // tokens t0 .. t0 + TILE + k - 1, flags 0 .. TILE + k
  for (int i = tid; i < span; i += TILE_THREADS) {
    long long t = t0 + i;
    s_tok[i] = t < n_tokens ? tokens[t] : 0;
    s_bnd[i] = 0;
  }
  if (tid == 0) s_bnd[span] = 0;
  __syncthreads();
  // read ends (exclusive) that fall in (t0, t0 + TILE + k]
  long long b = blockIdx.x;
  long long lo = tile_lo[b], hi = tile_lo[b + 2];
  for (long long j = lo + tid; j < hi; j += TILE_THREADS) {
    long long off = read_off[j] - t0;
    if (off <= span) s_bnd[off] = 1;
  }
  __syncthreads();
}

