// amg_device.h — device-side helpers shared by the build and pass kernels (gfx950).
#pragma once
#include "amg_internal.h"

#define AMG_WAVE 64
#define AMG_LAST_FLAG 0x80000000u
// per-window claims of the exact-key table passes: bit 30 marks the occurrence that CREATED the key.  Every key
// has exactly one, so occurrence counters start at 1 and the counting sweeps skip the marked occurrences: the
// millions of keys seen once (the error gene-mers of an uncorrected read set) then cost no atomic at all
#define AMG_MADE_FLAG 0x40000000u
#define AMG_FLAG_MASK (AMG_LAST_FLAG | AMG_MADE_FLAG)
#define AMG_SINGLE_BIT 0x20000000u  // on a node id handed to the edge pass: the node has coverage 1 (node ids stay below 2^29)

// relaxed, agent-scope accessors: L1-bypassing loads, coherent with the device-scope
// atomics that mutate the tables (MI355X_MICROARCH.md, "Inter-workgroup visibility").
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_i32(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return x;
}

// Canonical orientation of the window whose tokens are w[0..k-1] (any indexable view):
// rc[j] = two_v - 1 - w[k-1-j]; canonical = lexicographic min(w, rc); returns +1 when the
// window is the canonical one, -1 when its reverse complement is, 0 when they are equal
// (palindrome: the reference asserts, construct_gene_mer.py:23-25).
template <class View>
__device__ __forceinline__ int canon_dir(const View& w, int k, int flip) {
  for (int j = 0; j < k; ++j) {
    int a = w[j];
    int b = flip - w[k - 1 - j];
    if (a != b) return a < b ? 1 : -1;
  }
  return 0;
}

// j-th token of the canonical tuple given the window view and its direction
template <class View>
__device__ __forceinline__ int canon_tok(const View& w, int k, int flip, int dir, int j) {
  return dir > 0 ? w[j] : flip - w[k - 1 - j];
}

template <class View>
__device__ __forceinline__ unsigned long long canon_fingerprint(const View& w, int k, int flip,
                                                                int dir, unsigned long long seed) {
  unsigned long long h = seed;
  for (int j = 0; j < k; ++j) {
    unsigned long long c = (unsigned long long)(unsigned int)canon_tok(w, k, flip, dir, j);
    h = (h ^ c) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  h = mix64(h);
  return h ? h : 1ull;
}

// Insert-or-find `key` (non-zero) in an open-addressing table of 32-byte slots with
// linear probing; bump the slot's count and keep the minimum `first` value.
// Entirely lock-free: the key is claimed by one 64-bit CAS, the counter by atomicAdd,
// first-seen by atomicMax on the complement.  A stale (cached) read can only make us
// take the slow path (CAS / atomicMax), never a wrong decision, because a slot's key
// never changes once set and first_inv only grows.
// Returns the slot index, or -1 if `limit` probes did not find a home.
__device__ __forceinline__ long long table_upsert(Slot* tab, unsigned long long mask,
                                                  unsigned long long key,
                                                  unsigned long long start,
                                                  unsigned long long first, unsigned int limit,
                                                  bool count_inline = true,
                                                  const unsigned long long* abort_flag = nullptr) {
  unsigned long long idx = start & mask;
  for (unsigned int probes = 0;; ++probes) {
    // a table that turned out too small is abandoned quickly: once any thread has run out
    // of probes the others stop walking long chains (the host rebuilds with a larger table)
    if (abort_flag && (probes & 63u) == 63u && *reinterpret_cast<const volatile unsigned long long*>(abort_flag))
      return -1;
    Slot* s = tab + idx;
    unsigned long long cur = ld_u64(&s->key);
    if (cur == 0ull) {
      cur = atomicCAS(&s->key, 0ull, key);
      if (cur == 0ull) cur = key;
    }
    if (cur == key) {
      // scattered atomics run at ~27 G/s on MI355X against ~255 G/s for cached loads
      // (tools/ubench/atomic_bench.hip), so the single-GPU build counts occurrences in a
      // separate LDS-privatised pass (k_count_ids) and only the merge path counts here
      if (count_inline) atomicAdd(&s->count, 1u);
      unsigned long long fi = ~first;
      if (ld_u64(&s->first_inv) < fi) atomicMax(&s->first_inv, fi);
      return (long long)idx;
    }
    if (probes >= limit) return -1;
    idx = (idx + 1) & mask;
  }
}

// After node ranking a node slot is rewritten as a PACKED record so that the edge pass needs
// ONE 32-byte gather per window for both the node id and the exact tuple check:
//   bytes  0..19  tokens 0..9 as uint16      bytes 20..23  node id
//   bytes 24..31  tokens 10..13 as uint16
// (possible when every token fits 16 bits and k <= 14; otherwise the id stays in Slot::id and
// the tuple is gathered from node_tokens).
#define AMG_PACK_MAX_K 14
__device__ __forceinline__ void slot_pack(Slot* s, int id, const int* tok, int k) {
  unsigned int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < k; ++j) {
    const int word = j < 10 ? (j >> 1) : 6 + ((j - 10) >> 1);
    w[word] |= ((unsigned int)tok[j] & 0xffffu) << ((j & 1) * 16);
  }
  w[5] = (unsigned int)id;
  uint4* p = reinterpret_cast<uint4*>(s);
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ unsigned int packed_tok(const uint4& lo, const uint4& hi, int j) {
  const int word = j < 10 ? (j >> 1) : 6 + ((j - 10) >> 1);
  unsigned int w = word == 0 ? lo.x : word == 1 ? lo.y : word == 2 ? lo.z : word == 3 ? lo.w
                 : word == 4 ? hi.x : word == 6 ? hi.z : hi.w;
  return (w >> ((j & 1) * 16)) & 0xffffu;
}

// block-wide exclusive prefix of a per-thread count (WAVES waves of 64); returns the thread's
// offset inside the block and the block total in *total.
template <int WAVES>
__device__ __forceinline__ unsigned int block_exscan(unsigned int v, unsigned int* total,
                                                     unsigned int* s_wave /*[WAVES]*/) {
  unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned int y = __shfl_up(x, d, 64);
    if (lane >= (unsigned)d) x += y;
  }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  unsigned int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    unsigned int c = s_wave[w];
    if ((unsigned)w < wave) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}
__device__ __forceinline__ unsigned int block_exscan_256(unsigned int v, unsigned int* total,
                                                         unsigned int* s_wave /*[4]*/) {
  return block_exscan<4>(v, total, s_wave);
}

// ------------------------------------------------------------------ a hub row's ids in ascending order
// Rows of an adjacency list are put in order by rank (every id counts the smaller ids of its row): fine for the rows of
// a gene-mer graph, quadratic in the row.  A row longer than HUGE_ROW goes through a bitmap over the id space instead —
// clear, set one bit per id, prefix-count the words, read the ids off in order: O(ids / 32 + row) for a workgroup,
// whatever the row's length (a hub with a million neighbours would otherwise hold one workgroup for an hour).
// bits: this workgroup's scratch of `words` words; emit(rank, id).  All 256 threads of the workgroup call it.
// Rows of 3 .. 64 edge ids in ascending order by the WAVE, one row at a time: every lane that holds such a row (`mine`)
// is served in turn — lane j takes the row's j-th id, its rank is the number of smaller ids (one shuffle per id of the
// row), emit(row's tag, row's offset, row's length, rank, id) places it.  A thread per row with an insertion sort in a private array was the tail
// of the list kernels: a row of thirty ids is ~500 dependent scratch accesses of ONE lane (0.9 ms for the genome rows
// of a graph that carries the residual error nodes of eight read sets), here ~30 shuffles of a wave.
// Every lane of the wave must call (no early returns before).
#define WAVE_ROW_MAX 64
template <class Emit>
__device__ __forceinline__ void wave_rows_in_order(bool mine, unsigned int tag, long long off, int cnt,
                                                   const unsigned int* __restrict__ ids, Emit emit) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(mine);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const long long o = __shfl(off, leader, 64);
    const int n = __shfl(cnt, leader, 64);
    const unsigned int tg = __shfl(tag, leader, 64);
    const unsigned int x = lane < n ? ids[o + lane] : 0xffffffffu;
    int rank = 0;
    for (int q = 0; q < n; ++q) rank += __shfl(x, q, 64) < x ? 1 : 0;
    if (lane < n) emit(tg, o, n, rank, x);
  }
}

#define HUGE_ROW 1024
template <class Emit>
__device__ __forceinline__ void huge_row_in_order(const unsigned int* __restrict__ ids, long long cnt, unsigned int* bits,
                                                  long long words, unsigned int* s_wave /*[4]*/, Emit emit) {
  for (long long w = threadIdx.x; w < words; w += 256) bits[w] = 0u;
  __threadfence();
  __syncthreads();
  for (long long j = threadIdx.x; j < cnt; j += 256) atomicOr(&bits[ids[j] >> 5], 1u << (ids[j] & 31u));
  __threadfence();
  __syncthreads();
  const long long per = (words + 255) / 256;
  const long long w0 = (long long)threadIdx.x * per, w1 = w0 + per < words ? w0 + per : words;
  unsigned int mine = 0;
  for (long long w = w0; w < w1; ++w)
    mine += (unsigned int)__popc(__hip_atomic_load(bits + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  unsigned int total;
  unsigned int rank = block_exscan_256(mine, &total, s_wave);
  for (long long w = w0; w < w1; ++w) {
    unsigned int v = __hip_atomic_load(bits + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v) {
      const int b = __ffs((int)v) - 1;
      v &= v - 1u;
      emit((long long)rank++, (unsigned int)(w * 32 + b));
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ union-find helpers (components)
__device__ __forceinline__ int uf_ld(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // plain load, never hoisted
}
__device__ __forceinline__ int uf_find(int* parent, int x) {
  int p = uf_ld(parent + x);
  while (p != x) {
    const int g = uf_ld(parent + p);
    if (g != p) __hip_atomic_store(parent + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // halving
    x = p;
    p = g;
  }
  return x;
}


