// amg_match.hip — K6: batched exact sub-list search of P patterns in every read
// (is_sublist / find_sublist_indices, construct_graph.py:1957-1966, 2117-2123; the
// Tree.find_all call sites of path_finding_utils.py:244, 290, 300-308).
//
// Patterns are bucketed by their first symbol (direct index over the symbol domain: tokens for
// which = 0, node ids for which = 1).  One wave per read: each lane takes a position, looks up
// the bucket of the symbol there and verifies the candidates against the read (the read slice
// is L1/L2 resident).  Hits are packed as (pattern, read, position) keys, radix-sorted, and
// split per pattern — i.e. ordered by pattern, then read, then position.
#include <algorithm>

#include "amg_device.h"

static inline unsigned int nblk(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

#define POS_BITS 20
#define READ_BITS 28
#define PAT_BITS 16

struct MatchArgs {
  const int* seq;             // tokens or tok_node
  const long long* read_off;
  long long n_reads;
  int tail;                   // positions at the end of a read that are not symbols (k-1 for nodes)
  const int* pat;             // pattern symbols (device)
  const long long* pat_off;
  const long long* bucket_off;  // [domain + 1] -> range in bucket_pat
  const int* bucket_pat;
  const unsigned int* has_bits;  // bit s: some pattern starts with symbol s
  long long domain;
  int lds_words;              // words of has_bits staged in LDS (0: tested in global memory)
  unsigned long long* hits;   // nullptr: count only
  unsigned int* wave_count;   // count launch: hits of every wave of the grid
  const long long* wave_base; // fill launch (same grid): where every wave's hits begin
  unsigned long long cap;
};

// Waves walk the reads (grid stride); each lane takes positions of the read.  Nearly every position holds a symbol
// no pattern starts with: that is decided by one bit of a bitmap over the symbol domain staged in LDS once per
// workgroup (5 KB for 20 000 genes) — not by two 8-byte gathers from the bucket table per gene.  Hits are placed WITHOUT
// atomics: the count launch leaves every wave's number of hits, a scan turns them into the waves' first places, the
// fill launch (same grid, same reads per wave) writes from there.  (A single counter word takes ~100 returning atomics
// per microsecond: one per hit made a million hits cost 10 ms, one per read with hits still 3.3 ms for 335 k hits.)
#define MATCH_LDS_WORDS 12288  // 48 KB: 393 216 symbols
__global__ __launch_bounds__(256) void k_match(MatchArgs A) {
  extern __shared__ unsigned int s_has[];
  for (int w = threadIdx.x; w < A.lds_words; w += blockDim.x) s_has[w] = A.has_bits[w];
  __syncthreads();
  const unsigned int* has = A.lds_words ? s_has : A.has_bits;
  const int lane = threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  unsigned long long counted = 0;  // hits of this wave so far (fill launch: its next place is wave_base + counted)
  const unsigned long long my_base = A.hits ? (unsigned long long)A.wave_base[wave0] : 0ull;
  // MATCH_BATCH reads per iteration: their offsets and first 64 symbols are loaded for all of them before any is
  // looked at (a wave that walks one read at a time is bound by the chain offsets -> symbols of that one read); a read
  // none of whose symbols starts a pattern — nearly all of them — is done after that look
  constexpr int MATCH_BATCH = 4;
  for (long long rb = wave0 * MATCH_BATCH; rb < A.n_reads; rb += n_waves * MATCH_BATCH) {
    long long bt0[MATCH_BATCH], blen[MATCH_BATCH];
    int bsym[MATCH_BATCH];
#pragma unroll
    for (int q = 0; q < MATCH_BATCH; ++q) {
      const long long r = rb + q;
      bt0[q] = r < A.n_reads ? A.read_off[r] : 0;
      blen[q] = r < A.n_reads ? A.read_off[r + 1] - bt0[q] - A.tail : 0;
    }
#pragma unroll
    for (int q = 0; q < MATCH_BATCH; ++q) bsym[q] = lane < blen[q] ? A.seq[bt0[q] + lane] : -1;
    unsigned int todo = 0;
#pragma unroll
    for (int q = 0; q < MATCH_BATCH; ++q) {
      const int sym = bsym[q];
      const bool cand = sym >= 0 && sym < A.domain && ((has[sym >> 5] >> (sym & 31)) & 1u);
      if (__any(cand) || blen[q] > 64) todo |= 1u << q;
    }
   for (int q = 0; q < MATCH_BATCH; ++q) {
    if (!((todo >> q) & 1u)) continue;
    const long long r = rb + q;
    const long long t0 = A.read_off[r];
    const long long len = A.read_off[r + 1] - t0 - A.tail;
    // hits of a position: patterns of the bucket of its symbol that match there; fn(pattern) per hit
    auto scan = [&](auto fn) {
      for (long long i = lane; i < len; i += 64) {
        const int sym = A.seq[t0 + i];
        if (sym < 0 || sym >= A.domain) continue;
        if (!((has[sym >> 5] >> (sym & 31)) & 1u)) continue;
        for (long long b = A.bucket_off[sym]; b < A.bucket_off[sym + 1]; ++b) {
          const int p = A.bucket_pat[b];
          const long long po = A.pat_off[p], m = A.pat_off[p + 1] - po;
          if (i + m > len) continue;
          bool same = true;
          for (long long j = 1; j < m && same; ++j) same = (A.seq[t0 + i + j] == A.pat[po + j]);
          if (same) fn(p, i);
        }
      }
    };
    unsigned int mine = 0;
    scan([&](int, long long) { ++mine; });
    if (!A.hits) {
      counted += mine;
      continue;
    }
    // places: exclusive prefix over the lanes, one atomicAdd of the wave
    unsigned int x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned int y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    const unsigned int total = __shfl(x, 63, 64);
    if (total == 0) continue;
    unsigned long long at = my_base + counted + (unsigned long long)(x - mine);
    counted += total;
    scan([&](int p, long long i) {
      if (at < A.cap)
        A.hits[at] = ((unsigned long long)p << (READ_BITS + POS_BITS)) | ((unsigned long long)r << POS_BITS) |
                     (unsigned long long)i;
      ++at;
    });
   }
  }
  if (!A.hits) {
    for (int d = 32; d > 0; d >>= 1) counted += __shfl_down(counted, d, 64);
    if (lane == 0) A.wave_count[wave0] = (unsigned int)counted;
  }
}

// bit s of has_bits: hist[s] != 0 (some pattern starts with symbol s)
__global__ void k_pat_bits(const unsigned int* __restrict__ hist, long long domain, unsigned int* __restrict__ bits,
                           long long words) {
  const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  unsigned int v = 0;
  for (int b = 0; b < 32; ++b) {
    const long long s = 32 * w + b;
    if (s < domain && hist[s]) v |= 1u << b;
  }
  bits[w] = v;
}

__global__ void k_pat_first(const int* __restrict__ pat, const long long* __restrict__ pat_off,
                            long long n_pat, long long domain, unsigned int* __restrict__ first,
                            unsigned int* __restrict__ idx, unsigned int* __restrict__ hist) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pat) return;
  long long m = pat_off[p + 1] - pat_off[p];
  long long f = m > 0 ? (long long)pat[pat_off[p]] : -1;
  bool ok = f >= 0 && f < domain;
  unsigned int key = ok ? (unsigned int)f : (unsigned int)domain;  // unusable patterns go last
  first[p] = key;
  idx[p] = (unsigned int)p;
  if (ok) atomicAdd(&hist[key], 1u);
}

// sorted keys -> (read, position) per hit and the first hit of every pattern: hit i opens every pattern in
// (pattern of hit i - 1, pattern of hit i], the place after the last hit opens the rest and n_pat (no atomics: a counter
// per pattern took one atomic per hit, and a batch of few patterns with many hits queued them on a handful of words)
__global__ void k_hit_split(const unsigned long long* __restrict__ keys, long long n, long long n_pat,
                            int* __restrict__ hit_read, int* __restrict__ hit_pos, long long* __restrict__ pat_first) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const long long prev = i > 0 ? (long long)(keys[i - 1] >> (READ_BITS + POS_BITS)) : -1;
  long long cur = n_pat;
  if (i < n) {
    const unsigned long long k = keys[i];
    hit_pos[i] = (int)(k & ((1ull << POS_BITS) - 1));
    hit_read[i] = (int)((k >> POS_BITS) & ((1ull << READ_BITS) - 1));
    cur = (long long)(k >> (READ_BITS + POS_BITS));
  }
  for (long long p = prev + 1; p <= cur; ++p) pat_first[p] = i;
}

extern "C" int amg_match_patterns(amg_ctx* c, int which, const int32_t* pat, const int64_t* pat_offsets,
                                  int64_t n_pat, int64_t* hit_offsets, int32_t* hit_read, int32_t* hit_pos) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  if (which == 1 && !c->built) return amg_fail(AMG_E_STATE, "amg_build first");
  if (n_pat < 0 || !pat_offsets || !hit_offsets) return amg_fail(AMG_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  // second call of the two-call protocol: hand out the cached result
  if (hit_read || hit_pos) {
    if (!c->match_valid || c->match_npat != n_pat)
      return amg_fail(AMG_E_STATE, "call amg_match_patterns with NULL hit arrays first");
    if (c->match_total > 0) {
      if (hit_read)
        HIPCHK(hipMemcpyAsync(hit_read, c->match_read.p, (size_t)c->match_total * sizeof(int),
                              hipMemcpyDeviceToHost, st));
      if (hit_pos)
        HIPCHK(hipMemcpyAsync(hit_pos, c->match_pos.p, (size_t)c->match_total * sizeof(int),
                              hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipMemcpyAsync(hit_offsets, c->match_off.p, (size_t)(n_pat + 1) * sizeof(long long),
                          hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return AMG_OK;
  }
  c->match_valid = false;
  stages_reset(c);
  if (n_pat >= (1ll << PAT_BITS)) return amg_fail(AMG_E_ARG, "at most %d patterns per call", (1 << PAT_BITS) - 1);
  if (c->n_reads >= (1ll << READ_BITS)) return amg_fail(AMG_E_ARG, "too many reads for amg_match_patterns");
  const long long n_sym = pat_offsets[n_pat];
  const long long domain = which == 1 ? c->n_nodes : c->two_v;
  AMGCHK(c->match_off.ensure((size_t)(n_pat + 2) * sizeof(long long)));
  if (n_pat == 0) {
    hit_offsets[0] = 0;
    c->match_total = 0; c->match_npat = 0; c->match_valid = true;
    return AMG_OK;
  }
  // ---- upload patterns, bucket them by first symbol
  AMGCHK(c->s0.ensure((size_t)(n_sym + 1) * sizeof(int)));
  AMGCHK(c->s1.ensure((size_t)(n_pat + 2) * sizeof(long long)));
  AMGCHK(c->s2.ensure((size_t)(n_pat + 2) * sizeof(unsigned int) * 4));
  AMGCHK(c->s3.ensure((size_t)(domain + 2) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(domain + 2) * sizeof(long long)));
  if (n_sym > 0)
    HIPCHK(hipMemcpyAsync(c->s0.p, pat, (size_t)n_sym * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->s1.p, pat_offsets, (size_t)(n_pat + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
  unsigned int* first = c->s2.as<unsigned int>();
  unsigned int* idx = first + (n_pat + 2);
  unsigned int* first_sorted = idx + (n_pat + 2);
  unsigned int* idx_sorted = first_sorted + (n_pat + 2);
  HIPCHK(hipMemsetAsync(c->s3.p, 0, (size_t)(domain + 2) * sizeof(unsigned int), st));
  hipLaunchKernelGGL(k_pat_first, dim3(nblk(n_pat, 256)), dim3(256), 0, st, c->s0.as<int>(),
                     c->s1.as<long long>(), (long long)n_pat, domain, first, idx, c->s3.as<unsigned int>());
  AMGCHK(prim_sort_u32_u32(c, first, first_sorted, idx, idx_sorted, (size_t)n_pat,
                           ilog2_ceil((uint64_t)domain + 2) + 1));
  // symbols that start a pattern, as a bitmap (behind the per-symbol counts in s3: they are scanned below)
  const long long bit_words = (domain + 31) / 32 + 1;
  AMGCHK(c->gm_mask.ensure((size_t)bit_words * sizeof(unsigned int)));
  hipLaunchKernelGGL(k_pat_bits, dim3(nblk(bit_words, 256)), dim3(256), 0, st, c->s3.as<unsigned int>(), domain,
                     c->gm_mask.as<unsigned int>(), bit_words);
  AMGCHK(prim_exscan_u32_to_i64(c, c->s3.as<unsigned int>(), c->s4.as<long long>(), (size_t)domain + 1));
  MatchArgs A;
  A.seq = which == 1 ? c->tok_node.as<int>() : c->tokens.as<int>();
  A.read_off = c->read_off.as<long long>();
  A.n_reads = c->n_reads;
  A.tail = which == 1 ? c->k - 1 : 0;
  A.pat = c->s0.as<int>();
  A.pat_off = c->s1.as<long long>();
  A.bucket_off = c->s4.as<long long>();
  A.bucket_pat = reinterpret_cast<const int*>(idx_sorted);
  A.domain = domain;
  A.has_bits = c->gm_mask.as<unsigned int>();
  A.lds_words = bit_words <= MATCH_LDS_WORDS ? (int)bit_words : 0;
  const size_t lds_bytes = (size_t)A.lds_words * sizeof(unsigned int);
  // grid: enough waves to fill the device a few times over, each walking reads with a grid stride (the bitmap is staged
  // once per workgroup)
  const unsigned int match_blocks = (unsigned int)std::min<long long>(nblk(c->n_reads, 16), 256ll * 16);
  A.hits = nullptr;
  A.cap = 0;
  // ---- pass 1: count per wave, pass 2: fill from the waves' first places
  const long long n_waves = (long long)match_blocks * 4;
  AMGCHK(c->match_wave.ensure((size_t)(n_waves + 2) * (sizeof(unsigned int) + sizeof(long long)) + 16));
  long long* wave_base = c->match_wave.as<long long>();
  unsigned int* wave_count = reinterpret_cast<unsigned int*>(wave_base + (n_waves + 2));
  A.wave_count = wave_count;
  A.wave_base = wave_base;
  HIPCHK(hipMemsetAsync(wave_count, 0, (size_t)(n_waves + 1) * sizeof(unsigned int), st));
  stage_begin(c, "match_count");  // (the stage is the kernel alone: bench.py prices it against the HBM roofline)
  if (c->n_reads > 0) hipLaunchKernelGGL(k_match, dim3(match_blocks), dim3(256), lds_bytes, st, A);
  stage_end(c);
  AMGCHK(prim_exscan_u32_to_i64(c, wave_count, wave_base, (size_t)n_waves + 1));
  unsigned long long total = 0;
  {
    FetchList l;
    l.add(wave_base + n_waves);
    AMGCHK(fetch(c, l, &total));
  }
  AMGCHK(c->match_read.ensure((size_t)(total + 1) * sizeof(int)));
  AMGCHK(c->match_pos.ensure((size_t)(total + 1) * sizeof(int)));
  AMGCHK(c->s5.ensure((size_t)(total + 1) * sizeof(unsigned long long) * 2));
  unsigned long long* keys = c->s5.as<unsigned long long>();
  unsigned long long* keys_sorted = keys + (total + 1);
  if (total > 0) {
    A.hits = keys;
    A.cap = total;
    stage_begin(c, "match_fill");
    hipLaunchKernelGGL(k_match, dim3(match_blocks), dim3(256), lds_bytes, st, A);
    stage_end(c);
    // sort by (pattern, read, position); the values of the pair sort are not needed
    AMGCHK(c->s2.ensure((size_t)(total + 1) * sizeof(unsigned int) * 2 + (size_t)(n_pat + 2) * sizeof(unsigned int) * 4));
    // NB: s2 may have been reallocated — bucket arrays are no longer needed after pass 2
    unsigned int* dummy_in = c->s2.as<unsigned int>();
    unsigned int* dummy_out = dummy_in + (total + 1);
    AMGCHK(prim_sort_u64_u32(c, keys, keys_sorted, dummy_in, dummy_out, (size_t)total, 64));
    hipLaunchKernelGGL(k_hit_split, dim3(nblk((long long)total + 1, 256)), dim3(256), 0, st, keys_sorted,
                       (long long)total, (long long)n_pat, c->match_read.as<int>(), c->match_pos.as<int>(),
                       c->match_off.as<long long>());
  } else {
    HIPCHK(hipMemsetAsync(c->match_off.p, 0, (size_t)(n_pat + 1) * sizeof(long long), st));
  }
  HIPCHK(hipMemcpyAsync(hit_offsets, c->match_off.p, (size_t)(n_pat + 1) * sizeof(long long),
                        hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  c->match_total = (int64_t)total;
  c->match_npat = n_pat;
  c->match_valid = true;
  return AMG_OK;
}
