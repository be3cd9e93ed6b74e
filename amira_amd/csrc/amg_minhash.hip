// amg_minhash.hip — scaled MinHash sketches of nucleotide segments on the device (SURVEY section 8
// row f1: the containment test of bubble popping, reference construct_graph.py:2148-2194 and
// :1567-1575, which call sourmash.MinHash(n=0, ksize=K, scaled=S).add_sequence(seq, force=True)).
//
// sourmash is a third-party dependency of the reference (pyproject.toml:28, not vendored); its
// published sketch definition for DNA is implemented here: upper-cased sequence, every window of
// `ksize` bases without a character outside ACGT, canonical form = the bytewise smaller of the
// k-mer and its reverse complement, hash = first 64 bits of MurmurHash3_x64_128(k-mer, seed 42),
// kept when hash <= max_hash (2^64 - 1 for scaled = 1, round(2^64 / scaled) otherwise).
//
// One thread per k-mer start.  A block stages its 1024 + ksize - 1 bases in LDS (one coalesced
// read of the byte stream; upper case, 0 for anything outside ACGT), every thread takes its k-mer
// out of LDS as 64-bit words, reverse-complements, compares and hashes it a word at a time
// (amg_kmer.h) and the survivors are compacted with a block scan + one atomicAdd per block.  Segments are told apart by a
// per-base segment id found by binary search over the (few) segment offsets of the tile.
// Algorithmic traffic: 1 byte read per base, 12 bytes written per kept hash (1/scaled of them):
// an HBM stream, integer work only.
#include "amg_kmer.h"

#define MH_TILE 1024
#define MH_MAX_K KM_MAX_K

template <int NW>  // words of a k-mer: (ksize + 7) / 8
__global__ __launch_bounds__(256) void k_minhash(const unsigned char* __restrict__ bases, long long n_bases,
                                                 const long long* __restrict__ seg_off, const int* __restrict__ seg_set,
                                                 long long n_seg, int ksize, unsigned long long max_hash,
                                                 unsigned long long* counter, long long cap,
                                                 int* __restrict__ out_set, unsigned long long* __restrict__ out_hash) {
  __shared__ __attribute__((aligned(8))) unsigned char s_b[MH_TILE + MH_MAX_K + 24];
  __shared__ unsigned int s_wave[4];
  __shared__ unsigned long long s_base;
  const long long t0 = (long long)blockIdx.x * MH_TILE;
  for (int i = threadIdx.x; i < MH_TILE + MH_MAX_K + 24; i += 256) {
    const long long t = t0 + i;
    s_b[i] = (i < MH_TILE + ksize - 1 && t < n_bases) ? km_stage(bases[t]) : (unsigned char)0;
  }
  // the segments this tile's bases lie in: two searches over all offsets per block, then every thread looks among
  // those few (a search over all of them per k-mer start was half of this kernel's time)
  __shared__ long long s_seg[2];
  if (threadIdx.x < 2) {
    const long long t = threadIdx.x == 0 ? t0 : (t0 + MH_TILE - 1 < n_bases ? t0 + MH_TILE - 1 : n_bases - 1);
    long long lo = 0, hi = n_seg;  // seg_off[lo] <= t < seg_off[hi]
    while (hi - lo > 1) {
      const long long mid = (lo + hi) >> 1;
      if (seg_off[mid] <= t) lo = mid; else hi = mid;
    }
    s_seg[threadIdx.x] = lo;
  }
  __syncthreads();
  const long long seg_lo = s_seg[0], seg_hi = s_seg[1] + 1;
  unsigned long long h[4];
  int set[4];
  unsigned int keep = 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = threadIdx.x + it * 256;
    const long long t = t0 + i;
    if (t + ksize > n_bases) continue;
    // the segment this base belongs to: last offset <= t
    long long lo = seg_lo, hi = seg_hi;  // seg_off[lo] <= t < seg_off[hi]
    while (hi - lo > 1) {
      const long long mid = (lo + hi) >> 1;
      if (seg_off[mid] <= t) lo = mid; else hi = mid;
    }
    if (t + ksize > seg_off[lo + 1]) continue;  // the window runs over the end of its segment
    unsigned long long hv;
    if (!km_canonical_hash<NW>(s_b, i, ksize, &hv)) continue;  // force=True: windows with other characters are skipped
    if (hv <= max_hash) {
      h[it] = hv;
      set[it] = seg_set[lo];
      keep |= 1u << it;
    }
  }
  unsigned int total;
  const unsigned int off = block_exscan_256((unsigned int)__popc(keep), &total, s_wave);
  if (threadIdx.x == 0) s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
  __syncthreads();
  unsigned long long o = s_base + off;
#pragma unroll
  for (int it = 0; it < 4; ++it)
    if (keep & (1u << it)) {
      if ((long long)o < cap) {
        out_set[o] = set[it];
        out_hash[o] = h[it];
      }
      ++o;
    }
}

// Hashes of every valid k-mer of every segment that pass the scaled cut, one (set id, hash) pair per
// k-mer occurrence, in no particular order (the caller makes sets of them).  bases / seg_off /
// seg_set are HOST arrays; *n_out = number of pairs; out_set / out_hash (host, capacity cap) may be
// NULL to get the count only.  scaled = 1 keeps every hash.
extern "C" int amg_minhash(amg_ctx* c, const uint8_t* bases, const int64_t* seg_off, const int32_t* seg_set,
                           int64_t n_seg, int32_t ksize, uint64_t scaled, int32_t* out_set, uint64_t* out_hash,
                           int64_t cap, int64_t* n_out) {
  if (!c || !n_out) return amg_fail(AMG_E_ARG, "null argument");
  if (ksize < 1 || ksize > MH_MAX_K) return amg_fail(AMG_E_ARG, "ksize must be in [1, %d]", MH_MAX_K);
  if (n_seg < 0 || (n_seg > 0 && (!bases || !seg_off || !seg_set))) return amg_fail(AMG_E_ARG, "bad segments");
  *n_out = 0;
  if (n_seg == 0) return AMG_OK;
  const int64_t n_bases = seg_off[n_seg];
  if (seg_off[0] != 0) return amg_fail(AMG_E_ARG, "seg_off[0] must be 0");
  for (int64_t s = 0; s < n_seg; ++s)
    if (seg_off[s + 1] < seg_off[s]) return amg_fail(AMG_E_ARG, "seg_off not monotone");
  if (n_bases == 0) return AMG_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  // sourmash (>= 4, Rust core: max_hash_for_scaled): 2^64 - 1 for scaled 1, otherwise (u64::MAX as f64 / scaled as
  // f64) as u64 — a truncation.  (The old Python helper rounded; the two agree whenever the quotient is >= 2^53,
  // i.e. for scaled <= 2048, which covers the reference's 1 and 10.)
  unsigned long long max_hash = ~0ull;
  if (scaled == 0) return amg_fail(AMG_E_ARG, "scaled must be >= 1");
  if (scaled > 1) {
    const double q = 18446744073709551616.0 / (double)scaled;   // u64::MAX as f64 == 2^64
    max_hash = q >= 18446744073709551615.0 ? ~0ull : (unsigned long long)q;
  }
  // upper bound of the output: every k-mer start (scaled == 1) or a generous share of them
  const int64_t want = (out_set && out_hash) ? cap : 0;
  DevBuf &d_b = c->s0, &d_off = c->s1, &d_set = c->s2, &d_os = c->s3, &d_oh = c->s4;
  AMGCHK(d_b.ensure((size_t)n_bases + 64));
  AMGCHK(d_off.ensure((size_t)(n_seg + 1) * sizeof(long long)));
  AMGCHK(d_set.ensure((size_t)(n_seg + 1) * sizeof(int)));
  AMGCHK(d_os.ensure((size_t)(want + 1) * sizeof(int)));
  AMGCHK(d_oh.ensure((size_t)(want + 1) * sizeof(unsigned long long)));
  HIPCHK(hipMemcpyAsync(d_b.p, bases, (size_t)n_bases, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off.p, seg_off, (size_t)(n_seg + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_set.p, seg_set, (size_t)n_seg * sizeof(int), hipMemcpyHostToDevice, st));
  unsigned long long* ctr = c->status.as<unsigned long long>() + ST_MISC;
  HIPCHK(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), st));
  stages_reset(c);
  stage_begin(c, "minhash");
  const unsigned int blocks = (unsigned int)((n_bases + MH_TILE - 1) / MH_TILE);
  auto kern = ksize <= 8 ? k_minhash<1> : ksize <= 16 ? k_minhash<2> : ksize <= 24 ? k_minhash<3> : k_minhash<4>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, d_b.as<unsigned char>(), (long long)n_bases,
                     d_off.as<long long>(), d_set.as<int>(), (long long)n_seg, (int)ksize, max_hash, ctr,
                     (long long)want, d_os.as<int>(), d_oh.as<unsigned long long>());
  stage_end(c);
  unsigned long long n = 0;
  HIPCHK(hipMemcpyAsync(&n, ctr, sizeof(n), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *n_out = (int64_t)n;
  if (want > 0) {
    const int64_t m = (int64_t)n < want ? (int64_t)n : want;
    HIPCHK(hipMemcpyAsync(out_set, d_os.p, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_hash, d_oh.p, (size_t)m * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return AMG_OK;
}
