// amg_kmer.h — a nucleotide k-mer (k <= 32) as up to four 64-bit words, for the sketch kernels (amg_minhash.hip,
// amg_bubbles.hip): cut out of bases staged in LDS, reverse-complemented, compared and hashed a WORD at a time.
//
// sourmash's sketch (amg_minhash.hip holds the definition and its source): canonical k-mer = the bytewise smaller of
// the k-mer and its reverse complement, hash = first 64 bits of MurmurHash3_x64_128(canonical k-mer, seed 42).  A
// k-mer that holds a character outside ACGT is skipped (force = True).
//
// A byte loop per k-mer (cut the window, copy it, build the reverse complement, compare, feed the hash a byte at a
// time out of a private array) is ~400 instructions and a round trip through scratch memory per base.  Here the staged
// bases are ASCII upper case with 0 for anything outside ACGT; a k-mer is NW + 1 aligned LDS words funnel-shifted into
// NW; "no 0 byte" is one SWAR test per word; the complement of eight bases is three logic operations
// (A 0x41 <-> T 0x54 differ by 0x15, C 0x43 <-> G 0x47 by 0x04, and bit 1 tells the two pairs apart); the reversal is
// a byte swap per word and one funnel shift by the padding; MurmurHash3 takes its 8-byte blocks as they are.
#pragma once
#include "amg_device.h"

#define KM_MAX_K 32

// what the staging loop stores for a base: upper-case A / C / G / T, 0 for everything else
__device__ __forceinline__ unsigned char km_stage(unsigned char c) {
  if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
  return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : (unsigned char)0;
}

__device__ __forceinline__ unsigned long long km_rotl(unsigned long long x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ unsigned long long km_fmix(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xFF51AFD7ED558CCDull;
  k ^= k >> 33;
  k *= 0xC4CEB9FE1A85EC53ull;
  k ^= k >> 33;
  return k;
}

// bytes [0, n) of a word kept, the rest cleared (n in 0 .. 8)
__device__ __forceinline__ unsigned long long km_low_bytes(unsigned long long x, int n) {
  return n >= 8 ? x : (n <= 0 ? 0ull : x & ((1ull << (8 * n)) - 1ull));
}

// The k-mer that starts at byte i of `lds` (8-byte aligned, at least 8 readable bytes behind the k-mer's last word) as
// NW little-endian words, bytes beyond k cleared.  Returns false when a base outside ACGT is among its k.
template <int NW>
__device__ __forceinline__ bool km_load(const unsigned char* lds, int i, int k, unsigned long long (&f)[NW]) {
  const unsigned long long* W = reinterpret_cast<const unsigned long long*>(lds + (i & ~7));
  const int s8 = (i & 7) * 8;
  unsigned long long x[NW + 1];
#pragma unroll
  for (int j = 0; j <= NW; ++j) x[j] = W[j];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    unsigned long long v = s8 ? (x[j] >> s8) | (x[j + 1] << (64 - s8)) : x[j];
    const int n = k - 8 * j;   // bytes of the k-mer in this word (>= 1)
    v = km_low_bytes(v, n);
    f[j] = v;
    const unsigned long long t = n >= 8 ? v : v | (~0ull << (8 * n));   // padding must not look like a bad base
    ok = ok && (((t - 0x0101010101010101ull) & ~t & 0x8080808080808080ull) == 0ull);
  }
  return ok;
}

// complement of eight staged bases (bytes that are 0 come out as rubbish: the caller masks)
__device__ __forceinline__ unsigned long long km_comp8(unsigned long long x) {
  return x ^ 0x1515151515151515ull ^ (((x >> 1) & 0x0101010101010101ull) * 0x11ull);
}

// reverse complement of a k-mer of NW words: the 8 NW bytes reversed (word order + a byte swap each) put the k-mer's
// last base first after `pad` = 8 NW - k bytes of padding, which one funnel shift removes
template <int NW>
__device__ __forceinline__ void km_revcomp(const unsigned long long (&f)[NW], int k, unsigned long long (&r)[NW]) {
  const int pad8 = (8 * NW - k) * 8;
  unsigned long long t[NW + 1];
#pragma unroll
  for (int j = 0; j < NW; ++j) t[j] = __builtin_bswap64(f[NW - 1 - j]);
  t[NW] = 0ull;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const unsigned long long v = pad8 ? (t[j] >> pad8) | (t[j + 1] << (64 - pad8)) : t[j];
    r[j] = km_low_bytes(km_comp8(v), k - 8 * j);
  }
}

// a <= b as byte strings (the first byte is the low byte of word 0)
template <int NW>
__device__ __forceinline__ bool km_not_greater(const unsigned long long (&a)[NW], const unsigned long long (&b)[NW]) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    if (a[j] != b[j]) return __builtin_bswap64(a[j]) < __builtin_bswap64(b[j]);
  }
  return true;
}

// first 64 bits of MurmurHash3_x64_128 (Austin Appleby, public domain) of the k bytes held in NW words
template <int NW>
__device__ __forceinline__ unsigned long long km_murmur_h1(const unsigned long long (&x)[NW], int len, unsigned long long seed) {
  const unsigned long long c1 = 0x87C37B91114253D5ull, c2 = 0x4CF5AD432745937Full;
  unsigned long long h1 = seed, h2 = seed;
  int used = 0;
  if constexpr (NW >= 2) {
#pragma unroll
    for (int blk = 0; blk + 1 < NW; blk += 2) {
      if (len - 8 * blk >= 16) {
        unsigned long long k1 = x[blk], k2 = x[blk + 1];
        k1 *= c1; k1 = km_rotl(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = km_rotl(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52DCE729ull;
        k2 *= c2; k2 = km_rotl(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = km_rotl(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495AB5ull;
        used = blk + 2;
      }
    }
  }
  const int t = len - 8 * used;   // 0 .. 15 bytes of tail, in words used and used + 1 (cleared beyond the k-mer)
  unsigned long long k1 = 0ull, k2 = 0ull;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    if (j == used) k1 = x[j];
    if (j == used + 1) k2 = x[j];
  }
  if (t > 8) { k2 *= c2; k2 = km_rotl(k2, 33); k2 *= c1; h2 ^= k2; }
  if (t > 0) { k1 *= c1; k1 = km_rotl(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (unsigned long long)len;
  h2 ^= (unsigned long long)len;
  h1 += h2; h2 += h1;
  h1 = km_fmix(h1); h2 = km_fmix(h2);
  h1 += h2;
  return h1;
}

// hash of the canonical form of the k-mer at byte i of the staged bases; false: skipped (a base outside ACGT)
template <int NW>
__device__ __forceinline__ bool km_canonical_hash(const unsigned char* lds, int i, int k, unsigned long long* out) {
  unsigned long long f[NW], r[NW];
  if (!km_load<NW>(lds, i, k, f)) return false;
  km_revcomp<NW>(f, k, r);
  *out = km_not_greater<NW>(f, r) ? km_murmur_h1<NW>(f, k, 42ull) : km_murmur_h1<NW>(r, k, 42ull);
  return true;
}
