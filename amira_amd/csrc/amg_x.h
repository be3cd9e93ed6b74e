// amg_x.h — the 16-byte slot of the exact-key tables (amg_build_x.hip), shared with the
// multi-GPU merge (amg_dist.hip), which reads local keys back out of it.
#pragma once
#include "amg_device.h"
#include "amg_tile.h"

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


// ------------------------------------------------------------------ shared by amg_build_x.hip (two table passes,
// multi-GPU shards) and amg_build_f.hip (the fused single-GPU pass)
// canonical tuple -> (w1, tag): token j occupies bits [j*bits, (j+1)*bits) of a 94-bit value,
// w1 = (low 63 bits << 1) | 1, tag = (high 31 bits << 1) | 1 — both non-zero by construction
template <class View>
__device__ __forceinline__ void x_pack(const View& w, int k, int flip, int dir, int bits,
                                       unsigned long long& w1, unsigned int& tag) {
  unsigned long long lo = 0, hi = 0;
  int sh = 0;
  for (int j = 0; j < k; ++j, sh += bits) {
    const unsigned long long c = (unsigned long long)(unsigned int)canon_tok(w, k, flip, dir, j);
    if (sh < 63) {
      lo |= c << sh;
      if (sh + bits > 63) hi |= c >> (63 - sh);
    } else {
      hi |= c << (sh - 63);
    }
  }
  w1 = (lo << 1) | 1ull;  // bit 63 of lo (it belongs to hi) falls off here
  tag = ((unsigned int)hi << 1) | 1u;
}

// Compile-time k: direction, canonical tuple and packing as straight-line code (no early-exit
// loop, no per-token branches).  Returns the direction (0: palindrome).
template <int K, bool TWO>
__device__ __forceinline__ int x_canon_pack(const int* w, int flip, int bits, unsigned long long& w1,
                                            unsigned int& tag) {
  int a[K];
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = w[j];
  int dir = 0;  // the first differing position decides: walk from the last to the first
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
    const int d = a[j] - (flip - a[K - 1 - j]);
    dir = d != 0 ? (d < 0 ? 1 : -1) : dir;
  }
  if (TWO) {
    unsigned __int128 v = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const unsigned int c = (unsigned int)(dir > 0 ? a[j] : flip - a[K - 1 - j]);
      v |= (unsigned __int128)c << (j * bits);
    }
    w1 = ((unsigned long long)v << 1) | 1ull;
    tag = ((unsigned int)(unsigned long long)(v >> 63) << 1) | 1u;
  } else {
    unsigned long long v = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const unsigned int c = (unsigned int)(dir > 0 ? a[j] : flip - a[K - 1 - j]);
      v |= (unsigned long long)c << (j * bits);
    }
    w1 = (v << 1) | 1ull;
    tag = 1u;
  }
  return dir;
}

// Find or create the slot of key (w1, tag) starting at `idx`; `v` is the content of that first
// slot as a PLAIN load returned it.
//
// Plain loads are served by the issuing XCD's L2 and may be stale, but a slot only ever moves
// empty -> w1 -> tag -> id, each step once: a cached view that already shows a complete foreign
// key, or our key with its id, is final and is trusted (no fabric transaction: agent-scope
// loads cost one 64-byte fabric request each, ~100 G/s, atomics ~27 G/s, L2 hits ~255 G/s).
// Anything less (empty, tag or id missing) is settled by the CAS itself or by an agent-scope
// re-read of w2.  (Serving the first tiles from a separate launch with agent-scope loads only,
// so that no L2 caches a hot slot before it is complete, was measured and bought nothing.)
// A CAS on w1 takes the slot.  TWO: the key has a second word (tag), set by a second CAS by
// whichever thread needs it first; that thread owns the slot ("created").
// Returns the slot or -1; w2v = the slot's second word as seen (low 32 bits zero: the claim id
// is not published yet).
// BUCKET: the slots probed are those of ONE 128-byte line (8 slots), starting at idx and wrapping inside the line —
// the bucket region of the node table (k_nodes_m), probed `limit` + 1 slots deep, after which the key goes to its
// hashed slot.
template <bool TWO, bool BUCKET = false>
__device__ __forceinline__ int x_upsert(Slot16* tab, unsigned int mask,
                                              unsigned long long w1, unsigned int tag,
                                              unsigned int idx, ulonglong2 v,
                                              unsigned int limit, const unsigned long long* abort_flag,
                                              unsigned long long& w2v, bool& created, unsigned int off = 0u) {
  // off: the hashed slots are tab[off .. off + mask] (slots below `off` are addressed directly: home slots of the
  // edge pass, probed with limit 0 — taken, found or given up after that one slot)
  created = false;
  w2v = 0;
  unsigned int probes = 0;
  while (true) {
    Slot16* s = tab + idx;
    unsigned long long c1 = v.x, c2 = v.y;
    const bool mine = c1 == w1 && (!TWO || (unsigned int)(c2 >> 32) == tag);
    if (mine && (unsigned int)c2 != 0u) {
      w2v = c2;
      return (int)idx;
    }
    // The cached view does not decide.  No step below waits for another thread (lanes of one
    // wave must never wait for each other inside a loop).
    if (c1 == 0ull) {
      // looks empty: try to take it — the CAS returns the truth, no coherent re-read needed
      c1 = atomicCAS(&s->w1, 0ull, w1);
      if (c1 == 0ull) {
        if (!TWO) {
          created = true;
          return (int)idx;
        }
        c1 = w1;
        c2 = 0ull;
      } else if (c1 == w1) {
        c2 = ld_u64(&s->w2);  // somebody holds our w1: tag / id with agent scope
      }
    } else if (c1 == w1 && (mine || (TWO && (c2 >> 32) == 0ull))) {
      c2 = ld_u64(&s->w2);  // tag or id missing in the cached view
    }
    if (c1 == w1) {
      if (TWO) {
        // the slot belongs to whoever sets the tag (a thread that claimed w1 but lost w2 to a
        // different tag moves on, as does every later thread of its key at this slot)
        if ((c2 >> 32) == 0ull) {
          const unsigned long long old = atomicCAS(&s->w2, 0ull, (unsigned long long)tag << 32);
          if (old == 0ull) {
            created = true;
            return (int)idx;
          }
          c2 = old;
        }
        if ((unsigned int)(c2 >> 32) == tag) {
          w2v = c2;  // low word 0: the id is still on its way (x_claim waits for it)
          return (int)idx;
        }
      } else {
        w2v = c2;
        return (int)idx;
      }
    }
    if (probes >= limit) return -1;
    if ((probes & 63u) == 63u && *reinterpret_cast<const volatile unsigned long long*>(abort_flag))
      return -1;
    ++probes;
    idx = BUCKET ? ((idx & ~7u) | ((idx + 1u) & 7u)) : off + ((idx - off + 1u) & mask);
    v = *reinterpret_cast<const ulonglong2*>(tab + idx);
  }
}

// Second word of a slot.  TWO (the key spills into it): [63:32] tag, [31:ib] COARSE token position
// of the creating window (token index >> cshift), [ib-1:0] claim id + 1.  One-word keys: [63:32]
// the creator's exact first-seen (complemented), [31:0] claim id + 1.  Either way the probe load
// already tells almost every window that it comes after the creator and cannot be the first
// occurrence: the first-seen words of the claim are then not even read (one random access per
// window less; the table passes are bound by the L2 request rate).
struct XW2 {
  int ib;       // bits of the id field (TWO)
  int cshift;   // coarse position = token index >> cshift (TWO)
};
template <bool TWO>
__device__ __forceinline__ unsigned int xw2_id1(unsigned long long w2v, const XW2& f) {
  return TWO ? ((unsigned int)w2v & ((1u << f.ib) - 1u)) : (unsigned int)w2v;
}

// Claim ids for the slots this block created + first-seen bookkeeping.  slot[it] < 0: nothing.
// In: lw[it] / hw[it] = low / high half of the second slot word as seen by the probe (the high half
// only matters for one-word keys).  Out: id1[it] = claim id + 1 of every
// item with a slot.  Window `it` of the thread starts at token tbase + it * TILE_THREADS; its
// first-seen value is (token << FSH) | low bits (lowbits: FSH bits per item, packed), kept
// complemented.  (Positions and first-seen values are recomputed instead of kept in arrays: the
// table kernels run 8 waves per SIMD on 64 registers.)
template <bool TWO, int FSH, int STRIDE = TILE_THREADS>  // window `it` of a thread starts STRIDE tokens after window it - 1
__device__ __forceinline__ void x_claim(Slot16* tab, const int (&slot)[TILE_ITEMS],
                                        const unsigned int (&lw)[TILE_ITEMS],
                                        const unsigned int (&hw)[TWO ? 1 : TILE_ITEMS],
                                        unsigned int (&id1)[TILE_ITEMS], unsigned int created,
                                        const unsigned int (&tag)[TILE_ITEMS], unsigned int tbase,
                                        unsigned int lowbits, const XW2 f, unsigned int* first2,
                                        unsigned int* __restrict__ slot_by_claim,
                                        unsigned long long* counter, unsigned long long* stuck,
                                        unsigned int* s_wave, unsigned long long* s_base,
                                        bool skip_first) {
  auto tpos = [&](int it) { return tbase + (unsigned int)it * STRIDE; };
  auto fi = [&](int it) { return ~((tpos(it) << FSH) | ((lowbits >> (it * FSH)) & ((1u << FSH) - 1u))); };
  unsigned int total;
  const unsigned int off = block_exscan<TILE_THREADS / 64>((unsigned int)__popc(created), &total, s_wave);
  if (threadIdx.x == 0) *s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
  __syncthreads();
  unsigned int claim = (unsigned int)(*s_base) + off;
  if (created) {
#pragma unroll
    for (int it = 0; it < TILE_ITEMS; ++it)
      if (created & (1u << it)) {
        // the creator's first-seen goes to its own word with a plain store; everybody else
        // raises the claim's other word with atomicMax (both zero-initialised; first-seen = the larger
        // of the two), so nothing has to be ordered against the publication of the id (a
        // release fence here writes back the L2: measured 7x slower) and a creation costs no
        // read-modify-write beyond the CAS that took the slot
        first2[2u * claim + 1u] = fi(it);
        slot_by_claim[claim] = (unsigned int)slot[it];
        id1[it] = claim + 1u;
        const unsigned long long pub =
            TWO ? ((unsigned long long)tag[it] << 32) | (unsigned long long)((tpos(it) >> f.cshift) << f.ib) |
                      (unsigned long long)(claim + 1u)
                : ((unsigned long long)fi(it) << 32) | (unsigned long long)(claim + 1u);
        __hip_atomic_store(&tab[slot[it]].w2, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++claim;
      }
  }
  // found keys: wait for an id that is still on its way (its creator's block publishes without
  // waiting for anybody), then keep the minimum first-seen
  unsigned int check = 0;
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    if (slot[it] < 0 || (created & (1u << it))) continue;
    unsigned long long w = (unsigned long long)lw[it] | (TWO ? 0ull : (unsigned long long)hw[TWO ? 0 : it] << 32);
    for (unsigned int spins = 0; (unsigned int)w == 0u; ++spins) {
      w = ld_u64(&tab[slot[it]].w2);
      if ((unsigned int)w != 0u) break;
      if (spins > (1u << 22)) {  // seconds: never expected; fail the build instead of hanging
        *stuck = 1ull;
        w = 1ull;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    id1[it] = xw2_id1<TWO>(w, f);
    // can this window precede the creator's?  (coarse positions: same or earlier bucket)
    const bool maybe_first = TWO ? (tpos(it) >> f.cshift) <= (((unsigned int)w) >> f.ib)
                                 : fi(it) > (unsigned int)(w >> 32);
    if (maybe_first) check |= 1u << it;
  }
  if (skip_first) return;  // timing experiment only
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    if (!(check & (1u << it))) continue;
    // plain (possibly stale, at worst zero) reads: both words only grow, so a stale value can
    // only cause a superfluous atomicMax, never a missed one
    const unsigned int c = id1[it] - 1u;
    if (x_first_inv(first2, c) < fi(it)) atomicMax(first2 + 2u * c, fi(it));
  }
}


// Canonical orientation + packed key of the window a[0..K-1] for 16-bit tokens (two_v <= 65536),
// K odd.  Forward half-words a[j] | a[j+1] << 16, reverse-complement half-words
// (F | F << 16) - (a[j+1] | a[j] << 16).  Encoding == x_pack with bits = 16.
template <int K, bool TWO>
__device__ __forceinline__ int f_canon_pack16(const int* a, int flip, unsigned long long& w1, unsigned int& tag) {
  int dir = (2 * a[K / 2] < flip) ? 1 : -1;  // 2 x != 2V - 1: an odd k has no palindromes
#pragma unroll
  for (int j = K / 2 - 1; j >= 0; --j) {
    const int s = a[j] + a[K - 1 - j];
    dir = s != flip ? (s < flip ? 1 : -1) : dir;
  }
  const unsigned int ff = (unsigned int)flip | ((unsigned int)flip << 16);
  unsigned int word[K / 2 + 1];
#pragma unroll
  for (int m = 0; m < K / 2; ++m) {
    const unsigned int fw = (unsigned int)a[2 * m] | ((unsigned int)a[2 * m + 1] << 16);
    const unsigned int rc = ff - ((unsigned int)a[K - 1 - 2 * m] | ((unsigned int)a[K - 2 - 2 * m] << 16));
    word[m] = dir > 0 ? fw : rc;
  }
  word[K / 2] = (unsigned int)(dir > 0 ? a[K - 1] : flip - a[0]);
  const unsigned long long v = (unsigned long long)word[0] | ((unsigned long long)word[1] << 32);
  w1 = (v << 1) | 1ull;
  if constexpr (K == 3)
    tag = 1u;
  else  // K == 5
    tag = (((word[1] >> 31) | (word[K / 2] << 1)) << 1) | 1u;
  return dir;
}

// ---- one table phase for the four items of a thread: probe, insert, claim ids, first-seen.
//
// Claim ids come from F_SHARDS counters, one per shard (a wave belongs to one shard): a single
// counter word takes ~90 returning atomics per microsecond, which is what a tile per 1024 tokens
// asks of it at the speed of this pass; 64 words do not notice.  Claims are INTERLEAVED, claim =
// local index * F_SHARDS + shard, so that the early (hot) claims of every shard are small numbers
// and the claim space [0, F_SHARDS * largest local count) has few holes (entries of the per-claim
// arrays that nobody claimed keep first-seen == 0 and are skipped wherever claims are listed).
// The creators of a wave are counted with ballots and served by one atomicAdd of the wave:
// no LDS, no workgroup barrier.
#define F_SHARDS 64
#define F_CTR_STRIDE 16  // counters 128 bytes apart (u64 words)

// Chunked shards (the plain table passes, round 4): a WORKGROUP takes its claims from the counter of its shard
// (tile index & 63) with one atomicAdd, as it did from the single counter — 55 k returning atomics on one word are what
// a first-build pass waited for once its creations were cheap (0.42 of 0.72 ms in the edge pass; one word takes
// ≈ 100 per microsecond) — and shard s owns every 64th CHUNK of X_CHUNK claim ids (a tile's worth: a workgroup's stores
// into the per-claim arrays share lines).  The FIRST tiles of the stream — the head launch, which creates the genome's
// keys, and a few times as many tiles after it, which create the ones it missed — share a counter of their own
// (shard F_SHARDS) and take the ids [0, base) densely, as before: those keys hold the lowest claims, which the counting
// sweeps rely on (spread over the shards they lay scattered over 64 k ids and the two counts of a rebuild took 0.2 ms
// each instead of 0.07).  The shards' ids start at `base`; ids in use lie below
// base + X_CHUNK * F_SHARDS * ceil(largest shard count / X_CHUNK); ids nobody took keep first-seen == 0.
#define X_CHUNK 1024u
__device__ __forceinline__ unsigned int x_chunk_claim(unsigned int li, unsigned int shard) {
  return (li / X_CHUNK) * (X_CHUNK * F_SHARDS) + shard * X_CHUNK + (li % X_CHUNK);
}
// which counter a workgroup adds to and what its local index li becomes: no shards (shard < 0: one counter, claim = li),
// the head launch's counter, a shard's
struct XShard {
  int shard;          // -1, 0 .. F_SHARDS - 1, F_SHARDS (head)
  unsigned int base;  // ids of the head launch = capacity of its counter
  __device__ __forceinline__ unsigned long long* counter(unsigned long long* ctr) const {
    return shard >= 0 ? ctr + (unsigned int)shard * F_CTR_STRIDE : ctr;
  }
  __device__ __forceinline__ unsigned int limit(unsigned int cap) const { return shard == (int)F_SHARDS ? base : cap; }
  __device__ __forceinline__ unsigned int claim(unsigned int li) const {
    return shard < 0 || shard == (int)F_SHARDS ? li : base + x_chunk_claim(li, (unsigned int)shard);
  }
};

template <class T>
__device__ __forceinline__ T f_pick(const T (&a)[TILE_ITEMS], int w) {
  return w == 0 ? a[0] : w == 1 ? a[1] : w == 2 ? a[2] : a[3];
}

// STRIDE: window `it` of a thread starts STRIDE tokens after window it - 1.  homed / off: items whose idx[] is a
// HOME slot (a directly addressed slot below `off`: taken, found, or — when another key sits there — given up for the
// key's hashed slot in tab[off .. off + mask]); one-word keys only.
// HOME_PROBES: slots of its line a homed item looks at before it goes to its hashed slot (1: the home slot alone —
// the edge pass; > 1: a bucket line of the node table, k_nodes_m).
// LONE: items of the mask `lone` are keys the caller KNOWS to occur once in the whole input (edge classes with an end
// node of coverage 1): no probe, no compare-and-swap — they take a claim like every creator and write their slot,
// key and id in one 16-byte store, to tab[lone_base + claim] (a region behind the table that is never probed or
// cleared; the ranking reads the key back from there through slot_by_claim like any other).
template <bool TWO, int FSH, bool SHARDED, int STRIDE = 1, int HOME_PROBES = 1, bool LONE = false>
__device__ __forceinline__ void f_table_phase(Slot16* tab, unsigned int mask, unsigned int valid,
                                              const unsigned long long (&w1)[TILE_ITEMS],
                                              const unsigned int (&tag)[TILE_ITEMS],
                                              const unsigned int (&idx)[TILE_ITEMS],
                                              const ulonglong2 (&v)[TILE_ITEMS], unsigned int tbase,
                                              unsigned int lowbits, const XW2 f, unsigned int* first2,
                                              unsigned int* __restrict__ slot_by_claim,
                                              unsigned long long* ctr, unsigned int shard, unsigned int cap,
                                              unsigned int probe_limit, unsigned long long* status, int which,
                                              unsigned int (&id1)[TILE_ITEMS], unsigned int* s_wave = nullptr,
                                              unsigned int* made = nullptr, unsigned int homed = 0u,
                                              unsigned int off = 0u, unsigned int lone = 0u,
                                              unsigned int lone_base = 0u, XShard xs = XShard{-1, 0u},
                                              int lone_shard = -1) {
  // xs.shard >= 0 (not SHARDED): ctr is the array of shard counters, cap a shard's share (XShard)
  static_assert(!LONE || !TWO, "lone items: one-word keys");
  auto tpos = [&](int it) { return tbase + (unsigned int)it * (unsigned int)STRIDE; };
  auto fi = [&](int it) { return ~((tpos(it) << FSH) | ((lowbits >> (it * FSH)) & ((1u << FSH) - 1u))); };
  unsigned int lw[TILE_ITEMS], hw[TWO ? 1 : TILE_ITEMS];
  int slot[TILE_ITEMS];
  // ---- the key with its id, as the first probe load returned it: done (almost every window of a
  // rebuild).  Anything else goes through x_upsert below, one item at a time, in ONE copy of that code.
  unsigned int need = 0, created = 0;
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    id1[it] = 0;
    lw[it] = 0;
    slot[it] = (int)idx[it];
    if (!TWO) hw[TWO ? 0 : it] = 0;
    if (!(valid & (1u << it))) continue;
    if (LONE && (lone & (1u << it))) {  // v[it] was not even loaded
      created |= 1u << it;
      continue;
    }
    const unsigned long long c1 = v[it].x, c2 = v[it].y;
    const bool mine = c1 == w1[it] && (!TWO || (unsigned int)(c2 >> 32) == tag[it]);
    if (mine && (unsigned int)c2 != 0u) {
      lw[it] = (unsigned int)c2;
      if (!TWO) hw[TWO ? 0 : it] = (unsigned int)(c2 >> 32);
    } else {
      need |= 1u << it;
    }
  }
  while (need) {
    const int it = __ffs((int)need) - 1;
    need &= need - 1u;
    bool made;
    unsigned long long w2v;
    // (the slot is loaded again rather than picked out of v[]: a register array indexed at run time
    // lives in scratch memory)
    const unsigned int ix = f_pick(idx, it);
    const bool home = (homed >> it) & 1u;
    int sl;
    if (home)
      sl = x_upsert<TWO, (HOME_PROBES > 1)>(tab, mask, f_pick(w1, it), f_pick(tag, it), ix,
                                            *reinterpret_cast<const ulonglong2*>(tab + ix), (unsigned int)(HOME_PROBES - 1),
                                            status + ST_OVERFLOW, w2v, made, off);
    else
      sl = x_upsert<TWO>(tab, mask, f_pick(w1, it), f_pick(tag, it), ix,
                         *reinterpret_cast<const ulonglong2*>(tab + ix), probe_limit, status + ST_OVERFLOW, w2v, made, off);
    if (sl < 0 && home) {  // other keys live in the home slot(s): this one goes where its key hashes to
      const unsigned int ix2 =
          off + ((unsigned int)mix64(f_pick(w1, it) ^ (TWO ? (unsigned long long)f_pick(tag, it) * 0x9E3779B97F4A7C15ull : 0ull)) & mask);
      sl = x_upsert<TWO>(tab, mask, f_pick(w1, it), f_pick(tag, it), ix2,
                         *reinterpret_cast<const ulonglong2*>(tab + ix2), probe_limit, status + ST_OVERFLOW, w2v, made,
                         off);
    }
    if (sl < 0) {
      status[ST_OVERFLOW] = (unsigned long long)which;
      valid &= ~(1u << it);
    }
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j)
      if (j == it) {
        slot[j] = sl;
        lw[j] = (unsigned int)w2v;
        if (!TWO) hw[TWO ? 0 : j] = (unsigned int)(w2v >> 32);
      }
    if (made) created |= 1u << it;
  }
  if (made) *made = created;
  // ---- claim ids of the wave's creators
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned int n = 0, pre[TILE_ITEMS];
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    const unsigned long long m = __ballot((created >> it) & 1u);
    pre[it] = n + (unsigned int)__popcll(m & below);
    n += (unsigned int)__popcll(m);
  }
  unsigned int base = 0;
  // LONE: the lone classes of a workgroup take the ids after its table creations — or, in the first tiles of the stream
  // (xs = the dense counter), ids of the tile's SHARD like everywhere else: they are never counted (one occurrence, the
  // creator's), and with them kept out, the dense ids hold the genome's classes alone, within the one LDS range of the
  // counting sweeps (with them the head's classes reached 36 k ids and the first-build edge count needed a second
  // sweep of 0.13 ms)
  unsigned int lbase = 0;
  XShard lxs = xs;
  if constexpr (LONE) {
    unsigned int nl = 0;
#pragma unroll
    for (int it = 0; it < TILE_ITEMS; ++it) {  // pre[] so far ranks all creators of the wave: split it
      const unsigned long long ml = __ballot((lone >> it) & 1u);
      const unsigned int lone_before = nl + (unsigned int)__popcll(ml & below);  // lone creators ranked before this item
      pre[it] = ((lone >> it) & 1u) ? lone_before : pre[it] - lone_before;
      nl += (unsigned int)__popcll(ml);
    }
    const unsigned int nt = n - nl;
    const unsigned int wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = nt | (nl << 16);
    __syncthreads();
    unsigned int before_t = 0, before_l = 0, total_t = 0, total_l = 0;
#pragma unroll
    for (unsigned int w = 0; w < TILE_THREADS / 64; ++w) {
      const unsigned int cnt = s_wave[w];
      before_t += w < wave ? (cnt & 0xffffu) : 0u;
      before_l += w < wave ? (cnt >> 16) : 0u;
      total_t += cnt & 0xffffu;
      total_l += cnt >> 16;
    }
    const bool split = xs.shard == (int)F_SHARDS && lone_shard >= 0;  // workgroup-uniform
    if (split) lxs = XShard{lone_shard, xs.base};
    if (total_t + total_l) {
      if (threadIdx.x == 0) {
        unsigned int bt, bl;
        if (split) {
          bt = total_t ? (unsigned int)atomicAdd(xs.counter(ctr), (unsigned long long)total_t) : 0u;
          bl = total_l ? (unsigned int)atomicAdd(lxs.counter(ctr), (unsigned long long)total_l) : 0u;
        } else {
          bt = (unsigned int)atomicAdd(xs.counter(ctr), (unsigned long long)(total_t + total_l));
          bl = bt + total_t;
        }
        s_wave[TILE_THREADS / 64] = bt;
        s_wave[TILE_THREADS / 64 + 1] = bl;
      }
      __syncthreads();
      base = s_wave[TILE_THREADS / 64] + before_t;
      lbase = s_wave[TILE_THREADS / 64 + 1] + before_l;
    }
  } else if constexpr (SHARDED) {
    if (n) {  // wave-uniform
      unsigned int b = 0;
      if (lane == 0) b = (unsigned int)atomicAdd(ctr, (unsigned long long)n);
      base = (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
    }
  } else {
    // one atomicAdd per workgroup on the single counter (claims are then dense: 0 .. number of keys - 1);
    // s_wave: TILE_THREADS / 64 wave totals + the workgroup's base
    const unsigned int wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = n;
    __syncthreads();
    unsigned int before = 0, total = 0;
#pragma unroll
    for (unsigned int w = 0; w < TILE_THREADS / 64; ++w) {
      const unsigned int cnt = s_wave[w];
      before += w < wave ? cnt : 0u;
      total += cnt;
    }
    if (total) {  // workgroup-uniform
      if (threadIdx.x == 0)
        s_wave[TILE_THREADS / 64] =
            (unsigned int)atomicAdd(xs.counter(ctr), (unsigned long long)total);
      __syncthreads();
      base = s_wave[TILE_THREADS / 64] + before;
    }
  }
  if (created) {
#pragma unroll
    for (int it = 0; it < TILE_ITEMS; ++it)
      if (created & (1u << it)) {
        const bool is_lone = LONE && ((lone >> it) & 1u);
        unsigned int li = (is_lone ? lbase : base) + pre[it];
        if (li >= (SHARDED ? cap : is_lone ? lxs.limit(cap) : xs.limit(cap))) {  // the shard's share of the claim arrays is used up: the host rebuilds larger
          status[ST_OVERFLOW] = (unsigned long long)which;
          li = 0;
        }
        const unsigned int claim = SHARDED ? li * F_SHARDS + shard : is_lone ? lxs.claim(li) : xs.claim(li);
        // the creator's first-seen goes to its own word with a plain store; everybody else raises the
        // claim's other word with atomicMax (both zero-initialised, first-seen = the larger): nothing
        // has to be ordered against the publication of the id
        first2[2u * claim + 1u] = fi(it);
        id1[it] = claim + 1u;
        const unsigned long long pub =
            TWO ? ((unsigned long long)tag[it] << 32) | (unsigned long long)((tpos(it) >> f.cshift) << f.ib) |
                      (unsigned long long)(claim + 1u)
                : ((unsigned long long)fi(it) << 32) | (unsigned long long)(claim + 1u);
        if (LONE && (lone & (1u << it))) {
          slot_by_claim[claim] = lone_base + claim;
          *reinterpret_cast<ulonglong2*>(tab + lone_base + claim) = make_ulonglong2(w1[it], pub);
        } else {
          slot_by_claim[claim] = (unsigned int)slot[it];
          __hip_atomic_store(&tab[slot[it]].w2, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
  }
  // ---- found keys: wait for an id that is still on its way (a creator publishes without waiting
  // for anybody, after at most its own wave's atomicAdd), then keep the minimum first-seen
  unsigned int check = 0;
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    if (!(valid & (1u << it)) || (created & (1u << it))) continue;
    unsigned long long w = (unsigned long long)lw[it] | (TWO ? 0ull : (unsigned long long)hw[TWO ? 0 : it] << 32);
    for (unsigned int spins = 0; (unsigned int)w == 0u; ++spins) {
      w = ld_u64(&tab[slot[it]].w2);
      if ((unsigned int)w != 0u) break;
      if (spins > (1u << 22)) {  // seconds: never expected; fail the build instead of hanging
        status[ST_MISC] = 1ull;
        w = 1ull;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    id1[it] = xw2_id1<TWO>(w, f);
    // can this window precede the creator's?  (coarse positions: same or earlier bucket)
    const bool maybe_first = TWO ? (tpos(it) >> f.cshift) <= (((unsigned int)w) >> f.ib)
                                 : fi(it) > (unsigned int)(w >> 32);
    if (maybe_first) check |= 1u << it;
  }
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    if (!(check & (1u << it))) continue;
    // plain (possibly stale, at worst zero) reads: both words only grow, so a stale value can only
    // cause a superfluous atomicMax, never a missed one
    const unsigned int c = id1[it] - 1u;
    if (x_first_inv(first2, c) < fi(it)) atomicMax(first2 + 2u * c, fi(it));
  }
}

// ------------------------------------------------------------------ two-word keys, the slot owned by whoever takes w1
// (round 4).  f_table_phase above settles a two-word key with TWO compare-and-swaps (w1, then the tag in w2) and a
// third memory-side operation when the claim id is published: a first build makes 5.4 M keys, and those 16 M operations
// at the rate the memory side executes them are what it costs over a rebuild.  Here the thread whose CAS takes w1 OWNS
// the slot; the second word — tag, coarse position, claim id — arrives in ONE store when the id is published.  A slot
// is empty, owned (w1 set, w2 zero) or published (w2 complete, never changed again): one state fewer.  A thread that
// finds w1 equal to its own while w2 is still zero cannot tell yet whether the slot holds its key (the low 63 bits agree;
// the tag decides): it remembers the slot and looks again AFTER its own workgroup has published this round's creations —
// the owner publishes after its workgroup's barrier and one atomicAdd, waiting for nobody, so there is no cycle — and
// if the tag turns out to be another key's it goes on probing by itself (a key it then creates takes its claim id with
// an atomicAdd of its own).  That needs two keys that agree in 63 bits to meet in one probe chain while one of them
// is unpublished: rare, but it has to be right.
// state: 0 found (w2v complete), 1 created (this thread owns the slot), 2 pending (w1 equal, w2 not published yet)
template <bool BUCKET>
__device__ __forceinline__ int x_upsert_own(Slot16* tab, unsigned int mask, unsigned long long w1, unsigned int tag,
                                            unsigned int idx, ulonglong2 v, unsigned int limit,
                                            const unsigned long long* abort_flag, unsigned long long& w2v, int& state,
                                            unsigned int off = 0u) {
  state = 0;
  w2v = 0;
  unsigned int probes = 0;
  while (true) {
    Slot16* s = tab + idx;
    unsigned long long c1 = v.x, c2 = v.y;
    if (c1 == 0ull) {
      c1 = atomicCAS(&s->w1, 0ull, w1);  // looks empty: the CAS returns the truth
      if (c1 == 0ull) {
        state = 1;
        return (int)idx;
      }
      c2 = 0ull;  // (whatever the cached view said about w2 belongs to no key yet)
    }
    if (c1 == w1) {
      if (c2 == 0ull) c2 = ld_u64(&s->w2);  // not published in the view we have: one look with agent scope
      if (c2 == 0ull) {
        state = 2;
        return (int)idx;
      }
      if ((unsigned int)(c2 >> 32) == tag) {  // (a published second word is complete and final)
        w2v = c2;
        return (int)idx;
      }
    }
    if (probes >= limit) return -1;
    if ((probes & 63u) == 63u && *reinterpret_cast<const volatile unsigned long long*>(abort_flag)) return -1;
    ++probes;
    idx = BUCKET ? ((idx & ~7u) | ((idx + 1u) & 7u)) : off + ((idx - off + 1u) & mask);
    v = *reinterpret_cast<const ulonglong2*>(tab + idx);
  }
}

template <int FSH, int STRIDE = 1, int HOME_PROBES = 1>
__device__ __forceinline__ void f_table_phase_own(Slot16* tab, unsigned int mask, unsigned int valid,
                                                  const unsigned long long (&w1)[TILE_ITEMS],
                                                  const unsigned int (&tag)[TILE_ITEMS],
                                                  const unsigned int (&idx0)[TILE_ITEMS],
                                                  const ulonglong2 (&v)[TILE_ITEMS], unsigned int tbase,
                                                  unsigned int lowbits, const XW2 f, unsigned int* first2,
                                                  unsigned int* __restrict__ slot_by_claim, unsigned long long* ctr,
                                                  unsigned int cap, unsigned int probe_limit, unsigned long long* status,
                                                  int which, unsigned int (&id1)[TILE_ITEMS], unsigned int* s_wave,
                                                  unsigned int* made_out, unsigned int homed = 0u, unsigned int off = 0u,
                                                  XShard xs = XShard{-1, 0u}) {
  // a key that met a pending slot in its bucket line goes straight to its hashed slot in the redo rounds below: with
  // more than one slot of the line probed, a later thread could create the same key in the line's NEXT slot (two claims
  // for one key).  One probe per line is what was measured fastest anyway.
  static_assert(HOME_PROBES == 1, "the own-slot protocol looks at one slot of a bucket line");
  // xs.shard >= 0: ctr is the array of shard counters, cap a shard's share (XShard)
  unsigned long long* const myctr = xs.counter(ctr);
  cap = xs.limit(cap);
  auto claim_of = [&](unsigned int li) { return xs.claim(li); };
  auto tpos = [&](int it) { return tbase + (unsigned int)it * (unsigned int)STRIDE; };
  auto fi = [&](int it) { return ~((tpos(it) << FSH) | ((lowbits >> (it * FSH)) & ((1u << FSH) - 1u))); };
  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  int slot[TILE_ITEMS];  // where the item is: its slot once settled, the slot to go on from while it is not
  unsigned int need = 0, at_home = homed, made_all = 0;
  // (id1[] holds the low half of the published second word of a FOUND key until the end, claim id + 1 of a created one)
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    id1[it] = 0;
    slot[it] = (int)idx0[it];
    if (!(valid & (1u << it))) continue;
    const unsigned long long c1 = v[it].x, c2 = v[it].y;
    // the key with its id, as the first probe load returned it: done (almost every window of a rebuild)
    if (c1 == w1[it] && (unsigned int)(c2 >> 32) == tag[it] && (unsigned int)c2 != 0u)
      id1[it] = (unsigned int)c2;
    else
      need |= 1u << it;
  }
  // one item through the table from slot[it] on: found / created / pending; w1 of the item passed in
  auto upsert_item = [&](int it, unsigned long long kw, unsigned int& created, unsigned int& pending) {
    int state;
    unsigned long long w2v;
    const unsigned int ix = (unsigned int)f_pick(slot, it);
    const unsigned int tg = f_pick(tag, it);
    int sl;
    if ((at_home >> it) & 1u) {
      sl = x_upsert_own<(HOME_PROBES > 1)>(tab, mask, kw, tg, ix, *reinterpret_cast<const ulonglong2*>(tab + ix),
                                           (unsigned int)(HOME_PROBES - 1), status + ST_OVERFLOW, w2v, state, off);
      if (sl < 0) {  // other keys live in the home slot(s): this one goes where its key hashes to
        at_home &= ~(1u << it);
        const unsigned int ix2 = off + ((unsigned int)mix64(kw ^ ((unsigned long long)tg * 0x9E3779B97F4A7C15ull)) & mask);
        sl = x_upsert_own<false>(tab, mask, kw, tg, ix2, *reinterpret_cast<const ulonglong2*>(tab + ix2), probe_limit,
                                 status + ST_OVERFLOW, w2v, state, off);
      }
    } else {
      sl = x_upsert_own<false>(tab, mask, kw, tg, ix, *reinterpret_cast<const ulonglong2*>(tab + ix), probe_limit,
                               status + ST_OVERFLOW, w2v, state, off);
    }
    if (sl < 0) {
      status[ST_OVERFLOW] = (unsigned long long)which;
      valid &= ~(1u << it);
      state = 0;
      w2v = 0;
    }
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j)
      if (j == it) {
        slot[j] = sl;
        id1[j] = (unsigned int)w2v;
      }
    if (sl >= 0 && state == 1) created |= 1u << it;
    if (sl >= 0 && state == 2) pending |= 1u << it;
  };
  // claim ids of a round's creators (ballots, one atomicAdd per workgroup on the single counter) and their publication
  auto claim_and_publish = [&](unsigned int created) {
    unsigned int n = 0, pre[TILE_ITEMS];
#pragma unroll
    for (int it = 0; it < TILE_ITEMS; ++it) {
      const unsigned long long m = __ballot((created >> it) & 1u);
      pre[it] = n + (unsigned int)__popcll(m & below);
      n += (unsigned int)__popcll(m);
    }
    if (lane == 0) s_wave[wave] = n;
    __syncthreads();
    unsigned int before = 0, total = 0, base = 0;
#pragma unroll
    for (unsigned int w = 0; w < TILE_THREADS / 64; ++w) {
      const unsigned int cnt = s_wave[w];
      before += w < wave ? cnt : 0u;
      total += cnt;
    }
    if (total) {  // workgroup-uniform
      if (threadIdx.x == 0) s_wave[TILE_THREADS / 64] = (unsigned int)atomicAdd(myctr, (unsigned long long)total);
      __syncthreads();
      base = s_wave[TILE_THREADS / 64] + before;
    }
    if (created) {
#pragma unroll
      for (int it = 0; it < TILE_ITEMS; ++it)
        if (created & (1u << it)) {
          unsigned int li = base + pre[it];
          if (li >= cap) {  // the claim arrays are used up: the host rebuilds larger
            status[ST_OVERFLOW] = (unsigned long long)which;
            li = 0;
          }
          const unsigned int claim = claim_of(li);
          first2[2u * claim + 1u] = fi(it);  // the creator's own word, plain store (others raise the other word)
          slot_by_claim[claim] = (unsigned int)slot[it];
          id1[it] = claim + 1u;
          const unsigned long long pub = ((unsigned long long)tag[it] << 32) |
                                         (unsigned long long)((tpos(it) >> f.cshift) << f.ib) | (unsigned long long)(claim + 1u);
          __hip_atomic_store(&tab[slot[it]].w2, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
  };
  // the second word of a slot that was owned but not published when this thread met it (its owner publishes without
  // waiting for anybody, after at most its own workgroup's barrier)
  auto published = [&](int sl) {
    unsigned long long w = 0ull;
    for (unsigned int spins = 0;; ++spins) {
      w = ld_u64(&tab[sl].w2);
      if (w != 0ull) break;
      if (spins > (1u << 22)) {  // seconds: never expected; fail the build instead of hanging
        status[ST_MISC] = 1ull;
        w = ~0ull;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    return w;
  };

  // ---- the round every workgroup runs
  unsigned int redo = 0;
  {
    unsigned int created = 0, pending = 0;
    while (need) {
      const int it = __ffs((int)need) - 1;
      need &= need - 1u;
      upsert_item(it, f_pick(w1, it), created, pending);
    }
    made_all = created;
    claim_and_publish(created);
#pragma unroll
    for (int it = 0; it < TILE_ITEMS; ++it) {
      if (!(pending & (1u << it))) continue;
      const unsigned long long w = published(slot[it]);
      if ((unsigned int)(w >> 32) == tag[it])
        id1[it] = (unsigned int)w;
      else
        redo |= 1u << it;  // another key with the same low 63 bits lives there
    }
  }
  // ---- an item that met a half-equal key (the same low 63 bits, another tag) goes on from the next slot BY ITSELF:
  // rare, so a key it creates takes its claim id with an atomicAdd of its own instead of the workgroup's scan — no
  // barrier, nobody else involved.  One step per loop iteration and lane (probe a slot, or look once more at a second
  // word that is not there yet): a lane that has just taken a slot publishes in the same iteration, so no lane of a
  // wave ever spins waiting for another lane of the same wave.  (The slot the item stopped at holds its own w1.)
  while (redo) {
    const int it = __ffs((int)redo) - 1;
    redo &= redo - 1u;
    const unsigned int stopped = (unsigned int)f_pick(slot, it);
    const unsigned long long kw = tab[stopped].w1;
    const unsigned int tg = f_pick(tag, it);
    unsigned int at;
    if ((at_home >> it) & 1u) {
      at_home &= ~(1u << it);
      at = off + ((unsigned int)mix64(kw ^ ((unsigned long long)tg * 0x9E3779B97F4A7C15ull)) & mask);
    } else {
      at = off + ((stopped - off + 1u) & mask);
    }
    unsigned int got = 0, probes = 0, polls = 0;
    bool born = false, lost = false;
    while (true) {
      Slot16* sp = tab + at;
      unsigned long long c1 = sp->w1, c2 = 0ull;
      if (c1 == 0ull) {
        c1 = atomicCAS(&sp->w1, 0ull, kw);
        if (c1 == 0ull) {  // taken: claim id, first-seen, publication — all in this iteration
          unsigned int li = (unsigned int)atomicAdd(myctr, 1ull);
          if (li >= cap) {
            status[ST_OVERFLOW] = (unsigned long long)which;
            li = 0;
          }
          const unsigned int claim = claim_of(li);
          const unsigned int tp = tbase + (unsigned int)it * (unsigned int)STRIDE;
          first2[2u * claim + 1u] = ~((tp << FSH) | ((lowbits >> (it * FSH)) & ((1u << FSH) - 1u)));
          slot_by_claim[claim] = at;
          __hip_atomic_store(&sp->w2, ((unsigned long long)tg << 32) | (unsigned long long)((tp >> f.cshift) << f.ib) |
                                          (unsigned long long)(claim + 1u),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          got = claim + 1u;
          born = true;
          break;
        }
      }
      if (c1 == kw) {
        c2 = ld_u64(&sp->w2);
        if (c2 == 0ull) {  // owned, not published yet: look again in the next iteration
          if (++polls > (1u << 22)) {
            status[ST_MISC] = 1ull;
            lost = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        if ((unsigned int)(c2 >> 32) == tg) {
          got = (unsigned int)c2;
          break;
        }
      }
      if (++probes > probe_limit) {
        lost = true;
        break;
      }
      at = off + ((at - off + 1u) & mask);
    }
    if (lost) {
      status[ST_OVERFLOW] = (unsigned long long)which;
      valid &= ~(1u << it);
    }
    if (born) made_all |= 1u << it;
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j)
      if (j == it) {
        slot[j] = (int)at;
        id1[j] = got;
      }
  }
  if (made_out) *made_out = made_all;
  // ---- found keys: keep the minimum first-seen (can this window precede the creator's?  coarse positions: same or
  // earlier bucket); id1 turns from the slot's second word into claim id + 1
  unsigned int check = 0;
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    if (!(valid & (1u << it))) {
      id1[it] = 0;
      continue;
    }
    if (made_all & (1u << it)) continue;
    const unsigned int lw = id1[it];
    id1[it] = lw & ((1u << f.ib) - 1u);
    if ((tpos(it) >> f.cshift) <= (lw >> f.ib)) check |= 1u << it;
  }
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    if (!(check & (1u << it))) continue;
    const unsigned int c = id1[it] - 1u;
    if (x_first_inv(first2, c) < fi(it)) atomicMax(first2 + 2u * c, fi(it));
  }
}

// field widths of a two-word slot's second word for `max_claims` ids over T tokens
static inline XW2 xw2_for(size_t max_claims, long long T) {
  XW2 f;
  f.ib = ilog2_ceil((uint64_t)max_claims + 2);
  if (f.ib > 31) f.ib = 31;  // claims < 2^30 (slots are capped there)
  const int cb = 32 - f.ib;  // bits left for the coarse position (0: every window checks first-seen)
  const int tb = ilog2_ceil((uint64_t)(T > 0 ? T : 1) + 1);
  f.cshift = cb <= 0 ? 31 : (tb > cb ? tb - cb : 0);
  return f;
}

