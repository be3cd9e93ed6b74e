// amg_build_x.hip — the single-GPU build with EXACT keys and claim ids.
//
// Same result as the fingerprint path of amg_build.hip (GeneMerGraph.__init__, reference
// construct_graph.py:31-102), different bookkeeping, used whenever the canonical k-tuple
// fits 94 bits (k * ceil(log2(2V)) <= 94: k = 3 for any vocabulary, k = 5 up to 2^18 genes):
//
//   * a node slot is 16 bytes: w1 = low 63 bits of the packed tuple (+ a set bit 0), w2 =
//     {remaining tuple bits + a set bit, where the creating window was, claim id + 1} (struct
//     XW2).  The tuple itself is the key, so no fingerprint verification pass is needed and one
//     16-byte (plain, L2-served) load per probe decides it; see x_upsert for when a cached view
//     may be trusted.
//   * the thread that creates a slot gives it a CLAIM id: creators of a block are counted
//     with a block scan, one atomicAdd per block reserves the ids, the id is published in the
//     slot (threads that found the key before the id was there wait for it after their own
//     block has published — a block never waits before publishing, so there is no cycle).
//   * every per-key quantity lives in DENSE arrays indexed by claim id: first-seen (x_first, two
//     words per claim, read only by the few windows that might precede the creator), slot
//     (x_slot), final node id (x_final).  Claim order follows the token order, so the
//     frequently hit genome nodes own the first few ten thousand claims: the per-window
//     gathers of the edge pass (claim -> node id) and the first-seen updates hit a small
//     L2-resident region instead of one 128-byte line per hot slot, the occupied slots need
//     no table scan to be listed, and occurrence counting (k_count_ids) works on claim ids
//     directly, without a slot -> id gather.
//   * node id = rank of first-seen among the claims (32-bit keys: tokens < 2^29), as before.
// Edge classes ((lo id, hi id, sign), construct_edge.py:104-124) use the same scheme with a
// one-word key.
#include "amg_tile.h"
#include "amg_x.h"

// ------------------------------------------------------------------ tuples that do not fit the slot: 94-bit fingerprints
// k * ceil(log2 2V) > 94 (k >= 7 on a 20 000-gene vocabulary): the key of the SAME 16-byte slots, claim ids and dense
// per-claim arrays is a 94-bit fingerprint of the canonical tuple instead of the tuple itself (w1 = 63 bits, tag = 31).
// Nothing else of the build changes; what a slot no longer says is read from the token stream where it is needed:
//   * a node's tokens (k_x_assign_nodes*) are the window at its first-seen position, canonical by its direction bit;
//   * every window's tuple is compared with the tuple at its claim's first-seen position (k_x_verify_fp: the creator's
//     window — for the genome's nodes a few hundred kilobytes of the stream's head, L2-resident): two gene-mers that
//     share a fingerprint raise ST_COLLISION and the build is repeated with the next seed (AMG_TEST_WEAK_FP cuts the
//     fingerprint to 12 bits so that this happens).
// The round-1 path for such k — 32-byte slots, sorted compaction lists, a packed-tuple gather per window in the edge
// pass — took 3.5 to 6 times as long per build as the exact-key path (bench.py multi_k); it stays for inputs beyond
// 2^29 tokens, for the multi-GPU merge and behind AMG_KEY_MODE=fp.
template <class View>
__device__ __forceinline__ void x_fp94(const View& w, int k, int flip, int dir, unsigned long long seed, int weak,
                                       unsigned long long& w1, unsigned int& tag) {
  unsigned long long h1 = seed, h2 = seed ^ 0xC2B2AE3D27D4EB4Full;
  for (int j = 0; j < k; ++j) {
    const unsigned long long c = (unsigned long long)(unsigned int)canon_tok(w, k, flip, dir, j);
    h1 = (h1 ^ c) * 0x9E3779B97F4A7C15ull;
    h1 ^= h1 >> 29;
    h2 = (h2 + c) * 0xD6E8FEB86659FD93ull;
    h2 ^= h2 >> 31;
  }
  h1 = mix64(h1);
  h2 = mix64(h2 ^ (h1 >> 7));
  if (weak) {  // test hook: 12 bits
    h1 &= 0xfffull;
    h2 = 0ull;
  }
  w1 = (h1 << 1) | 1ull;
  tag = ((unsigned int)h2 << 1) | 1u;
}

// fingerprint keys: the tuple of every window against the tuple at its claim's first-seen position
__global__ __launch_bounds__(256) void k_x_verify_fp(const int* __restrict__ tokens, long long n_tokens, int k, int flip,
                                                      const int* __restrict__ tok_claim,
                                                      const signed char* __restrict__ tok_dir,
                                                      const unsigned int* __restrict__ first2,
                                                      unsigned long long* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tokens) return;
  const int raw = tok_claim[t];
  if (raw == -1) return;
  const unsigned int claim = (unsigned int)raw & ~AMG_FLAG_MASK;
  const unsigned int first = ~x_first_inv(first2, claim);
  const long long f = (long long)(first >> 1);
  if (f == t) return;  // the first occurrence itself
  const int dir = tok_dir[t], fdir = (first & 1u) ? -1 : 1;
  bool same = true;
  for (int j = 0; j < k; ++j) {
    const int a = dir > 0 ? tokens[t + j] : flip - tokens[t + k - 1 - j];
    const int b = fdir > 0 ? tokens[f + j] : flip - tokens[f + k - 1 - j];
    same = same && a == b;
  }
  if (!same) status[ST_COLLISION] = 1;  // benign race: every writer stores 1
}

// ------------------------------------------------------------------ nodes, four consecutive windows per thread
// Every gene-mer size (K = 0: k at run time): a thread's 4 + k - 1 tokens leave LDS in 128-bit reads, the
// common 16-bit packing shares half-words between the windows, the results leave as one 16-byte and one 4-byte
// store per thread, creators are counted with ballots, and the hit path (the key with its id in the first slot
// probed: nearly every window of a rebuild) is straight-line code.  Streams are non-temporal so that the L2s keep
// the table's hot lines (tools/ubench/pass_bench.hip: a pass of this shape is bound by the line requests of its
// probes, 0.2 ms per 56 M, plus its streams; whatever else it does has to hide behind those).
template <bool TWO, int K, bool B16, bool HEAD = false>  // HEAD: the short launch over the first tiles (a symbol of its own, as k_edges_v's)
__global__ __launch_bounds__(TILE_THREADS, 8) void k_nodes_v(
    const int* __restrict__ tokens, const unsigned int* __restrict__ bnd_bits, long long n_tokens, int k, int two_v,
    int bits, Slot16* tab, unsigned int mask, unsigned int probe_limit, int* __restrict__ tok_claim,
    signed char* __restrict__ tok_dir, unsigned long long* status, unsigned int* first2,
    unsigned int* __restrict__ slot_by_claim, unsigned int cap, XW2 xf, unsigned int tile0,
    unsigned long long* ctrs, unsigned int head_cap, unsigned long long fp_seed, int fp_weak) {
  // bits == 0 (K == 0 only): the key is a 94-bit fingerprint of the canonical tuple (x_fp94)
  typedef int i4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) int s_tok[TILE + AMG_MAX_K + 4];
  __shared__ unsigned int s_bits[TILE_BIT_WORDS];
  __shared__ unsigned int s_wave[TILE_THREADS / 64 + 1];
  const int tid = threadIdx.x;
  const long long t0 = (long long)(blockIdx.x + tile0) * TILE;
  // claims from the shard counters (XShard); the first tiles of the stream (the head launch and a few times as many
  // after it) share a counter of their own and take the ids below head_cap densely
  const XShard cshard{ctrs ? ((HEAD || (blockIdx.x + tile0) * (unsigned int)TILE < head_cap) ? (int)F_SHARDS
                                                                                              : (int)((blockIdx.x + tile0) & (F_SHARDS - 1u)))
                           : -1,
                      head_cap};
  const int flip = two_v - 1;
  {
    bool bad = false;
    if (t0 + TILE <= n_tokens && (reinterpret_cast<uintptr_t>(tokens) & 15) == 0) {  // (a borrowed array may sit anywhere)
      const i4 x = __builtin_nontemporal_load(reinterpret_cast<const i4*>(tokens + t0) + tid);
      bad = (unsigned int)x.x >= (unsigned int)two_v || (unsigned int)x.y >= (unsigned int)two_v ||
            (unsigned int)x.z >= (unsigned int)two_v || (unsigned int)x.w >= (unsigned int)two_v;
      reinterpret_cast<i4*>(s_tok)[tid] = x;
    } else {
      for (int i = tid; i < TILE; i += TILE_THREADS) {
        const long long t = t0 + i;
        const int x = t < n_tokens ? tokens[t] : 0;
        bad = bad || (unsigned int)x >= (unsigned int)two_v;
        s_tok[i] = x;
      }
    }
    if (tid < k + 3) {  // the k - 1 tokens the last windows reach into (+ padding read by the 128-bit loads)
      const long long t = t0 + TILE + tid;
      const int x = t < n_tokens ? tokens[t] : 0;
      bad = bad || (unsigned int)x >= (unsigned int)two_v;
      s_tok[TILE + tid] = x;
    }
    if (bad) status[ST_BADINPUT] = 2;  // a token outside [0, two_v) would alias another tuple
    if (tid < TILE_BIT_WORDS) s_bits[tid] = bnd_bits[(t0 >> 5) + tid];
  }
  __syncthreads();
  const int i0 = 4 * tid;
  unsigned int id1[TILE_ITEMS];
  unsigned int last = 0, ndir = 0, valid = 0;  // per window: last of its read; direction -1; has a node
  unsigned int made = 0;                       // per window: it created its node's key
  {
    unsigned long long w1[TILE_ITEMS];
    unsigned int idx[TILE_ITEMS], tag[TILE_ITEMS];
    ulonglong2 v[TILE_ITEMS];
    // windows whose k tokens lie in one read: no read ends at the positions t + 1 .. t + k - 1 (a read that ends
    // right after the window makes it the last of its read)
    const unsigned int b = tile_bits(s_bits, i0 + 1, k + 3);
    constexpr int NA = K > 0 ? 4 + K - 1 : 1;
    int a[NA];
    if constexpr (K > 0) {
#pragma unroll
      for (int j = 0; j < (NA + 3) / 4; ++j) {
        const int4 x = reinterpret_cast<const int4*>(s_tok + i0)[j];
        if (4 * j + 0 < NA) a[4 * j + 0] = x.x;
        if (4 * j + 1 < NA) a[4 * j + 1] = x.y;
        if (4 * j + 2 < NA) a[4 * j + 2] = x.z;
        if (4 * j + 3 < NA) a[4 * j + 3] = x.w;
      }
    }
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      w1[w] = 0;
      tag[w] = 0;
      idx[w] = 0;
      const long long t = t0 + i0 + w;
      const bool inside = ((b >> w) & ((1u << (k - 1)) - 1u)) == 0u;
      if (!((t + k <= n_tokens) && inside)) continue;
      int dir;
      if constexpr (K > 0 && B16) {
        dir = f_canon_pack16<K, TWO>(a + w, flip, w1[w], tag[w]);
      } else if constexpr (K > 0) {
        dir = x_canon_pack<K, TWO>(a + w, flip, bits, w1[w], tag[w]);
      } else {
        LdsView win{s_tok + i0 + w};
        dir = canon_dir(win, k, flip);
        if (dir != 0) {
          if (bits == 0)
            x_fp94(win, k, flip, dir, fp_seed, fp_weak, w1[w], tag[w]);
          else
            x_pack(win, k, flip, dir, bits, w1[w], tag[w]);
        }
      }
      if (dir == 0) {
        status[ST_PALINDROME] = 1;  // benign race: every writer stores 1
        continue;
      }
      idx[w] = (unsigned int)mix64(w1[w] ^ ((unsigned long long)tag[w] * 0x9E3779B97F4A7C15ull)) & mask;
      v[w] = *reinterpret_cast<const ulonglong2*>(tab + idx[w]);  // in flight while the next window is prepared
      if (dir < 0) ndir |= 1u << w;
      valid |= 1u << w;
      if ((b >> (w + k - 1)) & 1u) last |= 1u << w;
    }
    if constexpr (TWO)  // the slot belongs to whoever takes w1: two memory-side operations per creation instead of three
      f_table_phase_own<1>(tab, mask, valid, w1, tag, idx, v, (unsigned int)t0 + i0, ndir, xf, first2, slot_by_claim,
                           ctrs ? ctrs : status + ST_NODE_INSERTS, cap, probe_limit, status, 1, id1, s_wave, &made, 0u, 0u,
                           cshard);
    else
    f_table_phase<TWO, 1, false>(tab, mask, valid, w1, tag, idx, v, (unsigned int)t0 + i0, ndir, xf, first2,
                                 slot_by_claim, ctrs ? ctrs : status + ST_NODE_INSERTS, 0u, cap, probe_limit, status, 1,
                                 id1, s_wave, &made, 0u, 0u, 0u, 0u, cshard);
  }
  i4 oc;
  unsigned int od = 0;
  {
    int o[TILE_ITEMS];
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      o[w] = id1[w] ? (int)((id1[w] - 1u) | ((last & (1u << w)) ? AMG_LAST_FLAG : 0u) |
                            ((made & (1u << w)) ? AMG_MADE_FLAG : 0u))
                    : -1;
      od |= (id1[w] ? ((ndir & (1u << w)) ? 0xffu : 1u) : 0u) << (8 * w);
    }
    oc = i4{o[0], o[1], o[2], o[3]};
  }
  const long long t = t0 + i0;
  if (t + TILE_ITEMS <= n_tokens) {
    __builtin_nontemporal_store(oc, reinterpret_cast<i4*>(tok_claim + t));
    __builtin_nontemporal_store(od, reinterpret_cast<unsigned int*>(tok_dir + t));
  } else {
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w)
      if (t + w < n_tokens) {
        tok_claim[t + w] = oc[w];
        tok_dir[t + w] = (signed char)(od >> (8 * w));
      }
  }
}

// ------------------------------------------------------------------ nodes, minimiser buckets
// k_nodes_v is bound by the 128-byte line requests of its probes (one hashed slot = one line per window, DESIGN.md
// section 4).  Consecutive windows of a read share k - 1 genes, so a key function that consecutive windows mostly
// AGREE on turns neighbouring probes into requests for the same line: the gene of a window with the smallest rank
// (its minimiser; ranks follow sha256 order, i.e. they are a random permutation of the genes) is shared by about
// (k + 1) / 2 consecutive windows.  The table gets a BUCKET REGION in front of its hashed slots: one 128-byte line
// (8 slots) per gene rank.  A gene-mer lives in the line of its minimiser, starting at slot p0 = the position of
// that gene in the tuple read along the gene's own strand: the up to k gene-mers of a genome that share a
// minimiser g hold g at k different positions, so every one of them sits in its FIRST slot (p0 is a function of
// the canonical tuple alone: the key decides where it lives, whatever the orientation of the read).  Whatever
// does not find its slot free or its own (AMG_BUCKET_PROBES = 1 slot looked at; error gene-mers, genes in many copies) goes to its
// hashed slot as before.  A thread's four windows lie 256 apart, so the 64 lanes of one probe instruction look at
// 64 consecutive windows: ~64 * 2 / (k + 1) distinct lines instead of 64.
#define AMG_BUCKET_PROBES 1  // measured on cfg 3 (first build / rebuild, ms): 1: 0.92 / 0.35, 2: 0.97 / 0.36, 3: 1.00 / 0.37, 8: 1.16 / 0.43
template <bool TWO, int K, bool B16, bool HEAD = false>
__global__ __launch_bounds__(TILE_THREADS, 8) void k_nodes_m(
    const int* __restrict__ tokens, const unsigned int* __restrict__ bnd_bits, long long n_tokens, int two_v,
    int bits, Slot16* tab, unsigned int mask, unsigned int probe_limit, int* __restrict__ tok_claim,
    signed char* __restrict__ tok_dir, unsigned long long* status, unsigned int* first2,
    unsigned int* __restrict__ slot_by_claim, unsigned int cap, XW2 xf, unsigned int tile0, unsigned int home_n,
    unsigned long long* ctrs, unsigned int head_cap) {
  typedef int i4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) int s_tok[TILE + AMG_MAX_K + 4];
  __shared__ unsigned int s_bits[TILE_BIT_WORDS];
  __shared__ unsigned int s_wave[TILE_THREADS / 64 + 1];
  const int tid = threadIdx.x;
  const long long t0 = (long long)(blockIdx.x + tile0) * TILE;
  // claims from the shard counters (XShard); the first tiles of the stream (the head launch and a few times as many
  // after it) share a counter of their own and take the ids below head_cap densely
  const XShard cshard{ctrs ? ((HEAD || (blockIdx.x + tile0) * (unsigned int)TILE < head_cap) ? (int)F_SHARDS
                                                                                              : (int)((blockIdx.x + tile0) & (F_SHARDS - 1u)))
                           : -1,
                      head_cap};
  const int flip = two_v - 1, V = two_v >> 1;
  {
    bool bad = false;
    if (t0 + TILE <= n_tokens && (reinterpret_cast<uintptr_t>(tokens) & 15) == 0) {  // (a borrowed array may sit anywhere)
      const i4 x = __builtin_nontemporal_load(reinterpret_cast<const i4*>(tokens + t0) + tid);
      bad = (unsigned int)x.x >= (unsigned int)two_v || (unsigned int)x.y >= (unsigned int)two_v ||
            (unsigned int)x.z >= (unsigned int)two_v || (unsigned int)x.w >= (unsigned int)two_v;
      reinterpret_cast<i4*>(s_tok)[tid] = x;
    } else {
      for (int i = tid; i < TILE; i += TILE_THREADS) {
        const long long t = t0 + i;
        const int x = t < n_tokens ? tokens[t] : 0;
        bad = bad || (unsigned int)x >= (unsigned int)two_v;
        s_tok[i] = x;
      }
    }
    if (tid < K + 3) {  // the k - 1 tokens the last windows reach into
      const long long t = t0 + TILE + tid;
      const int x = t < n_tokens ? tokens[t] : 0;
      bad = bad || (unsigned int)x >= (unsigned int)two_v;
      s_tok[TILE + tid] = x;
    }
    if (bad) status[ST_BADINPUT] = 2;  // a token outside [0, two_v) would alias another tuple
    if (tid < TILE_BIT_WORDS) s_bits[tid] = bnd_bits[(t0 >> 5) + tid];
  }
  __syncthreads();
  unsigned int id1[TILE_ITEMS];
  unsigned int last = 0, ndir = 0, valid = 0;  // per window: last of its read; direction -1; has a node
  unsigned int made = 0;                       // per window: it created its node's key
  {
    unsigned long long w1[TILE_ITEMS];
    unsigned int idx[TILE_ITEMS], tag[TILE_ITEMS];
    ulonglong2 v[TILE_ITEMS];
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      w1[w] = 0;
      tag[w] = 0;
      idx[w] = 0;
      const int i = w * TILE_THREADS + tid;
      const long long t = t0 + i;
      // no read ends at the positions t + 1 .. t + k - 1; one that ends right after the window makes it the last of its read
      const unsigned int b = tile_bits(s_bits, i + 1, K);
      if (!((t + K <= n_tokens) && (b & ((1u << (K - 1)) - 1u)) == 0u)) continue;
      int a[K];
#pragma unroll
      for (int j = 0; j < K; ++j) a[j] = s_tok[i + j];
      int dir;
      if constexpr (B16)
        dir = f_canon_pack16<K, TWO>(a, flip, w1[w], tag[w]);
      else
        dir = x_canon_pack<K, TWO>(a, flip, bits, w1[w], tag[w]);
      if (dir == 0) {
        status[ST_PALINDROME] = 1;  // benign race: every writer stores 1
        continue;
      }
      // minimiser of the CANONICAL tuple (first occurrence of the smallest gene rank) and where it sits along its own strand
      int best = 0x7fffffff, bj = 0;
      bool plus = true;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int c = dir > 0 ? a[j] : flip - a[K - 1 - j];
        const bool p = c >= V;
        const int r = p ? c - V : V - 1 - c;
        if (r < best) {
          best = r;
          bj = j;
          plus = p;
        }
      }
      idx[w] = (unsigned int)best * 8u + (unsigned int)((plus ? bj : K - 1 - bj) & 7);
      v[w] = *reinterpret_cast<const ulonglong2*>(tab + idx[w]);  // in flight while the next window is prepared
      if (dir < 0) ndir |= 1u << w;
      valid |= 1u << w;
      if ((b >> (K - 1)) & 1u) last |= 1u << w;
    }
    if constexpr (TWO)
      f_table_phase_own<1, TILE_THREADS, AMG_BUCKET_PROBES>(tab, mask, valid, w1, tag, idx, v, (unsigned int)t0 + tid, ndir,
                                                            xf, first2, slot_by_claim, ctrs ? ctrs : status + ST_NODE_INSERTS,
                                                            cap, probe_limit, status, 1, id1, s_wave, &made, valid, home_n,
                                                            cshard);
    else
    f_table_phase<TWO, 1, false, TILE_THREADS, AMG_BUCKET_PROBES>(
        tab, mask, valid, w1, tag, idx, v, (unsigned int)t0 + tid, ndir, xf, first2, slot_by_claim,
        ctrs ? ctrs : status + ST_NODE_INSERTS, 0u, cap, probe_limit, status, 1, id1, s_wave, &made, valid, home_n, 0u, 0u,
        cshard);
  }
#pragma unroll
  for (int w = 0; w < TILE_ITEMS; ++w) {
    const long long t = t0 + w * TILE_THREADS + tid;
    if (t >= n_tokens) continue;
    const int o = id1[w] ? (int)((id1[w] - 1u) | ((last & (1u << w)) ? AMG_LAST_FLAG : 0u) |
                                 ((made & (1u << w)) ? AMG_MADE_FLAG : 0u))
                         : -1;
    __builtin_nontemporal_store(o, tok_claim + t);
    __builtin_nontemporal_store(id1[w] ? ((ndir & (1u << w)) ? (signed char)-1 : (signed char)1) : (signed char)0, tok_dir + t);
  }
}

// ---- build with the coverage filter applied on the way (amg_build_filtered): claims below the threshold are
// taken out of the claim space (first-seen words zeroed = "unclaimed", which every ranking kernel skips) before
// anything is ranked, so only the survivors get ids, arrays, edges
__global__ void k_x_drop_claims(const unsigned int* __restrict__ cnt, long long n, unsigned int min_cnt,
                                unsigned int* __restrict__ first2, int* __restrict__ final_of_claim,
                                unsigned long long* __restrict__ n_kept, unsigned int* __restrict__ first_all) {
  long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  if (c < n) {
    if (first_all) first_all[c] = x_first_inv(first2, c);  // component labels are those of the UNFILTERED graph
    keep = cnt[c] >= min_cnt && x_first_inv(first2, c) != 0u;
    if (!keep) {
      first2[2 * c] = 0u;
      first2[2 * c + 1] = 0u;
      if (final_of_claim) final_of_claim[c] = -2;  // windows of a dropped node read None (remove_node_from_reads)
    }
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_kept, (unsigned long long)__popcll(m));
}

// tagged != nullptr: claim -> node id | AMG_SINGLE_BIT where the node's coverage is 1 (what k_edges_v<.., LONE> reads)
__global__ void k_x_cov_from_claims(const unsigned int* __restrict__ cnt, const unsigned int* __restrict__ first2,
                                    const int* __restrict__ final_of_claim, long long n,
                                    unsigned int* __restrict__ node_cov, int* __restrict__ tagged) {
  long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int id = final_of_claim[c];
  const bool used = x_first_inv(first2, c) != 0u;
  if (tagged) tagged[c] = (used && id >= 0 && cnt[c] == 1u) ? (int)((unsigned int)id | AMG_SINGLE_BIT) : id;
  if (used) node_cov[id] = cnt[c];
}

// reads that lost a window to the filter join _readsToCorrect (remove_node_from_reads :442-461).  Sixteen lanes per
// read, four loads per lane in flight: a wave per read kept ONE 240-byte load in flight and ran at 1.2 TB/s.
#define FD_READS 16  // reads per 256-thread workgroup
__global__ __launch_bounds__(256) void k_x_flag_dead_reads(const int* __restrict__ tok_node,
                                                            const long long* __restrict__ read_off, long long n_reads,
                                                            unsigned char* __restrict__ read_fix) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const long long r = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;
  const bool have = r < n_reads;
  const long long a = have ? read_off[r] : 0, b = have ? read_off[r + 1] : 0;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = a + sub + 16 * j < b ? tok_node[a + sub + 16 * j] : 0;
  bool hit = v[0] == -2 || v[1] == -2 || v[2] == -2 || v[3] == -2;
  for (long long t = a + 64 + sub; t < b; t += 16) hit = hit || (tok_node[t] == -2);  // reads longer than 64 genes
  const unsigned long long m = __ballot(hit);
  if (have && sub == 0 && ((m >> (grp * 16)) & 0xffffull)) read_fix[r] = 1;
}

// ---- ranking by first-seen without a sort.  A token
// position opens at most one window / adjacency, so the first-seen token indices of the claims are
// distinct: set one bit per claim in a bitmap over the tokens, prefix-count the bitmap words, and
// the rank of a claim is the number of bits before its own.  (A radix sort of 0.5 - 1 M pairs
// costs ~10 launches and ~0.2 ms however small the input; this costs ~0.06 ms.)
// (one BYTE per token first, set with plain stores: 5.4 M scattered atomicOr on bitmap words ran at the
// memory-side atomic rate, 0.2 ms per ranking of the first build; the bytes are folded into the
// bitmap words by the pass that counts them)
__global__ void k_x_rank_setflags(const unsigned int* __restrict__ first2, long long n, int shift,
                                  unsigned char* __restrict__ flags) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int fi = x_first_inv(first2, i);
  if (fi == 0u) return;  // a claim id nobody took (interleaved shards, amg_build_f.hip)
  flags[(~fi) >> shift] = 1;
}

__device__ __forceinline__ long long x_rank_of(unsigned int t, const unsigned int* __restrict__ bits,
                                               const long long* __restrict__ prefix) {
  const unsigned int w = bits[t >> 5];
  return prefix[t >> 5] + (long long)__popc(w & ((1u << (t & 31)) - 1u));
}

__global__ void k_x_assign_nodes_ranked(const unsigned int* __restrict__ first2, long long n_nodes,
                                        const unsigned int* __restrict__ bits, const long long* __restrict__ prefix,
                                        const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                                        int k, int nbits, int two, int* __restrict__ final_of_claim,
                                        int* __restrict__ node_tokens, long long* __restrict__ node_first,
                                        unsigned char* __restrict__ node_alive, const int* __restrict__ tokens, int flip) {
  long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_nodes) return;
  if (x_first_inv(first2, c) == 0u) return;  // unclaimed
  const unsigned int first = ~x_first_inv(first2, c);
  const long long i = x_rank_of(first >> 1, bits, prefix);
  final_of_claim[c] = (int)i;
  node_first[i] = (long long)first;
  node_alive[i] = 1;
  if (nbits == 0) {  // fingerprint keys: the tuple is the window at the first-seen position, canonical by its direction bit
    const long long f = (long long)(first >> 1);
    for (int j = 0; j < k; ++j) node_tokens[i * k + j] = (first & 1u) ? flip - tokens[f + k - 1 - j] : tokens[f + j];
    return;
  }
  const Slot16 s = tab[slot_by_claim[c]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;  // one-word keys keep the creator's first-seen there
  for (int j = 0; j < k; ++j) node_tokens[i * k + j] = x_unpack(s.w1, tag, nbits, j);
}

// An edge class recorded by the fused table pass is keyed by the CLAIM ids of its two nodes; here,
// where the classes are ranked, key and orientation bit are re-expressed in final node ids:
// key = (sign, smaller id, larger id + 1), bit 0 of first-seen = "the first event ran smaller -> larger"
__device__ __forceinline__ void pair_to_final(unsigned long long& key, unsigned int& first,
                                              const int* __restrict__ fin) {
  const unsigned int lo_c = (unsigned int)((key >> 32) & 0x7fffffffull);
  const unsigned int hi_c = (unsigned int)(key & 0xffffffffull) - 1u;
  const bool a_is_lo = (first & 1u) != 0u;
  const unsigned int X = (unsigned int)fin[a_is_lo ? lo_c : hi_c], Y = (unsigned int)fin[a_is_lo ? hi_c : lo_c];
  const unsigned int lo = X < Y ? X : Y, hi = X < Y ? Y : X;
  key = (key & (1ull << 63)) | ((unsigned long long)lo << 32) | (unsigned long long)(hi + 1u);
  first = (first & ~1u) | (X == lo ? 1u : 0u);
}

__global__ void k_x_gather_pairs_ranked(const unsigned int* __restrict__ first2, long long n_pairs,
                                        const unsigned int* __restrict__ bits, const long long* __restrict__ prefix,
                                        const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                                        const unsigned int* __restrict__ cnt_by_claim,
                                        unsigned long long* __restrict__ pkey, unsigned long long* __restrict__ pfirst,
                                        unsigned int* __restrict__ pcnt, const int* __restrict__ fin,
                                        int* __restrict__ efinal) {
  long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_pairs) return;
  if (x_first_inv(first2, c) == 0u) return;  // unclaimed
  unsigned int first = ~x_first_inv(first2, c);
  const long long i = x_rank_of(first >> 3, bits, prefix);
  unsigned long long key = etab[slot_by_claim[c]].w1;
  if (fin) pair_to_final(key, first, fin);
  pkey[i] = key;
  pfirst[i] = (unsigned long long)first;
  if (efinal)
    efinal[c] = (int)i;  // the occurrences are counted afterwards, per class id (bf_finish)
  else
    pcnt[i] = cnt_by_claim[c];
}

__global__ void k_x_sort_keys(const unsigned int* __restrict__ first2, long long n,
                              unsigned int* __restrict__ keys, unsigned int* __restrict__ vals) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = ~x_first_inv(first2, i);
  vals[i] = (unsigned int)i;
}

__global__ void k_x_assign_nodes(const unsigned int* __restrict__ first_sorted,
                                 const unsigned int* __restrict__ claim_sorted, long long n_nodes,
                                 const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                                 int k, int bits, int two, int* __restrict__ final_of_claim,
                                 int* __restrict__ node_tokens, long long* __restrict__ node_first,
                                 unsigned char* __restrict__ node_alive, const int* __restrict__ tokens, int flip) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const unsigned int c = claim_sorted[i];
  final_of_claim[c] = (int)i;
  node_first[i] = (long long)first_sorted[i];
  node_alive[i] = 1;
  if (bits == 0) {  // fingerprint keys (see k_x_assign_nodes_ranked)
    const unsigned int first = first_sorted[i];
    const long long f = (long long)(first >> 1);
    for (int j = 0; j < k; ++j) node_tokens[i * k + j] = (first & 1u) ? flip - tokens[f + k - 1 - j] : tokens[f + j];
    return;
  }
  const Slot16 s = tab[slot_by_claim[c]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;  // one-word keys keep the creator's first-seen there
  for (int j = 0; j < k; ++j) node_tokens[i * k + j] = x_unpack(s.w1, tag, bits, j);
}

// ------------------------------------------------------------------ components of a filtered build
// The reference labels components once, in __init__, on the graph of ALL nodes (construct_graph.py:101-102,
// :911-927) and keeps the labels through filter_graph.  A filtered build never makes the nodes and edges
// below the threshold, so the labels are made in CLAIM space instead, from what the node pass left behind:
// every window's claim (tok_slot) — two consecutive windows of a read are an edge of the unfiltered graph —
// and every claim's first-seen value (saved by k_x_drop_claims).  Union-find over the claims; component id =
// 1 + rank of the component's earliest first-seen among the components (= DFS discovery order over _nodes).
__global__ void k_xc_init(int* parent, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = (int)i;
}

__global__ void k_xc_union(const int* __restrict__ tok_claim, long long n_tokens, int* parent) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t + 1 >= n_tokens) return;
  const int raw = tok_claim[t];
  if (raw == -1 || ((unsigned int)raw & AMG_LAST_FLAG)) return;
  const int nxt = tok_claim[t + 1];
  if (nxt == -1) return;
  int a = (int)((unsigned int)raw & ~AMG_FLAG_MASK), b = (int)((unsigned int)nxt & ~AMG_FLAG_MASK);
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) break;
    if (a > b) { int x = a; a = b; b = x; }
    const int old = atomicCAS(parent + b, b, a);  // hook the larger root under the smaller
    if (old == b) break;
    b = old;
  }
}

__global__ void k_xc_flatten(int* parent, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = uf_find(parent, (int)i);  // only thread i writes entry i; roots keep parent == self
}

__device__ __forceinline__ int xc_root(const int* __restrict__ parent, long long c) {
  int r = parent[c];
  while (parent[r] != r) r = parent[r];  // entries of non-roots may still name an intermediate ancestor
  return r;
}

// best2[2 r] = the largest complemented first-seen (= the earliest) among the claims of root r; roots counted
__global__ void k_xc_best(const int* __restrict__ parent, const unsigned int* __restrict__ first_all, long long n,
                          unsigned int* __restrict__ best2, unsigned long long* __restrict__ n_roots) {
  long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool root = false;
  if (c < n && first_all[c] != 0u) {
    const int r = xc_root(parent, c);
    root = r == (int)c;
    atomicMax(best2 + 2ll * r, first_all[c]);
  }
  const unsigned long long m = __ballot(root);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_roots, (unsigned long long)__popcll(m));
}

__global__ void k_xc_label(const int* __restrict__ parent, const unsigned int* __restrict__ first_all, long long n,
                           const unsigned int* __restrict__ best2, const unsigned int* __restrict__ bits,
                           const long long* __restrict__ prefix, const int* __restrict__ final_of_claim,
                           int* __restrict__ node_comp) {
  long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n || first_all[c] == 0u) return;
  const int node = final_of_claim[c];
  if (node < 0) return;  // the filter dropped this node
  const unsigned int first = ~best2[2ll * xc_root(parent, c)];
  node_comp[node] = (int)x_rank_of(first >> 1, bits, prefix) + 1;
}

// ------------------------------------------------------------------ edges, four adjacencies per thread
// HOME = 0: a thread's four adjacencies are consecutive (one LDS read of its neighbours' words, one 16-byte store of its
// results), every class lives where its key hashes to.
// HOME > 0 (`home_n` slots in front of the hashed ones): node ids are first-seen ranks, so nine adjacencies in ten join
// ids n and n + 1 — their class lives in slot n, addressed directly (a second class of the same two nodes, or any other
// pair, goes to the hashed slots behind), and a thread's adjacencies lie 256 windows apart so that the 64 lanes of a
// probe instruction look at 64 consecutive windows: consecutive ids, eight slots to a 128-byte line instead of one
// line per lane (the passes are bound by the line requests of their probes).
// LONE (with HOME): `final_of_claim` carries AMG_SINGLE_BIT on the nodes of coverage 1 (k_x_cov_from_claims).  Every
// adjacency of such a node lies next to its one window, so a class with a single end occurs once — or twice when
// windows t - 1 and t + 1 of the single window t are the same node in the same direction (a period-2 stretch), which
// the two neighbouring words show.  A class known to occur once needs no table: f_table_phase's lone items.
template <bool HEAD, bool HOME, bool LONE = false>  // HEAD: the short launch over the first tiles (its own symbol: per-kernel statistics keep the main launch apart)
__global__ __launch_bounds__(TILE_THREADS, 8) void k_edges_v(
    long long n_tokens, const int* __restrict__ tok_claim, const signed char* __restrict__ tok_dir,
    const int* __restrict__ final_of_claim, int* __restrict__ tok_node, Slot16* etab, unsigned int emask,
    unsigned int probe_limit, unsigned long long* status, int* __restrict__ tok_pair, unsigned int* first2,
    unsigned int* __restrict__ slot_by_claim, unsigned int cap, XW2 xf, unsigned int tile0, unsigned int home_n,
    unsigned int lone_base, unsigned long long* ctrs, unsigned int head_cap) {
  static_assert(!LONE || HOME, "lone classes come with the home-slot layout");
  typedef int i4 __attribute__((ext_vector_type(4)));
  // word per window: node id | single << 29 (LONE) | (direction -1) << 30 | last-of-read << 31, -1: no node
  constexpr unsigned int DIRBIT = 0x40000000u;
  constexpr unsigned int IDMASK = LONE ? AMG_SINGLE_BIT - 1u : DIRBIT - 1u;
  __shared__ __attribute__((aligned(16))) int s_w[TILE + 4];
  __shared__ unsigned int s_wave[TILE_THREADS / 64 + 2];
  const int tid = threadIdx.x;
  const long long t0 = (long long)(blockIdx.x + tile0) * TILE;
  // claims from the shard counters (XShard); the first tiles of the stream (the head launch and a few times as many
  // after it) share a counter of their own and take the ids below head_cap densely
  const XShard cshard{ctrs ? ((HEAD || (blockIdx.x + tile0) * (unsigned int)TILE < head_cap) ? (int)F_SHARDS
                                                                                              : (int)((blockIdx.x + tile0) & (F_SHARDS - 1u)))
                           : -1,
                      head_cap};
  const int i0 = 4 * tid;
  const long long t = t0 + i0;
  // node id of a window: -1 no node, -2 a node the merge's fused filter dropped (amg_dist.hip)
  auto node_of = [&](int raw) { return raw == -1 ? -1 : final_of_claim[(unsigned int)raw & ~AMG_FLAG_MASK]; };
  auto word = [&](int raw, int id, unsigned int d) {  // d: the direction byte
    if (id < 0) return -1;
    return (int)((unsigned int)id | ((unsigned int)raw & AMG_LAST_FLAG) | ((d & 0x80u) ? DIRBIT : 0u));
  };
  auto word_at = [&](long long tw) {  // a window outside the tile
    if (tw < 0 || tw >= n_tokens) return -1;
    const int raw = tok_claim[tw];
    return word(raw, node_of(raw), (unsigned int)(unsigned char)tok_dir[tw]);
  };
  int cw[TILE_ITEMS + 1];
  {
    i4 x = {-1, -1, -1, -1};
    unsigned int d = 0;
    if (t + TILE_ITEMS <= n_tokens) {
      x = __builtin_nontemporal_load(reinterpret_cast<const i4*>(tok_claim + t));
      d = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(tok_dir + t));
    } else {
#pragma unroll
      for (int w = 0; w < TILE_ITEMS; ++w)
        if (t + w < n_tokens) {
          x[w] = tok_claim[t + w];
          d |= (unsigned int)(unsigned char)tok_dir[t + w] << (8 * w);
        }
    }
    // node id per window (construct_read.py get_geneMers order), as every later stage wants it
    i4 ids;
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      const int id = node_of(x[w]);
      cw[w] = word(x[w], id, d >> (8 * w));
      ids[w] = (LONE && id >= 0) ? (int)((unsigned int)id & ~AMG_SINGLE_BIT) : id;
    }
    if (t + TILE_ITEMS <= n_tokens) {
      __builtin_nontemporal_store(ids, reinterpret_cast<i4*>(tok_node + t));
    } else {
#pragma unroll
      for (int w = 0; w < TILE_ITEMS; ++w)
        if (t + w < n_tokens) tok_node[t + w] = ids[w];
    }
  }
  reinterpret_cast<int4*>(s_w)[tid] = make_int4(cw[0], cw[1], cw[2], cw[3]);
  if (tid == 0) s_w[TILE] = word_at(t0 + TILE);  // the next tile's first window: right-hand neighbour of this tile's last
  if constexpr (LONE) {                          // and the two windows the period-2 check of the tile's ends looks at
    if (tid == 64) s_w[TILE + 1] = word_at(t0 + TILE + 1);
    if (tid == 128) s_w[TILE + 2] = word_at(t0 - 1);
  }
  __syncthreads();
  if constexpr (!HOME) cw[TILE_ITEMS] = s_w[i0 + TILE_ITEMS];

  // adjacency (A, dA) -> (B, dB) of windows t and t + 1 of one read (create_edges :246-262); class key =
  // (smaller id, larger id, dA * dB), first-seen = (token << 3) | orientation
  unsigned long long key[TILE_ITEMS];
  unsigned int idx[TILE_ITEMS], etag[TILE_ITEMS], id1[TILE_ITEMS];
  ulonglong2 v[TILE_ITEMS];
  unsigned int valid = 0, orient3 = 0, homed = 0, lone = 0;
#pragma unroll
  for (int w = 0; w < TILE_ITEMS; ++w) {
    key[w] = 0;
    idx[w] = 0;
    etag[w] = 0;
    const int A = HOME ? s_w[w * TILE_THREADS + tid] : cw[w], B = HOME ? s_w[w * TILE_THREADS + tid + 1] : cw[w + 1];
    if (A == -1 || ((unsigned int)A & AMG_LAST_FLAG) || B == -1) continue;
    const unsigned int a = (unsigned int)A & IDMASK, b = (unsigned int)B & IDMASK;
    const bool negA = ((unsigned int)A & DIRBIT) != 0u, negB = ((unsigned int)B & DIRBIT) != 0u;
    const unsigned int lo = a < b ? a : b, hi = a < b ? b : a;
    const unsigned long long sign = negA != negB ? 1ull : 0ull;
    key[w] = (sign << 63) | ((unsigned long long)lo << 32) | (unsigned long long)(hi + 1u);
    const unsigned int orient = (a == lo ? 1u : 0u) | (negA ? 0u : 2u) | (negB ? 0u : 4u);
    orient3 |= orient << (3 * w);
    valid |= 1u << w;
    if constexpr (LONE) {
      if (((unsigned int)A | (unsigned int)B) & AMG_SINGLE_BIT) {
        const int i = w * TILE_THREADS + tid;
        const int Wm = i == 0 ? s_w[TILE + 2] : s_w[i - 1], Wn = s_w[i + 2];
        // the other adjacency of a single node is the same class: same neighbour node, same direction
        const bool dup_prev = Wm != -1 && !((unsigned int)Wm & AMG_LAST_FLAG) &&
                              (((unsigned int)Wm ^ (unsigned int)B) & (IDMASK | DIRBIT)) == 0u;
        const bool dup_next = !((unsigned int)B & AMG_LAST_FLAG) && Wn != -1 &&
                              (((unsigned int)A ^ (unsigned int)Wn) & (IDMASK | DIRBIT)) == 0u;
        if ((((unsigned int)A & AMG_SINGLE_BIT) && !dup_prev) || (((unsigned int)B & AMG_SINGLE_BIT) && !dup_next)) {
          lone |= 1u << w;
          continue;
        }
      }
    }
    if (HOME && hi == lo + 1u && lo < home_n) {
      idx[w] = lo;
      homed |= 1u << w;
    } else {
      idx[w] = (HOME ? home_n : 0u) + ((unsigned int)mix64(key[w]) & emask);
    }
    v[w] = *reinterpret_cast<const ulonglong2*>(etab + idx[w]);
  }
  unsigned int made = 0;
  if constexpr (HOME)
    f_table_phase<false, 3, false, TILE_THREADS, 1, LONE>(etab, emask, valid, key, etag, idx, v, (unsigned int)t0 + tid,
                                                          orient3, xf, first2, slot_by_claim,
                                                          ctrs ? ctrs : status + ST_PAIR_INSERTS, 0u, cap, probe_limit, status,
                                                          2, id1, s_wave, &made, homed, home_n, lone, lone_base, cshard,
                                                          ctrs ? (int)((blockIdx.x + tile0) & (F_SHARDS - 1u)) : -1);
  else
    f_table_phase<false, 3, false>(etab, emask, valid, key, etag, idx, v, (unsigned int)t0 + i0, orient3, xf, first2,
                                   slot_by_claim, ctrs ? ctrs : status + ST_PAIR_INSERTS, 0u, cap, probe_limit, status, 2,
                                   id1, s_wave, &made, 0u, 0u, 0u, 0u, cshard);
  if constexpr (HOME) {
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      const long long tw = t0 + w * TILE_THREADS + tid;
      if (tw < n_tokens)
        __builtin_nontemporal_store(id1[w] ? (int)((id1[w] - 1u) | ((made & (1u << w)) ? AMG_MADE_FLAG : 0u)) : -1,
                                    tok_pair + tw);
    }
  } else {
    i4 op;
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w)
      op[w] = id1[w] ? (int)((id1[w] - 1u) | ((made & (1u << w)) ? AMG_MADE_FLAG : 0u)) : -1;
    if (t + TILE_ITEMS <= n_tokens) {
      __builtin_nontemporal_store(op, reinterpret_cast<i4*>(tok_pair + t));
    } else {
#pragma unroll
      for (int w = 0; w < TILE_ITEMS; ++w)
        if (t + w < n_tokens) tok_pair[t + w] = op[w];
    }
  }
}

// edge classes in first-seen order: key, first, count (counted per claim by k_count_ids)
__global__ void k_x_gather_pairs(const unsigned int* __restrict__ first_sorted,
                                 const unsigned int* __restrict__ claim_sorted, long long n_pairs,
                                 const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                                 const unsigned int* __restrict__ cnt_by_claim,
                                 unsigned long long* __restrict__ pkey, unsigned long long* __restrict__ pfirst,
                                 unsigned int* __restrict__ pcnt, const int* __restrict__ fin) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const unsigned int c = claim_sorted[i];
  unsigned long long key = etab[slot_by_claim[c]].w1;
  unsigned int first = first_sorted[i];
  if (fin) pair_to_final(key, first, fin);
  pkey[i] = key;
  pfirst[i] = (unsigned long long)first;
  pcnt[i] = cnt_by_claim[c];
}

#define X_CTRS (F_SHARDS + 1)  // counters of a plain pass: F_SHARDS shards + the first tiles'

// ------------------------------------------------------------------ host side
static inline unsigned int blocks_for(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

static const unsigned int kProbeLimitX = 1024;

// tiles of the head launch of a table pass (0: none: small inputs): a few coverages of a genome of at most
// two_v / 2 genes, never more than 1/16 of the tiles.  Every build of a large input gets one, rebuilds with few keys
// included: the keys that almost every later window hits then hold the lowest claims in stream order, which is what
// the counting sweeps and the rank bitmaps like (measured on the cleaning sweep: 8.8 -> 8.5 ms against head launches
// for first builds only; 48 / 64 / 96 / 128 / 156 / 256 / 512 tiles: 9.27 / 9.02 / 8.69 / 8.62 / 8.46 / 8.70 / 8.84 ms).
static long long head_tiles(const amg_ctx* c, long long n_tiles) {
  const char* e = getenv("AMG_X_HEAD_TILES");  // A/B switch
  if (e) return atoll(e) < n_tiles ? atoll(e) : n_tiles;
  if (n_tiles < 4096) return 0;
  long long h = 4ll * c->two_v / TILE;
  if (h < 64) h = 64;
  if (h > n_tiles / 16) h = n_tiles / 16;
  return h;
}
static const long long kRankBitmapMax = 16ll << 20;  // claims up to which the bitmap ranking is used (5.4 M: 0.38 vs 0.44 ms;
                                                    // 0.5 M: 0.09 vs 0.16 ms); the radix sort beyond (AMG_X_RANK_SORT=1 forces it)

// bitmap over the tokens (s1) with one bit per claim + exclusive prefix of the word popcounts (s5)
static int x_rank_bitmap(amg_ctx* c, const unsigned int* first2, long long n, int shift) {
  hipStream_t st = c->stream;
  const long long words = (c->n_tokens >> 5) + 2;
  AMGCHK(c->s1.ensure((size_t)words * sizeof(unsigned int)));
  AMGCHK(c->s5.ensure((size_t)(words + 2) * sizeof(long long)));
  AMGCHK(c->s0.ensure((size_t)words * 32 + 64));
  if (c->rank_flags_clean != words) {  // (else: zeroed behind the table pass's read-back, read_status)
    ClearList cl;
    cl.add(c->s0.p, (size_t)words * 32);
    AMGCHK(clear_many(c, cl));
  }
  c->rank_flags_clean = 0;
  hipLaunchKernelGGL(k_x_rank_setflags, dim3(blocks_for(n, 256)), dim3(256), 0, st, first2, n, shift,
                     c->s0.as<unsigned char>());
  // the flag bytes are folded into the bitmap words by the scan that counts them (one launch, no count array)
  return prim_exscan_flag_words(c, c->s0.as<unsigned char>(), c->s1.as<unsigned int>(), c->s5.as<long long>(), (size_t)words);
}

// the status words after a table pass; ctrs != nullptr: the pass took its claims from the X_CTRS counters there —
// their sum replaces host[count_word], the fullest shard's count goes to *most.  rank_filler: the flag bytes of the
// ranking bitmap that follows (x_rank_bitmap) are zeroed behind the read-back kernel, while the host waits
static int read_status(amg_ctx* c, unsigned long long* host, const unsigned long long* ctrs = nullptr, int count_word = 0,
                       unsigned long long* most = nullptr, bool rank_filler = false) {
  FetchList l;
  l.add_words(c->status.p, ST_WORDS);
  if (ctrs)
    for (int i = 0; i < X_CTRS; ++i) l.add(ctrs + (size_t)i * F_CTR_STRIDE);
  unsigned long long v[ST_WORDS + X_CTRS];
  ClearList fill;
  const long long words = (c->n_tokens >> 5) + 2;
  if (rank_filler) {
    AMGCHK(c->s0.ensure((size_t)words * 32 + 64));
    fill.add(c->s0.p, (size_t)words * 32);
  }
  AMGCHK(fetch(c, l, v, rank_filler ? &fill : nullptr));
  if (rank_filler) c->rank_flags_clean = words;
  for (int i = 0; i < ST_WORDS; ++i) host[i] = v[i];
  if (ctrs) {
    unsigned long long sum = 0, mx = 0;
    for (int i = 0; i < X_CTRS; ++i) {
      sum += v[ST_WORDS + i];
      if (i < (int)F_SHARDS && v[ST_WORDS + i] > mx) mx = v[ST_WORDS + i];
    }
    host[count_word] = sum;
    if (most) *most = mx;
  }
  return AMG_OK;
}

// claims from shard counters (XShard) for the inputs that get a head launch (4 M tokens and more: below that one
// counter serves a pass's workgroups in the time the pass takes anyway); AMG_CLAIM_SHARDS = 0 / 1: never / always
// (A/B + test switch)
static bool shard_claims(long long n_tiles) {
  if (const char* e = getenv("AMG_CLAIM_SHARDS")) return atoi(e) != 0 && n_tiles > 0;
  return n_tiles >= 4096;
}
// a shard's share of `bound` claims: an even split, a quarter of slack, what one workgroup creates in one go; whole chunks
static unsigned int shard_share(long long bound) {
  const long long even = (bound + F_SHARDS - 1) / F_SHARDS;
  const long long share = even + even / 4 + TILE + 1;
  return (unsigned int)((share + X_CHUNK - 1) / X_CHUNK * X_CHUNK);
}
// tiles at the start of the stream that take their claims densely from ONE counter: the head launch's and three times
// as many after it.  A genome key the head launch has not seen (a dozen of 20 k on cfg 3) is created by one of the
// next tiles; from a shard counter its claim would lie far above the ids the counting sweeps keep in LDS, and its
// thousands of occurrences would each be a global atomic on one word (measured: 8 k such windows, +55 us per count).
static long long dense_tiles(const amg_ctx* c, long long n_tiles) {
  const long long d = 4 * head_tiles(c, n_tiles);
  return d < n_tiles ? d : n_tiles;
}
// claim ids in use lie below this when the fullest shard handed out `most` (head_cap ids in front: the first tiles')
static long long shard_space(unsigned long long most, long long head_cap, size_t max_claims) {
  const long long s = head_cap + (long long)((most + X_CHUNK - 1) / X_CHUNK) * (long long)(X_CHUNK * F_SHARDS);
  return s < (long long)max_claims ? s : (long long)max_claims;
}

// bits per token of the packed tuple: 16 whenever that fits (constant shifts in the kernels); AMG_X_TIGHT_BITS=1:
// as few as the vocabulary needs (test switch: the kernels' general packing, which large vocabularies take)
int bx_bits(const amg_ctx* c, int k) {
  const int need = ilog2_ceil((uint64_t)(c->two_v > 2 ? c->two_v : 2));
  if (need <= 16 && (long long)k * 16 <= 94 && !getenv("AMG_X_TIGHT_BITS")) return 16;
  return need;
}

bool bx_applicable(const amg_ctx* c, int k) {
  if (c->dist_mode || c->count_inline) return false;
  return bx_fits(c, k);
}

// the tuple fits the slot and first-seen fits 32 bits (per shard in a merged build)
// the canonical tuple itself fits the 94 key bits of a slot
bool bx_tuple_fits(const amg_ctx* c, int k) {
  if (c->weak_fp_builds > 0) return false;
  const char* e = getenv("AMG_KEY_MODE");  // A/B + test switch: "fp" forces the 32-byte fingerprint path
  if (e && e[0] == 'f') return false;
  if ((long long)k * bx_bits(c, k) > 94) return false;
  if (c->n_tokens >= (1ll << 29)) return false;  // 32-bit first-seen: (token << 3) | orientation
  return true;
}

// the 16-byte-slot build applies: the tuple fits, or its 94-bit fingerprint is the key (x_fp94)
bool bx_fits(const amg_ctx* c, int k) {
  if (bx_tuple_fits(c, k)) return true;
  const char* e = getenv("AMG_KEY_MODE");
  if (e && e[0] == 'f') return false;
  if (c->n_tokens >= (1ll << 29)) return false;
  // (a tuple that WOULD fit, with the weak-fingerprint test hook set, keeps exercising the 32-byte path)
  return (long long)k * bx_bits(c, k) > 94;
}

// windows -> node table, claim ids, node ids, node arrays.  AMG_E_OVERFLOW + *which = 1: table full
int bx_nodes(amg_ctx* c, int k, int* which) {
  AMGCHK(bx_nodes_upsert(c, k, which, true));
  return bx_nodes_rank(c);
}

// the table pass alone: claims 0 .. n_local_nodes-1 with their first-seen / slot arrays
// sharded: claim ids may come from shard counters (then x_nspace > n_nodes: ids nobody took in between)
// rank_follows: the caller ranks the claims next (x_rank_bitmap): its flag bytes are zeroed behind the status read-back
int bx_nodes_upsert(amg_ctx* c, int k, int* which, bool sharded, bool rank_follows) {
  *which = 0;
  hipStream_t st = c->stream;
  const long long T = c->n_tokens;
  unsigned long long hs[ST_WORDS];
  c->exact_keys = true;
  c->packed_nodes = false;
  c->x_bits = bx_bits(c, k);
  c->x_fp = (long long)k * c->x_bits > 94;  // the tuple does not fit the slot: its 94-bit fingerprint is the key (x_fp94)
  const long long n_tiles = (T + TILE - 1) / TILE;

  // bucket region of the node table (k_nodes_m): one 8-slot line per gene rank in front of the hashed slots, for the
  // gene-mer sizes that have a compile-time kernel; AMG_NODE_BUCKETS=0: hashed slots only (k_nodes_v; A/B switch)
  const char* nb = getenv("AMG_NODE_BUCKETS");
  const bool buckets = !(nb && atoi(nb) == 0) && !getenv("AMG_X_GENERIC_K") && !c->x_fp &&
                       (k == 3 || k == 5 || k == 7) && n_tiles > 0;
  const size_t home_n = buckets ? (size_t)4 * (size_t)c->two_v : 0;  // 8 slots x (two_v / 2) gene ranks
  const size_t tab_slots = (size_t)c->node_slots + home_n;
  const long long claim_bound = ((long long)tab_slots < T ? (long long)tab_slots : T) + 1;
  const bool plain = sharded && rank_follows;  // the caller is the plain build: ranking follows the read-back directly
  c->rank_flags_clean = 0;
  sharded = sharded && shard_claims(n_tiles);
  const unsigned int cap = sharded ? shard_share(claim_bound) : (unsigned int)claim_bound;  // per counter
  const long long head_cap = sharded ? dense_tiles(c, n_tiles) * TILE : 0;                  // ids of the first tiles
  const size_t max_claims = sharded ? (size_t)head_cap + (size_t)cap * F_SHARDS + 1 : (size_t)claim_bound;
  unsigned long long* ctrs = nullptr;
  if (sharded) {
    AMGCHK(c->f_ctrs.ensure(2 * X_CTRS * F_CTR_STRIDE * sizeof(unsigned long long)));
    ctrs = c->f_ctrs.as<unsigned long long>();
  }
  AMGCHK(c->tok_slot.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->tok_node.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->tok_dir.ensure((size_t)(T + 8)));
  AMGCHK(c->node_tab.ensure(tab_slots * sizeof(Slot16)));
  AMGCHK(c->x_first.ensure(2 * max_claims * sizeof(unsigned int)));  // {raised by others, creator's} per claim
  AMGCHK(c->x_slot.ensure(max_claims * sizeof(unsigned int)));
  AMGCHK(c->x_final.ensure(max_claims * sizeof(int)));

  {  // status words, table, first-seen words and the read-end bitmap are zeroed by ONE launch
    ClearList cl;
    cl.add(c->status.p, ST_WORDS * sizeof(unsigned long long));
    cl.add(c->node_tab.p, tab_slots * sizeof(Slot16));
    cl.add(c->x_first.p, 2 * max_claims * sizeof(unsigned int));
    if (ctrs) cl.add(ctrs, X_CTRS * F_CTR_STRIDE * sizeof(unsigned long long));
    AMGCHK(bs_read_stats(c, k, &cl));
  }

  stage_begin(c, (n_tiles > 0 && head_tiles(c, n_tiles) > 0) ? "node_upsert_head" : "node_upsert");
  if (n_tiles > 0) {
    const bool two = c->x_fp || (long long)k * c->x_bits > 63;  // tuple spills into w2?
    if (buckets) {
      const bool b16 = c->x_bits == 16 && (k == 3 || k == 5);
      auto kern = k_nodes_m<false, 3, false>;
      auto kern_head = k_nodes_m<false, 3, false, true>;
      if (b16 && k == 3) kern = k_nodes_m<false, 3, true>, kern_head = k_nodes_m<false, 3, true, true>;
      else if (b16 && k == 5) kern = k_nodes_m<true, 5, true>, kern_head = k_nodes_m<true, 5, true, true>;
      else if (k == 3) kern = two ? k_nodes_m<true, 3, false> : k_nodes_m<false, 3, false>,
                       kern_head = two ? k_nodes_m<true, 3, false, true> : k_nodes_m<false, 3, false, true>;
      else if (k == 5) kern = two ? k_nodes_m<true, 5, false> : k_nodes_m<false, 5, false>,
                       kern_head = two ? k_nodes_m<true, 5, false, true> : k_nodes_m<false, 5, false, true>;
      else kern = two ? k_nodes_m<true, 7, false> : k_nodes_m<false, 7, false>,
           kern_head = two ? k_nodes_m<true, 7, false, true> : k_nodes_m<false, 7, false, true>;
      const long long head = head_tiles(c, n_tiles);  // (see below: the genome's keys get the lowest claims — and the first slots of their lines)
      for (int part = 0; part < 2; ++part) {
        const long long lo = part == 0 ? 0 : head, cnt = part == 0 ? head : n_tiles - head;
        if (cnt <= 0) continue;
        if (part == 1 && head > 0) {  // the head launch is a stage of its own
          stage_end(c);
          stage_begin(c, "node_upsert");
        }
        hipLaunchKernelGGL(part == 0 ? kern_head : kern, dim3((unsigned)cnt), dim3(TILE_THREADS), 0, st, c->tokens.as<int>(),
                           c->bnd_bits.as<unsigned int>(), T, c->two_v, c->x_bits, c->node_tab.as<Slot16>(),
                           (unsigned int)(c->node_slots - 1), kProbeLimitX, c->tok_slot.as<int>(),
                           c->tok_dir.as<signed char>(), c->status.as<unsigned long long>(),
                           c->x_first.as<unsigned int>(), c->x_slot.as<unsigned int>(), cap,
                           xw2_for(max_claims, T), (unsigned int)lo, (unsigned int)home_n, ctrs, (unsigned int)head_cap);
      }
    } else {
      const bool b16 = c->x_bits == 16 && (k == 3 || k == 5);
      auto kern = two ? k_nodes_v<true, 0, false> : k_nodes_v<false, 0, false>;
      auto kern_head = two ? k_nodes_v<true, 0, false, true> : k_nodes_v<false, 0, false, true>;
      if (!getenv("AMG_X_GENERIC_K") && !c->x_fp) {  // A/B switch
        if (b16 && k == 3) kern = k_nodes_v<false, 3, true>, kern_head = k_nodes_v<false, 3, true, true>;
        else if (b16 && k == 5) kern = k_nodes_v<true, 5, true>, kern_head = k_nodes_v<true, 5, true, true>;
        else if (k == 3) kern = two ? k_nodes_v<true, 3, false> : k_nodes_v<false, 3, false>,
                         kern_head = two ? k_nodes_v<true, 3, false, true> : k_nodes_v<false, 3, false, true>;
        else if (k == 5) kern = two ? k_nodes_v<true, 5, false> : k_nodes_v<false, 5, false>,
                         kern_head = two ? k_nodes_v<true, 5, false, true> : k_nodes_v<false, 5, false, true>;
        else if (k == 7) kern = two ? k_nodes_v<true, 7, false> : k_nodes_v<false, 7, false>,
                         kern_head = two ? k_nodes_v<true, 7, false, true> : k_nodes_v<false, 7, false, true>;
      }
      // Claim ids follow the order in which the ~2000 concurrently running tiles create keys: with many creations
      // per tile (a first build: one window in ten) the genome's keys, which almost every later window hits,
      // get claims scattered over the first few hundred thousand, and whoever counts by claim (k_count_ids, one
      // LDS range of 32 k ids per sweep) needs several sweeps.  A short head launch over the first few genome
      // coverages creates them first: their claims are then the lowest.
      const long long head = head_tiles(c, n_tiles);
      for (int part = 0; part < 2; ++part) {
        const long long lo = part == 0 ? 0 : head, cnt = part == 0 ? head : n_tiles - head;
        if (cnt <= 0) continue;
        if (part == 1 && head > 0) {  // the head launch is a stage of its own
          stage_end(c);
          stage_begin(c, "node_upsert");
        }
        hipLaunchKernelGGL(part == 0 ? kern_head : kern, dim3((unsigned)cnt), dim3(TILE_THREADS), 0, st, c->tokens.as<int>(),
                           c->bnd_bits.as<unsigned int>(), T, k, c->two_v, c->x_fp ? 0 : c->x_bits,
                           c->node_tab.as<Slot16>(), (unsigned int)(c->node_slots - 1), kProbeLimitX,
                           c->tok_slot.as<int>(), c->tok_dir.as<signed char>(),
                           c->status.as<unsigned long long>(), c->x_first.as<unsigned int>(),
                           c->x_slot.as<unsigned int>(), cap, xw2_for(max_claims, T), (unsigned int)lo, ctrs,
                           (unsigned int)head_cap, (unsigned long long)c->seed, c->weak_fp_builds > 0 ? 1 : 0);
      }
    }
  }
  stage_end(c);  // the stage is the kernel alone: its time is what bench.py prices against the roofline
  if (c->x_fp && n_tiles > 0) {
    // fingerprint keys: every window's tuple against its claim's first occurrence; a mismatch raises ST_COLLISION,
    // which the next read-back of the status words reports (bx_edges_upsert / bx_nodes_filtered): which = 3
    stage_begin(c, "node_verify");
    hipLaunchKernelGGL(k_x_verify_fp, dim3(blocks_for(T, 256)), dim3(256), 0, st, c->tokens.as<int>(), T, k, c->two_v - 1,
                       c->tok_slot.as<int>(), c->tok_dir.as<signed char>(), c->x_first.as<unsigned int>(),
                       c->status.as<unsigned long long>());
    stage_end(c);
  }
  unsigned long long most = 0;
  AMGCHK(read_status(c, hs, ctrs, ST_NODE_INSERTS, &most, plain));
  if (hs[ST_BADINPUT])
    return amg_fail(AMG_E_ARG, hs[ST_BADINPUT] == 1 ? "read_offsets must start at 0, never decrease and end at the token count"
                                                    : "a token lies outside [0, two_v)");
  if (hs[ST_PALINDROME])
    return amg_fail(AMG_E_PALINDROME, "Gene-mer and reverse complement gene-mer are identical");
  if (hs[ST_MISC]) return amg_fail(AMG_E_HIP, "node pass: a claim id was never published");
  if (hs[ST_COLLISION]) {  // fingerprint keys: k_x_verify_fp found two gene-mers under one key — nothing is built on this table
    *which = 3;
    return AMG_E_OVERFLOW;
  }
  if (hs[ST_OVERFLOW]) {
    *which = 1;
    return AMG_E_OVERFLOW;
  }
  c->n_windows = (int64_t)hs[ST_N_WINDOWS];
  c->n_short = (int64_t)hs[ST_N_SHORT];
  c->n_local_nodes = c->n_nodes = (int64_t)hs[ST_NODE_INSERTS];
  c->x_nspace = ctrs ? shard_space(most, head_cap, max_claims) : c->n_nodes;
  c->x_max_claims = (int64_t)max_claims;
  return AMG_OK;
}

// node table pass, then only the nodes with coverage >= min_cov are kept: ids, arrays and coverages of the
// survivors; x_final = -2 for the others
int bx_nodes_filtered(amg_ctx* c, int k, unsigned int min_cov, int* which) {
  hipStream_t st = c->stream;
  c->filtered_build = true;
  const int r0 = bx_nodes_upsert(c, k, which, true);
  c->filtered_build = false;
  AMGCHK(r0);
  // claim ids in use lie below n (shard counters: with ids nobody took in between — first-seen 0, like the claims
  // dropped here)
  const long long n = c->x_nspace, T = c->n_tokens;
  stage_begin(c, "node_count");  // per claim, straight from the per-window claims (construct_node.py:33-36)
  AMGCHK(c->x_ecnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  AMGCHK(count_ids(c, c->tok_slot.as<int>(), T, nullptr, n, c->x_ecnt.as<unsigned int>(), 4));
  stage_end(c);
  stage_begin(c, "node_filter");
  AMGCHK(c->x_first_all.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  c->comp_from_claims = true;
  unsigned long long* kept = c->status.as<unsigned long long>() + ST_COMPACT_A;
  HIPCHK(hipMemsetAsync(kept, 0, sizeof(unsigned long long), st));
  if (n > 0)
    hipLaunchKernelGGL(k_x_drop_claims, dim3(blocks_for(n, 256)), dim3(256), 0, st, c->x_ecnt.as<unsigned int>(), n,
                       min_cov, c->x_first.as<unsigned int>(), c->x_final.as<int>(), kept,
                       c->x_first_all.as<unsigned int>());
  unsigned long long D = 0;
  {
    FetchList l;
    l.add(kept);
    l.add(c->status.as<unsigned long long>() + ST_COLLISION);
    unsigned long long v[2] = {0, 0};
    AMGCHK(fetch(c, l, v));
    D = v[0];
    if (v[1]) {  // two gene-mers share a fingerprint (k_x_verify_fp): the build is repeated with the next seed
      stage_end(c);
      *which = 3;
      return AMG_E_OVERFLOW;
    }
  }
  stage_end(c);
  c->n_nodes = (int64_t)D;
  AMGCHK(bx_nodes_rank(c));
  if (n > 0 && D > 0)
    hipLaunchKernelGGL(k_x_cov_from_claims, dim3(blocks_for(n, 256)), dim3(256), 0, st, c->x_ecnt.as<unsigned int>(),
                       c->x_first.as<unsigned int>(), c->x_final.as<int>(), n, c->node_cov.as<unsigned int>(), (int*)nullptr);
  return AMG_OK;
}

// node id = rank of first-seen among the claims; node arrays
int bx_nodes_rank(amg_ctx* c) {
  hipStream_t st = c->stream;
  const long long T = c->n_tokens;
  const int k = c->k;
  stage_begin(c, "node_rank");
  const long long D = c->n_nodes;
  AMGCHK(c->s1.ensure((size_t)(D + 1) * sizeof(unsigned int)));
  AMGCHK(c->s2.ensure((size_t)(D + 1) * sizeof(unsigned int)));
  AMGCHK(c->s3.ensure((size_t)(D + 1) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(D + 1) * sizeof(unsigned int)));
  AMGCHK(bs_alloc_nodes(c, D));
  const long long S = c->x_nspace;  // claim ids in use (== D unless handed out in interleaved shards)
  if (D > 0 && ((D <= kRankBitmapMax && !getenv("AMG_X_RANK_SORT")) || S != D)) {
    AMGCHK(x_rank_bitmap(c, c->x_first.as<unsigned int>(), S, 1));
    hipLaunchKernelGGL(k_x_assign_nodes_ranked, dim3(blocks_for(S, 256)), dim3(256), 0, st,
                       c->x_first.as<unsigned int>(), S,
                       c->s1.as<unsigned int>(), c->s5.as<long long>(), c->node_tab.as<Slot16>(),
                       c->x_slot.as<unsigned int>(), k, c->x_fp ? 0 : c->x_bits, (c->x_fp || (long long)k * c->x_bits > 63) ? 1 : 0,
                       c->x_final.as<int>(), c->node_tokens.as<int>(),
                       c->node_first.as<long long>(), c->node_alive.as<unsigned char>(), c->tokens.as<int>(), c->two_v - 1);
  } else if (D > 0) {
    c->rank_flags_clean = 0;  // (s0 is about to be reused by whoever comes next: nothing of it is known to be zero)
    hipLaunchKernelGGL(k_x_sort_keys, dim3(blocks_for(D, 256)), dim3(256), 0, st, c->x_first.as<unsigned int>(),
                       D, c->s1.as<unsigned int>(),
                       c->s3.as<unsigned int>());
    AMGCHK(prim_sort_u32_u32(c, c->s1.as<unsigned int>(), c->s2.as<unsigned int>(), c->s3.as<unsigned int>(),
                             c->s4.as<unsigned int>(), (size_t)D, ilog2_ceil((uint64_t)T * 2 + 2) + 1));
    hipLaunchKernelGGL(k_x_assign_nodes, dim3(blocks_for(D, 256)), dim3(256), 0, st, c->s2.as<unsigned int>(),
                       c->s4.as<unsigned int>(), D, c->node_tab.as<Slot16>(), c->x_slot.as<unsigned int>(),
                       k, c->x_fp ? 0 : c->x_bits, (c->x_fp || (long long)k * c->x_bits > 63) ? 1 : 0, c->x_final.as<int>(),
                       c->node_tokens.as<int>(), c->node_first.as<long long>(), c->node_alive.as<unsigned char>(),
                       c->tokens.as<int>(), c->two_v - 1);
  }
  stage_end(c);
  return AMG_OK;
}

// adjacencies -> edge-class table, claims, pair arrays in first-seen order, coverages.
// AMG_E_OVERFLOW + *which = 2: edge table full
int bx_edges(amg_ctx* c, int* which, unsigned int min_edge_cov) {
  // a plain build counts its nodes BEFORE the edge pass: the nodes of coverage 1 (nine in ten of an uncorrected
  // graph) mark the edge classes that occur once, and those need no table (k_edges_v<.., LONE>)
  bool lone = false;
  if (min_edge_cov == 0) {
    const char* e = getenv("AMG_EDGE_LONE");  // A/B + test switch: 0 never, 1 whenever the kernel allows it
    const char* eh = getenv("AMG_EDGE_HOME");
    const bool can = !(eh && atoi(eh) == 0) && c->n_nodes > 0;
    // worth its extra words per window where many nodes are single: more than one node per 16 windows
    lone = can && (e ? atoi(e) != 0 : c->n_nodes * 16 > c->n_tokens);
    AMGCHK(bx_node_count(c, lone));
  }
  AMGCHK(bx_edges_upsert(c, which, lone, min_edge_cov == 0));
  return bx_edges_rank(c, min_edge_cov, min_edge_cov == 0);
}

// node coverage of a plain build (construct_node.py:33-36): occurrences per CLAIM from the node pass's per-window
// claims — the occurrence that created a key is marked there, so the keys seen once (most of an uncorrected graph)
// cost nothing — then one store per claim into the node's counter.  tag: x_ftag = x_final with AMG_SINGLE_BIT
int bx_node_count(amg_ctx* c, bool tag) {
  const long long T = c->n_tokens, D = c->n_nodes;
  stage_begin(c, "node_count");
  const long long S = c->x_nspace;
  AMGCHK(c->x_ncnt.ensure((size_t)(S + 2) * sizeof(unsigned int)));
  if (tag) AMGCHK(c->x_ftag.ensure((size_t)(S + 2) * sizeof(int)));
  AMGCHK(count_ids(c, c->tok_slot.as<int>(), T, nullptr, S, c->x_ncnt.as<unsigned int>(), 4));
  if (S > 0 && D > 0)
    hipLaunchKernelGGL(k_x_cov_from_claims, dim3(blocks_for(S, 256)), dim3(256), 0, c->stream,
                       c->x_ncnt.as<unsigned int>(), c->x_first.as<unsigned int>(), c->x_final.as<int>(), S,
                       c->node_cov.as<unsigned int>(), tag ? c->x_ftag.as<int>() : (int*)nullptr);
  stage_end(c);
  return AMG_OK;
}

// the table pass alone: tok_node from x_final, edge-class claims 0 .. n_local_pairs-1
// lone: x_ftag marks the nodes of coverage 1 (bx_node_count); their classes bypass the table
// sharded, rank_follows: as bx_nodes_upsert's
int bx_edges_upsert(amg_ctx* c, int* which, bool lone, bool sharded, bool rank_follows) {
  *which = 0;
  hipStream_t st = c->stream;
  const long long T = c->n_tokens, D = c->n_nodes;
  const long long n_tiles = (T + TILE - 1) / TILE;
  unsigned long long hs[ST_WORDS];
  // home slots (k_edges_v<.., true>): one per node id in front of the hashed slots; AMG_EDGE_HOME=0: none (A/B switch)
  const char* eh = getenv("AMG_EDGE_HOME");
  const long long home_n = (eh && atoi(eh) == 0) ? 0 : ((D + 7) & ~7ll);
  // hashed slots: with home slots only the classes that do not join ids n and n + 1 (one in ten on gene-call reads)
  // need one — sized for a quarter of the nodes; an input that needs more overflows once and is rebuilt 4x larger
  const int64_t want_slots = (int64_t)slots_for((uint64_t)(home_n ? D / 4 + 1 : D));
  if (c->edge_slots < want_slots) c->edge_slots = want_slots;
  const size_t tab_slots = (size_t)c->edge_slots + (size_t)home_n;
  if (!home_n) lone = false;
  // claims: at most one per table slot, plus (lone) two classes per single node, never more than the windows
  long long claim_bound = (long long)tab_slots + (lone ? 2 * D : 0);
  claim_bound = (claim_bound < T ? claim_bound : T) + 1;
  const bool plain = sharded && rank_follows;  // the caller is the plain build: counting and ranking follow, nobody else writes s0
  c->rank_flags_clean = 0;
  sharded = sharded && shard_claims(n_tiles);
  const unsigned int cap = sharded ? shard_share(claim_bound) : (unsigned int)claim_bound;  // per counter
  const long long head_cap = sharded ? dense_tiles(c, n_tiles) * TILE : 0;                  // ids of the first tiles
  const size_t max_claims = sharded ? (size_t)head_cap + (size_t)cap * F_SHARDS + 1 : (size_t)claim_bound;
  unsigned long long* ctrs = nullptr;
  if (sharded) {
    AMGCHK(c->f_ctrs.ensure(2 * X_CTRS * F_CTR_STRIDE * sizeof(unsigned long long)));
    ctrs = c->f_ctrs.as<unsigned long long>() + X_CTRS * F_CTR_STRIDE;
  }
  AMGCHK(c->tok_pair.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->edge_tab.ensure((tab_slots + (lone ? max_claims : 0)) * sizeof(Slot16)));  // (lone classes: slot = tab_slots + claim)
  AMGCHK(c->x_efirst.ensure(2 * max_claims * sizeof(unsigned int)));
  AMGCHK(c->x_eslot.ensure(max_claims * sizeof(unsigned int)));
  stage_begin(c, "edge_table_clear");
  {
    ClearList cl;
    cl.add(c->edge_tab.p, tab_slots * sizeof(Slot16));
    cl.add(c->x_efirst.p, 2 * max_claims * sizeof(unsigned int));
    cl.add(c->status.as<unsigned long long>() + ST_PAIR_INSERTS, 2 * sizeof(unsigned long long));
    if (ctrs) cl.add(ctrs, X_CTRS * F_CTR_STRIDE * sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
  }
  stage_end(c);
  stage_begin(c, (n_tiles > 0 && head_tiles(c, n_tiles) > 0) ? "edge_upsert_head" : "edge_upsert");
  if (n_tiles > 0) {
    const long long head = head_tiles(c, n_tiles);  // (see bx_nodes_upsert: the genome's classes get the lowest claims)
    for (int part = 0; part < 2; ++part) {
      const long long lo = part == 0 ? 0 : head, cnt = part == 0 ? head : n_tiles - head;
      if (cnt <= 0) continue;
      if (part == 1 && head > 0) {  // the head launch is a stage of its own
        stage_end(c);
        stage_begin(c, "edge_upsert");
      }
      auto kern = lone     ? (part == 0 ? k_edges_v<true, true, true> : k_edges_v<false, true, true>)
                  : home_n ? (part == 0 ? k_edges_v<true, true> : k_edges_v<false, true>)
                           : (part == 0 ? k_edges_v<true, false> : k_edges_v<false, false>);
      hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(TILE_THREADS), 0, st, T, c->tok_slot.as<int>(),
                         c->tok_dir.as<signed char>(), lone ? c->x_ftag.as<int>() : c->x_final.as<int>(),
                         c->tok_node.as<int>(), c->edge_tab.as<Slot16>(), (unsigned int)(c->edge_slots - 1), kProbeLimitX,
                         c->status.as<unsigned long long>(), c->tok_pair.as<int>(),
                         c->x_efirst.as<unsigned int>(), c->x_eslot.as<unsigned int>(), cap,
                         xw2_for(max_claims, T), (unsigned int)lo, (unsigned int)home_n, (unsigned int)tab_slots, ctrs,
                         (unsigned int)head_cap);
    }
  }
  stage_end(c);  // the stage is the kernel alone: its time is what bench.py prices against the roofline
  unsigned long long most = 0;
  AMGCHK(read_status(c, hs, ctrs, ST_PAIR_INSERTS, &most, plain));
  if (hs[ST_MISC]) return amg_fail(AMG_E_HIP, "edge pass: a claim id was never published");
  if (hs[ST_COLLISION]) {  // fingerprint keys: k_x_verify_fp found two gene-mers under one key
    *which = 3;
    return AMG_E_OVERFLOW;
  }
  if (hs[ST_OVERFLOW]) {
    *which = 2;
    return AMG_E_OVERFLOW;
  }
  c->n_local_pairs = c->n_pairs = (int64_t)hs[ST_PAIR_INSERTS];
  c->x_espace = ctrs ? shard_space(most, head_cap, max_claims) : c->n_pairs;
  c->x_max_eclaims = (int64_t)max_claims;
  return AMG_OK;
}

// coverages, edge classes in first-seen order.  min_edge_cov > 0: the build applies the coverage filter on the
// way (bx_nodes_filtered has the node coverages already; classes below min_edge_cov are dropped before ranking)
int bx_edges_rank(amg_ctx* c, unsigned int min_edge_cov, bool nodes_counted) {
  const long long T = c->n_tokens, P = c->x_espace;  // (claim ids in use lie below P)
  if (min_edge_cov == 0 && !nodes_counted) AMGCHK(bx_node_count(c, false));
  // edge-class coverage per claim
  stage_begin(c, "edge_count");
  AMGCHK(c->x_ecnt.ensure((size_t)(P + 2) * sizeof(unsigned int)));
  AMGCHK(count_ids(c, c->tok_pair.as<int>(), T, nullptr, P, c->x_ecnt.as<unsigned int>(), 5));
  stage_end(c);
  if (min_edge_cov > 1 && P > 0) {  // filter_graph's edge threshold (:531-535)
    hipStream_t st = c->stream;
    unsigned long long* kept = c->status.as<unsigned long long>() + ST_COMPACT_A;
    HIPCHK(hipMemsetAsync(kept, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_x_drop_claims, dim3(blocks_for(P, 256)), dim3(256), 0, st, c->x_ecnt.as<unsigned int>(), P,
                       min_edge_cov, c->x_efirst.as<unsigned int>(), (int*)nullptr, kept, (unsigned int*)nullptr);
    unsigned long long left = 0;
    FetchList l;
    l.add(kept);
    AMGCHK(fetch(c, l, &left));
    c->n_pairs = (int64_t)left;  // x_espace stays: the dropped classes are holes of the claim space
  }
  return bx_pairs_rank(c, nullptr, nullptr);
}

// after bs_finish_from_pairs of a filtered build: the reads the filter touched
int bx_flag_dead_reads(amg_ctx* c) {
  if (c->n_reads > 0)
    hipLaunchKernelGGL(k_x_flag_dead_reads, dim3(blocks_for(c->n_reads, FD_READS)), dim3(256), 0, c->stream,
                       c->tok_node.as<int>(), c->read_off.as<long long>(), c->n_reads, c->read_fix.as<unsigned char>());
  return AMG_OK;
}

// edge classes in first-seen order (pair_key / pair_first / pair_cnt) from the claims' arrays;
// final_of_claim != nullptr: the classes are keyed by node CLAIM ids (fused table pass)
// efinal != nullptr: class id per claim is written there and pair_cnt is left to the caller
int bx_pairs_rank(amg_ctx* c, const int* final_of_claim, int* efinal) {
  hipStream_t st = c->stream;
  const long long T = c->n_tokens, P = c->n_pairs;
  stage_begin(c, "edge_rank");
  AMGCHK(bs_alloc_pairs(c, P));
  AMGCHK(c->s1.ensure((size_t)(P + 1) * sizeof(unsigned int)));
  AMGCHK(c->s2.ensure((size_t)(P + 1) * sizeof(unsigned int)));
  AMGCHK(c->s3.ensure((size_t)(P + 1) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(P + 1) * sizeof(unsigned int)));
  const long long S = c->x_espace;
  if (P > 0 && ((P <= kRankBitmapMax && !getenv("AMG_X_RANK_SORT")) || S != P || efinal)) {
    AMGCHK(x_rank_bitmap(c, c->x_efirst.as<unsigned int>(), S, 3));
    hipLaunchKernelGGL(k_x_gather_pairs_ranked, dim3(blocks_for(S, 256)), dim3(256), 0, st,
                       c->x_efirst.as<unsigned int>(), S,
                       c->s1.as<unsigned int>(), c->s5.as<long long>(), c->edge_tab.as<Slot16>(),
                       c->x_eslot.as<unsigned int>(), c->x_ecnt.as<unsigned int>(),
                       c->pair_key.as<unsigned long long>(), c->pair_first.as<unsigned long long>(),
                       c->pair_cnt.as<unsigned int>(), final_of_claim, efinal);
  } else if (P > 0) {
    c->rank_flags_clean = 0;
    if (efinal) return amg_fail(AMG_E_STATE, "bx_pairs_rank: class ids per claim need the bitmap ranking");
    hipLaunchKernelGGL(k_x_sort_keys, dim3(blocks_for(P, 256)), dim3(256), 0, st, c->x_efirst.as<unsigned int>(),
                       P, c->s1.as<unsigned int>(),
                       c->s3.as<unsigned int>());
    AMGCHK(prim_sort_u32_u32(c, c->s1.as<unsigned int>(), c->s2.as<unsigned int>(), c->s3.as<unsigned int>(),
                             c->s4.as<unsigned int>(), (size_t)P, ilog2_ceil((uint64_t)T * 8 + 8) + 1));
    hipLaunchKernelGGL(k_x_gather_pairs, dim3(blocks_for(P, 256)), dim3(256), 0, st, c->s2.as<unsigned int>(),
                       c->s4.as<unsigned int>(), P, c->edge_tab.as<Slot16>(), c->x_eslot.as<unsigned int>(),
                       c->x_ecnt.as<unsigned int>(), c->pair_key.as<unsigned long long>(),
                       c->pair_first.as<unsigned long long>(), c->pair_cnt.as<unsigned int>(), final_of_claim);
  }
  stage_end(c);
  return AMG_OK;
}

// component ids of a filtered build (see k_xc_*)
int bx_components_from_claims(amg_ctx* c) {
  hipStream_t st = c->stream;
  const long long S = c->x_nspace, T = c->n_tokens, D = c->n_nodes;
  stage_begin(c, "components");
  AMGCHK(c->node_comp.ensure((size_t)(D + 1) * sizeof(int)));
  AMGCHK(c->s3.ensure((size_t)(S + 2) * sizeof(int)));
  AMGCHK(c->s4.ensure((size_t)(2 * S + 4) * sizeof(unsigned int)));
  int* parent = c->s3.as<int>();
  unsigned int* best2 = c->s4.as<unsigned int>();
  unsigned long long* n_roots = c->status.as<unsigned long long>() + ST_COMPACT_A;
  long long ncomp = 0;
  if (S > 0) {
    ClearList cl;
    cl.add(best2, (size_t)(2 * S + 2) * sizeof(unsigned int));
    cl.add(n_roots, sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
    hipLaunchKernelGGL(k_xc_init, dim3(blocks_for(S, 256)), dim3(256), 0, st, parent, S);
    if (T > 1)
      hipLaunchKernelGGL(k_xc_union, dim3(blocks_for(T, 256)), dim3(256), 0, st, c->tok_slot.as<int>(), T, parent);
    hipLaunchKernelGGL(k_xc_flatten, dim3(blocks_for(S, 256)), dim3(256), 0, st, parent, S);
    hipLaunchKernelGGL(k_xc_best, dim3(blocks_for(S, 256)), dim3(256), 0, st, parent, c->x_first_all.as<unsigned int>(), S,
                       best2, n_roots);
    AMGCHK(x_rank_bitmap(c, best2, S, 1));  // non-roots keep {0, 0}: skipped like unclaimed ids
    hipLaunchKernelGGL(k_xc_label, dim3(blocks_for(S, 256)), dim3(256), 0, st, parent, c->x_first_all.as<unsigned int>(), S,
                       best2, c->s1.as<unsigned int>(), c->s5.as<long long>(), c->x_final.as<int>(),
                       c->node_comp.as<int>());
    FetchList l;
    l.add(n_roots);
    AMGCHK(fetch(c, l, reinterpret_cast<unsigned long long*>(&ncomp)));
  }
  stage_end(c);
  c->n_components = ncomp;
  c->comp_valid = true;
  return AMG_OK;
}
