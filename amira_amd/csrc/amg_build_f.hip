// amg_build_f.hip — the single-GPU build's table pass, FUSED: one kernel takes every window of a
// tile through the node table AND every adjacency through the edge-class table
// (GeneMerGraph.__init__, reference construct_graph.py:31-102: Read.get_geneMers
// construct_read.py:37-59, define_geneMer construct_gene_mer.py:42-56, add_node :196-212,
// add_edge :300-324 with the edge classes of construct_edge.py:104-124).
//
// What changed against the two passes of amg_build_x.hip (which stay for the multi-GPU shards):
//   * edge classes are keyed by CLAIM ids of their two nodes, not by final node ids.  A claim id is
//     stable from the moment it is published, so the adjacency of windows t and t + 1 can be
//     inserted right after their nodes, while both still sit in registers / LDS: no second pass
//     over the tokens that re-reads (claim, direction), gathers claim -> node id and writes node
//     ids per window.  The (lo, hi) order of a class and its "first event ran lo -> hi" bit are
//     re-expressed in final node ids where the classes are ranked (E-sized, k_x_gather_pairs*).
//     Per-window node ids are produced by the coverage count's first sweep, which reads the
//     claims anyway (count_ids with a remap).
//   * a thread takes FOUR CONSECUTIVE windows: its 4 + k - 1 tokens come out of LDS with 128-bit
//     reads, canonical orientation is decided from the pair sums x[j] + x[k-1-j] (the window is
//     below its reverse complement at the first j where the sum is below 2V - 1 — both orders
//     compare the same two sums), the packed tuple of the common 16-bit case is assembled from
//     shared half-words, three of the four adjacencies never leave the thread, and the results
//     leave as one 16-byte store per array.  The two-pass kernels spent ~270 vector + ~240 scalar
//     instructions per window on this front end, more than on the table probe.
//   * tiles advance by 1020 windows and compute 1024: the last four belong to the next tile and
//     only the first of them is used (as the right-hand neighbour of window 1019); its insert is
//     the same find-or-create the owning tile performs, so whichever comes first creates the slot.
#include <type_traits>
#include <vector>
#include "amg_tile.h"
#include "amg_x.h"

#define F_STRIDE 1020  // windows a tile owns (threads 0..254 x 4)
#define F_SPAN 1024    // windows a tile computes
#define F_BIT_WORDS ((31 + F_SPAN + AMG_MAX_K + 2 + 31) / 32 + 1)
#define F_DIRBIT 0x40000000u  // in the LDS exchange word: the window's direction is -1

static_assert(F_BIT_WORDS <= BND_PAD_WORDS, "read-end bitmap padding too small for the fused tiles");

// PHASE 0: both halves in one launch.  PHASE 1 / 2: the node half / the edge half alone (two launches, the
// per-window claims and directions travel through tok_claim / tok_dir): each launch then keeps ONE
// table's hot lines in the L2s.
template <int K, bool TWO, bool B16, int PHASE>  // K > 0: k known at compile time; B16: 16 bits per token (K = 3 or 5)
__global__ __launch_bounds__(TILE_THREADS, 8) void k_graph_x(
    const int* __restrict__ tokens, const unsigned int* __restrict__ bnd_bits, long long n_tokens, int k, int two_v,
    int bits, Slot16* ntab, unsigned int nmask, Slot16* etab, unsigned int emask, unsigned int probe_limit,
    int* __restrict__ tok_claim, signed char* __restrict__ tok_dir, int* __restrict__ tok_pair,
    unsigned long long* status, unsigned int* nfirst2, unsigned int* __restrict__ nslot_by_claim,
    unsigned int* efirst2, unsigned int* __restrict__ eslot_by_claim, XW2 nxf, XW2 exf,
    unsigned long long* ctrs, unsigned int ncap, unsigned int ecap, unsigned long long* stamps) {
  // timing experiment (make EXPERIMENTS=1, AMG_F_STAMPS=1): wave 0 of every block stamps its phases
#define F_STAMP(i)                                                                     \
  if (AMG_EXPERIMENTS && stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime()
  F_STAMP(0);
  __shared__ __attribute__((aligned(16))) int s_tok[PHASE == 2 ? 4 : F_SPAN + AMG_MAX_K + 4];
  __shared__ unsigned int s_bits[PHASE == 2 ? 1 : F_BIT_WORDS];
  __shared__ __attribute__((aligned(16))) int s_claim[PHASE == 1 ? 4 : F_SPAN + 4];
  const int tid = threadIdx.x;
  const long long t0 = (long long)blockIdx.x * F_STRIDE;  // 16-byte aligned in every per-token array
  const int flip = two_v - 1;
  const unsigned int shard = (blockIdx.x * (TILE_THREADS / 64) + (tid >> 6)) & (F_SHARDS - 1);
  // ---- stage tokens t0 .. t0 + F_SPAN + k - 2 and the tile's slice of the read-end bitmap
  if constexpr (PHASE != 2) {
    bool bad = false;
    if (t0 + F_SPAN <= n_tokens) {
      const int4 x = reinterpret_cast<const int4*>(tokens + t0)[tid];
      bad = (unsigned int)x.x >= (unsigned int)two_v || (unsigned int)x.y >= (unsigned int)two_v ||
            (unsigned int)x.z >= (unsigned int)two_v || (unsigned int)x.w >= (unsigned int)two_v;
      reinterpret_cast<int4*>(s_tok)[tid] = x;
    } else {
      for (int i = tid; i < F_SPAN; i += TILE_THREADS) {
        const long long t = t0 + i;
        const int x = t < n_tokens ? tokens[t] : 0;
        bad = bad || (unsigned int)x >= (unsigned int)two_v;
        s_tok[i] = x;
      }
    }
    if (tid < k + 3) {  // the k - 1 tokens the last windows reach into (+ padding read by 128-bit loads)
      const long long t = t0 + F_SPAN + tid;
      const int x = t < n_tokens ? tokens[t] : 0;
      bad = bad || (unsigned int)x >= (unsigned int)two_v;
      s_tok[F_SPAN + tid] = x;
    }
    if (bad) status[ST_BADINPUT] = 2;  // a token outside [0, two_v) would alias another tuple
    if (tid < F_BIT_WORDS) s_bits[tid] = bnd_bits[(t0 >> 5) + tid];
    __syncthreads();
  }
  F_STAMP(1);

  // ---- nodes: four consecutive windows per thread
  const int i0 = 4 * tid;
  unsigned int id1[TILE_ITEMS];
  unsigned int last = 0, ndir = 0;  // per window: last of its read; direction -1
  int cw[TILE_ITEMS + 1];
  if constexpr (PHASE != 2) {
    unsigned long long w1[TILE_ITEMS];
    unsigned int idx[TILE_ITEMS], tag[TILE_ITEMS];
    ulonglong2 v[TILE_ITEMS];
    unsigned int valid = 0;
    // windows whose k tokens lie in one read: no read ends at the positions t + 1 .. t + k - 1
    // (a read that ends right after the window makes it the last of its read)
    const unsigned int b = tile_bits(s_bits, (int)(t0 & 31) + i0 + 1, k + 3);
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
      const bool ok = (t + k <= n_tokens) && inside && (tid < TILE_THREADS - 1 || (w == 0 && PHASE == 0));
      if (!ok) continue;
      int dir;
      if constexpr (K > 0 && B16) {
        dir = f_canon_pack16<K, TWO>(a + w, flip, w1[w], tag[w]);
      } else if constexpr (K > 0) {
        dir = x_canon_pack<K, TWO>(a + w, flip, bits, w1[w], tag[w]);
      } else {
        LdsView win{s_tok + i0 + w};
        dir = canon_dir(win, k, flip);
        if (dir != 0) x_pack(win, k, flip, dir, bits, w1[w], tag[w]);
      }
      if (dir == 0) {
        status[ST_PALINDROME] = 1;  // benign race: every writer stores 1
        continue;
      }
      idx[w] = (unsigned int)mix64(w1[w] ^ ((unsigned long long)tag[w] * 0x9E3779B97F4A7C15ull)) & nmask;
      v[w] = *reinterpret_cast<const ulonglong2*>(ntab + idx[w]);  // in flight while the next window is prepared
      if (dir < 0) ndir |= 1u << w;
      valid |= 1u << w;
      if ((b >> (w + k - 1)) & 1u) last |= 1u << w;
    }
    F_STAMP(2);
    f_table_phase<TWO, 1, true>(ntab, nmask, valid, w1, tag, idx, v, (unsigned int)t0 + i0, ndir, nxf, nfirst2,
                          nslot_by_claim, ctrs + (size_t)shard * F_CTR_STRIDE, shard, ncap, probe_limit, status, 1, id1);
  }
  F_STAMP(3);

  // ---- hand the claims to the neighbours: word = claim | last-of-read << 31 | (direction -1) << 30
  if constexpr (PHASE != 2) {
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w)
      cw[w] = id1[w] ? (int)((id1[w] - 1u) | ((last & (1u << w)) ? AMG_LAST_FLAG : 0u) |
                             ((ndir & (1u << w)) ? F_DIRBIT : 0u))
                     : -1;
  } else {
    // the node half's outputs of this tile (the halo thread takes the next tile's first window)
    const long long t = t0 + i0;
    int4 x = make_int4(-1, -1, -1, -1);
    unsigned int d = 0;
    if (t + TILE_ITEMS <= n_tokens) {
      x = *reinterpret_cast<const int4*>(tok_claim + t);
      d = *reinterpret_cast<const unsigned int*>(tok_dir + t);
    } else {
      if (t + 0 < n_tokens) { x.x = tok_claim[t + 0]; d |= (unsigned int)(unsigned char)tok_dir[t + 0]; }
      if (t + 1 < n_tokens) { x.y = tok_claim[t + 1]; d |= (unsigned int)(unsigned char)tok_dir[t + 1] << 8; }
      if (t + 2 < n_tokens) { x.z = tok_claim[t + 2]; d |= (unsigned int)(unsigned char)tok_dir[t + 2] << 16; }
      if (t + 3 < n_tokens) { x.w = tok_claim[t + 3]; d |= (unsigned int)(unsigned char)tok_dir[t + 3] << 24; }
    }
    cw[0] = x.x == -1 ? -1 : (int)((unsigned int)x.x | ((d & 0x80u) ? F_DIRBIT : 0u));
    cw[1] = x.y == -1 ? -1 : (int)((unsigned int)x.y | ((d & 0x8000u) ? F_DIRBIT : 0u));
    cw[2] = x.z == -1 ? -1 : (int)((unsigned int)x.z | ((d & 0x800000u) ? F_DIRBIT : 0u));
    cw[3] = x.w == -1 ? -1 : (int)((unsigned int)x.w | ((d & 0x80000000u) ? F_DIRBIT : 0u));
    if (tid == TILE_THREADS - 1) cw[1] = cw[2] = cw[3] = -1;  // they are the next tile's
  }
  if constexpr (PHASE != 1) {
    reinterpret_cast<int4*>(s_claim)[tid] = make_int4(cw[0], cw[1], cw[2], cw[3]);
    __syncthreads();
    cw[TILE_ITEMS] = tid < TILE_THREADS - 1 ? s_claim[i0 + TILE_ITEMS] : -1;
  }

  // ---- per-window outputs of the node half (the halo thread owns none)
  if (PHASE != 2 && tid < TILE_THREADS - 1) {
    int oc[TILE_ITEMS];
    unsigned int od = 0;
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      oc[w] = cw[w] == -1 ? -1 : (int)((unsigned int)cw[w] & ~F_DIRBIT);
      const unsigned int d = cw[w] == -1 ? 0u : (((unsigned int)cw[w] & F_DIRBIT) ? 0xffu : 1u);
      od |= d << (8 * w);
    }
    const long long t = t0 + i0;
    if (t + TILE_ITEMS <= n_tokens) {
      *reinterpret_cast<int4*>(tok_claim + t) = make_int4(oc[0], oc[1], oc[2], oc[3]);
      *reinterpret_cast<unsigned int*>(tok_dir + t) = od;
    } else {
#pragma unroll
      for (int w = 0; w < TILE_ITEMS; ++w)
        if (t + w < n_tokens) {
          tok_claim[t + w] = oc[w];
          tok_dir[t + w] = (signed char)(od >> (8 * w));
        }
    }
  }
  F_STAMP(4);

  // ---- edges: adjacency (A, dA) -> (B, dB) of windows t and t + 1 of one read (create_edges :246-262);
  // class key = (smaller claim, larger claim, dA * dB), first-seen = (token << 3) | orientation
  if constexpr (PHASE != 1) {
    unsigned long long key[TILE_ITEMS];
    unsigned int idx[TILE_ITEMS], etag[TILE_ITEMS];
    ulonglong2 v[TILE_ITEMS];
    unsigned int valid = 0, orient3 = 0;
#pragma unroll
    for (int w = 0; w < TILE_ITEMS; ++w) {
      key[w] = 0;
      idx[w] = 0;
      etag[w] = 0;
      const int A = cw[w], B = cw[w + 1];
      if (A == -1 || ((unsigned int)A & AMG_LAST_FLAG) || B == -1) continue;
      const unsigned int ca = (unsigned int)A & 0x3fffffffu, cb = (unsigned int)B & 0x3fffffffu;
      const bool negA = ((unsigned int)A & F_DIRBIT) != 0u, negB = ((unsigned int)B & F_DIRBIT) != 0u;
      const unsigned int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
      const unsigned long long sign = negA != negB ? 1ull : 0ull;
      key[w] = (sign << 63) | ((unsigned long long)lo << 32) | (unsigned long long)(hi + 1u);
      const unsigned int orient = (ca == lo ? 1u : 0u) | (negA ? 0u : 2u) | (negB ? 0u : 4u);
      orient3 |= orient << (3 * w);
      idx[w] = (unsigned int)mix64(key[w]) & emask;
      v[w] = *reinterpret_cast<const ulonglong2*>(etab + idx[w]);
      valid |= 1u << w;
    }
    F_STAMP(5);
    f_table_phase<false, 3, true>(etab, emask, valid, key, etag, idx, v, (unsigned int)t0 + i0, orient3, exf, efirst2,
                            eslot_by_claim, ctrs + (size_t)(F_SHARDS + shard) * F_CTR_STRIDE, shard, ecap, probe_limit,
                            status, 2, id1);
  }
  F_STAMP(6);
  if (PHASE != 1 && tid < TILE_THREADS - 1) {
    const long long t = t0 + i0;
    if (t + TILE_ITEMS <= n_tokens) {
      *reinterpret_cast<int4*>(tok_pair + t) =
          make_int4((int)id1[0] - 1, (int)id1[1] - 1, (int)id1[2] - 1, (int)id1[3] - 1);
    } else {
#pragma unroll
      for (int w = 0; w < TILE_ITEMS; ++w)
        if (t + w < n_tokens) tok_pair[t + w] = (int)id1[w] - 1;
    }
  }
  F_STAMP(7);
#undef F_STAMP
}

// claims handed out: sum and largest per-shard count of the node / edge-class counters
__global__ void k_f_ctr_reduce(const unsigned long long* __restrict__ ctrs, unsigned long long* status) {
  const int which = blockIdx.x;  // 0 nodes, 1 edge classes
  unsigned long long v = ctrs[(size_t)(which * F_SHARDS + threadIdx.x) * F_CTR_STRIDE];
  unsigned long long sum = v, mx = v;
  for (int d = 32; d > 0; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    const unsigned long long o = __shfl_xor(mx, d, 64);
    mx = mx > o ? mx : o;
  }
  if (threadIdx.x == 0) {
    status[which == 0 ? ST_NODE_INSERTS : ST_PAIR_INSERTS] = sum;
    status[which == 0 ? ST_COMPACT_A : ST_COMPACT_B] = mx;
  }
}

// ------------------------------------------------------------------ host side
static const unsigned int kProbeLimitF = 1024;

// bits per token of the packed tuple: 16 whenever that fits (constant shifts in the kernels)
int bx_bits(const amg_ctx* c, int k) {
  const int need = ilog2_ceil((uint64_t)(c->two_v > 2 ? c->two_v : 2));
  if (need <= 16 && (long long)k * 16 <= 94 && !getenv("AMG_X_TIGHT_BITS")) return 16;
  return need;
}

// AMG_FUSED: 0 = the two passes of amg_build_x.hip, 1 = one fused launch, 2 = this file's kernel as two
// launches (node half, edge half)
static int f_mode() {
  const char* e = getenv("AMG_FUSED");
  if (!e) return 0;
  return e[0] == '1' ? 1 : e[0] == '2' ? 2 : 0;
}

bool bf_applicable(const amg_ctx* c, int k) {
  if (!bx_applicable(c, k)) return false;
  // Measured (DESIGN.md "One table pass or two"): the fused pass costs as many wave-cycles as the two
  // passes together — a tile's time is its chain of dependent L2 / fabric round trips, which fusion
  // does not shorten — while both tables' hot lines now compete for the same 4 MB of L2 per XCD
  // (misses 30 M -> 62 M per pass) and the per-window node ids need a remap in the counting sweep.
  // cfg 3 sweep: 15.5 ms fused against 14.2 ms with two passes.  It is therefore OFF unless
  // AMG_FUSED=1 asks for it (tests run both).
  if (f_mode() == 0) return false;
  if (getenv("AMG_X_RANK_SORT")) return false;  // test switch of the two-pass path's sort ranking
  return c->n_tokens < (1ll << 29);  // claims carry two flag bits in the LDS exchange
}

static int f_read_status(amg_ctx* c, unsigned long long* host) {
  HIPCHK(hipMemcpyAsync(host, c->status.p, ST_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                        c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// node table + edge-class table in one pass: node claims 0 .. n_nodes-1 and edge-class claims
// 0 .. n_pairs-1 with their first-seen / slot arrays, per-window node claims, directions and
// edge-class claims.  AMG_E_OVERFLOW + *which = 1 / 2: node / edge table full.
int bf_tables(amg_ctx* c, int k, int* which) {
  *which = 0;
  hipStream_t st = c->stream;
  const long long T = c->n_tokens;
  unsigned long long hs[ST_WORDS];
  c->exact_keys = true;
  c->packed_nodes = false;
  c->x_bits = bx_bits(c, k);
  HIPCHK(hipMemsetAsync(c->status.p, 0, ST_WORDS * sizeof(unsigned long long), st));
  AMGCHK(bs_read_stats(c, k));
  const long long n_tiles = (T + F_STRIDE - 1) / F_STRIDE;

  // the edge table is sized before the node count is known: ~1.1 classes (2.2 directed edges) per
  // node on gene-call graphs; an overflow rebuilds with a table four times the size
  {
    const uint64_t est_nodes = c->node_hint > 0 ? (uint64_t)c->node_hint : (uint64_t)T / 8;
    const int64_t want = (int64_t)slots_for(est_nodes * 5 / 4 + 16);  // classes, not directed edges
    if (c->edge_slots < want) c->edge_slots = want;
    if (c->edge_slots > (1ll << 30)) c->edge_slots = 1ll << 30;
  }
  // claim arrays: F_SHARDS interleaved shares; a share holds at least what one wave can create
  auto share = [&](long long slots) {
    const long long bound = (slots < T ? slots : T) + 1;
    // every shard sees every 64th wave of the stream: shares fill evenly up to statistics (a quarter
    // of slack) and up to what one wave creates in one go
    const long long even = (bound + F_SHARDS - 1) / F_SHARDS;
    return (unsigned int)(even + even / 4 + (T < 256 ? T : 256) + 1);
  };
  const unsigned int ncap = share(c->node_slots), ecap = share(c->edge_slots);
  const size_t max_claims = (size_t)ncap * F_SHARDS + 1, max_eclaims = (size_t)ecap * F_SHARDS + 1;
  AMGCHK(c->f_ctrs.ensure(2 * F_SHARDS * F_CTR_STRIDE * sizeof(unsigned long long)));
  AMGCHK(c->tok_slot.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->tok_node.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->tok_dir.ensure((size_t)(T + 8)));
  AMGCHK(c->tok_pair.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->node_tab.ensure((size_t)c->node_slots * sizeof(Slot16)));
  AMGCHK(c->edge_tab.ensure((size_t)c->edge_slots * sizeof(Slot16)));
  AMGCHK(c->x_first.ensure(2 * max_claims * sizeof(unsigned int)));
  AMGCHK(c->x_slot.ensure(max_claims * sizeof(unsigned int)));
  AMGCHK(c->x_final.ensure(max_claims * sizeof(int)));
  AMGCHK(c->x_efirst.ensure(2 * max_eclaims * sizeof(unsigned int)));
  AMGCHK(c->x_eslot.ensure(max_eclaims * sizeof(unsigned int)));

  stage_begin(c, "table_clear");
  {
    ClearList cl;
    cl.add(c->node_tab.p, (size_t)c->node_slots * sizeof(Slot16));
    cl.add(c->edge_tab.p, (size_t)c->edge_slots * sizeof(Slot16));
    cl.add(c->x_first.p, 2 * max_claims * sizeof(unsigned int));
    cl.add(c->x_efirst.p, 2 * max_eclaims * sizeof(unsigned int));
    cl.add(c->f_ctrs.p, 2 * F_SHARDS * F_CTR_STRIDE * sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
  }
  stage_end(c);

  unsigned long long* stamps = nullptr;
  if (AMG_EXPERIMENTS && getenv("AMG_F_STAMPS")) {
    AMGCHK(c->s0.ensure((size_t)(n_tiles + 1) * 8 * sizeof(unsigned long long)));
    stamps = c->s0.as<unsigned long long>();
  }
  stage_begin(c, f_mode() == 1 ? "graph_upsert" : "node_upsert");
  if (n_tiles > 0) {
    const bool two = (long long)k * c->x_bits > 63;  // tuple spills into the second slot word?
    const bool b16 = c->x_bits == 16 && (k == 3 || k == 5);
    const bool generic = getenv("AMG_X_GENERIC_K") != nullptr;  // A/B switch
    const char* sp = getenv("AMG_F_STAMPS");  // "1": the fused launch or the node half, "2": the edge half
    auto launch = [&](auto phase) {
      constexpr int PH = decltype(phase)::value;
      auto kern = two ? k_graph_x<0, true, false, PH> : k_graph_x<0, false, false, PH>;
      if (!generic) {
        if (b16 && k == 3) kern = k_graph_x<3, false, true, PH>;
        else if (b16 && k == 5) kern = k_graph_x<5, true, true, PH>;
        else if (k == 3) kern = two ? k_graph_x<3, true, false, PH> : k_graph_x<3, false, false, PH>;
        else if (k == 5) kern = two ? k_graph_x<5, true, false, PH> : k_graph_x<5, false, false, PH>;
        else if (k == 7) kern = two ? k_graph_x<7, true, false, PH> : k_graph_x<7, false, false, PH>;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)n_tiles), dim3(TILE_THREADS), 0, st, c->tokens.as<int>(),
                         c->bnd_bits.as<unsigned int>(), T, k, c->two_v, c->x_bits, c->node_tab.as<Slot16>(),
                         (unsigned int)(c->node_slots - 1), c->edge_tab.as<Slot16>(),
                         (unsigned int)(c->edge_slots - 1), kProbeLimitF, c->tok_slot.as<int>(),
                         c->tok_dir.as<signed char>(), c->tok_pair.as<int>(), c->status.as<unsigned long long>(),
                         c->x_first.as<unsigned int>(), c->x_slot.as<unsigned int>(), c->x_efirst.as<unsigned int>(),
                         c->x_eslot.as<unsigned int>(), xw2_for(max_claims, T), xw2_for(max_eclaims, T),
                         c->f_ctrs.as<unsigned long long>(), ncap, ecap,
                         (stamps && sp && (PH == 2) == (sp[0] == '2')) ? stamps : (unsigned long long*)nullptr);
    };
    if (f_mode() == 1) {
      launch(std::integral_constant<int, 0>{});
    } else {
      launch(std::integral_constant<int, 1>{});
      stage_end(c);
      stage_begin(c, "edge_upsert");
      launch(std::integral_constant<int, 2>{});
    }
    hipLaunchKernelGGL(k_f_ctr_reduce, dim3(2), dim3(64), 0, st, c->f_ctrs.as<unsigned long long>(),
                       c->status.as<unsigned long long>());
  }
  stage_end(c);  // the stage is the kernel alone: its time is what bench.py prices against the roofline
  AMGCHK(f_read_status(c, hs));
  if (stamps && n_tiles > 0) {
    std::vector<unsigned long long> h((size_t)n_tiles * 8);
    HIPCHK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static const char* names[7] = {"stage", "node probe", "node claim", "exchange+node stores", "edge probe", "edge claim", "edge stores"};
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long b = 0; b < n_tiles; ++b)
      for (int i = 0; i < 7; ++i) sum[i] += (double)(h[b * 8 + i + 1] - h[b * 8 + i]);
    fprintf(stderr, "[amg] k_graph_x phases, mean cycles of wave 0 per tile (%lld tiles):", n_tiles);
    double tot = 0;
    for (int i = 0; i < 7; ++i) { fprintf(stderr, " %s %.0f;", i < 7 ? names[i] : "?", sum[i] / n_tiles); tot += sum[i] / n_tiles; }
    fprintf(stderr, " total %.0f\n", tot);
  }
  if (hs[ST_BADINPUT])
    return amg_fail(AMG_E_ARG, hs[ST_BADINPUT] == 1 ? "read_offsets must start at 0, never decrease and end at the token count"
                                                    : "a token lies outside [0, two_v)");
  if (hs[ST_PALINDROME])
    return amg_fail(AMG_E_PALINDROME, "Gene-mer and reverse complement gene-mer are identical");
  if (hs[ST_MISC]) return amg_fail(AMG_E_HIP, "table pass: a claim id was never published");
  if (hs[ST_OVERFLOW]) {
    *which = (int)hs[ST_OVERFLOW];
    return AMG_E_OVERFLOW;
  }
  c->n_windows = (int64_t)hs[ST_N_WINDOWS];
  c->n_short = (int64_t)hs[ST_N_SHORT];
  c->n_local_nodes = c->n_nodes = (int64_t)hs[ST_NODE_INSERTS];
  c->n_local_pairs = c->n_pairs = (int64_t)hs[ST_PAIR_INSERTS];
  c->x_nspace = (int64_t)hs[ST_COMPACT_A] * F_SHARDS;  // claim ids in use lie below these bounds
  c->x_espace = (int64_t)hs[ST_COMPACT_B] * F_SHARDS;
  c->x_max_claims = (int64_t)max_claims;
  c->x_max_eclaims = (int64_t)max_eclaims;
  return AMG_OK;
}

// after bx_nodes_rank: coverages, per-window node ids, edge classes in first-seen order keyed by
// final node ids.  Both counts run on RANKED ids (node ids, edge-class ids): first-seen order puts
// the genome's nodes / classes first, so the LDS-privatised ranges of count_ids hold them.
int bf_finish(amg_ctx* c) {
  const long long T = c->n_tokens, D = c->n_nodes, P = c->n_pairs;
  // node coverage (construct_node.py:33-36); the first sweep turns the per-window claims into node ids
  stage_begin(c, "node_count");
  AMGCHK(count_ids_remap(c, c->tok_slot.as<int>(), T, c->x_final.as<int>(), D, c->node_cov.as<unsigned int>(), 0));
  std::swap(c->tok_slot, c->tok_node);  // tok_node: node id per window; tok_slot: scratch again
  stage_end(c);
  AMGCHK(c->x_efinal.ensure((size_t)(c->x_espace + 2) * sizeof(int)));
  AMGCHK(bx_pairs_rank(c, c->x_final.as<int>(), c->x_efinal.as<int>()));
  // edge-class coverage (construct_edge.py:87-90), per class id
  stage_begin(c, "edge_count");
  AMGCHK(count_ids_remap(c, c->tok_pair.as<int>(), T, c->x_efinal.as<int>(), P, c->pair_cnt.as<unsigned int>(), 1));
  stage_end(c);
  return AMG_OK;
}
