// amg_build.hip — GeneMerGraph.__init__ on the device (reference construct_graph.py:31-102).
//
// Launch sequence of amg_build (all on ctx->stream):
//   k_read_stats      per-read window / short-read counts          (construct_graph.py:53-55)
//   k_tile_reads      first read boundary of every token tile
//   k_node_upsert     K1+K2: LDS-staged sliding windows, canonical orientation
//                     (construct_gene_mer.py:4-56), fingerprint, open-address upsert:
//                     count (+=1, construct_node.py:33-36) and first-seen (atomicMax of ~first)
//   k_compact_slots   wave-ballot / prefix-sum compaction of occupied slots
//   radix sort        by first-seen  -> node id = insertion order of _nodes (:188-190)
//   k_assign_nodes    dense node arrays (canonical tokens, coverage, first direction)
//   k_edges           K3/K4: slot -> node id per window (get_readNodes, :165-178), exact
//                     verification of the fingerprint against the node's canonical tuple,
//                     and upsert of one record per adjacency into the edge-class table
//                     (create_edges / add_edge_to_edges, :246-277; Edge.__hash__ classes,
//                     construct_edge.py:104-124)
//   k_compact_slots + sort + k_pair_width + scan + k_emit_edges
//                     directed edges in _edges insertion order, E1 then E2 (:279-285)
//   k_adj_keys + stable radix sort + k_row_offsets
//                     forwardEdgeHashes / backwardEdgeHashes lists (:287-298)
//   k_uf_*            connected components, ids in DFS discovery order (:911-927)
#include "amg_device.h"

#include "amg_tile.h"

// ------------------------------------------------------------------ K1 + K2
__global__ __launch_bounds__(TILE_THREADS) void k_node_upsert(
    const int* __restrict__ tokens, const unsigned int* __restrict__ bnd_bits, long long n_tokens, int k,
    int two_v, unsigned long long seed, Slot* __restrict__ tab, unsigned long long mask,
    unsigned int probe_limit, long long tok_base, int* __restrict__ tok_slot,
    signed char* __restrict__ tok_dir, unsigned long long* status, int count_inline,
    unsigned long long fp_mask) {
  __shared__ int s_tok[TILE + AMG_MAX_K];
  __shared__ unsigned int s_bits[TILE_BIT_WORDS];
  const long long t0 = (long long)blockIdx.x * TILE;
  stage_tile(tokens, bnd_bits, n_tokens, k, t0, s_tok, s_bits, two_v, status);
  const int flip = two_v - 1;
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    int i = threadIdx.x + it * TILE_THREADS;
    long long t = t0 + i;
    if (t >= n_tokens) continue;
    bool inside, is_last;
    tile_window(s_bits, i, k, inside, is_last);
    const bool valid = (t + k <= n_tokens) && inside;
    int out_slot = -1;
    signed char out_dir = 0;
    if (valid) {
      LdsView w{s_tok + i};
      int dir = canon_dir(w, k, flip);
      if (dir == 0) {
        status[ST_PALINDROME] = 1;  // benign race: every writer stores 1
      } else {
        unsigned long long fp = canon_fingerprint(w, k, flip, dir, seed) & fp_mask;  // mask: test hook
        fp = fp ? fp : 1ull;
        unsigned long long first = ((unsigned long long)(tok_base + t) << 1) | (dir < 0 ? 1ull : 0ull);
        long long slot = table_upsert(tab, mask, fp, fp >> 20, first, probe_limit, count_inline != 0,
                                      status + ST_OVERFLOW);
        if (slot < 0) {
          status[ST_OVERFLOW] = 1;
        } else {
          out_slot = (int)((unsigned int)slot | (is_last ? AMG_LAST_FLAG : 0u));
          out_dir = (signed char)dir;
        }
      }
    }
    tok_slot[t] = out_slot;
    tok_dir[t] = out_dir;
  }
}

// ------------------------------------------------------------------ compaction
// (first_seen, slot) of every occupied slot, any order; one atomicAdd per block.
__global__ __launch_bounds__(256) void k_compact_slots(const Slot* __restrict__ tab,
                                                       unsigned long long n_slots,
                                                       unsigned long long* __restrict__ out_first,
                                                       unsigned int* __restrict__ out_slot,
                                                       unsigned long long* counter) {
  __shared__ unsigned int s_wave[4];
  __shared__ unsigned long long s_base;
  const int ITEMS = 8;
  unsigned long long base = (unsigned long long)blockIdx.x * (256 * ITEMS);
  unsigned long long firsts[ITEMS];
  unsigned int have = 0, cnt = 0;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    unsigned long long s = base + it * 256 + threadIdx.x;
    if (s < n_slots && tab[s].key != 0ull) {
      firsts[it] = ~tab[s].first_inv;
      have |= 1u << it;
      ++cnt;
    }
  }
  unsigned int total;
  unsigned int off = block_exscan_256(cnt, &total, s_wave);
  if (threadIdx.x == 0) s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
  __syncthreads();
  unsigned long long o = s_base + off;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (have & (1u << it)) {
      out_first[o] = firsts[it];
      out_slot[o] = (unsigned int)(base + it * 256 + threadIdx.x);
      ++o;
    }
  }
}

// ------------------------------------------------------------------ node arrays
__global__ void k_assign_nodes(const unsigned long long* __restrict__ first_sorted,
                               const unsigned int* __restrict__ slot_sorted, long long n_nodes,
                               Slot* __restrict__ tab, const int* __restrict__ tokens, int k,
                               int two_v, long long tok_base, int packed, int* __restrict__ node_tokens,
                               unsigned int* __restrict__ node_cov,
                               long long* __restrict__ node_first,
                               unsigned char* __restrict__ node_alive) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  unsigned long long first = first_sorted[i];
  unsigned int slot = slot_sorted[i];
  tab[slot].id = (int)i;
  node_cov[i] = tab[slot].count;  // 0 when counting is deferred to k_count_ids
  node_first[i] = (long long)first;
  node_alive[i] = 1;
  long long t = (long long)(first >> 1) - tok_base;
  int dir = (first & 1ull) ? -1 : 1;
  const int flip = two_v - 1;
  int canon[AMG_MAX_K];
  for (int j = 0; j < k; ++j) {
    canon[j] = dir > 0 ? tokens[t + j] : flip - tokens[t + k - 1 - j];
    node_tokens[i * k + j] = canon[j];
  }
  if (packed) slot_pack(tab + slot, (int)i, canon, k);
}

// ------------------------------------------------------------------ K3 + K4
__global__ __launch_bounds__(TILE_THREADS) void k_edges(
    const int* __restrict__ tokens, long long n_tokens, int k, int two_v,
    const Slot* __restrict__ node_tab, const int* __restrict__ node_tokens,
    const int* __restrict__ tok_slot, const signed char* __restrict__ tok_dir,
    int* __restrict__ tok_node, Slot* __restrict__ edge_tab, unsigned long long edge_mask,
    unsigned int probe_limit, int verify, long long tok_base, unsigned long long* status,
    int count_inline, int* __restrict__ tok_pair, int packed) {
  __shared__ int s_id[TILE + 1];
  __shared__ int s_raw[TILE + 1];
  __shared__ signed char s_dir[TILE + 1];
  const long long t0 = (long long)blockIdx.x * TILE;
  const int flip = two_v - 1;
  for (int i = threadIdx.x; i < TILE + 1; i += TILE_THREADS) {
    long long t = t0 + i;
    int raw = -1;
    signed char d = 0;
    if (t < n_tokens) {
      raw = tok_slot[t];
      d = tok_dir[t];
    }
    int id = -1;
    if (raw != -1 && packed) {
      // one 32-byte gather: node id + the node's canonical tuple (16-bit tokens)
      const uint4* rec = reinterpret_cast<const uint4*>(node_tab + ((unsigned int)raw & ~AMG_LAST_FLAG));
      const uint4 lo = rec[0], hi = rec[1];
      id = (int)hi.y;
      if (verify && i < TILE && id >= 0) {
        const int* w = tokens + t;
        bool same = true;
        for (int j = 0; j < k; ++j) {
          int cj = d > 0 ? w[j] : flip - w[k - 1 - j];
          same = same && ((unsigned int)cj == packed_tok(lo, hi, j));
        }
        if (!same) status[ST_COLLISION] = 1;
      }
    } else if (raw != -1) {
      id = node_tab[(unsigned int)raw & ~AMG_LAST_FLAG].id;
      if (verify && i < TILE && id >= 0) {
        // exact check: the window's canonical tuple must equal the node's tuple
        const int* w = tokens + t;
        const int* nt = node_tokens + (long long)id * k;
        bool same = true;
        for (int j = 0; j < k; ++j) {
          int c = d > 0 ? w[j] : flip - w[k - 1 - j];
          same = same && (c == nt[j]);
        }
        if (!same) status[ST_COLLISION] = 1;
      }
    }
    s_id[i] = id;
    s_raw[i] = raw;
    s_dir[i] = d;
    if (i < TILE && t < n_tokens) tok_node[t] = id;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < TILE_ITEMS; ++it) {
    int i = threadIdx.x + it * TILE_THREADS;
    int raw = s_raw[i];
    if (raw == -1 || ((unsigned int)raw & AMG_LAST_FLAG)) {
      if (tok_pair && t0 + i < n_tokens) tok_pair[t0 + i] = -1;
      continue;
    }
    // adjacency (A, dA) -> (B, dB): windows t and t + 1 of the same read.  (ids < 0 only in a
    // merged build with a fused coverage filter: the node was dropped, no edge is recorded)
    if (s_id[i] < 0 || s_id[i + 1] < 0) {
      if (tok_pair && t0 + i < n_tokens) tok_pair[t0 + i] = -1;
      continue;
    }
    unsigned int a = (unsigned int)s_id[i], b = (unsigned int)s_id[i + 1];
    int dA = s_dir[i], dB = s_dir[i + 1];
    unsigned int lo = a < b ? a : b, hi = a < b ? b : a;
    unsigned long long sign = (dA * dB < 0) ? 1ull : 0ull;
    unsigned long long key = (sign << 63) | ((unsigned long long)lo << 32) |
                             (unsigned long long)(hi + 1u);
    unsigned long long orient = (a == lo ? 1ull : 0ull) | (dA > 0 ? 2ull : 0ull) |
                                (dB > 0 ? 4ull : 0ull);
    unsigned long long first = ((unsigned long long)(tok_base + t0 + i) << 3) | orient;
    long long slot = table_upsert(edge_tab, edge_mask, key, mix64(key), first, probe_limit,
                                  count_inline != 0);
    if (slot < 0) status[ST_OVERFLOW] = 2;
    if (tok_pair) tok_pair[t0 + i] = (int)slot;
  }
}

// ------------------------------------------------------------------ edge emission
// edge classes ("pairs") in first-seen order as plain arrays: key, count, first
__global__ void k_gather_pairs(const unsigned int* __restrict__ slot_sorted, long long n_pairs,
                               const Slot* __restrict__ edge_tab, unsigned long long* __restrict__ pkey,
                               unsigned int* __restrict__ pcnt) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const Slot* s = edge_tab + slot_sorted[i];
  pkey[i] = s->key;
  pcnt[i] = s->count;
}

__global__ void k_emit_edges(const unsigned long long* __restrict__ pkey,
                             const unsigned int* __restrict__ pcnt,
                             const unsigned long long* __restrict__ pfirst, long long n_pairs,
                             const long long* __restrict__ base, int* __restrict__ e_src,
                             int* __restrict__ e_tgt, signed char* __restrict__ e_sdir,
                             signed char* __restrict__ e_tdir, unsigned int* __restrict__ e_cov,
                             unsigned char* __restrict__ e_alive) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  unsigned long long key = pkey[i], first = pfirst[i];
  int lo = (int)((key >> 32) & 0x7fffffffull);
  int hi = (int)((key & 0xffffffffull) - 1ull);
  int X = (first & 1ull) ? lo : hi, Y = (first & 1ull) ? hi : lo;
  signed char dX = (first & 2ull) ? 1 : -1, dY = (first & 4ull) ? 1 : -1;
  long long e = base[i];
  unsigned int cnt = pcnt[i];
  if (lo == hi) {
    // E1 and E2 fall in the same class: one edge, +2 per traversal
    e_src[e] = X; e_tgt[e] = Y; e_sdir[e] = dX; e_tdir[e] = dY;
    e_cov[e] = cnt * 2u; e_alive[e] = 1;
  } else {
    e_src[e] = X; e_tgt[e] = Y; e_sdir[e] = dX; e_tdir[e] = dY;
    e_cov[e] = cnt; e_alive[e] = 1;
    e_src[e + 1] = Y; e_tgt[e + 1] = X; e_sdir[e + 1] = (signed char)-dY;
    e_tdir[e + 1] = (signed char)-dX; e_cov[e + 1] = cnt; e_alive[e + 1] = 1;
  }
}

// adjacency rows: row = 2 * src + (sdir == +1 ? 0 : 1); edge ids ascending inside a row
__global__ void k_adj_keys(const int* __restrict__ e_src, const signed char* __restrict__ e_sdir,
                           long long n_edges, unsigned int* __restrict__ keys,
                           unsigned int* __restrict__ vals) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  keys[e] = 2u * (unsigned int)e_src[e] + (e_sdir[e] > 0 ? 0u : 1u);
  vals[e] = (unsigned int)e;
}

// CSR offsets from the sorted row keys, no atomics: position i opens every row in
// (key[i - 1], key[i]]; the position after the last edge opens the remaining rows and n_rows
__global__ void k_row_offsets(const unsigned int* __restrict__ keys, long long n_edges, long long n_rows,
                              long long* __restrict__ off) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_edges) return;
  const long long prev = i > 0 ? (long long)keys[i - 1] : -1;
  const long long cur = i < n_edges ? (long long)keys[i] : n_rows;
  for (long long r = prev + 1; r <= cur; ++r) off[r] = i;
}

// The same lists WITHOUT a sort, for graphs of up to a few million edges (every graph of a cleaning sweep after the first
// filter): a library radix sort is a dozen launches of ~5 us whatever it sorts.
//   k_adjc_ticket  every edge draws a ticket of its row (rows zeroed before): row sizes and a place inside the row
//   (scan)         row sizes -> CSR offsets
//   k_adjc_fill    every edge drops its id at offset + ticket (any order within the row)
//   k_adjc_rows    a thread per row puts the row's ids in ascending order (= list order: edge ids follow insertion
//                  order; rows of 3 .. 64 by the wave); rows longer than a wave are left to k_adjc_long, a workgroup per long row (hub nodes)
__device__ __forceinline__ unsigned int adj_row_of(const int* __restrict__ e_src, const signed char* __restrict__ e_sdir,
                                                   long long e) {
  return 2u * (unsigned int)e_src[e] + (e_sdir[e] > 0 ? 0u : 1u);
}

__global__ void k_adjc_ticket(const int* __restrict__ e_src, const signed char* __restrict__ e_sdir, long long n_edges,
                              unsigned int* __restrict__ cnt, unsigned int* __restrict__ tick) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_edges) tick[e] = atomicAdd(&cnt[adj_row_of(e_src, e_sdir, e)], 1u);
}

__global__ void k_adjc_fill(const int* __restrict__ e_src, const signed char* __restrict__ e_sdir, long long n_edges,
                            const long long* __restrict__ off, const unsigned int* __restrict__ tick,
                            unsigned int* __restrict__ tmp) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_edges) tmp[off[adj_row_of(e_src, e_sdir, e)] + tick[e]] = (unsigned int)e;
}

__global__ __launch_bounds__(256) void k_adjc_rows(const long long* __restrict__ off, long long n_rows,
                                                   const unsigned int* __restrict__ tmp, int* __restrict__ adj_edge,
                                                   unsigned int* __restrict__ long_rows, unsigned long long* n_long) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long o = 0;
  int cnt = 0;
  if (r < n_rows) {
    o = off[r];
    cnt = (int)(off[r + 1] - o);
  }
  bool mine = cnt > 2;
  if (cnt == 1) {
    adj_edge[o] = (int)tmp[o];
  } else if (cnt == 2) {
    const unsigned int a = tmp[o], b = tmp[o + 1];
    adj_edge[o] = (int)(a < b ? a : b);
    adj_edge[o + 1] = (int)(a < b ? b : a);
  } else if (cnt > WAVE_ROW_MAX) {
    long_rows[atomicAdd(n_long, 1ull)] = (unsigned int)r;
    mine = false;
  }
  // rows of 3 .. 64 ids: by the wave, one row at a time (wave_rows_in_order, amg_device.h)
  wave_rows_in_order(mine, 0u, o, cnt, tmp,
                     [&](unsigned int, long long ro, int, int rank, unsigned int x) { adj_edge[ro + rank] = (int)x; });
}

// a workgroup per long row: every element finds its rank among the row's (distinct) edge ids; rows beyond HUGE_ROW are
// left to the first HUB_BLOCKS workgroups, which put them in order through a bitmap (huge_row_in_order, amg_device.h)
#define HUB_BLOCKS 8
__global__ __launch_bounds__(256) void k_adjc_long(const unsigned int* __restrict__ long_rows,
                                                   const unsigned long long* __restrict__ n_long,
                                                   const long long* __restrict__ off, const unsigned int* __restrict__ tmp,
                                                   int* __restrict__ adj_edge, unsigned int* hub_bits, long long hub_words) {
  __shared__ unsigned int s_wave[4];
  const unsigned long long n = *n_long;
  for (unsigned long long q = blockIdx.x; q < n; q += gridDim.x) {
    const unsigned int r = long_rows[q];
    const long long o = off[r];
    const int cnt = (int)(off[r + 1] - o);
    if (cnt > HUGE_ROW) continue;
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const unsigned int x = tmp[o + j];
      int rank = 0;
      for (int i = 0; i < cnt; ++i) rank += tmp[o + i] < x ? 1 : 0;
      adj_edge[o + rank] = (int)x;
    }
  }
  if (blockIdx.x >= HUB_BLOCKS) return;
  for (unsigned long long q = blockIdx.x; q < n; q += HUB_BLOCKS) {  // (block-uniform: every thread takes the same rows)
    const unsigned int r = long_rows[q];
    const long long o = off[r];
    const long long cnt = off[r + 1] - o;
    if (cnt <= HUGE_ROW) continue;
    huge_row_in_order(tmp + o, cnt, hub_bits + (long long)blockIdx.x * hub_words, hub_words, s_wave,
                      [&](long long rank, unsigned int id) { adj_edge[o + rank] = (int)id; });
  }
}

// ------------------------------------------------------------------ components
// Union-find with parent[x] <= x.  Only the hook (a root gets a smaller parent) is an atomic;
// every other access is a PLAIN load or store that the issuing XCD's L2 may serve stale.  That
// is safe: a node's parent only ever moves to another member of its set with a smaller id, a
// stale value is an older such ancestor, and a node that has been hooked never becomes a root
// again — so a walk over stale parents still ends at a member of the set, a hook attempted on a
// node that only LOOKED like a root fails and returns the truth, and a path-halving store can
// at worst undo some compression.  (With agent-scope loads and atomicMin halving every step was
// a fabric transaction: 0.8 ms for 6.3 M pairs.)
__global__ void k_uf_init(int* parent, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = (int)i;
}

// Node ids are first-seen ranks: nine edge classes in ten join ids n and n + 1, so a component is mostly a few long
// RUNS of consecutive ids.  The runs are linked before any union: starts[n] = 0 where a class (n - 1, n) exists, a
// prefix sum numbers the runs, every node's parent is its run's first node (a flat forest, parent <= self), and the
// union-find proper only sees the classes that do NOT join consecutive ids.  (One pass of hooks over all classes built
// long chains along those runs first and then halved them: 0.11 ms per call for 0.5 M classes, twice per cleaning sweep,
// and at W emulated ranks the merged graph's 0.5 M x W classes on every rank.)
__global__ void k_uf_links(const unsigned long long* __restrict__ pkey, long long n_pairs, unsigned int* __restrict__ starts) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const unsigned long long key = pkey[i];
  const unsigned int a = (unsigned int)((key >> 32) & 0x7fffffffull), b = (unsigned int)(key & 0xffffffffull) - 1u;
  if (b == a + 1u) starts[b] = 0u;  // (both classes of such a pair, the two signs, store the same word)
}

__global__ void k_uf_run_starts(const unsigned int* __restrict__ starts, const long long* __restrict__ run_of, long long n,
                                int* __restrict__ run_start) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && starts[i]) run_start[run_of[i]] = (int)i;
}

__global__ void k_uf_init_runs(const unsigned int* __restrict__ starts, const long long* __restrict__ run_of,
                               const int* __restrict__ run_start, long long n, int* __restrict__ parent) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = run_start[run_of[i] + (long long)starts[i] - 1];  // (run_of = runs started BEFORE i)
}

__global__ void k_uf_union(const unsigned long long* __restrict__ pkey, long long n_pairs, int* parent) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  unsigned long long key = pkey[i];
  int a = (int)((key >> 32) & 0x7fffffffull);
  int b = (int)((key & 0xffffffffull) - 1ull);
  if (b == a + 1) return;  // linked as a run already (k_uf_links)
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) break;
    if (a > b) { int t = a; a = b; b = t; }
    int old = atomicCAS(parent + b, b, a);  // hook the larger root under the smaller
    if (old == b) break;
    b = old;
  }
}

// root_copy: the roots once more, for the labelling that overwrites parent[] (was a copy launch of its own);
// is_root[n] = 0 closes the array for the scan
__global__ void k_uf_roots(int* parent, long long n, unsigned int* __restrict__ is_root, int* __restrict__ root_copy) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) is_root[n] = 0u;
  if (i >= n) return;
  int r = uf_find(parent, (int)i);
  parent[i] = r;  // only thread i writes entry i with its final root; roots keep parent==self
  root_copy[i] = r;
  is_root[i] = (r == (int)i) ? 1u : 0u;
}

// component id = 1 + rank of the component's smallest node id == DFS discovery order
__global__ void k_uf_label(const int* __restrict__ root, const long long* __restrict__ root_rank,
                           long long n, int* __restrict__ comp) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = root[i];
  // root[] entries of non-roots may still point at an intermediate ancestor written by
  // another thread's k_uf_roots; chase to the fixed point (roots satisfy root[r] == r)
  while (root[r] != r) r = root[r];
  comp[i] = (int)(root_rank[r] + 1);
}


// ------------------------------------------------------------------ deferred counting
// Occurrence counts without one global atomic per window.  Persistent 1024-thread blocks keep
// HOT_IDS counters in LDS for the id range [lo, lo + HOT_IDS) and sweep the whole id array;
// ids are first-seen ranks (or claim order), so the frequently hit (genome) nodes / edges are
// the LOW ids and a few ranges absorb almost every increment.  Every sweep also counts the
// ids that lie beyond its range (state[r]); the next sweep reads that number and, when at
// most 1/8 of the array is left, finishes the job with global atomics (27 G/s, cheaper than
// further 4-byte-per-id sweeps at that point); sweeps after that see its done flag and exit.
// The first sweep takes the same decision from what the first sweep of the PREVIOUS count of
// this kind left behind (hint = {beyond, n}; rebuilds of a cleaning sweep look alike).
// With GATHER the array holds table slots on entry and is rewritten to dense ids
// (tab[slot].id) during the first sweep.
// (HOT_IDS: 156 of the CU's 160 KB of LDS — the head launch of cfg 3's first build hands out 36 k claims, four
// thousand more than the 32 k counters of rounds 1-3 held, and their windows were what a second sweep was for)
#define HOT_IDS 39936
#define COUNT_MAX_SWEEPS 4
// The first sweep also LISTS the ids it finds beyond its range while they are few — every workgroup in a segment of
// its own (COUNT_LIST_SEG ids, filled through a counter in LDS: one shared list cost 20 k returning atomics on one word,
// 0.5 ms) — so that a count whose first sweep had no hint to finish the job itself (the first count of a read set) ends
// with a second launch that walks the segments, microseconds, instead of a second sweep over the whole array for a few
// thousand increments.  list: [0] a segment ran over (the list is then not used), [2 + b] ids in workgroup b's segment,
// segments from COUNT_LIST_HEAD on.
#define COUNT_LIST_SEG 256
#define COUNT_MAX_BLOCKS 256
#define COUNT_LIST_HEAD (2 + COUNT_MAX_BLOCKS)
// state: [0..3] ids beyond the range of sweep r, [4..7] sweep r finished the job
template <bool GATHER>
__global__ __launch_bounds__(1024) void k_count_ids(int* __restrict__ ids, long long n,
                                                    const Slot* __restrict__ tab, long long lo,
                                                    int sweep, int last, unsigned long long* state,
                                                    unsigned long long* hint, unsigned int* __restrict__ out,
                                                    int strip, const int* __restrict__ remap, unsigned int* list,
                                                    unsigned int seg_cap) {
  __shared__ unsigned int s_cnt[HOT_IDS];
  __shared__ unsigned int s_listed;
  bool tail_all = last != 0;
  if (sweep > 0) {
    for (int q = 0; q < sweep; ++q)
      if (state[COUNT_MAX_SWEEPS + q]) return;  // an earlier sweep already finished
    const unsigned long long left = state[sweep - 1];
    if (sweep == 1 && blockIdx.x == 0 && threadIdx.x == 0) {  // what the next count of this kind starts from
      hint[0] = left;
      hint[1] = (unsigned long long)n;
    }
    if (left == 0ull) return;
    if (sweep == 1 && list && list[1] == 1u && list[0] == 0u) {  // everything left is in the first sweep's segments
      const unsigned int mine = list[2 + blockIdx.x];
      for (unsigned int i = threadIdx.x; i < mine; i += 1024u)
        atomicAdd(&out[list[COUNT_LIST_HEAD + blockIdx.x * COUNT_LIST_SEG + i]], 1u);
      if (blockIdx.x == 0 && threadIdx.x == 0) state[COUNT_MAX_SWEEPS + sweep] = 1ull;
      return;
    }
    if (left * 8ull <= (unsigned long long)n) tail_all = true;
  } else if (hint[1] != 0ull && hint[0] * 8ull <= hint[1]) {
    tail_all = true;
  }
  const bool listing = sweep == 0 && !tail_all && list != nullptr && gridDim.x <= COUNT_MAX_BLOCKS;
  if (threadIdx.x == 0) s_listed = 0u;  // (ordered before its first use by the barrier below)
  auto list_id = [&](int id) {
    const unsigned int at = atomicAdd(&s_listed, 1u);
    if (at < seg_cap) list[COUNT_LIST_HEAD + blockIdx.x * COUNT_LIST_SEG + at] = (unsigned int)id;
  };
  for (int i = threadIdx.x; i < HOT_IDS; i += 1024) s_cnt[i] = 0;
  __syncthreads();
  const long long stride = (long long)gridDim.x * 1024;
  unsigned int beyond = 0;
  // strip: 1 = claims as the table pass wrote them (flags in the top bits), 2 = and the occurrence that created a
  // key is not counted (every counter started at 1: count_ids)
  auto tally = [&](int id, long long t) {
    bool made = false;
    if (strip && id != -1) {
      made = strip == 2 && ((unsigned int)id & AMG_MADE_FLAG) != 0u;
      id = (int)((unsigned int)id & ~AMG_FLAG_MASK);
    }
    if (GATHER) {
      id = id < 0 ? -1 : (remap ? remap[id] : tab[id].id);
      ids[t] = id;
    }
    if (id < 0 || made) return;
    const long long rel = (long long)id - lo;
    if (rel < 0) return;
    if (rel < HOT_IDS) {
      atomicAdd(&s_cnt[rel], 1u);
    } else {
      ++beyond;
      if (tail_all) atomicAdd(&out[id], 1u);
      else if (listing) list_id(id);
    }
  };
  // one block per CU (the counters fill the LDS), so the bytes in flight have to come from the threads themselves:
  // 16-byte loads, four of them in flight per thread (64 MB chip-wide; with 4-byte loads the sweep ran at 2.2 TB/s)
  typedef int i4 __attribute__((ext_vector_type(4)));
  const long long n4 = ((reinterpret_cast<uintptr_t>(ids) & 15) == 0) ? (n >> 2) : 0;
  i4* ids4 = reinterpret_cast<i4*>(ids);
  long long q = (long long)blockIdx.x * 1024 + threadIdx.x;
  auto tally4 = [&](i4 x, long long qi) {
    unsigned int mades = 0;
    if (strip == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (x[j] != -1 && ((unsigned int)x[j] & AMG_MADE_FLAG)) mades |= 1u << j;
    }
    if (GATHER) {
      i4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int id = x[j];
        if (strip && id != -1) id = (int)((unsigned int)id & ~AMG_FLAG_MASK);
        y[j] = id < 0 ? -1 : (remap ? remap[id] : tab[id].id);
      }
      ids4[qi] = y;
      x = y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int id = x[j];
      if (!GATHER && strip && id != -1) id = (int)((unsigned int)id & ~AMG_FLAG_MASK);
      if (id < 0 || (mades & (1u << j))) continue;
      const long long rel = (long long)id - lo;
      if (rel < 0) continue;
      if (rel < HOT_IDS) {
        atomicAdd(&s_cnt[rel], 1u);
      } else {
        ++beyond;
        if (tail_all) atomicAdd(&out[id], 1u);
        else if (listing) list_id(id);
      }
    }
  };
  for (; q + 3 * stride < n4; q += 4 * stride) {
    i4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = GATHER ? ids4[q + j * stride] : __builtin_nontemporal_load(ids4 + q + j * stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) tally4(v[j], q + j * stride);
  }
  for (; q < n4; q += stride) tally4(GATHER ? ids4[q] : __builtin_nontemporal_load(ids4 + q), q);
  // what the 16-byte chunks leave (at most three ids; everything when the array is not 16-byte aligned)
  for (long long t = 4 * n4 + (long long)blockIdx.x * 1024 + threadIdx.x; t < n; t += stride) tally(ids[t], t);
  for (int d = 32; d > 0; d >>= 1) beyond += __shfl_down(beyond, d, 64);
  if ((threadIdx.x & 63) == 0 && beyond) atomicAdd(&state[sweep], (unsigned long long)beyond);
  if (tail_all && blockIdx.x == 0 && threadIdx.x == 0) state[COUNT_MAX_SWEEPS + sweep] = 1ull;
  __syncthreads();
  if (listing && threadIdx.x == 0) {
    const unsigned int got = s_listed;
    list[2 + blockIdx.x] = got < seg_cap ? got : seg_cap;
    if (got > seg_cap) list[0] = 1u;
    if (blockIdx.x == 0) list[1] = 1u;  // "the first sweep listed"
  }
  for (int i = threadIdx.x; i < HOT_IDS; i += 1024) {
    const unsigned int cnt = s_cnt[i];
    if (cnt) atomicAdd(&out[lo + i], cnt);
  }
}

// counts[id] = occurrences of id in ids[0..n); n_ids distinct ids; kind: 0 nodes, 1 edge classes,
// 2 / 3 node / edge-class claims as a table pass wrote them (flag bits on top), 4 / 5 the same with the creating
// occurrence marked (AMG_MADE_FLAG: the exact-key passes)
int count_ids_remap(amg_ctx* c, int* claims, long long n, const int* remap, long long n_ids, unsigned int* out,
                    int edges) {
  return count_ids(c, claims, n, nullptr, n_ids, out, edges ? 3 : 2, remap);
}

int count_ids(amg_ctx* c, int* ids, long long n, const Slot* gather_tab, long long n_ids,
              unsigned int* out, int kind, const int* remap) {
  hipStream_t st = c->stream;
  // kinds 4 / 5: the ids carry AMG_MADE_FLAG on the occurrence that created their key — exactly one per id — so every
  // counter starts at 1 and the sweeps leave those occurrences out: an id seen once costs nothing
  const bool made = kind >= 4;
  ClearList cl;
  cl.add(out, (size_t)(n_ids + 1) * sizeof(unsigned int), made ? 1u : 0u);
  const bool fresh = !c->cnt_state.p || c->cnt_hint_reset;
  // per kind (nodes / edge classes) a block of 2 * COUNT_MAX_SWEEPS state words; the two hints after both blocks
  AMGCHK(c->cnt_state.ensure((4 * COUNT_MAX_SWEEPS + 4) * sizeof(unsigned long long)));
  if (fresh) c->cnt_sweeps[0] = c->cnt_sweeps[1] = COUNT_MAX_SWEEPS;
  c->cnt_hint_reset = false;
  const int kslot = (kind == 1 || kind == 3 || kind == 5) ? 1 : 0;
  unsigned long long* state = c->cnt_state.as<unsigned long long>() + kslot * 2 * COUNT_MAX_SWEEPS;
  unsigned long long* hint = c->cnt_state.as<unsigned long long>() + 4 * COUNT_MAX_SWEEPS + 2 * kslot;
  const int strip = made ? 2 : (kind >= 2 ? 1 : 0);
  cl.add(state, 2 * COUNT_MAX_SWEEPS * sizeof(unsigned long long));
  if (fresh) cl.add(c->cnt_state.as<unsigned long long>() + 4 * COUNT_MAX_SWEEPS, 4 * sizeof(unsigned long long));
  AMGCHK(c->cnt_list.ensure((size_t)(COUNT_LIST_HEAD + COUNT_MAX_BLOCKS * COUNT_LIST_SEG) * sizeof(unsigned int)));
  unsigned int* list = c->cnt_list.as<unsigned int>();
  cl.add(list, COUNT_LIST_HEAD * sizeof(unsigned int));
  unsigned int seg_cap = COUNT_LIST_SEG;  // AMG_COUNT_LIST_SEG: test switch (a small segment runs over: the second launch sweeps)
  if (const char* e = getenv("AMG_COUNT_LIST_SEG")) seg_cap = (unsigned int)std::min(std::max(atoi(e), 0), COUNT_LIST_SEG);
  AMGCHK(clear_many(c, cl));
  if (n <= 0 || n_ids <= 0) return AMG_OK;
  long long ranges = (n_ids + HOT_IDS - 1) / HOT_IDS;
  if (ranges > COUNT_MAX_SWEEPS) ranges = COUNT_MAX_SWEEPS;
  // no more sweeps than the previous count of this kind made use of (count_learn): the last one launched finishes
  // with global atomics whatever is left, so too few sweeps cost time, never counts
  if (ranges > c->cnt_sweeps[kslot]) ranges = c->cnt_sweeps[kslot];
  // every block flushes up to HOT_IDS counters with global atomics at the end of a sweep: give a
  // block at least twice that many ids to count (small inputs: fewer blocks, not a shorter sweep)
  long long want_blocks = (n + 2 * HOT_IDS - 1) / (2 * HOT_IDS);
  unsigned int blocks = (unsigned int)(want_blocks < 1 ? 1 : (want_blocks < COUNT_MAX_BLOCKS ? want_blocks : COUNT_MAX_BLOCKS));
  for (long long r = 0; r < ranges; ++r) {
    const long long lo = r * HOT_IDS;
    const int last = (r == ranges - 1) ? 1 : 0;
    if ((gather_tab || remap) && r == 0)
      hipLaunchKernelGGL(k_count_ids<true>, dim3(blocks), dim3(1024), 0, st, ids, n, gather_tab, lo,
                         (int)r, last, state, hint, out, strip, remap, list, seg_cap);
    else
      hipLaunchKernelGGL(k_count_ids<false>, dim3(blocks), dim3(1024), 0, st, ids, n, gather_tab, lo,
                         (int)r, last, state, hint, out, remap ? 0 : strip, remap, list, seg_cap);
  }
  if (getenv("AMG_COUNT_DEBUG")) {  // what every sweep left beyond its range, which one finished (synchronises: debugging only)
    unsigned long long h[2 * COUNT_MAX_SWEEPS];
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(h, state, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[amg] count kind %d: n %lld ids %lld sweeps %lld beyond %llu %llu %llu %llu done %llu %llu %llu %llu\n", kind, n,
            n_ids, ranges, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  return AMG_OK;
}

__global__ void k_set_pair_ids(const unsigned int* __restrict__ slot_sorted, long long n_pairs,
                               Slot* __restrict__ edge_tab) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pairs) edge_tab[slot_sorted[i]].id = (int)i;
}

__global__ void k_slots_to_ids(const int* __restrict__ slots, long long n, const Slot* __restrict__ tab,
                               int* __restrict__ ids) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int raw = slots[t];
  ids[t] = raw == -1 ? -1 : tab[(unsigned int)raw & ~AMG_LAST_FLAG].id;
}

// Occurrences per table entry without per-window atomics, for the merge path: entries get
// dense ids in first-seen order (slot_sorted), the per-window slots are turned into ids
// (ids_scratch may alias slots) and counted by k_count_ids.  out[i] = count of entry i.
int bs_count_by_slot(amg_ctx* c, const int* slots, int* ids_scratch, long long n, Slot* tab,
                     const unsigned int* slot_sorted, long long n_ids, unsigned int* out, int kind) {
  hipStream_t st = c->stream;
  if (n_ids > 0)
    hipLaunchKernelGGL(k_set_pair_ids, dim3((unsigned)((n_ids + 255) / 256)), dim3(256), 0, st, slot_sorted,
                       n_ids, tab);
  if (n > 0)
    hipLaunchKernelGGL(k_slots_to_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slots, n, tab,
                       ids_scratch);
  return count_ids(c, ids_scratch, n, nullptr, n_ids, out, kind);
}

// ------------------------------------------------------------------ host orchestration
// The build is split into stages so that the multi-GPU path (amg_dist.hip) can put its
// exchanges between them:
//   bs_nodes_pass        local windows -> local node table (+ compaction list in s1 / s3)
//   bs_nodes_rank_local  single GPU: node ids from the local table
//   bs_edges_pass        local adjacencies -> local edge-class table (+ compaction list)
//   bs_pairs_from_local  single GPU: edge classes in first-seen order as arrays
//   bs_finish_from_pairs directed edges, components, adjacency lists
static inline unsigned int blocks_for(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

static int read_status(amg_ctx* c, unsigned long long* host) {
  return fetch_status(c, host);
}

uint64_t pow2_at_least(uint64_t x) {
  uint64_t p = 1024;
  while (p < x) p <<= 1;
  return p;
}

static const unsigned int kProbeLimit = 1024;

// window / short-read counts into the status words + the read-end bitmap the tile kernels use
int bs_read_stats(amg_ctx* c, int k, const ClearList* also) {
  hipStream_t st = c->stream;
  const long long T = c->n_tokens, R = c->n_reads;
  stage_begin(c, "read_stats");
  const size_t words = (size_t)(T >> 5) + BND_PAD_WORDS;
  AMGCHK(c->bnd_bits.ensure(words * sizeof(unsigned int)));
  {  // the read-end bitmap and whatever else the caller wants zeroed before its table pass: one launch
    ClearList cl;
    if (also) cl = *also;
    cl.add(c->bnd_bits.p, words * sizeof(unsigned int));
    AMGCHK(clear_many(c, cl));
  }
  if (R > 0)
    hipLaunchKernelGGL(k_read_stats, dim3(blocks_for(R, 256) < 512u ? blocks_for(R, 256) : 512u), dim3(256), 0, st,
                       c->read_off.as<long long>(), R, T, k, c->status.as<unsigned long long>(),
                       c->bnd_bits.as<unsigned int>());
  stage_end(c);
  return AMG_OK;
}

// returns AMG_OK, or AMG_E_OVERFLOW with *which = 1 (node table too small)
int bs_nodes_pass(amg_ctx* c, int k, int* which) {
  *which = 0;
  hipStream_t st = c->stream;
  const long long T = c->n_tokens;
  unsigned long long hs[ST_WORDS];
  HIPCHK(hipMemsetAsync(c->status.p, 0, ST_WORDS * sizeof(unsigned long long), st));

  AMGCHK(bs_read_stats(c, k));
  const long long n_tiles = (T + TILE - 1) / TILE;

  AMGCHK(c->tok_slot.ensure((size_t)(T + 1) * sizeof(int)));
  AMGCHK(c->tok_node.ensure((size_t)(T + 1) * sizeof(int)));
  AMGCHK(c->tok_dir.ensure((size_t)(T + 1)));
  AMGCHK(c->node_tab.ensure((size_t)c->node_slots * sizeof(Slot)));

  stage_begin(c, "node_table_clear");
  HIPCHK(hipMemsetAsync(c->node_tab.p, 0, (size_t)c->node_slots * sizeof(Slot), st));
  stage_end(c);

  stage_begin(c, "node_upsert");
  if (n_tiles > 0)
    hipLaunchKernelGGL(k_node_upsert, dim3((unsigned)n_tiles), dim3(TILE_THREADS), 0, st,
                       c->tokens.as<int>(), c->bnd_bits.as<unsigned int>(), T, k,
                       c->two_v, c->seed, c->node_tab.as<Slot>(),
                       (unsigned long long)(c->node_slots - 1), kProbeLimit, (long long)c->tok_base,
                       c->tok_slot.as<int>(), c->tok_dir.as<signed char>(),
                       c->status.as<unsigned long long>(), c->count_inline ? 1 : 0,
                       c->weak_fp_builds > 0 ? 0x00000FFF00000000ull : ~0ull);
  stage_end(c);

  stage_begin(c, "node_rank");
  // worst case every slot is occupied; size scratch by min(slots, windows upper bound)
  size_t max_nodes = (size_t)((long long)c->node_slots < T ? c->node_slots : T) + 1;
  AMGCHK(c->s1.ensure(max_nodes * sizeof(unsigned long long)));
  AMGCHK(c->s2.ensure(max_nodes * sizeof(unsigned long long)));
  AMGCHK(c->s3.ensure(max_nodes * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure(max_nodes * sizeof(unsigned int)));
  hipLaunchKernelGGL(k_compact_slots, dim3(blocks_for(c->node_slots, 2048)), dim3(256), 0, st,
                     c->node_tab.as<Slot>(), (unsigned long long)c->node_slots,
                     c->s1.as<unsigned long long>(), c->s3.as<unsigned int>(),
                     c->status.as<unsigned long long>() + ST_COMPACT_A);
  AMGCHK(read_status(c, hs));
  stage_end(c);
  if (hs[ST_BADINPUT])
    return amg_fail(AMG_E_ARG, hs[ST_BADINPUT] == 1 ? "read_offsets must start at 0, never decrease and end at the token count"
                                                    : "a token lies outside [0, two_v)");
  if (hs[ST_PALINDROME])
    return amg_fail(AMG_E_PALINDROME, "Gene-mer and reverse complement gene-mer are identical");
  if (hs[ST_OVERFLOW]) {
    *which = 1;
    return AMG_E_OVERFLOW;
  }
  c->n_windows = (int64_t)hs[ST_N_WINDOWS];
  c->n_short = (int64_t)hs[ST_N_SHORT];
  c->n_local_nodes = (int64_t)hs[ST_COMPACT_A];
  return AMG_OK;
}

int bs_alloc_nodes(amg_ctx* c, long long D) {
  AMGCHK(c->node_tokens.ensure((size_t)(D * c->k + 1) * sizeof(int)));
  AMGCHK(c->node_cov.ensure((size_t)(D + 1) * sizeof(unsigned int)));
  AMGCHK(c->node_first.ensure((size_t)(D + 1) * sizeof(long long)));
  AMGCHK(c->node_comp.ensure((size_t)(D + 1) * sizeof(int)));
  AMGCHK(c->node_alive.ensure((size_t)(D + 1)));
  return AMG_OK;
}

int bs_nodes_rank_local(amg_ctx* c) {
  hipStream_t st = c->stream;
  stage_begin(c, "node_rank");
  c->packed_nodes = (c->two_v <= 65536 && c->k <= AMG_PACK_MAX_K);
  c->n_nodes = c->n_local_nodes;
  const long long D = c->n_nodes;
  int first_bits = ilog2_ceil((uint64_t)(c->tok_total > 0 ? c->tok_total : 1) * 2 + 2) + 1;
  AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->s2.as<unsigned long long>(),
                           c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)D,
                           first_bits));
  AMGCHK(bs_alloc_nodes(c, D));
  if (D > 0)
    hipLaunchKernelGGL(k_assign_nodes, dim3(blocks_for(D, 256)), dim3(256), 0, st,
                       c->s2.as<unsigned long long>(), c->s4.as<unsigned int>(), D,
                       c->node_tab.as<Slot>(), c->tokens.as<int>(), c->k, c->two_v,
                       (long long)c->tok_base, c->packed_nodes ? 1 : 0, c->node_tokens.as<int>(),
                       c->node_cov.as<unsigned int>(), c->node_first.as<long long>(),
                       c->node_alive.as<unsigned char>());
  stage_end(c);
  return AMG_OK;
}

// returns AMG_OK, or AMG_E_OVERFLOW with *which = 2 (edge table) / 3 (fingerprint collision)
int bs_edges_pass(amg_ctx* c, int* which) {
  *which = 0;
  hipStream_t st = c->stream;
  const long long T = c->n_tokens, D = c->n_nodes;
  const long long n_tiles = (T + TILE - 1) / TILE;
  unsigned long long hs[ST_WORDS];
  if (c->edge_slots < (int64_t)slots_for((uint64_t)D)) c->edge_slots = (int64_t)slots_for((uint64_t)D);
  if (!c->count_inline) AMGCHK(c->tok_pair.ensure((size_t)(T + 4) * sizeof(int)));
  AMGCHK(c->edge_tab.ensure((size_t)c->edge_slots * sizeof(Slot)));
  stage_begin(c, "edge_table_clear");
  HIPCHK(hipMemsetAsync(c->edge_tab.p, 0, (size_t)c->edge_slots * sizeof(Slot), st));
  HIPCHK(hipMemsetAsync(c->status.as<unsigned long long>() + ST_OVERFLOW, 0, sizeof(unsigned long long), st));
  HIPCHK(hipMemsetAsync(c->status.as<unsigned long long>() + ST_COMPACT_B, 0, sizeof(unsigned long long), st));
  stage_end(c);
  stage_begin(c, "edge_upsert");
  if (n_tiles > 0)
    hipLaunchKernelGGL(k_edges, dim3((unsigned)n_tiles), dim3(TILE_THREADS), 0, st,
                       c->tokens.as<int>(), T, c->k, c->two_v, c->node_tab.as<Slot>(),
                       c->node_tokens.as<int>(), c->tok_slot.as<int>(),
                       c->tok_dir.as<signed char>(), c->tok_node.as<int>(),
                       c->edge_tab.as<Slot>(), (unsigned long long)(c->edge_slots - 1),
                       kProbeLimit, 1, (long long)c->tok_base, c->status.as<unsigned long long>(),
                       c->count_inline ? 1 : 0, c->count_inline ? (int*)nullptr : c->tok_pair.as<int>(),
                       c->packed_nodes ? 1 : 0);
  stage_end(c);

  stage_begin(c, "edge_rank");
  size_t max_pairs = (size_t)((long long)c->edge_slots < T ? c->edge_slots : T) + 1;
  AMGCHK(c->s1.ensure(max_pairs * sizeof(unsigned long long)));
  AMGCHK(c->s2.ensure(max_pairs * sizeof(unsigned long long)));
  AMGCHK(c->s3.ensure(max_pairs * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure(max_pairs * sizeof(unsigned int)));
  hipLaunchKernelGGL(k_compact_slots, dim3(blocks_for(c->edge_slots, 2048)), dim3(256), 0, st,
                     c->edge_tab.as<Slot>(), (unsigned long long)c->edge_slots,
                     c->s1.as<unsigned long long>(), c->s3.as<unsigned int>(),
                     c->status.as<unsigned long long>() + ST_COMPACT_B);
  AMGCHK(read_status(c, hs));
  stage_end(c);
  if (hs[ST_COLLISION]) {
    *which = 3;
    return AMG_E_OVERFLOW;
  }
  if (hs[ST_OVERFLOW]) {
    *which = 2;
    return AMG_E_OVERFLOW;
  }
  c->n_local_pairs = (int64_t)hs[ST_COMPACT_B];
  if (!c->count_inline && !c->dist_mode) {
    // node coverage (construct_node.py:33-36) from the per-window node ids
    stage_begin(c, "node_count");
    AMGCHK(count_ids(c, c->tok_node.as<int>(), T, nullptr, D, c->node_cov.as<unsigned int>(), 0));
    stage_end(c);
  }
  return AMG_OK;
}

int bs_alloc_pairs(amg_ctx* c, long long P) {
  AMGCHK(c->pair_key.ensure((size_t)(P + 2) * sizeof(unsigned long long)));
  AMGCHK(c->pair_first.ensure((size_t)(P + 2) * sizeof(unsigned long long)));
  AMGCHK(c->pair_cnt.ensure((size_t)(P + 2) * sizeof(unsigned int)));
  return AMG_OK;
}

int bs_pairs_from_local(amg_ctx* c) {
  hipStream_t st = c->stream;
  stage_begin(c, "edge_rank");
  const long long P = c->n_local_pairs;
  c->n_pairs = P;
  AMGCHK(bs_alloc_pairs(c, P));
  int efirst_bits = ilog2_ceil((uint64_t)(c->tok_total > 0 ? c->tok_total : 1) * 8 + 8) + 1;
  AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->pair_first.as<unsigned long long>(),
                           c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)P,
                           efirst_bits));
  if (P > 0)
    hipLaunchKernelGGL(k_gather_pairs, dim3(blocks_for(P, 256)), dim3(256), 0, st,
                       c->s4.as<unsigned int>(), P, c->edge_tab.as<Slot>(),
                       c->pair_key.as<unsigned long long>(), c->pair_cnt.as<unsigned int>());
  stage_end(c);
  if (!c->count_inline && P > 0) {
    // edge-class coverage: pair ids into the table, then count the per-adjacency slots
    stage_begin(c, "edge_count");
    hipLaunchKernelGGL(k_set_pair_ids, dim3(blocks_for(P, 256)), dim3(256), 0, st,
                       c->s4.as<unsigned int>(), P, c->edge_tab.as<Slot>());
    AMGCHK(count_ids(c, c->tok_pair.as<int>(), c->n_tokens, c->edge_tab.as<Slot>(), P,
                     c->pair_cnt.as<unsigned int>(), 1));
    stage_end(c);
  }
  return AMG_OK;
}

// pair_key / pair_cnt / pair_first hold the c->n_pairs edge classes in first-seen order: directed
// edges in _edges order.  Component ids and the forward / backward edge lists are made when
// somebody asks for them (ensure_components / ensure_adjacency; amg_finalize does both): of the three
// graphs of a cleaning sweep only the second needs its components (tip clipping) and none needs the
// lists of removed edges — the correction walks the LIVE adjacency, built from the live edges alone.
int bs_finish_from_pairs(amg_ctx* c) {
  hipStream_t st = c->stream;
  const long long P = c->n_pairs, R = c->n_reads;
  stage_begin(c, "edge_emit");
  AMGCHK(c->s5.ensure((size_t)(P + 2) * sizeof(long long)));
  long long* base = c->s5.as<long long>();
  // at most two directed edges per class: the arrays are sized before the exact count is known
  const long long cap = 2 * P;
  AMGCHK(c->edge_src.ensure((size_t)(cap + 2) * sizeof(int)));
  AMGCHK(c->edge_tgt.ensure((size_t)(cap + 2) * sizeof(int)));
  AMGCHK(c->edge_sdir.ensure((size_t)(cap + 2)));
  AMGCHK(c->edge_tdir.ensure((size_t)(cap + 2)));
  AMGCHK(c->edge_cov.ensure((size_t)(cap + 2) * sizeof(unsigned int)));
  AMGCHK(c->edge_alive.ensure((size_t)(cap + 2)));
  AMGCHK(c->read_fix.ensure((size_t)R + 1));
  {
    ClearList cl;
    cl.add(c->read_fix.p, (size_t)R + 1);
    AMGCHK(clear_many(c, cl));
  }
  long long total = 0;
  if (P > 0) {
    // a class is one directed edge (self-loop) or two: the widths are made by the scan that sums them
    AMGCHK(prim_exscan_pair_width(c, c->pair_key.as<unsigned long long>(), base, (size_t)P));
    hipLaunchKernelGGL(k_emit_edges, dim3(blocks_for(P, 256)), dim3(256), 0, st,
                       c->pair_key.as<unsigned long long>(), c->pair_cnt.as<unsigned int>(),
                       c->pair_first.as<unsigned long long>(), P, base, c->edge_src.as<int>(),
                       c->edge_tgt.as<int>(), c->edge_sdir.as<signed char>(),
                       c->edge_tdir.as<signed char>(), c->edge_cov.as<unsigned int>(),
                       c->edge_alive.as<unsigned char>());
  }
  {  // the build's final synchronisation; the done flags of its counting sweeps ride along
    FetchList l;
    l.add(P > 0 ? static_cast<const void*>(base + P) : c->status.p);
    const bool learn = c->cnt_state.p && !c->cnt_hint_reset;
    if (learn)
      for (int s = 0; s < 2; ++s)
        l.add_words(c->cnt_state.as<unsigned long long>() + s * 2 * COUNT_MAX_SWEEPS + COUNT_MAX_SWEEPS, COUNT_MAX_SWEEPS);
    unsigned long long v[1 + 2 * COUNT_MAX_SWEEPS] = {0};
    AMGCHK(fetch(c, l, v));
    if (P > 0) total = (long long)v[0];
    if (learn)
      for (int s = 0; s < 2; ++s) {
        int used = COUNT_MAX_SWEEPS;
        for (int q = COUNT_MAX_SWEEPS - 1; q >= 0; --q)
          if (v[1 + s * COUNT_MAX_SWEEPS + q]) used = q + 1;
        c->cnt_sweeps[s] = used;
      }
  }
  c->n_edges = total;
  stage_end(c);
  c->ladj_valid = false;
  c->ladj_stale = false;
  c->pristine = !c->comp_from_claims;  // (a filtered build's labels are those of the graph BEFORE its filter)
  c->edge_own_deaths = false;
  c->comp_valid = false;
  c->adj_valid = false;
  c->n_components = 0;
  return AMG_OK;
}

// assign_component_ids (construct_graph.py:920-927) of the graph AS BUILT (all edge classes, whatever was
// removed since: the reference labels once, in __init__)
int ensure_components(amg_ctx* c) {
  if (c->comp_valid) return AMG_OK;
  if (c->comp_from_claims) return bx_components_from_claims(c);  // a filtered build: the UNFILTERED graph's labels
  hipStream_t st = c->stream;
  const long long P = c->n_pairs, D = c->n_nodes;
  stage_begin(c, "components");
  AMGCHK(c->node_comp.ensure((size_t)(D + 1) * sizeof(int)));
  int* parent = c->node_comp.as<int>();  // holds roots until k_uf_label rewrites it
  AMGCHK(c->s1.ensure((size_t)(D + 2) * sizeof(long long)));  // root ranks
  AMGCHK(c->s2.ensure((size_t)(D + 2) * sizeof(unsigned int) + (size_t)(D + 2) * sizeof(int)));
  unsigned int* is_root = c->s2.as<unsigned int>();
  int* root_copy = reinterpret_cast<int*>(is_root + (D + 2));
  long long ncomp = 0;
  if (D > 0) {
    if (P > 0) {
      // runs of consecutive ids first (k_uf_links), the other classes through the union-find
      AMGCHK(c->s3.ensure((size_t)(D + 2) * sizeof(unsigned int)));
      AMGCHK(c->s4.ensure((size_t)(D + 2) * sizeof(int)));
      AMGCHK(c->s5.ensure((size_t)(D + 2) * sizeof(long long)));
      unsigned int* starts = c->s3.as<unsigned int>();
      int* run_start = c->s4.as<int>();
      long long* run_of = c->s5.as<long long>();
      ClearList cl;
      cl.add(starts, (size_t)D * sizeof(unsigned int), 1u);
      AMGCHK(clear_many(c, cl));
      hipLaunchKernelGGL(k_uf_links, dim3(blocks_for(P, 256)), dim3(256), 0, st, c->pair_key.as<unsigned long long>(), P, starts);
      AMGCHK(prim_exscan_u32_to_i64(c, starts, run_of, (size_t)D));
      hipLaunchKernelGGL(k_uf_run_starts, dim3(blocks_for(D, 256)), dim3(256), 0, st, starts, run_of, D, run_start);
      hipLaunchKernelGGL(k_uf_init_runs, dim3(blocks_for(D, 256)), dim3(256), 0, st, starts, run_of, run_start, D, parent);
      hipLaunchKernelGGL(k_uf_union, dim3(blocks_for(P, 256)), dim3(256), 0, st,
                         c->pair_key.as<unsigned long long>(), P, parent);
    } else {
      hipLaunchKernelGGL(k_uf_init, dim3(blocks_for(D, 256)), dim3(256), 0, st, parent, D);
    }
    hipLaunchKernelGGL(k_uf_roots, dim3(blocks_for(D, 256)), dim3(256), 0, st, parent, D, is_root, root_copy);
    AMGCHK(prim_exscan_u32_to_i64(c, is_root, c->s1.as<long long>(), (size_t)D + 1));
    hipLaunchKernelGGL(k_uf_label, dim3(blocks_for(D, 256)), dim3(256), 0, st, root_copy,
                       c->s1.as<long long>(), D, parent);
    FetchList l;
    l.add(c->s1.as<long long>() + D);
    AMGCHK(fetch(c, l, reinterpret_cast<unsigned long long*>(&ncomp)));
  }
  stage_end(c);
  c->n_components = ncomp;
  c->comp_valid = true;
  return AMG_OK;
}

// forwardEdgeHashes / backwardEdgeHashes of every node (construct_node.py:79-101): all edges ever
// inserted, in list order; removed edges stay listed (test `alive`)
int ensure_adjacency(amg_ctx* c) {
  if (c->adj_valid) return AMG_OK;
  hipStream_t st = c->stream;
  const long long D = c->n_nodes, E = c->n_edges;
  stage_begin(c, "adjacency");
  AMGCHK(c->adj_off.ensure((size_t)(2 * D + 2) * sizeof(long long)));
  AMGCHK(c->adj_edge.ensure((size_t)(E + 2) * sizeof(int)));
  AMGCHK(c->s1.ensure((size_t)(E + 2) * sizeof(unsigned int)));
  AMGCHK(c->s2.ensure((size_t)(E + 2) * sizeof(unsigned int)));
  AMGCHK(c->s3.ensure((size_t)(E + 2) * sizeof(unsigned int)));
  const char* force_sort = getenv("AMG_ADJ_SORT");  // A/B switch and test hook: the sorted route for every graph
  if (E > 0 && E <= (4ll << 20) && !(force_sort && force_sort[0] == '1')) {
    AMGCHK(c->s1.ensure((size_t)(2 * D + 2 > E + 2 ? 2 * D + 2 : E + 2) * sizeof(unsigned int)));
    unsigned int* cnt = c->s1.as<unsigned int>();
    unsigned int* tick = c->s2.as<unsigned int>();
    unsigned int* tmp = c->s3.as<unsigned int>();
    AMGCHK(c->s4.ensure((size_t)(E / WAVE_ROW_MAX + 2) * sizeof(unsigned int)));
    unsigned int* long_rows = c->s4.as<unsigned int>();
    unsigned long long* n_long = c->status.as<unsigned long long>() + ST_COMPACT_B;
    {
      ClearList cl;
      cl.add(cnt, (size_t)(2 * D + 2) * sizeof(unsigned int));
      cl.add(n_long, sizeof(unsigned long long));
      AMGCHK(clear_many(c, cl));
    }
    hipLaunchKernelGGL(k_adjc_ticket, dim3(blocks_for(E, 256)), dim3(256), 0, st, c->edge_src.as<int>(),
                       c->edge_sdir.as<signed char>(), E, cnt, tick);
    AMGCHK(prim_exscan_u32_to_i64(c, cnt, c->adj_off.as<long long>(), (size_t)(2 * D + 1)));
    hipLaunchKernelGGL(k_adjc_fill, dim3(blocks_for(E, 256)), dim3(256), 0, st, c->edge_src.as<int>(),
                       c->edge_sdir.as<signed char>(), E, c->adj_off.as<long long>(), tick, tmp);
    hipLaunchKernelGGL(k_adjc_rows, dim3(blocks_for(2 * D, 256)), dim3(256), 0, st, c->adj_off.as<long long>(), 2 * D, tmp,
                       c->adj_edge.as<int>(), long_rows, n_long);
    const long long hub_words = (E + 31) / 32 + 1;  // (scratch of the hub rows: HUB_BLOCKS bitmaps over the edge ids)
    AMGCHK(c->hub_bits.ensure((size_t)HUB_BLOCKS * (size_t)hub_words * sizeof(unsigned int)));
    hipLaunchKernelGGL(k_adjc_long, dim3(256), dim3(256), 0, st, long_rows, n_long, c->adj_off.as<long long>(), tmp,
                       c->adj_edge.as<int>(), c->hub_bits.as<unsigned int>(), hub_words);
    stage_end(c);
    c->adj_valid = true;
    return AMG_OK;
  }
  if (E > 0) {
    hipLaunchKernelGGL(k_adj_keys, dim3(blocks_for(E, 256)), dim3(256), 0, st,
                       c->edge_src.as<int>(), c->edge_sdir.as<signed char>(), E,
                       c->s1.as<unsigned int>(), c->s2.as<unsigned int>());
    AMGCHK(prim_sort_u32_u32(c, c->s1.as<unsigned int>(), c->s3.as<unsigned int>(),
                             c->s2.as<unsigned int>(),
                             reinterpret_cast<unsigned int*>(c->adj_edge.p), (size_t)E,
                             ilog2_ceil((uint64_t)2 * D + 2) + 1));
  }
  hipLaunchKernelGGL(k_row_offsets, dim3(blocks_for(E + 1, 256)), dim3(256), 0, st,
                     c->s3.as<unsigned int>(), E, 2 * D, c->adj_off.as<long long>());
  stage_end(c);
  c->adj_valid = true;
  return AMG_OK;
}

extern "C" int amg_finalize(amg_ctx* c) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (!c->built) return amg_fail(AMG_E_STATE, "amg_build first");
  HIPCHK(hipSetDevice(c->device));
  stages_reset(c);
  AMGCHK(ensure_components(c));
  AMGCHK(ensure_adjacency(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// Slots for n expected keys.  All 64 lanes of a wave wait for the longest probe chain among
// them, so a low load factor pays even when every probe is an L2 hit (measured on 20 000 hot
// keys: 0.99 ms per pass at load 0.31, 0.67 ms at 0.02): 32 slots per key while that stays
// within 4 M slots (a 64 MB table clears in ~25 us), never less than 3 per key.
uint64_t slots_for(uint64_t n) {
  const uint64_t lo = n * 3, hi = n * 32, cap = 4ull << 20;
  const uint64_t want = hi < cap ? hi : cap;
  return pow2_at_least(want > lo ? want : lo);
}

// table sizing: previous distinct-node count when known, otherwise the window bound
void bs_size_tables(amg_ctx* c) {
  // no history: a quarter of a slot per token.  Real gene-call data repeat every gene-mer tens to
  // thousands of times, so this is already generous; inputs with more distinct gene-mers than
  // that overflow once (cheaply, see table_upsert's abort flag) and are rebuilt 4x larger.
  uint64_t want = c->node_hint > 0 ? slots_for((uint64_t)c->node_hint) : (uint64_t)c->n_tokens / 4;
  c->node_slots = (int64_t)pow2_at_least(want);
  if (c->node_slots > (1ll << 30)) c->node_slots = 1ll << 30;
  c->edge_slots = 1024;
}

static int build_impl(amg_ctx* c, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov, bool* fused);

extern "C" int amg_build(amg_ctx* c, int32_t k) {
  bool fused = false;
  return build_impl(c, k, 0, 0, &fused);
}

// GeneMerGraph.__init__ followed by filter_graph(min_node_cov, min_edge_cov) (graph_utils.py:147-149 — what every
// cleaning iteration does with a freshly built graph).  On the exact-key path the filter is applied ON THE WAY:
// nodes below the threshold never get an id, an array entry or an edge (an uncorrected graph is ~99 % such
// nodes), their windows read None and their reads are queued for correction — the state a caller of
// amg_build + amg_filter finds, except that ids number the survivors only (first-seen order among them).
// Elsewhere (fingerprint keys) it IS amg_build + amg_filter.
extern "C" int amg_build_filtered(amg_ctx* c, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov) {
  bool fused = false;
  const int r = build_impl(c, k, min_node_cov < 1 ? 1 : min_node_cov, min_edge_cov < 1 ? 1 : min_edge_cov, &fused);
  if (r != AMG_OK || fused) return r;
  return amg_filter(c, min_node_cov, min_edge_cov);
}

static int build_impl(amg_ctx* c, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov, bool* fused) {
  *fused = false;
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (k < 1 || k > AMG_MAX_K) return amg_fail(AMG_E_ARG, "k must be in [1, %d]", AMG_MAX_K);
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  HIPCHK(hipSetDevice(c->device));
  stages_reset(c);
  // the reads are what the last correction left of the reads of the graph still held, nothing re-threaded: that graph's
  // live part IS the graph to build (amg_derive.hip; AMG_NO_DERIVE=1: A/B + test switch)
  const bool derive = c->derive_ready && k == c->k && !c->dist_mode && !getenv("AMG_NO_DERIVE");
  c->derive_ready = c->dist_candidate = false;
  c->derived = false;
  c->built = false;
  c->have_corrected = false;
  c->match_valid = false;
  c->retries = 0;
  c->tok_base = 0;
  c->tok_total = c->n_tokens;
  if (derive) {
    bool done = false;
    AMGCHK(derive_from_previous(c, k, &done));
    if (done) return AMG_OK;  // (amg_build_filtered goes on with amg_filter)
  }
  c->k = k;
  c->dist_mode = false;
  c->comp_from_claims = false;
  if (c->hint_bound > 0) {  // reads taken over from another ctx's correction: its bound holds for a graph at ITS gene-mer size
    if (c->hint_bound_k == k) c->node_hint = c->hint_bound > 256 ? c->hint_bound : 256;
    c->hint_bound = 0;
  }
  {
    // test hook: the first AMG_TEST_WEAK_FP attempts use a 12-bit fingerprint, which is
    // certain to collide; the exact verification must catch it and the retry must succeed
    const char* e = getenv("AMG_TEST_WEAK_FP");
    c->weak_fp_builds = e ? atoi(e) : 0;
  }
  {
    const char* e = getenv("AMG_COUNT_INLINE");  // A/B switch: 1 = one global atomic per window
    c->count_inline = e && e[0] == '1';
  }
  bs_size_tables(c);
  c->exact_keys = false;
  const bool exact = bx_applicable(c, k);  // tuple fits the slot: exact keys + claim ids
  for (int attempt = 0; attempt < 12; ++attempt) {
    int which = 0;
    int r;
    if (exact && min_node_cov > 0) {
      r = bx_nodes_filtered(c, k, min_node_cov, &which);
      if (r == AMG_OK) r = bx_edges(c, &which, min_edge_cov);
      *fused = true;
    } else if (exact) {
      r = bx_nodes(c, k, &which);
      if (r == AMG_OK) r = bx_edges(c, &which);
    } else {
      r = bs_nodes_pass(c, k, &which);
      if (r == AMG_OK) r = bs_nodes_rank_local(c);
      if (r == AMG_OK) r = bs_edges_pass(c, &which);
      if (r == AMG_OK) r = bs_pairs_from_local(c);
    }
    if (r == AMG_OK) r = bs_finish_from_pairs(c);
    if (r == AMG_OK && *fused) r = bx_flag_dead_reads(c);
    if (r == AMG_OK) {
      c->built = true;
      // (a filtered build keeps only the survivors: the next table is sized by what the pass saw)
      const int64_t seen = *fused ? c->n_local_nodes : c->n_nodes;
      c->node_hint = seen > 256 ? seen : 256;
      return AMG_OK;
    }
    if (r != AMG_E_OVERFLOW || which == 0) return r;
    ++c->retries;
    if (which == 1) {
      if (c->node_slots >= (1ll << 30)) return amg_fail(AMG_E_OVERFLOW, "node table at maximum size");
      c->node_slots = c->node_slots * 4 > (1ll << 30) ? (1ll << 30) : c->node_slots * 4;
    } else if (which == 2) {
      c->edge_slots *= 4;
    } else {
      c->seed = c->seed * 6364136223846793005ull + 1442695040888963407ull;  // new fingerprint
      if (c->weak_fp_builds > 0) --c->weak_fp_builds;
    }
  }
  return amg_fail(AMG_E_OVERFLOW, "build did not converge after 12 attempts");
}

// ------------------------------------------------------------------ several k over one read set
// choose_kmer_size (graph_utils.py:258-296) builds the graph of the SAME reads for k = 3, 5, ..., 15.  amg_build_multi
// puts the reads on the device ONCE — graph i's ctx borrows the first ctx's token arrays — and builds every graph with
// the ordinary amg_build on its own ctx, one after the other on the first ctx's stream.  (Rounds 2-5 also staged every
// tile of tokens once for the node passes of all k and once for their edge passes, k_node_upsert_multi / k_edges_multi:
// measured on cfg 3, `multi_k` in bench.py, the seven graphs took 28.2 ms that way and 28.0 ms as seven builds — the
// table work of a pass, not the reading of the tokens, is what a build costs — so those kernels are gone.)
#define MULTI_MAX 8
extern "C" int amg_build_multi(amg_ctx* const* ctxs, const int32_t* ks, int32_t n) {
  if (!ctxs || !ks || n < 1 || n > MULTI_MAX) return amg_fail(AMG_E_ARG, "amg_build_multi: 1 .. %d graphs", MULTI_MAX);
  amg_ctx* c0 = ctxs[0];
  if (!c0 || c0->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads on the first ctx first");
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return amg_fail(AMG_E_ARG, "null ctx");
    if (ctxs[i]->device != c0->device) return amg_fail(AMG_E_ARG, "amg_build_multi: one device");
    if (ks[i] < 1 || ks[i] > AMG_MAX_K) return amg_fail(AMG_E_ARG, "k must be in [1, %d]", AMG_MAX_K);
    for (int j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return amg_fail(AMG_E_ARG, "amg_build_multi: one ctx per graph");
  }
  HIPCHK(hipSetDevice(c0->device));
  HIPCHK(hipStreamSynchronize(c0->stream));
  const long long T = c0->n_tokens, R = c0->n_reads;
  // the other graphs read the first ctx's token arrays (borrowed: nothing is copied) and, for the length of this
  // call, work on its stream
  std::vector<hipStream_t> own(n);
  for (int i = 0; i < n; ++i) {
    amg_ctx* c = ctxs[i];
    own[i] = c->stream;
    if (i == 0) continue;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->tokens.borrow(c0->tokens.p, (size_t)T * sizeof(int32_t));
    c->read_off.borrow(c0->read_off.p, (size_t)(R + 1) * sizeof(int64_t));
    c->n_reads = R;
    c->n_tokens = T;
    c->two_v = c0->two_v;
    c->have_pos = c->have_read_len = false;
    c->node_hint = 0;
    c->cnt_hint_reset = true;
    c->derive_ready = c->dist_candidate = false;
    c->stream = c0->stream;
  }
  struct Restore {
    amg_ctx* const* ctxs;
    std::vector<hipStream_t>& own;
    int n;
    ~Restore() {
      for (int i = 0; i < n; ++i) ctxs[i]->stream = own[i];
    }
  } restore{ctxs, own, n};
  for (int i = 0; i < n; ++i) AMGCHK(amg_build(ctxs[i], ks[i]));
  HIPCHK(hipStreamSynchronize(c0->stream));
  return AMG_OK;
}
