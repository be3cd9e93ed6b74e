// amg_passes.hip — coverage filter, node removal, tip clipping, component filter and
// per-read correction on the device (reference construct_graph.py:402-540, 679-861,
// 950-958, 1123-1396).
#include "amg_device.h"

#define NEED_BUILT(c)                                                     \
  do {                                                                    \
    if (!(c)) return amg_fail(AMG_E_ARG, "null ctx");                     \
    if (!(c)->built) return amg_fail(AMG_E_STATE, "amg_build first");     \
    HIPCHK(hipSetDevice((c)->device));                                    \
  } while (0)

static inline unsigned int nblk(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

// ------------------------------------------------------------------ filter_graph (:523-540)
// list_nodes_to_remove (:496-503): coverage < minNodeCoverage
__global__ void k_filter_nodes(const unsigned int* __restrict__ cov, unsigned char* __restrict__ alive,
                               long long n, unsigned int min_cov) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && alive[i] && cov[i] < min_cov) alive[i] = 0;
}

// list_edges_to_remove (:505-521): coverage < minEdgeCoverage or a doomed endpoint
__global__ void k_filter_edges(const int* __restrict__ src, const int* __restrict__ tgt,
                               const unsigned int* __restrict__ cov,
                               const unsigned char* __restrict__ node_alive,
                               unsigned char* __restrict__ alive, long long n, unsigned int min_cov) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || !alive[e]) return;
  // (an edge's coverage is at least 1: with the usual threshold of 1 the coverage array is not read at all)
  if ((min_cov > 1 && cov[e] < min_cov) || !node_alive[src[e]] || !node_alive[tgt[e]]) alive[e] = 0;
}

// remove_node_from_reads (:442-461): one wave per read; windows of removed nodes become
// None (-2) and the read joins _readsToCorrect
// Wave-per-read kernels move ~60 windows per read: one read per wave is bound by the chain
// offsets -> ids -> flags of a single short read.  Each wave therefore takes READS_PER_WAVE
// consecutive reads and issues every load of one stage for all of them before using any.
#define READS_PER_WAVE 8  // (2 / 4 / 8: filter stage 0.27 / 0.22 / 0.21 ms)
__global__ __launch_bounds__(256) void k_mask_reads(int* __restrict__ tok_node,
                                                    const long long* __restrict__ read_off,
                                                    long long n_reads,
                                                    const unsigned char* __restrict__ node_alive,
                                                    unsigned char* __restrict__ read_fix) {
  const long long rbase = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * READS_PER_WAVE;
  if (rbase >= n_reads) return;
  const int lane = threadIdx.x & 63;
  const long long off_l = (lane <= READS_PER_WAVE && rbase + lane <= n_reads) ? read_off[rbase + lane] : 0;
  long long a[READS_PER_WAVE], b[READS_PER_WAVE];
  int v[READS_PER_WAVE];
#pragma unroll
  for (int j = 0; j < READS_PER_WAVE; ++j) {
    a[j] = __shfl(off_l, j, 64);
    b[j] = rbase + j < n_reads ? __shfl(off_l, j + 1, 64) : a[j];
  }
#pragma unroll
  for (int j = 0; j < READS_PER_WAVE; ++j) v[j] = a[j] + lane < b[j] ? tok_node[a[j] + lane] : -1;
  bool dead[READS_PER_WAVE];
#pragma unroll
  for (int j = 0; j < READS_PER_WAVE; ++j) dead[j] = v[j] >= 0 && !node_alive[v[j]];
#pragma unroll
  for (int j = 0; j < READS_PER_WAVE; ++j) {
    if (dead[j]) tok_node[a[j] + lane] = -2;
    bool hit = dead[j];
    for (long long t = a[j] + 64 + lane; t < b[j]; t += 64) {  // reads longer than one wave
      const int n = tok_node[t];
      if (n >= 0 && !node_alive[n]) {
        tok_node[t] = -2;
        hit = true;
      }
    }
    if (__any(hit) && lane == 0) read_fix[rbase + j] = 1;
  }
}

static int apply_removals(amg_ctx* c, unsigned int min_edge_cov) {
  hipStream_t st = c->stream;
  // NODES died, and every edge that dies with them has a dead end (no edge falls to a coverage threshold of its own):
  // live lists that exist stay and are brought up to date when somebody walks them again (ensure_live_adj: k_lr_patch)
  if (c->ladj_valid && min_edge_cov <= 1 && !getenv("AMG_NO_LADJ_PATCH")) c->ladj_stale = true;  // (A/B + test switch)
  else if (!c->ladj_stale || min_edge_cov > 1) c->ladj_stale = false;
  c->ladj_valid = false;
  c->pristine = false;
  if (min_edge_cov > 1) c->edge_own_deaths = true;
  c->match_valid = false;  // node-id patterns of a cached K6 result may name removed nodes
  if (c->n_edges > 0)
    hipLaunchKernelGGL(k_filter_edges, dim3(nblk(c->n_edges, 256)), dim3(256), 0, st,
                       c->edge_src.as<int>(), c->edge_tgt.as<int>(), c->edge_cov.as<unsigned int>(),
                       c->node_alive.as<unsigned char>(), c->edge_alive.as<unsigned char>(),
                       c->n_edges, min_edge_cov);
  if (c->n_reads > 0)
    hipLaunchKernelGGL(k_mask_reads, dim3(nblk(c->n_reads, 4 * READS_PER_WAVE)), dim3(256), 0, st,
                       c->tok_node.as<int>(), c->read_off.as<long long>(), c->n_reads,
                       c->node_alive.as<unsigned char>(), c->read_fix.as<unsigned char>());
  return AMG_OK;
}

extern "C" int amg_filter(amg_ctx* c, uint32_t min_node_cov, uint32_t min_edge_cov) {
  NEED_BUILT(c);
  stages_reset(c);
  stage_begin(c, "filter");
  if (c->n_nodes > 0)
    hipLaunchKernelGGL(k_filter_nodes, dim3(nblk(c->n_nodes, 256)), dim3(256), 0, c->stream,
                       c->node_cov.as<unsigned int>(), c->node_alive.as<unsigned char>(),
                       c->n_nodes, min_node_cov);
  AMGCHK(apply_removals(c, min_edge_cov));
  stage_end(c);
  c->have_corrected = false;
  return AMG_OK;
}

// ------------------------------------------------------------------ remove_node (:463-484)
__global__ void k_kill_listed(const int* __restrict__ ids, long long n, long long n_nodes,
                              unsigned char* __restrict__ alive) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v = ids[i];
  if (v >= 0 && v < n_nodes) alive[v] = 0;
}

extern "C" int amg_remove_nodes(amg_ctx* c, const int32_t* node_ids, int64_t n) {
  NEED_BUILT(c);
  if (n < 0 || (n > 0 && !node_ids)) return amg_fail(AMG_E_ARG, "bad node list");
  if (n == 0) return AMG_OK;
  AMGCHK(c->s0.ensure((size_t)n * sizeof(int)));
  HIPCHK(hipMemcpyAsync(c->s0.p, node_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_kill_listed, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->s0.as<int>(),
                     (long long)n, c->n_nodes, c->node_alive.as<unsigned char>());
  AMGCHK(apply_removals(c, 0));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->have_corrected = false;
  return AMG_OK;
}


// ------------------------------------------------------------------ graph view for walkers
// (GView: amg_internal.h)

// Live adjacency straight from the live edges: flag + scan squeezes the removed edges out in edge
// order, a stable sort by row (2 * source + side) groups them, so a row lists its live edges in
// the order of the reference's forward / backward lists (edge ids ascend in insertion order).
__global__ void k_live_keys(const unsigned char* __restrict__ e_alive, const int* __restrict__ e_src,
                            const signed char* __restrict__ e_sdir, const long long* __restrict__ pos,
                            long long n_edges, unsigned int* __restrict__ keys, unsigned int* __restrict__ vals) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges || !e_alive[e]) return;
  const long long o = pos[e];
  keys[o] = 2u * (unsigned int)e_src[e] + (e_sdir[e] > 0 ? 0u : 1u);
  vals[o] = (unsigned int)e;
}

// Rows from the dense list of live edges WITHOUT a sort (a library radix sort of a few ten thousand pairs is a
// dozen launches).  The live edges of a row are few (one side of one node), so:
//   k_lr_count   every live edge adds 1 to its row's counter                          (rows zeroed before)
//   k_lr_alloc   every live edge takes a ticket of its row; ticket 0 reserves the row's stretch of the entry
//                array — stretches are handed out per workgroup with ONE atomicAdd (their order is irrelevant)
//   k_lr_fill    every live edge drops its edge id into its row's stretch at its ticket (any order)
//   k_lr_finish  the ticket-0 edge of a row sorts the row's ids ascending (= list order of the reference: edge ids
//                follow insertion order) and writes the entries {target, direction} and the row record; rows longer
//                than a wave's 64 lanes are left to k_lr_long, a workgroup per long row (hub nodes)
__global__ void k_lr_count(const unsigned int* __restrict__ keys, long long n_live, int4* __restrict__ lrows) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_live) atomicAdd(&lrows[keys[i]].y, 1);
}

// (LR_PER edges per thread: the pool word takes ~90 returning atomics per microsecond, and a workgroup per 256 of a
// million live edges spent 45 of this kernel's 64 us queueing there)
#define LR_PER 8
__global__ __launch_bounds__(256) void k_lr_alloc(const unsigned int* __restrict__ keys, long long n_live,
                                                  int4* __restrict__ lrows, unsigned int* __restrict__ tick,
                                                  unsigned long long* pool) {
  __shared__ unsigned int s_wave[4];
  __shared__ unsigned long long s_base;
  const long long i0 = (long long)blockIdx.x * (256 * LR_PER) + threadIdx.x;
  unsigned int cnt[LR_PER], key[LR_PER], mine = 0;
#pragma unroll
  for (int j = 0; j < LR_PER; ++j) {
    const long long i = i0 + (long long)j * 256;
    cnt[j] = 0;
    key[j] = 0;
    if (i < n_live) {
      key[j] = keys[i];
      const unsigned int t = (unsigned int)atomicAdd(&lrows[key[j]].z, 1);
      tick[i] = t;
      if (t == 0) cnt[j] = (unsigned int)lrows[key[j]].y | 0x80000000u;  // final: k_lr_count is a launch of its own
      mine += cnt[j] & 0x7fffffffu;
    }
  }
  unsigned int total;
  unsigned int off = block_exscan_256(mine, &total, s_wave);
  if (threadIdx.x == 0) s_base = total ? atomicAdd(pool, (unsigned long long)total) : 0ull;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < LR_PER; ++j)
    if (cnt[j] & 0x80000000u) {  // this edge drew ticket 0 of its row: it places the row's stretch
      lrows[key[j]].x = (int)(s_base + off);
      off += cnt[j] & 0x7fffffffu;
    }
}

// (a row with ONE live edge — most rows: a node of a cleaned graph has one forward and one backward neighbour — is
// finished right here: entry and row record written by the edge that drew its only ticket, which is marked done; the
// ticket-0 pass that puts rows in order then touches the rows with two edges and more alone.  On the 7.4 M rows of a
// rebuilt graph of eight read sets that pass took 0.92 ms with every row going through it.)
#define LR_DONE 0xffffffffu
__global__ void k_lr_fill(const unsigned int* __restrict__ keys, const unsigned int* __restrict__ edge_of,
                          unsigned int* __restrict__ tick, long long n_live, int4* __restrict__ lrows,
                          const int* __restrict__ e_tgt, const signed char* __restrict__ e_tdir,
                          unsigned int* __restrict__ tmp, int2* __restrict__ lent) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_live) return;
  const unsigned int key = keys[i], e = edge_of[i];
  const int4 rw = lrows[key];
  if (rw.y == 1) {
    const int t = e_tgt[e], d = (int)e_tdir[e];
    lent[rw.x] = make_int2(t, d);
    lrows[key] = make_int4(rw.x, 1, t, d);
    tick[i] = LR_DONE;
    return;
  }
  tmp[rw.x + tick[i]] = e;
}

__global__ __launch_bounds__(256) void k_lr_finish(const unsigned int* __restrict__ keys, const unsigned int* __restrict__ tick,
                                                   long long n_live, const unsigned int* __restrict__ tmp,
                                                   const int* __restrict__ e_tgt, const signed char* __restrict__ e_tdir,
                                                   int4* __restrict__ lrows, int2* __restrict__ lent,
                                                   unsigned int* __restrict__ long_rows, unsigned long long* n_long) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool mine = i < n_live && tick[i] == 0u;  // the ticket-0 edge of a row with two live edges or more
  unsigned int key = 0;
  int off = 0, cnt = 0;
  if (mine) {
    key = keys[i];
    off = lrows[key].x;
    cnt = lrows[key].y;
    if (cnt > WAVE_ROW_MAX) {
      long_rows[atomicAdd(n_long, 1ull)] = key;
      mine = false;
    } else if (cnt == 2) {  // (most of what is left)
      unsigned int a = tmp[off], b = tmp[off + 1];
      if (a > b) {
        const unsigned int t = a;
        a = b;
        b = t;
      }
      const int ta = e_tgt[a], da = (int)e_tdir[a];
      lent[off] = make_int2(ta, da);
      lent[off + 1] = make_int2(e_tgt[b], (int)e_tdir[b]);
      lrows[key] = make_int4(off, 2, ta, da);
      mine = false;
    }
  }
  wave_rows_in_order(mine, key, (long long)off, cnt, tmp, [&](unsigned int row, long long o, int n, int rank, unsigned int x) {
    const int t = e_tgt[x], d = (int)e_tdir[x];
    lent[o + rank] = make_int2(t, d);
    if (rank == 0) lrows[row] = make_int4((int)o, n, t, d);
  });
}

// a workgroup per long row: every element finds its rank among the row's (distinct) edge ids; rows beyond HUGE_ROW are
// left to the first LR_HUB_BLOCKS workgroups, which put them in order through a bitmap (huge_row_in_order, amg_device.h)
#define LR_HUB_BLOCKS 8
__global__ __launch_bounds__(256) void k_lr_long(const unsigned int* __restrict__ long_rows,
                                                 const unsigned long long* __restrict__ n_long,
                                                 const unsigned int* __restrict__ tmp, const int* __restrict__ e_tgt,
                                                 const signed char* __restrict__ e_tdir, int4* __restrict__ lrows,
                                                 int2* __restrict__ lent, unsigned int* hub_bits, long long hub_words) {
  __shared__ unsigned int s_wave[4];
  const unsigned long long n = *n_long;
  for (unsigned long long r = blockIdx.x; r < n; r += gridDim.x) {
    const unsigned int key = long_rows[r];
    const int off = lrows[key].x, cnt = lrows[key].y;
    __syncthreads();  // (the row record is rewritten below: everybody has read it)
    if (cnt > HUGE_ROW) continue;
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const unsigned int x = tmp[off + j];
      int rank = 0;
      for (int q = 0; q < cnt; ++q) rank += tmp[off + q] < x ? 1 : 0;
      lent[off + rank] = make_int2(e_tgt[x], (int)e_tdir[x]);
      if (rank == 0) lrows[key] = make_int4(off, cnt, e_tgt[x], (int)e_tdir[x]);
    }
  }
  if (blockIdx.x >= LR_HUB_BLOCKS) return;
  for (unsigned long long r = blockIdx.x; r < n; r += LR_HUB_BLOCKS) {  // (block-uniform)
    const unsigned int key = long_rows[r];
    const int off = lrows[key].x, cnt = lrows[key].y;
    if (cnt <= HUGE_ROW) continue;  // (huge_row_in_order starts with a barrier: the record has been read by then)
    huge_row_in_order(tmp + off, (long long)cnt, hub_bits + (long long)blockIdx.x * hub_words, hub_words, s_wave,
                      [&](long long rank, unsigned int x) {
                        lent[off + rank] = make_int2(e_tgt[x], (int)e_tdir[x]);
                        if (rank == 0) lrows[key] = make_int4(off, cnt, e_tgt[x], (int)e_tdir[x]);
                      });
  }
}

// forward/backward edge lists with the removed edges squeezed out: the walkers below then
// never touch a dead edge (a hub node of an uncorrected graph lists hundreds of them), and the
// lists of the removed edges are never made at all
// The live lists after NODES died: a row of a dead node empties, the other rows drop their entries with a dead
// target, in place and in order — one pass over the row records with the targets' alive bytes out of the L2, where
// making the lists again from the live edges is a flag, a scan and five passes over them
__global__ void k_lr_patch(int4* __restrict__ lrows, int2* __restrict__ lent, long long n_rows,
                           const unsigned char* __restrict__ n_alive) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int4 rw = lrows[r];
  if (rw.y == 0) return;
  if (!n_alive[r >> 1]) {
    lrows[r] = make_int4(rw.x, 0, 0, 0);
    return;
  }
  if (rw.y == 1) {
    if (!n_alive[rw.z]) lrows[r] = make_int4(rw.x, 0, 0, 0);
    return;
  }
  int w = 0;
  int2 first = make_int2(0, 0);
  for (int j = 0; j < rw.y; ++j) {
    const int2 e = lent[rw.x + j];
    if (!n_alive[e.x]) continue;
    if (w != j) lent[rw.x + w] = e;
    if (w == 0) first = e;
    ++w;
  }
  if (w != rw.y) lrows[r] = make_int4(rw.x, w, first.x, first.y);
}

int ensure_live_adj(amg_ctx* c) {
  if (c->ladj_valid) return AMG_OK;
  hipStream_t st = c->stream;
  if (c->ladj_stale) {
    hipLaunchKernelGGL(k_lr_patch, dim3(nblk(2 * c->n_nodes, 256)), dim3(256), 0, st, c->ladj_rows.as<int4>(),
                       c->ladj.as<int2>(), 2 * c->n_nodes, c->node_alive.as<unsigned char>());
    c->ladj_stale = false;
    c->ladj_valid = true;
    return AMG_OK;
  }
  const long long rows = 2 * c->n_nodes, E = c->n_edges;
  AMGCHK(c->ladj_rows.ensure((size_t)(rows + 2) * sizeof(int4)));
  AMGCHK(c->ladj_pos.ensure((size_t)(E + 2) * sizeof(long long)));
  unsigned long long* ctr = c->status.as<unsigned long long>() + ST_COMPACT_A;  // [0] entries handed out, [1] long rows
  {
    ClearList cl;
    cl.add(c->ladj_rows.p, (size_t)(rows + 2) * sizeof(int4));
    cl.add(ctr, 2 * sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
  }
  AMGCHK(prim_exscan_bytes_set(c, c->edge_alive.as<unsigned char>(), c->ladj_pos.as<long long>(), (size_t)E));
  long long total = 0;
  {
    FetchList l;
    l.add(c->ladj_pos.as<long long>() + E);
    AMGCHK(fetch(c, l, reinterpret_cast<unsigned long long*>(&total)));
  }
  AMGCHK(c->ladj.ensure((size_t)(total + 1) * sizeof(int2)));
  AMGCHK(c->ladj_keys.ensure(5 * (size_t)(total + 2) * sizeof(unsigned int)));
  unsigned int* keys = c->ladj_keys.as<unsigned int>();
  unsigned int* edge_of = keys + (total + 2);
  unsigned int* tick = edge_of + (total + 2);
  unsigned int* tmp = tick + (total + 2);
  unsigned int* long_rows = tmp + (total + 2);
  if (total > 0) {
    hipLaunchKernelGGL(k_live_keys, dim3(nblk(E, 256)), dim3(256), 0, st, c->edge_alive.as<unsigned char>(),
                       c->edge_src.as<int>(), c->edge_sdir.as<signed char>(), c->ladj_pos.as<long long>(), E, keys,
                       edge_of);
    hipLaunchKernelGGL(k_lr_count, dim3(nblk(total, 256)), dim3(256), 0, st, keys, total, c->ladj_rows.as<int4>());
    hipLaunchKernelGGL(k_lr_alloc, dim3(nblk(total, 256 * LR_PER)), dim3(256), 0, st, keys, total, c->ladj_rows.as<int4>(),
                       tick, ctr);
    hipLaunchKernelGGL(k_lr_fill, dim3(nblk(total, 256)), dim3(256), 0, st, keys, edge_of, tick, total,
                       c->ladj_rows.as<int4>(), c->edge_tgt.as<int>(), c->edge_tdir.as<signed char>(), tmp, c->ladj.as<int2>());
    hipLaunchKernelGGL(k_lr_finish, dim3(nblk(total, 256)), dim3(256), 0, st, keys, tick, total, tmp, c->edge_tgt.as<int>(),
                       c->edge_tdir.as<signed char>(), c->ladj_rows.as<int4>(), c->ladj.as<int2>(), long_rows, ctr + 1);
    const long long hub_words = (E + 31) / 32 + 1;  // (scratch of the hub rows: LR_HUB_BLOCKS bitmaps over the edge ids)
    AMGCHK(c->hub_bits.ensure((size_t)LR_HUB_BLOCKS * (size_t)hub_words * sizeof(unsigned int)));
    hipLaunchKernelGGL(k_lr_long, dim3(64), dim3(256), 0, st, long_rows, ctr + 1, tmp, c->edge_tgt.as<int>(),
                       c->edge_tdir.as<signed char>(), c->ladj_rows.as<int4>(), c->ladj.as<int2>(),
                       c->hub_bits.as<unsigned int>(), hub_words);
  }
  c->ladj_valid = true;
  return AMG_OK;
}

GView make_view(amg_ctx* c) {
  GView g;
  g.lent = c->ladj.as<int2>();
  g.lrows = c->ladj_rows.as<int4>();
  g.n_alive = c->node_alive.as<unsigned char>();
  g.n_cov = c->node_cov.as<unsigned int>();
  g.n_tok = c->node_tokens.as<int>();
  g.n_first = c->node_first.as<long long>();
  g.n_comp = c->node_comp.as<int>();
  g.k = c->k;
  g.flip = c->two_v - 1;
  return g;
}

// get_degree (:326-329): live edge classes on both sides
__device__ __forceinline__ int node_degree(const GView& g, int n) {
  return g.lrows[2ll * n].y + g.lrows[2ll * n + 1].y;
}

// get_forward_node_from_node (:722-741) / get_backward_node_from_node (:781-802):
// forward needs EXACTLY one live forward edge, backward takes the FIRST live backward edge.
// returns 0 = no edge, 1 = edge but cannot extend, 2 = extend
__device__ __forceinline__ int lin_step(const GView& g, int n, bool use_forward, int* tgt, int* tdir) {
  const long long row = 2ll * n + (use_forward ? 0 : 1);
  const int4 rw = g.lrows[row];
  if (rw.y == 0 || (use_forward && rw.y != 1)) return 0;
  *tgt = rw.z;
  *tdir = rw.w;
  const int deg = node_degree(g, *tgt);
  return ((deg == 1 || deg == 2) && *tgt != n) ? 2 : 1;
}

// ------------------------------------------------------------------ remove_edge (:409-428)
// one DIRECTED edge per listed id leaves the graph (and its source node's forward / backward list); its reverse
// twin stays, as in the reference, until it is removed by its own call
extern "C" int amg_remove_edges(amg_ctx* c, const int32_t* edge_ids, int64_t n) {
  NEED_BUILT(c);
  if (n < 0 || (n > 0 && !edge_ids)) return amg_fail(AMG_E_ARG, "bad edge list");
  if (n == 0) return AMG_OK;
  AMGCHK(c->s0.ensure((size_t)n * sizeof(int)));
  HIPCHK(hipMemcpyAsync(c->s0.p, edge_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_kill_listed, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->s0.as<int>(), (long long)n,
                     c->n_edges, c->edge_alive.as<unsigned char>());
  c->ladj_valid = false;
  c->pristine = false;
  c->edge_own_deaths = true;
  c->ladj_stale = false;  // (an edge left with both ends alive: the lists are made again)
  c->match_valid = false;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->have_corrected = false;
  return AMG_OK;
}

// ------------------------------------------------------------------ remove_short_linear_paths (:679-720)
#define CLIP_MAX 64
// acc = {sum of the live nodes' coverages, number of live nodes} (k_comp_hist): the threshold is 1.5 x their mean
// (:868-871, statistics.mean over live nodes) — the quotient of the two integers as doubles is correctly rounded,
// == float(Fraction(sum, n)), on the device as on the host
__global__ void k_clip_mark(GView g, long long n_nodes, int min_length, const unsigned long long* __restrict__ acc,
                            const unsigned int* __restrict__ comp_live,
                            const unsigned char* __restrict__ protect,
                            unsigned char* __restrict__ kill) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || !g.n_alive[i]) return;
  int n = (int)i;
  if (node_degree(g, n) != 1) return;
  int d0 = (g.n_first[n] & 1ll) ? -1 : 1;  // direction of the node's first occurrence (:852-858)
  int path[CLIP_MAX];
  int len = 0;
  path[len++] = n;
  // backward walk, started with -d0 (get_backward_path_from_node, :804-847)
  int tgt = -1, td = 0;
  int r = lin_step(g, n, d0 == -1, &tgt, &td);
  while (r == 2 && tgt != n) {
    if (len >= min_length) return;  // already too long to be clipped
    path[len++] = tgt;
    r = lin_step(g, tgt, td == 1, &tgt, &td);
  }
  // forward walk, started with d0 (get_forward_path_from_node, :743-779)
  r = lin_step(g, n, d0 == 1, &tgt, &td);
  while (r == 2 && tgt != n) {
    if (len >= min_length) return;
    path[len++] = tgt;
    r = lin_step(g, tgt, td == 1, &tgt, &td);
  }
  if (!(len > 0 && len < min_length)) return;
  const double thr = ((double)acc[0] / (double)acc[1]) * 1.5;
  bool all_high = true;
  for (int j = 0; j < len; ++j) all_high = all_high && ((double)g.n_cov[path[j]] > thr);
  if (all_high) return;
  // a tip that IS its whole component is kept (:710-713)
  if (comp_live) {
    int distinct = 0;
    for (int j = 0; j < len; ++j) {
      bool dup = false;
      for (int q = 0; q < j; ++q) dup = dup || (path[q] == path[j]);
      distinct += dup ? 0 : 1;
    }
    if ((unsigned int)distinct == comp_live[g.n_comp[n]]) return;
  } else {
    // nothing has been removed since the build: the component of the path's nodes is the path exactly when no live
    // edge leaves it (its nodes have at most two edges each: the walk only enters nodes of degree one or two)
    bool closed = true;
    for (int j = 0; j < len && closed; ++j)
      for (int side = 0; side < 2 && closed; ++side) {
        const int4 rw = g.lrows[2ll * path[j] + side];
        for (int q = 0; q < rw.y && closed; ++q) {
          const int t = q == 0 ? rw.z : g.lent[rw.x + q].x;
          bool in = false;
          for (int m = 0; m < len; ++m) in = in || (path[m] == t);
          closed = in;
        }
      }
    if (closed) return;
  }
  for (int j = 0; j < len; ++j)
    if (!protect || !protect[path[j]]) kill[path[j]] = 1;
}

// (grid-stride: a thread first sums up runs of equal ids along its own nodes, so that the giant
// component costs one atomic per WAVE of a small grid — one per wave of a node-sized grid still put
// thousands of atomics on one address, ~90 per microsecond)
__global__ void k_comp_hist(const int* __restrict__ comp, const unsigned char* __restrict__ alive,
                            const unsigned int* __restrict__ cov, long long n, unsigned int min_cov,
                            unsigned int* __restrict__ live_cnt, unsigned int* __restrict__ high_cnt,
                            unsigned long long* __restrict__ acc /* or null: {sum of live coverages, live nodes} */) {
  int cid = -1;
  unsigned int n_live = 0, n_high = 0;
  unsigned long long s_cov = 0, s_n = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (!alive[i]) continue;
    if (acc) {
      s_cov += cov[i];
      s_n += 1;
    }
    const int id = comp[i];
    if (id != cid) {
      if (cid >= 0) {
        atomicAdd(&live_cnt[cid], n_live);
        if (high_cnt && n_high) atomicAdd(&high_cnt[cid], n_high);
      }
      cid = id;
      n_live = n_high = 0;
    }
    ++n_live;
    n_high += (high_cnt && cov[i] >= min_cov) ? 1u : 0u;
  }
  const int lane = threadIdx.x & 63;
  if (acc) {  // one pair of atomics per WORKGROUP (a pair per wave: two thousand atomics on one line, ~15 us)
    __shared__ unsigned long long s_part[2][4];
    for (int d = 32; d > 0; d >>= 1) {
      s_cov += __shfl_xor(s_cov, d, 64);
      s_n += __shfl_xor(s_n, d, 64);
    }
    if (lane == 0) {
      s_part[0][threadIdx.x >> 6] = s_cov;
      s_part[1][threadIdx.x >> 6] = s_n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long n_all = s_part[1][0] + s_part[1][1] + s_part[1][2] + s_part[1][3];
      if (n_all) {
        atomicAdd(&acc[0], s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3]);
        atomicAdd(&acc[1], n_all);
      }
    }
  }
  bool active = cid >= 0;
  // most nodes share one giant component: aggregate equal ids inside the wave so that a
  // wave issues one atomic per distinct component instead of one per node
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lc = __shfl(cid, leader, 64);
    const bool same = active && cid == lc;
    const unsigned long long m = __ballot(same);
    unsigned int sl = same ? n_live : 0u, sh = same ? n_high : 0u;
    for (int d = 32; d > 0; d >>= 1) {
      sl += __shfl_xor(sl, d, 64);
      sh += __shfl_xor(sh, d, 64);
    }
    if (lane == leader) {
      atomicAdd(&live_cnt[lc], sl);
      if (high_cnt && sh) atomicAdd(&high_cnt[lc], sh);
    }
    active = active && !same;
    todo &= ~m;
  }
}

// {sum of the live nodes' coverages, number of live nodes} alone (a clip that needs no component labels)
__global__ __launch_bounds__(256) void k_cov_acc(const unsigned char* __restrict__ alive, const unsigned int* __restrict__ cov,
                                                 long long n, unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long s_part[2][4];
  unsigned long long s_cov = 0, s_n = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (alive[i]) {
      s_cov += cov[i];
      s_n += 1;
    }
  for (int d = 32; d > 0; d >>= 1) {
    s_cov += __shfl_xor(s_cov, d, 64);
    s_n += __shfl_xor(s_n, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_part[0][threadIdx.x >> 6] = s_cov;
    s_part[1][threadIdx.x >> 6] = s_n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long n_all = s_part[1][0] + s_part[1][1] + s_part[1][2] + s_part[1][3];
    if (n_all) {
      atomicAdd(&acc[0], s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3]);
      atomicAdd(&acc[1], n_all);
    }
  }
}

__global__ void k_scatter_ids(const unsigned char* __restrict__ killed, const long long* __restrict__ pos,
                              long long n, int* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && killed[i]) out[pos[i]] = (int)i;
}

// kill[] (s0) -> node_alive, removal side effects, ascending list of removed ids
static int finish_kill(amg_ctx* c, int64_t* n_removed, int32_t* removed_ids) {
  hipStream_t st = c->stream;
  const long long D = c->n_nodes;
  AMGCHK(c->s2.ensure((size_t)(D + 2) * sizeof(long long)));
  // the scan applies what it counts: a marked live node dies, kill[] is left as "removed by this call"
  AMGCHK(prim_exscan_apply_kill(c, c->s0.as<unsigned char>(), c->node_alive.as<unsigned char>(), c->s2.as<long long>(),
                                (size_t)D));
  long long total = 0;
  {
    FetchList l;
    l.add(c->s2.as<long long>() + D);
    AMGCHK(fetch(c, l, reinterpret_cast<unsigned long long*>(&total)));
  }
  if (n_removed) *n_removed = total;
  if (removed_ids && total > 0) {
    AMGCHK(c->s3.ensure((size_t)total * sizeof(int)));
    hipLaunchKernelGGL(k_scatter_ids, dim3(nblk(D, 256)), dim3(256), 0, st, c->s0.as<unsigned char>(),
                       c->s2.as<long long>(), D, c->s3.as<int>());
    HIPCHK(hipMemcpyAsync(removed_ids, c->s3.p, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  if (total > 0) AMGCHK(apply_removals(c, 0));
  if (removed_ids && total > 0) HIPCHK(hipStreamSynchronize(st));  // (nothing else here is read by the host)
  return AMG_OK;
}

extern "C" int amg_remove_short_linear_paths(amg_ctx* c, int32_t min_length, const uint8_t* protect,
                                             int64_t* n_removed, int32_t* removed_ids) {
  NEED_BUILT(c);
  if (min_length < 1 || min_length > CLIP_MAX)
    return amg_fail(AMG_E_ARG, "min_length must be in [1, %d]", CLIP_MAX);
  hipStream_t st = c->stream;
  const long long D = c->n_nodes;
  if (n_removed) *n_removed = 0;
  if (D == 0) return AMG_OK;
  stages_reset(c);
  // "a tip that is its whole component is kept" (:710-713) compares the path with the LIVE nodes of its component as
  // labelled at build time.  While nothing has been removed since the build that is the question whether a live edge
  // leaves the path, which the walk's own rows answer: no labels are made for the clip of a freshly built graph
  // (the cleaning sweep's case; AMG_CLIP_COMPONENTS=1: A/B + test switch)
  const bool by_labels = !c->pristine || getenv("AMG_CLIP_COMPONENTS");
  if (by_labels) AMGCHK(ensure_components(c));
  stage_begin(c, "clip");
  // live nodes per component, and the mean node coverage's two integers (:868-871), in one pass
  unsigned long long* acc = c->status.as<unsigned long long>() + ST_COV_SUM;  // (the live adjacency below uses ST_COMPACT_*)
  AMGCHK(c->s0.ensure((size_t)D + 8));
  if (by_labels) AMGCHK(c->s4.ensure((size_t)(c->n_components + 2) * sizeof(unsigned int)));
  {
    ClearList cl;
    cl.add(acc, 2 * sizeof(unsigned long long));
    cl.add(c->s0.p, (size_t)D + 8);
    if (by_labels) cl.add(c->s4.p, (size_t)(c->n_components + 2) * sizeof(unsigned int));
    AMGCHK(clear_many(c, cl));
  }
  if (by_labels)
    hipLaunchKernelGGL(k_comp_hist, dim3(nblk(D, 256) < 256u ? nblk(D, 256) : 256u), dim3(256), 0, st, c->node_comp.as<int>(),
                       c->node_alive.as<unsigned char>(), c->node_cov.as<unsigned int>(), D, 0u,
                       c->s4.as<unsigned int>(), (unsigned int*)nullptr, acc);
  else
    hipLaunchKernelGGL(k_cov_acc, dim3(nblk(D, 2048) < 256u ? nblk(D, 2048) : 256u), dim3(256), 0, st,
                       c->node_alive.as<unsigned char>(), c->node_cov.as<unsigned int>(), D, acc);
  unsigned char* d_protect = nullptr;
  if (protect) {
    AMGCHK(c->s5.ensure((size_t)D + 8));
    HIPCHK(hipMemcpyAsync(c->s5.p, protect, (size_t)D, hipMemcpyHostToDevice, st));
    d_protect = c->s5.as<unsigned char>();
  }
  AMGCHK(ensure_live_adj(c));
  hipLaunchKernelGGL(k_clip_mark, dim3(nblk(D, 128)), dim3(128), 0, st, make_view(c), D, (int)min_length,
                     acc, by_labels ? c->s4.as<unsigned int>() : (const unsigned int*)nullptr, d_protect,
                     c->s0.as<unsigned char>());
  int r = finish_kill(c, n_removed, removed_ids);
  stage_end(c);
  c->have_corrected = false;
  return r;
}

// ------------------------------------------------------------------ remove_low_coverage_components (:950-958)
__global__ void k_kill_low_components(const int* __restrict__ comp, const unsigned char* __restrict__ alive,
                                      const unsigned int* __restrict__ high_cnt, long long n,
                                      unsigned char* __restrict__ kill) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && alive[i] && high_cnt[comp[i]] == 0) kill[i] = 1;
}

extern "C" int amg_remove_low_coverage_components(amg_ctx* c, uint32_t min_cov) {
  NEED_BUILT(c);
  hipStream_t st = c->stream;
  const long long D = c->n_nodes;
  if (D == 0) return AMG_OK;
  AMGCHK(ensure_components(c));
  size_t nc = (size_t)(c->n_components + 2);
  AMGCHK(c->s0.ensure((size_t)D + 8));
  AMGCHK(c->s4.ensure(2 * nc * sizeof(unsigned int)));
  {
    ClearList cl;
    cl.add(c->s0.p, (size_t)D + 8);
    cl.add(c->s4.p, 2 * nc * sizeof(unsigned int));
    AMGCHK(clear_many(c, cl));
  }
  unsigned int* live = c->s4.as<unsigned int>();
  unsigned int* high = live + nc;
  hipLaunchKernelGGL(k_comp_hist, dim3(nblk(D, 256) < 256u ? nblk(D, 256) : 256u), dim3(256), 0, st, c->node_comp.as<int>(),
                     c->node_alive.as<unsigned char>(), c->node_cov.as<unsigned int>(), D, min_cov, live, high,
                     (unsigned long long*)nullptr);
  hipLaunchKernelGGL(k_kill_low_components, dim3(nblk(D, 256)), dim3(256), 0, st, c->node_comp.as<int>(),
                     c->node_alive.as<unsigned char>(), high, D, c->s0.as<unsigned char>());
  c->have_corrected = false;
  return finish_kill(c, nullptr, nullptr);
}

// ------------------------------------------------------------------ correct_reads (:1123-1396)
// Pipeline (all device):
//   k_corr_classify   one wave per read: class, [start,end] (find_read_boundaries :1153-1164),
//                     number of None runs inside (identify_path_terminals :1375-1386), output bound
//   scan(bound)       temp offsets
//   k_corr_simple     one wave per read: unmarked reads are copied, reads that only lost their
//                     ends are sliced ([start : end + k], :1277-1285)
//   k_corr_gapped     one thread per gapped read: bounded DFS per None run
//                     (new_find_paths_between_nodes :2292-2342), cartesian product of the
//                     replacements (insert_elements :1166-1203), best candidate by shared genes,
//                     then mean coverage (:1297-1310), genes via get_annotation_for_read (:1331-1373)
//   k_corr_nw         one wave per gapped read (positions only): needleman_wunsch (:1433-1480) by
//                     anti-diagonals, traceback, position carry-over (:1314-1325) and
//                     replace_invalid_gene_positions (:1669-1691)
//   scan + k_corr_pack  compaction into the corrected CSR
enum { RC_SKIP = 0, RC_COPY = 1, RC_DROP = 2, RC_TRIM = 3, RC_GAPPED = 4, RC_KEEP_ORIG = 5 };

struct CorrArgs {
  const int* tokens;
  const long long* read_off;
  const int* tok_node;
  const signed char* tok_dir;
  const unsigned char* read_fix;
  // gene positions are NOT moved with the genes when reads are corrected: a read carries an offset
  // (pos_off[r]; nullptr = its token offset) into one of two pools — the caller's arrays as handed to
  // amg_set_positions (indices < n0) or the positions the carry-over kernels produced (p1*, indices
  // from n0 on).  An untouched read keeps its offset, a trimmed one adds its start, a re-threaded one
  // points at its new positions; the corrected set is gathered only when the host asks for it.
  const long long* p0s;
  const long long* p0e;
  const long long* p1s;
  const long long* p1e;
  long long n0;
  const long long* pos_off;
  const long long* read_len;
  long long n_reads;
  int k, flip, have_pos;
  // per read
  unsigned int* gflag;            // 1: the read is re-threaded (RC_GAPPED)
  unsigned long long* max_bound;  // largest `bound` of a re-threaded read
  unsigned long long* lmask;      // live-window mask of a flagged read with <= 64 windows (0 otherwise)
  unsigned long long* n_runs;     // None runs over all re-threaded reads: 16 partial sums, 16 words apart
  unsigned char* cls;
  unsigned char* cls_final;  // starts as a copy of cls (k_corr_classify writes both); the re-threading may turn a read into RC_KEEP_ORIG
  int* r_start;
  int* r_end;
  unsigned int* bound;
  const long long* tmp_off;
  unsigned int* new_len;
  // staged genes of the re-threaded reads
  int* tmp_tok;
};

// positions of the genes that start at pool index `off`
__device__ __forceinline__ void pos_base(const CorrArgs& a, long long off, const long long*& gs, const long long*& ge) {
  if (off < a.n0) {
    gs = a.p0s + off;
    ge = a.p0e + off;
  } else {
    gs = a.p1s + (off - a.n0);
    ge = a.p1e + (off - a.n0);
  }
}

__device__ __forceinline__ long long bcast_i64(long long v, int lane) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// One wave classifies 64 consecutive reads.  Lane l owns read l: its offsets, fix flag and results are loaded and
// stored coalesced, one read per lane.  What needs the read's windows — which are live — is a 64-bit mask per read:
// the wave loads the windows of one flagged read at a time (lanes = windows, four reads in flight), ballots, and
// hands the mask to the owning lane; everything else is bit arithmetic on that mask.  (A wave per four reads
// stored every result with its own one-lane instruction: ~8 vector-memory instructions per read.)
#define CLS_READS 64
__global__ __launch_bounds__(256) void k_corr_classify(CorrArgs a) {
  const long long rbase = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * CLS_READS;
  if (rbase >= a.n_reads) return;
  const int lane = threadIdx.x & 63;
  const long long r = rbase + lane;
  const bool have = r < a.n_reads;
  const long long t0 = have ? a.read_off[r] : 0;
  const long long n = have ? (a.read_off[r + 1] - t0) - a.k + 1 : 0;  // windows; L = n + k - 1
  const bool look = have && n > 0 && a.read_fix[r] != 0;
  // live-window masks of the flagged reads with <= 64 windows
  unsigned long long lv = 0;
  unsigned long long todo = __ballot(look && n <= 64);
  while (todo) {
    int who[4];
    long long tj[4];
    int nj[4], v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      who[q] = todo ? __ffsll((long long)todo) - 1 : -1;
      if (todo) todo &= todo - 1ull;
      tj[q] = who[q] >= 0 ? bcast_i64(t0, who[q]) : 0;
      nj[q] = who[q] >= 0 ? __builtin_amdgcn_readlane((int)n, who[q]) : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = lane < nj[q] ? a.tok_node[tj[q] + lane] : -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned long long m = __ballot(v[q] >= 0);  // lanes >= n hold -1
      if (lane == who[q]) lv = m;
    }
  }
  long long first = n, last = -1;
  unsigned int runs = 0, live = 0;
  if (look && n <= 64) {
    first = lv ? (long long)__ffsll((long long)lv) - 1 : n;
    last = lv ? 63 - (long long)__clzll((long long)lv) : -1;
    live = (unsigned int)__popcll(lv);
    // a None run ends where the next window is live (windows past `last` are not live)
    const unsigned long long inside =
        lv ? ((last == 63 ? ~0ull : ((1ull << (last + 1)) - 1ull)) & ~((1ull << first) - 1ull)) : 0ull;
    runs = (unsigned int)__popcll(~lv & inside & (lv >> 1));
  }
  // flagged reads longer than one wave: the wave walks each of them
  unsigned long long big = __ballot(look && n > 64);
  while (big) {
    const int w = __ffsll((long long)big) - 1;
    big &= big - 1ull;
    const long long tw = bcast_i64(t0, w), nw = bcast_i64(n, w);
    long long f = nw, l = -1;
    for (long long i = lane; i < nw; i += 64)
      if (a.tok_node[tw + i] >= 0) {
        f = f < i ? f : i;
        l = l > i ? l : i;
      }
    for (int d = 32; d > 0; d >>= 1) {
      const long long f2 = __shfl_xor(f, d, 64), l2 = __shfl_xor(l, d, 64);
      f = f < f2 ? f : f2;
      l = l > l2 ? l : l2;
    }
    unsigned int ru = 0, li = 0;
    if (l >= 0) {
      for (long long i = f + lane; i <= l; i += 64) {
        const bool none = a.tok_node[tw + i] < 0;
        li += none ? 0u : 1u;
        if (none && a.tok_node[tw + i + 1] >= 0) ++ru;
      }
      for (int d = 32; d > 0; d >>= 1) {
        ru += __shfl_xor(ru, d, 64);
        li += __shfl_xor(li, d, 64);
      }
    }
    if (lane == w) {
      first = f;
      last = l;
      runs = ru;
      live = li;
    }
  }
  const long long L = n + a.k - 1;
  unsigned char cls;
  int start = 0, end = -1;
  unsigned int bound = 0;
  unsigned int len_out = 0;  // genes of the corrected read, known here unless it has None runs
  if (n <= 0) {
    cls = RC_SKIP;  // no entry in _readNodes: correct_reads never sees the read (:1128)
  } else if (!look) {
    cls = RC_COPY;
    len_out = (unsigned int)L;
  } else if (last < 0) {
    cls = RC_DROP;  // every node filtered: the read is dropped (:1141,:1150)
  } else {
    start = (int)first;
    end = (int)last;
    if (runs == 0) {
      cls = RC_TRIM;
      len_out = (unsigned int)(end - start + a.k);
    } else {
      cls = RC_GAPPED;
      const unsigned int b1 = live + runs * (unsigned int)(2 * a.k) + (unsigned int)a.k;
      bound = b1 > (unsigned int)L ? b1 : (unsigned int)L;  // may fall back to the original genes
    }
  }
  if (have) {
    a.cls[r] = cls;
    a.cls_final[r] = cls;
    a.r_start[r] = start;
    a.r_end[r] = end;
    a.bound[r] = bound;    // temp space: only re-threaded reads are staged
    a.new_len[r] = len_out;
    a.gflag[r] = cls == RC_GAPPED ? 1u : 0u;  // list of re-threaded reads (scan input)
    a.lmask[r] = (cls == RC_GAPPED && n <= 64) ? lv : 0ull;
  }
  // largest staging bound of a re-threaded read and the number of None runs: one atomic per wave each (the maximum
  // only when it raises a plain — possibly stale, never too large — read of it); lanes past the last read hold zeros
  unsigned int mb = (have && cls == RC_GAPPED) ? bound : 0u;
  unsigned int nr = (have && cls == RC_GAPPED) ? runs : 0u;
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned int o = (unsigned int)__shfl_xor((int)mb, d, 64);
    mb = mb > o ? mb : o;
    nr += (unsigned int)__shfl_xor((int)nr, d, 64);
  }
  if (lane == 0 && (unsigned long long)mb > *a.max_bound) atomicMax(a.max_bound, (unsigned long long)mb);
  // (16 counter words 128 bytes apart: one word takes ~90 atomics per microsecond, a wave per 64 reads asks more)
  if (lane == 0 && nr) atomicAdd(a.n_runs + 16 * (blockIdx.x & 15), (unsigned long long)nr);
}

// ---- gapped reads
#define DFS_MAX (2 * AMG_MAX_K + 4)

struct PathSink {
  int* buf;        // nullptr => count only
  long long used;  // ints
  int n_paths;
};

// new_find_paths_between_nodes(start, end, distance, direction): simple paths following the
// forward list when the current direction is +1, the backward list when -1, in list order;
// a path is accepted when it reaches `end` with <= distance nodes.  Emits [len, node*len, dir*len].
__device__ void dfs_paths(const GView& g, int s, int sdir, int e, int distance, PathSink* sink) {
  int node[DFS_MAX], dir[DFS_MAX];
  long long cur[DFS_MAX], lim[DFS_MAX];
  int depth = 0;
  node[0] = s;
  dir[0] = sdir;
  bool entering = true;
  while (depth >= 0) {
    if (entering) {
      int L = depth + 1;
      if (node[depth] == e && L <= distance) {
        if (sink->buf) {
          int* o = sink->buf + sink->used;
          o[0] = L;
          for (int j = 0; j < L; ++j) {
            o[1 + j] = node[j];
            o[1 + L + j] = dir[j];
          }
        }
        sink->used += 1 + 2 * L;
        sink->n_paths += 1;
        --depth;
        entering = false;
        continue;
      }
      if (L - 1 > distance) {
        --depth;
        entering = false;
        continue;
      }
      long long row = 2ll * node[depth] + (dir[depth] == 1 ? 0 : 1);
      const int4 rw = g.lrows[row];
      cur[depth] = rw.x;
      lim[depth] = rw.x + rw.y;
      entering = false;
    }
    bool pushed = false;
    while (cur[depth] < lim[depth]) {
      const int2 ent = g.lent[cur[depth]++];
      int t = ent.x;
      bool seen = false;
      for (int j = 0; j <= depth; ++j) seen = seen || (node[j] == t);
      if (seen) continue;
      node[depth + 1] = t;
      dir[depth + 1] = ent.y;
      ++depth;
      entering = true;
      pushed = true;
      break;
    }
    if (!pushed) --depth;
  }
}

// last gene of node n taken in direction d (get_gene_mer_genes / get_reverse_gene_mer_genes)
__device__ __forceinline__ int oriented_tok(const GView& g, int n, int d, int j) {
  const int* nt = g.n_tok + (long long)n * g.k;
  return d == 1 ? nt[j] : g.flip - nt[g.k - 1 - j];
}

struct GapIter {
  long long t0;
  const int* tok_node;
  int start, end, i;
  int ps, pe;
  __device__ bool next() {
    // identify_path_terminals: for i in [start, end] with a None at i: path_start = i-1 if live,
    // pair emitted when i+1 is live
    while (i <= end) {
      int idx = i++;
      if (tok_node[t0 + idx] < 0) {
        if (tok_node[t0 + idx - 1] >= 0) ps = idx - 1;
        if (tok_node[t0 + idx + 1] >= 0) {
          pe = idx + 1;
          return true;
        }
      }
    }
    return false;
  }
};

// what the wave-per-read kernel needs to start on gapped read gi, in one 32-byte record
// (written by k_scatter_gapped) instead of a chain of dependent per-read loads
struct __attribute__((aligned(16))) GapRec {
  int r, L0, start, end;
  long long t0, dst;
  unsigned long long mask;  // live windows of a read with <= 64 windows (k_corr_classify), 0 otherwise
  long long pad;
};

struct GapArgs {
  CorrArgs a;
  GView g;
  const GapRec* rec;
  const int* gapped_reads;
  long long n_gapped;
  int* pool;               // path pool (ints)
  unsigned long long pool_cap;
  unsigned long long* pool_used;  // bump pointer
  unsigned long long* status;
  int* cand;               // candidate scratch, [grid threads * cand_stride]
  unsigned int cand_stride;
  unsigned char* final_cls;
  unsigned char* need_slow;  // per gapped read: 1 = the wave-per-read fast kernel gave up
  // path memo (k_gap_queries / k_gap_dfs): the same (start node, direction, end node) question is asked by every read
  // that lost the same stretch of the genome — about nine times each at 3 000x depth — and answered once
  const int* gq;             // per gapped read GF_MAXGAP query slots in run order; [0] < 0: no memo for this read
  const int4* qres;          // per query slot {pool offset, ints, paths, -}; ints < 0: the answer did not fit
  const int* qpool;          // [run = 0, len, nodes, dirs] records in DFS order
};

// build candidate `combo` (mixed radix over the gaps' path choices) into (out_node, out_dir);
// returns its node count.  paths of gap q start at pool[gap_off[q]] as [len, nodes, dirs] records.
__device__ int build_candidate(const GapArgs& A, long long t0, int start, int end, const int* rec,
                               int n_gaps, unsigned long long combo, int* out_node, signed char* out_dir) {
  // rec layout: for each gap q: [ps, pe, n_paths, first_record_offset] (4 ints)
  // product(*lists): the LAST gap varies fastest
  int n = 0;
  int q = 0;
  int i = start;
  int prev_pe = -1;
  // choice for gap q = (combo / prod_{j>q} n_j) % n_q
  while (i <= end) {
    if (q < n_gaps && rec[4 * q] == i) {
      int ps = rec[4 * q], pe = rec[4 * q + 1], np = rec[4 * q + 2];
      unsigned long long div = 1;
      for (int j = q + 1; j < n_gaps; ++j) div *= (unsigned long long)rec[4 * j + 2];
      int pick = (int)((combo / div) % (unsigned long long)np);
      const int* p = A.pool + rec[4 * q + 3];
      for (int s = 0; s < pick; ++s) p += 1 + 2 * p[0];
      int L = p[0];
      if (prev_pe == ps && n > 0) --n;  // shared endpoint: the later replacement overwrites it
      for (int j = 0; j < L; ++j) {
        out_node[n] = p[1 + j];
        out_dir[n] = (signed char)p[1 + L + j];
        ++n;
      }
      prev_pe = pe;
      i = pe;
      ++q;
      if (!(q < n_gaps && rec[4 * q] == pe)) i = pe + 1;
    } else {
      out_node[n] = A.a.tok_node[t0 + i];
      out_dir[n] = A.a.tok_dir[t0 + i];
      ++n;
      ++i;
    }
  }
  return n;
}

__global__ __launch_bounds__(64) void k_corr_gapped(GapArgs A) {
  const CorrArgs& a = A.a;
  const GView& g = A.g;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  int* my = A.cand + gtid * (long long)A.cand_stride;
  for (long long gi = gtid; gi < A.n_gapped; gi += gstride) {
    if (!A.need_slow[gi]) continue;
    const long long r = A.gapped_reads[gi];
    const long long t0 = a.read_off[r];
    const int L0 = (int)(a.read_off[r + 1] - t0);
    const int start = a.r_start[r], end = a.r_end[r];
    const long long dst = a.tmp_off[r];
    // ---- pass 1: count paths per None run
    int n_gaps = 0;
    long long need = 0;
    bool dead_end = false;
    {
      GapIter it{t0, a.tok_node, start, end, start, -1, -1};
      while (it.next()) {
        PathSink sink{nullptr, 0, 0};
        dfs_paths(g, a.tok_node[t0 + it.ps], a.tok_dir[t0 + it.ps], a.tok_node[t0 + it.pe], 2 * g.k, &sink);
        if (sink.n_paths == 0) dead_end = true;
        need += sink.used;
        ++n_gaps;
      }
    }
    bool keep_orig = dead_end;  // product over an empty list: possible_paths == [] (:1292-1293)
    int* rec = nullptr;
    if (!keep_orig) {
      // ---- reserve pool space: 4 ints per gap + the path records
      unsigned long long want = (unsigned long long)need + 4ull * n_gaps;
      unsigned long long base = atomicAdd(A.pool_used, want);
      if (base + want > A.pool_cap) {
        A.status[ST_OVERFLOW] = 3;  // host grows the pool and re-runs
        a.new_len[r] = 0;
        continue;
      }
      rec = A.pool + base;
      int* wr = rec + 4 * n_gaps;
      GapIter it{t0, a.tok_node, start, end, start, -1, -1};
      int q = 0;
      while (it.next()) {
        PathSink sink{wr, 0, 0};
        dfs_paths(g, a.tok_node[t0 + it.ps], a.tok_dir[t0 + it.ps], a.tok_node[t0 + it.pe], 2 * g.k, &sink);
        rec[4 * q] = it.ps;
        rec[4 * q + 1] = it.pe;
        rec[4 * q + 2] = sink.n_paths;
        rec[4 * q + 3] = (int)(wr - A.pool);
        wr += sink.used;
        ++q;
      }
    }
    if (keep_orig) {  // the pack step copies the original genes and positions
      a.new_len[r] = (unsigned int)L0;
      A.final_cls[r] = RC_KEEP_ORIG;
      continue;
    }
    // ---- enumerate the cartesian product in itertools.product order
    unsigned long long n_combo = 1;
    for (int q = 0; q < n_gaps; ++q) {
      n_combo *= (unsigned long long)rec[4 * q + 2];
      if (n_combo > (1ull << 40)) n_combo = 1ull << 40;  // unreachable in practice; bounds the loop
    }
    const int cap_nodes = (int)a.bound[r];
    int* c_node = my;                                         // [cap_nodes]
    signed char* c_dir = reinterpret_cast<signed char*>(my + cap_nodes);  // [cap_nodes]
    int* c_gene = my + cap_nodes + (cap_nodes + 3) / 4;       // [cap_nodes + k]
    int best_shared = 0;
    unsigned long long best_sum = 0, best_len = 1;  // mean coverage 0
    int best_n = -1;
    for (unsigned long long combo = 0; combo < n_combo; ++combo) {
      int n = build_candidate(A, t0, start, end, rec, n_gaps, combo, c_node, c_dir);
      // genes (get_annotation_for_read): k-1 genes of the first node + last gene of every node
      int ng = 0;
      for (int j = 0; j < g.k - 1; ++j) c_gene[ng++] = oriented_tok(g, c_node[0], c_dir[0], j);
      unsigned long long csum = 0;
      for (int j = 0; j < n; ++j) {
        c_gene[ng++] = oriented_tok(g, c_node[j], c_dir[j], g.k - 1);
        csum += g.n_cov[c_node[j]];
      }
      // len(set(genes) & set(original genes))
      int shared = 0;
      for (int j = 0; j < ng; ++j) {
        int tk = c_gene[j];
        bool dup = false;
        for (int q = 0; q < j && !dup; ++q) dup = (c_gene[q] == tk);
        if (dup) continue;
        bool hit = false;
        for (int q = 0; q < L0 && !hit; ++q) hit = (a.tokens[t0 + q] == tk);
        shared += hit ? 1 : 0;
      }
      // strictly more shared genes, or equal and strictly higher mean coverage (:1301-1308)
      bool better = shared > best_shared ||
                    (shared == best_shared && csum * best_len > best_sum * (unsigned long long)n);
      if (better) {
        best_shared = shared;
        best_sum = csum;
        best_len = (unsigned long long)n;
        best_n = ng;
        for (int j = 0; j < ng; ++j) a.tmp_tok[dst + j] = c_gene[j];
      }
    }
    a.new_len[r] = (unsigned int)best_n;
  }
}

#define NWF_MAX_M 64
#define NWF_MAX_N 128

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool nw_fast_ok(long long N, long long M) {
  return N <= NWF_MAX_N && M <= NWF_MAX_M && N > 0 && M > 0;
}


// ---- fast path of the gapped-read kernel: one wave per read, everything staged in LDS.
// Reads that exceed any of its fixed capacities are flagged (need_slow) and left to the
// general one-thread-per-read kernel above; results are identical by construction (same
// DFS order, same product order, same comparisons).
#define GF_MAXW 128     // windows per read
#define GF_MAXGAP 16    // None runs per read
#define GF_POOL 384     // ints of path records per read
#define GF_CAND 136     // nodes of a candidate (LDS of the block stays under 20 KB: 8 blocks per CU)
#define GF_MAXCOMBO 256

// DFS of one None run, executed COOPERATIVELY by the whole wave: control flow is uniform,
// stack level d lives in the registers of lane d (node, direction, row cursor), levels are
// read with v_readlane, the "already on the path" test is one ballot, and an accepted path is
// written to the pool by lanes 0..len-1 at once.  (A one-lane DFS with its stack in LDS or
// scratch spends its time in dependent LDS/scratch round trips.)
// Emits [run, len, nodes, dirs] records; returns the number of paths, -1 on pool overflow.
__device__ int dfs_paths_wave(const GView& g, int s, int sdir, int e, int distance, int run, int* pool,
                              int* used, int lane) {
  int my_node = 0, my_dir = 0, my_cur = 0, my_lim = 0, my_off = 0;
  int depth = 0, n_paths = 0;
  bool overflow = false;
  if (lane == 0) {
    my_node = s;
    my_dir = sdir;
  }
  bool entering = true;
  int2 first_ent = make_int2(-1, 0);
  while (depth >= 0) {
    const int d = __builtin_amdgcn_readfirstlane(depth);
    if (entering) {
      const int L = d + 1;
      const int cur_node = __builtin_amdgcn_readlane(my_node, d);
      if (cur_node == e && L <= distance) {
        const int off = *used;
        if (off + 2 + 2 * L <= GF_POOL) {
          if (lane == 0) {
            pool[off] = run;
            pool[off + 1] = L;
          }
          if (lane < L) {
            pool[off + 2 + lane] = my_node;
            pool[off + 2 + L + lane] = my_dir;
          }
        } else {
          overflow = true;
        }
        wave_sync();
        if (lane == 0) *used = off + 2 + 2 * L;
        wave_sync();
        ++n_paths;
        --depth;
        entering = false;
        first_ent.x = -1;
        continue;
      }
      if (L - 1 > distance) {
        --depth;
        entering = false;
        first_ent.x = -1;
        continue;
      }
      const int cur_dir = __builtin_amdgcn_readlane(my_dir, d);
      const int4 rw = g.lrows[2ll * cur_node + (cur_dir == 1 ? 0 : 1)];  // uniform address
      if (lane == d) {
        my_cur = 0;
        my_lim = rw.y;
        my_off = rw.x;
      }
      first_ent = make_int2(rw.z, rw.w);
      entering = false;
    }
    int c = __builtin_amdgcn_readlane(my_cur, d);
    const int lim = __builtin_amdgcn_readlane(my_lim, d);
    const int row_off = __builtin_amdgcn_readlane(my_off, d);
    bool pushed = false;
    while (c < lim) {
      int2 ent = first_ent;
      if (!(c == 0 && first_ent.x >= 0)) ent = g.lent[row_off + c];  // uniform address
      ++c;
      const int t = __builtin_amdgcn_readfirstlane(ent.x);
      const int td = __builtin_amdgcn_readfirstlane(ent.y);
      if (__ballot(lane <= d && my_node == t) != 0ull) continue;  // no node twice on a path
      if (lane == d) my_cur = c;
      if (lane == d + 1) {
        my_node = t;
        my_dir = td;
      }
      ++depth;
      entering = true;
      pushed = true;
      break;
    }
    if (!pushed) {
      --depth;
      first_ent.x = -1;  // back in an older row: its first entry was consumed long ago
    }
  }
  return overflow ? -1 : n_paths;
}

// ---- path memo.  k_gap_queries: one LANE per re-threaded read walks the read's None runs on its live-window mask
// (k_corr_classify kept it), looks up the three node words of each run and enters the question (start node, start
// direction, end node) into an open-addressing table; the slot index is the question's id, the lane that created
// the slot lists it.  k_gap_dfs answers every listed question once (the wave-cooperative search below, result copied
// to a global pool); k_corr_gapped_fast copies answers instead of searching.  Reads with more than 64 windows or
// more than GF_MAXGAP runs take no part (gq[0] = -1: they search for themselves, as before).
#define GM_INLINE 64
__device__ __forceinline__ unsigned long long gap_query_key(int s, int sdir, int e) {
  return (1ull << 63) | ((unsigned long long)(unsigned int)s << 32) | ((unsigned long long)(unsigned int)e << 1) |
         (sdir == 1 ? 1ull : 0ull);
}

__global__ __launch_bounds__(256) void k_gap_queries(const GapRec* __restrict__ rec, long long n_gapped, int k,
                                                      const int* __restrict__ tok_node,
                                                      const signed char* __restrict__ tok_dir,
                                                      unsigned long long* qtab, unsigned int qmask,
                                                      unsigned long long* ctr /*[0] questions listed*/,
                                                      int* __restrict__ qlist, int* __restrict__ gq,
                                                      unsigned long long* status) {
  // the questions this workgroup creates are collected in LDS and listed with ONE atomicAdd (a counter word takes
  // ~90 returning atomics per microsecond; there are ~100 k questions)
  __shared__ int s_list[256 * GF_MAXGAP];
  __shared__ unsigned int s_n;
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi < n_gapped) {
    const GapRec q = rec[gi];
    int* my = gq + gi * GF_MAXGAP;
    const int nwin = q.L0 - k + 1;
    const unsigned long long lv = q.mask;
    // a None run ends at window i when i is not live and i + 1 is, first <= i < last (k_corr_classify's `runs`)
    const int first = q.start, last = q.end;
    const unsigned long long inside =
        (last >= 63 ? ~0ull : ((1ull << (last + 1)) - 1ull)) & ~((1ull << (first & 63)) - 1ull);
    unsigned long long ends = ~lv & inside & (lv >> 1);
    if (nwin > 64 || lv == 0ull || __popcll(ends) > GF_MAXGAP) {
      my[0] = -1;
    } else {
      int j = 0;
      while (ends) {
        const int i = __ffsll((long long)ends) - 1;
        ends &= ends - 1ull;
        const int ps = 63 - __clzll((long long)(lv & ((1ull << i) - 1ull)));  // the live window before the run
        const int pe = i + 1;
        const unsigned long long key = gap_query_key(tok_node[q.t0 + ps], (int)tok_dir[q.t0 + ps], tok_node[q.t0 + pe]);
        unsigned int idx = (unsigned int)mix64(key) & qmask;
        int slot = -1;
        for (unsigned int probes = 0; probes <= qmask; ++probes) {
          unsigned long long cur = qtab[idx];  // plain: a stale view can only show "empty", which the CAS settles
          if (cur == 0ull) {
            cur = atomicCAS(qtab + idx, 0ull, key);
            if (cur == 0ull) {
              s_list[atomicAdd(&s_n, 1u)] = (int)idx;
              cur = key;
            }
          }
          if (cur == key) {
            slot = (int)idx;
            break;
          }
          idx = (idx + 1) & qmask;
        }
        if (slot < 0) status[ST_OVERFLOW] = 7;  // the table holds two slots per run: cannot fill up
        my[j++] = slot;
      }
    }
  }
  __syncthreads();
  const unsigned int n = s_n;
  if (n == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(ctr, (unsigned long long)n);
  __syncthreads();
  for (unsigned int i = threadIdx.x; i < n; i += 256) qlist[s_base + i] = s_list[i];
}

__global__ __launch_bounds__(64, 8) void k_gap_dfs(GView g, const int* __restrict__ qlist, long long n_queries,
                                                    const unsigned long long* __restrict__ qtab,
                                                    unsigned long long* pool_used, unsigned long long pool_cap,
                                                    int* __restrict__ qpool, int4* __restrict__ qres,
                                                    int* __restrict__ qgene) {
  __shared__ int s_pool[GF_POOL];
  __shared__ int s_used;
  const long long qi = blockIdx.x;
  if (qi >= n_queries) return;
  const int lane = threadIdx.x;
  const int slot = qlist[qi];
  const unsigned long long key = qtab[slot];
  const int s = (int)((key >> 32) & 0x7fffffffull), e = (int)((key >> 1) & 0x7fffffffull);
  const int sdir = (key & 1ull) ? 1 : -1;
  if (lane == 0) s_used = 0;
  wave_sync();
  const int np = dfs_paths_wave(g, s, sdir, e, 2 * g.k, 0, s_pool, &s_used, lane);
  wave_sync();
  const int used = s_used;
  int4 res = make_int4(0, -1, 0, 0);
  if (np >= 0) {
    // an answer of up to GM_INLINE ints lives in the question's own stretch of the pool (a few paths of ~7 nodes: nearly
    // all of them); longer ones take space behind those stretches, one atomicAdd each
    unsigned long long off = (unsigned long long)qi * GM_INLINE;
    if (used > GM_INLINE) {
      if (lane == 0) off = (unsigned long long)n_queries * GM_INLINE + atomicAdd(pool_used, (unsigned long long)used);
      off = (unsigned long long)bcast_i64((long long)off, 0);
    }
    if (off + (unsigned long long)used <= pool_cap) {
      for (int i = lane; i < used; i += 64) qpool[off + i] = s_pool[i];
      res = make_int4((int)off, used, np, 0);
      // a question with ONE answer (nearly all of them) also keeps the last gene of every node of its path, taken in the
      // path's direction (get_gene_mer_genes / get_reverse_gene_mer_genes :588-598): what k_corr_gapped_lean writes out
      if (np == 1 && used <= GM_INLINE) {
        const int len = s_pool[1];
        if (lane < len) qgene[off + 2 + lane] = oriented_tok(g, s_pool[2 + lane], s_pool[2 + len + lane], g.k - 1);
      }
    }
  }
  if (lane == 0) qres[slot] = res;
}

// ---- re-threading, the common case: SIXTEEN LANES per read.
// k_corr_gapped_fast gives a read a whole wave and spends ~900 instructions on it, most of them with a handful of
// useful lanes, and its waves wait two thirds of their time on a chain of five dependent loads (rocprofv3 counters,
// profiles/r5_*): the kernel is bound by instruction issue and by that chain, not by bytes.  Nearly every read asks
// questions the path memo answered with exactly ONE path (after filter_graph the error bubbles are gone: between two
// terminals of a read there is the genome's path and nothing else).  Then nothing has to be chosen — no cartesian
// product, no shared-gene count, no mean coverage — and the corrected read is the original one with, for every None
// run (ps, pe), the genes k + ps .. k + pe - 1 replaced by the last genes of the path's nodes 1 .. len - 1:
//   * a live window w of the read spells the read's own genes w .. w + k - 1, so every gene that comes from a live
//     window is a token of the read itself (no node-token gather);
//   * the path's first node is window ps; its last node is window pe's NODE in whatever direction the path arrives
//     (new_find_paths_between_nodes :2292-2342 accepts any), so its gene comes from the memo like the interior ones —
//     unless the next run starts at pe: then the later replacement overwrites the shared terminal (insert_elements
//     :1166-1203) and the gene is the read's own again.
// A group of 16 lanes takes one read: lane q owns None run q (k_gap_queries entered at most 16 per read), the group's
// prefix sums run over DPP row shifts (a DPP row IS 16 lanes), and the output genes are written 16 at a time.  Four
// reads per wave share every instruction and keep four chains of loads in flight.  A read that does not qualify (no
// memo entry, a question with no or several answers, an answer beyond the inline stretch, a path of one node) is
// flagged for k_corr_gapped_fast: same results by construction, checked against the oracle through the whole sweep at
// full size and by the fuzzers.
#define GL_GROUP 16
#define GL_THREADS 256
__device__ __forceinline__ int row_scan_incl(int v) {  // inclusive prefix sum inside a DPP row of 16 lanes
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  return v;
}

__global__ __launch_bounds__(GL_THREADS) void k_corr_gapped_lean(GapArgs A, const int* __restrict__ qgene,
                                                                  unsigned char* __restrict__ left) {
  // per read and run: {first output index of the run's genes, their number, output - token shift behind them, where
  // the path's genes are in qgene}
  __shared__ int4 s_run[GL_THREADS / GL_GROUP][GL_GROUP];
  __shared__ int s_tk[GL_THREADS / GL_GROUP][64 + AMG_MAX_K];  // the read's own genes (a qualifying read has <= 64 windows)
  const CorrArgs& a = A.a;
  const int k = A.g.k;
  const int lane = threadIdx.x & 63, l16 = threadIdx.x & (GL_GROUP - 1), grp = threadIdx.x / GL_GROUP;
  const int sh = lane & ~(GL_GROUP - 1);  // first lane of this group inside its wave
  const long long gi = (long long)blockIdx.x * (GL_THREADS / GL_GROUP) + grp;
  const bool have = gi < A.n_gapped;
  GapRec rec;
  rec.r = 0; rec.L0 = 0; rec.start = 0; rec.end = -1; rec.t0 = 0; rec.dst = 0; rec.mask = 0ull; rec.pad = 0;
  // the chain of dependent loads is what a read costs here: the record and the read's question slots leave together
  // (the slots of runs the read does not have are not initialised and not looked at), the read's genes and the
  // answers' headers follow, the answers' genes last
  int slot = have ? A.gq[gi * GF_MAXGAP + l16] : -1;
  if (have) rec = A.rec[gi];
  const unsigned long long lv = rec.mask;
  const int nwin = rec.L0 - k + 1;
  bool ok = have && nwin <= 64 && lv != 0ull;
  // the read's None runs as k_gap_queries numbered them: run q ends at the q-th window that is not live while the
  // next one is (identify_path_terminals :1375-1386)
  const int first = rec.start, last = rec.end;
  const unsigned long long inside =
      (last >= 63 ? ~0ull : ((1ull << (last + 1)) - 1ull)) & ~((1ull << (first & 63)) - 1ull);
  unsigned long long ends = ok ? (~lv & inside & (lv >> 1)) : 0ull;
  const int n_gaps = __popcll(ends);
  ok = ok && n_gaps >= 1 && n_gaps <= GF_MAXGAP;
  // (k_gap_queries left gq[0] = -1 on a read it did not enter)
  const bool mine = ok && l16 < n_gaps;
  if (!mine) slot = -1;
  int4 res = make_int4(0, -1, 0, 0);
  if (mine && slot >= 0) res = A.qres[slot];
  if (ok) {  // genes first .. last + k - 1 of the read, staged while the answers' headers are on their way
    const int span = last + k - first;
    for (int i = l16; i < span; i += GL_GROUP) s_tk[grp][i] = a.tokens[rec.t0 + first + i];
  }
  int ps = 0, pe = 0;
  if (mine) {
    unsigned long long e = ends;
    for (int j = 0; j < l16; ++j) e &= e - 1ull;
    const int i = __ffsll((long long)e) - 1;
    ps = 63 - __clzll((long long)(lv & ((1ull << i) - 1ull)));
    pe = i + 1;
  }
  const int len = (mine && res.y >= 4) ? (res.y - 2) >> 1 : 0;  // one record [run, len, nodes, dirs]
  const bool good = !mine || (slot >= 0 && res.z == 1 && res.y >= 6 && res.y <= GM_INLINE && len >= 2);
  // every run of the read has to qualify: the group's 16 bits of the wave's ballot
  const unsigned int bad16 = (unsigned int)((__ballot(!good) >> sh) & 0xffffull);
  ok = ok && bad16 == 0u;
  // does the next run start where this one ends?  (its ps from the neighbouring lane: row_shl:1)
  const int ps_next = __builtin_amdgcn_update_dpp(-1, ps, 0x101, 0xf, 0xf, false);
  const int shared = (mine && l16 + 1 < n_gaps && ps_next == pe) ? 1 : 0;
  const int c = mine ? len - 1 - shared : 0;            // genes the run brings
  const int rep = mine ? pe - ps - shared : 0;          // genes of the read they replace: k + ps .. k + pe - 1 - shared
  const int incl = row_scan_incl(c - rep);
  const int d_before = incl - (c - rep);
  if (mine) s_run[grp][l16] = make_int4(ps + k - first + d_before, c, incl, res.x + 3);  // (path node 1 sits at res.x + 2 + 1)
  const int total_delta = __shfl(incl, sh + (n_gaps > 0 ? n_gaps - 1 : 0), 64);
  __syncthreads();
  const int ng = ok ? (last + k - first) + total_delta : 0;
  if (ok) {
    for (int o = l16; o < ng; o += GL_GROUP) {
      int tok_shift = 0, src = -1;
      for (int q = 0; q < n_gaps; ++q) {
        const int4 rq = s_run[grp][q];
        if (o >= rq.x) {
          src = o < rq.x + rq.y ? rq.w + (o - rq.x) : -1;
          tok_shift = rq.z;
        }
      }
      const int v = src >= 0 ? qgene[src] : s_tk[grp][o - tok_shift];
      a.tmp_tok[rec.dst + o] = v;
    }
    if (l16 == 0) a.new_len[rec.r] = (unsigned int)ng;
  }
  // the reads left to the wave-per-read kernel: a flag per read (a list would need a counter, and one counter word
  // takes ~100 returning atomics per microsecond: 30 k waves with a read to hand over were 0.3 ms of this kernel)
  if (have && l16 == 0) left[gi] = ok ? 0 : 1;
}

// GF_WPB reads (waves) per workgroup.  The LDS of a workgroup is held until its LAST wave is done
// and reads differ a lot in work (runs, paths): with four waves per workgroup the kernel ran at
// half its occupancy limit waiting for stragglers (2.05 ms; 1.83 ms with two, 1.80 ms with one).
#define GF_WPB 1
struct GfLds {  // one read's staging
  int node[GF_MAXW];
  signed char dir[GF_MAXW];
  int tok[GF_MAXW + AMG_MAX_K];
  int gap[GF_MAXGAP * 3];  // ps, pe, n_paths
  int pool[GF_POOL];
  int used;
  int cnode[GF_CAND];
  signed char cdir[GF_CAND];
  int gene[GF_CAND + AMG_MAX_K];
  int best[GF_CAND + AMG_MAX_K];
};

__device__ __forceinline__ void gapped_fast_read(const GapArgs& A, long long gi, int lane, GfLds& S) {
  const CorrArgs& a = A.a;
  const GView& g = A.g;
  const int wv = 0;
  int (*s_node)[GF_MAXW] = &S.node;
  signed char (*s_dir)[GF_MAXW] = &S.dir;
  int (*s_tok)[GF_MAXW + AMG_MAX_K] = &S.tok;
  int (*s_gap)[GF_MAXGAP * 3] = &S.gap;
  int (*s_pool)[GF_POOL] = &S.pool;
  int* s_used = &S.used;
  int (*s_cnode)[GF_CAND] = &S.cnode;
  signed char (*s_cdir)[GF_CAND] = &S.cdir;
  int (*s_gene)[GF_CAND + AMG_MAX_K] = &S.gene;
  int (*s_best)[GF_CAND + AMG_MAX_K] = &S.best;
  const GapRec rec = A.rec[gi];
  const long long r = rec.r, t0 = rec.t0, dst = rec.dst;
  const int L0 = rec.L0;
  const int nwin = L0 - g.k + 1;
  const int start = rec.start, end = rec.end;
  if (nwin > GF_MAXW) {
    if (lane == 0) A.need_slow[gi] = 1;
    return;
  }
  int* W = s_node[wv];
  signed char* Dr = s_dir[wv];
  int* TK = s_tok[wv];
  int* GAP = s_gap[wv];
  int* POOL = s_pool[wv];
  for (int i = lane; i < nwin; i += 64) {
    W[i] = a.tok_node[t0 + i];
    Dr[i] = a.tok_dir[t0 + i];
  }
  for (int i = lane; i < L0; i += 64) TK[i] = a.tokens[t0 + i];
  // the read's question slots in the path memo ([0] < 0: none; entries past its runs are not initialised)
  const int myslot = (A.gq && lane < GF_MAXGAP) ? A.gq[gi * GF_MAXGAP + lane] : -1;
  if (lane == 0) s_used[wv] = 0;
  wave_sync();
  // ---- None runs in [start, end] (identify_path_terminals), in read order
  int n_gaps = 0;
  for (int c0 = start; c0 <= end; c0 += 64) {
    const int i = c0 + lane;
    const bool is_end = i <= end && W[i] < 0 && W[i + 1] >= 0;  // i < end whenever W[i] < 0
    const unsigned long long m = __ballot(is_end);
    if (is_end) {
      int q = n_gaps + __popcll(m & ((1ull << lane) - 1ull));
      if (q < GF_MAXGAP) {
        int ps = i - 1;
        while (W[ps] < 0) --ps;
        GAP[3 * q] = ps;
        GAP[3 * q + 1] = i + 1;
        GAP[3 * q + 2] = 0;
      }
    }
    n_gaps += __popcll(m);
  }
  if (n_gaps > GF_MAXGAP) {
    if (lane == 0) A.need_slow[gi] = 1;
    return;
  }
  wave_sync();
  // ---- the paths of every run, runs in read order: copied from the memo when the read's questions were entered
  // there (k_gap_queries) — lane q looks run q up, then all copies are in flight together — otherwise one
  // wave-cooperative DFS per run
  const bool memo = __builtin_amdgcn_readfirstlane(myslot) >= 0;
  bool bad = false;
  if (memo) {
    int4 res = make_int4(0, 0, 0, 0);
    if (lane < n_gaps) res = A.qres[myslot];
    const int len = res.y > 0 ? res.y : 0;
    int at = len;  // inclusive prefix over the runs (lanes < n_gaps <= 16)
#pragma unroll
    for (int d = 1; d < GF_MAXGAP; d <<= 1) {
      const int o = __shfl_up(at, d, 64);
      if (lane >= d) at += o;
    }
    const int total = __shfl(at, n_gaps - 1, 64);
    at -= len;
    bad = __any(res.y < 0) || total > GF_POOL;
    if (!bad) {
      for (int q = 0; q < n_gaps; ++q) {
        const int src = __shfl(res.x, q, 64), ln = __shfl(len, q, 64), dst0 = __shfl(at, q, 64);
        for (int i = lane; i < ln; i += 64) POOL[dst0 + i] = A.qpool[src + i];
      }
      wave_sync();
      if (lane < n_gaps) {
        for (int o = at; o < at + len; o += 2 + 2 * POOL[o + 1]) POOL[o] = lane;  // the records' run field
        GAP[3 * lane + 2] = res.z;
      }
      if (lane == 0) s_used[wv] = total;
      wave_sync();
    }
  } else {
    for (int q = 0; q < n_gaps && !bad; ++q) {
      const int ps = GAP[3 * q], pe = GAP[3 * q + 1];
      const int np = dfs_paths_wave(g, W[ps], Dr[ps], W[pe], 2 * g.k, q, POOL, &s_used[wv], lane);
      if (lane == 0) GAP[3 * q + 2] = np < 0 ? 0 : np;
      bad = np < 0;
    }
  }
  if (bad) {
    if (lane == 0) A.need_slow[gi] = 1;
    return;
  }
  wave_sync();
  unsigned long long n_combo = 1;
  bool dead_end = false;
  for (int q = 0; q < n_gaps; ++q) {
    int np = GAP[3 * q + 2];
    dead_end = dead_end || np == 0;
    n_combo *= (unsigned long long)np;
    if (n_combo > GF_MAXCOMBO) break;
  }
  if (dead_end) {
    // possible_paths == []: the original genes (and positions) are kept (:1292-1293);
    // the pack step copies them
    if (lane == 0) {
      a.new_len[r] = (unsigned int)L0;
      A.final_cls[r] = RC_KEEP_ORIG;
    }
    return;
  }
  if (n_combo > GF_MAXCOMBO) {
    if (lane == 0) A.need_slow[gi] = 1;
    return;
  }
  int* CN = s_cnode[wv];
  signed char* CD = s_cdir[wv];
  int* GN = s_gene[wv];
  int* BEST = s_best[wv];
  const int used = s_used[wv];
  int best_shared = 0, best_ng = -1;
  unsigned long long best_sum = 0, best_len = 1;
  for (unsigned long long combo = 0; combo < n_combo; ++combo) {
    // ---- candidate node list: live windows + the chosen path of every run.  Control flow is
    // wave-uniform (one step per run, not per window), the copies are lane-parallel.
    int n = 0;
    {
      int i = start, prev_pe = -1;
      bool over = false;
      for (int q = 0; q < n_gaps && !over; ++q) {
        const int ps = GAP[3 * q], pe = GAP[3 * q + 1], np = GAP[3 * q + 2];
        // windows [i, ps) are live (a None run is maximal): copied as they are
        const int cnt = ps - i;
        if (cnt > 0) {
          if (n + cnt > GF_CAND) { over = true; break; }
          for (int j = lane; j < cnt; j += 64) {
            CN[n + j] = W[i + j];
            CD[n + j] = Dr[i + j];
          }
          n += cnt;
        }
        unsigned long long div = 1;
        for (int j = q + 1; j < n_gaps; ++j) div *= (unsigned long long)GAP[3 * j + 2];
        int pick = (int)((combo / div) % (unsigned long long)np);
        int off = 0;
        while (off < used) {  // records of run q appear in DFS order
          if (POOL[off] == q) {
            if (pick == 0) break;
            --pick;
          }
          off += 2 + 2 * POOL[off + 1];
        }
        const int L = POOL[off + 1];
        if (prev_pe == ps && n > 0) --n;  // consecutive runs share their terminal node
        if (n + L > GF_CAND) { over = true; break; }
        for (int j = lane; j < L; j += 64) {
          CN[n + j] = POOL[off + 2 + j];
          CD[n + j] = (signed char)POOL[off + 2 + L + j];
        }
        n += L;
        prev_pe = pe;
        i = (q + 1 < n_gaps && GAP[3 * (q + 1)] == pe) ? pe : pe + 1;
      }
      if (!over) {
        const int cnt = end - i + 1;
        if (cnt > 0) {
          if (n + cnt > GF_CAND) {
            over = true;
          } else {
            for (int j = lane; j < cnt; j += 64) {
              CN[n + j] = W[i + j];
              CD[n + j] = Dr[i + j];
            }
            n += cnt;
          }
        }
      }
      if (over) n = -1;
    }
    wave_sync();
    if (n < 0) {
      if (lane == 0) A.need_slow[gi] = 1;
      return;
    }
    const int ng = n + g.k - 1;
    // ---- genes (get_annotation_for_read) and coverage sum, lane-parallel
    unsigned long long csum = 0;
    for (int q = lane; q < ng; q += 64) {
      const int idx = q < g.k - 1 ? 0 : q - (g.k - 1);
      const int j = q < g.k - 1 ? q : g.k - 1;
      GN[q] = oriented_tok(g, CN[idx], CD[idx], j);
    }
    if (n_combo > 1) {  // the mean coverage only ranks candidates against each other
      for (int q = lane; q < n; q += 64) csum += g.n_cov[CN[q]];
      for (int d = 32; d > 0; d >>= 1) csum += __shfl_xor(csum, d, 64);
    }
    wave_sync();
    bool better = true;
    if (n_combo > 1) {
      // len(set(genes) & set(original genes))
      int shared = 0;
      for (int q = lane; q < ng; q += 64) {
        const int tk = GN[q];
        bool dup = false;
        for (int w = 0; w < q && !dup; ++w) dup = (GN[w] == tk);
        bool hit = false;
        if (!dup)
          for (int w = 0; w < L0 && !hit; ++w) hit = (TK[w] == tk);
        shared += hit ? 1 : 0;
      }
      for (int d = 32; d > 0; d >>= 1) shared += __shfl_xor(shared, d, 64);
      better = shared > best_shared ||
               (shared == best_shared && csum * best_len > best_sum * (unsigned long long)n);
      if (better) best_shared = shared;
    }
    if (better) {
      best_sum = csum;
      best_len = (unsigned long long)n;
      best_ng = ng;
      for (int q = lane; q < ng; q += 64) BEST[q] = GN[q];
    }
    wave_sync();
  }
  for (int q = lane; q < best_ng; q += 64) a.tmp_tok[dst + q] = BEST[q];
  if (lane == 0) a.new_len[r] = (unsigned int)best_ng;
}

// left == nullptr: every re-threaded read, one per workgroup; else the reads k_corr_gapped_lean flagged: a workgroup
// takes LEAN_CHUNK consecutive reads, lane l looks at read l's flag and the wave works through the flagged ones
#define LEAN_CHUNK 32
__global__ __launch_bounds__(64 * GF_WPB, 8) void k_corr_gapped_fast(GapArgs A, const unsigned char* __restrict__ left) {
  __shared__ GfLds s_lds;
  const int lane = (int)threadIdx.x;
  if (!left) {
    if ((long long)blockIdx.x < A.n_gapped) gapped_fast_read(A, (long long)blockIdx.x, lane, s_lds);
    return;
  }
  const long long base = (long long)blockIdx.x * LEAN_CHUNK;
  const bool mine = lane < LEAN_CHUNK && base + lane < A.n_gapped && left[base + lane] != 0;
  unsigned long long todo = __ballot(mine);
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    gapped_fast_read(A, base + b, lane, s_lds);
    wave_sync();  // the next read reuses the staging
  }
}


// ---- positions for gapped reads: one wave per read (general path: any N, M)
#define NW_LDS_N 1024       // rows kept in LDS (rolling anti-diagonals, op list)
#define NW_LDS_CELLS 16384  // pointer-matrix cells kept in LDS (one byte each)

struct __attribute__((aligned(16))) NwRec {
  int r, M, N, pad;
  long long t0, dst;  // first token of the read, first staged gene of its corrected version
  long long pdst;     // where the read's new positions go in the pool of produced positions
  long long poff;     // pool index of the read's ORIGINAL positions
};

struct NwArgs {
  CorrArgs a;
  const NwRec* rec;  // per gapped read, written by k_nw_sizes
  long long* o_gs;   // gene positions of the corrected set: the carry-over writes its reads' entries
  long long* o_ge;   // directly (no staging copy for the pack step to move again)
  const int* gapped_reads;
  long long n_gapped;
  const unsigned char* final_cls;
  const long long* big_off;  // per gapped read: byte offset of its global scratch (big reads only)
  unsigned char* big_buf;
  int allow_fast;            // 0: every read takes the general kernel (debugging / A-B switch)
  int shortcuts;             // 0: k_corr_nw_fast fills a matrix for every read (AMG_NW_NO_SHORTCUT=1: test switch)
};

__global__ __launch_bounds__(64) void k_corr_nw(NwArgs A) {
  __shared__ unsigned char s_ptr[NW_LDS_CELLS];
  __shared__ int s_diag[3 * (NW_LDS_N + 1)];
  __shared__ unsigned char s_ops[2 * NW_LDS_N];
  const CorrArgs& a = A.a;
  const long long gi = blockIdx.x;
  if (gi >= A.n_gapped) return;
  const long long r = A.rec[gi].r;
  const long long pdst = A.rec[gi].pdst;
  if (A.final_cls[r] == RC_KEEP_ORIG) return;  // original genes kept: positions untouched
  const int lane = threadIdx.x;
  const long long t0 = a.read_off[r];
  const int M = (int)(a.read_off[r + 1] - t0);  // y = original genes
  const int N = (int)a.new_len[r];              // x = corrected genes
  if (A.allow_fast && nw_fast_ok(N, M)) return;  // k_corr_nw_fast handles it
  const long long dst = a.tmp_off[r];
  const int* x = a.tmp_tok + dst;
  const int* y = a.tokens + t0;
  const long long *ogs, *oge;
  pos_base(a, A.rec[gi].poff, ogs, oge);
  unsigned char* P = s_ptr;
  int* dg = s_diag;
  unsigned char* ops = s_ops;
  const bool small = (N <= NW_LDS_N && M <= NW_LDS_N && (long long)N * M <= NW_LDS_CELLS);
  if (!small) {
    unsigned char* base = A.big_buf + A.big_off[gi];
    P = base;
    ops = base + (long long)N * M;
    dg = reinterpret_cast<int*>(base + (((long long)N * M + N + M + 15) & ~15ll));
  }
  int* d0 = dg;            // anti-diagonal d-2, entry i+1 holds F[i, d-2-i]
  int* d1 = d0 + (N + 1);  // anti-diagonal d-1
  int* d2 = d1 + (N + 1);  // anti-diagonal d
  // borders (:1439-1445): F[-1,-1] = 0, F[i,-1] = -i, F[-1,j] = -j
  for (int d = 0; d <= N + M - 2; ++d) {
    int ilo = d - (M - 1) > 0 ? d - (M - 1) : 0;
    int ihi = d < N - 1 ? d : N - 1;
    for (int i = ilo + lane; i <= ihi; i += 64) {
      int j = d - i;
      int f_dd = (i == 0 && j == 0) ? 0 : (i == 0 ? -(j - 1) : (j == 0 ? -(i - 1) : d0[i]));
      int f_im1 = (i == 0) ? -j : d1[i];      // F[i-1, j]
      int f_jm1 = (j == 0) ? -i : d1[i + 1];  // F[i, j-1]
      int s_diag_ = f_dd + (x[i] == y[j] ? 1 : 0);
      int s_left = f_im1 - 1;  // pointer LEFT = (-1, 0)
      int s_up = f_jm1 - 1;    // pointer UP   = (0, -1)
      // max over (score, pointer) tuples: on ties UP (0,-1) > LEFT (-1,0) > DIAG (-1,-1)
      int best = s_diag_;
      unsigned char ptr = 0;
      if (s_left >= best) { best = s_left; ptr = 1; }
      if (s_up >= best) { best = s_up; ptr = 2; }
      d2[i + 1] = best;
      P[(long long)i * M + j] = ptr;
    }
    __syncthreads();
    int* t = d0; d0 = d1; d1 = d2; d2 = t;
  }
  if (lane != 0) return;
  // traceback (:1458-1480); ops are collected back to front
  int n_ops = 0;
  int i = N - 1, j = M - 1;
  while (i >= 0 && j >= 0) {
    unsigned char p = P[(long long)i * M + j];
    ops[n_ops++] = p;
    if (p == 0) { --i; --j; }
    else if (p == 1) --i;
    else --j;
  }
  while (i >= 0) { ops[n_ops++] = 1; --i; }
  while (j >= 0) { ops[n_ops++] = 2; --j; }
  // carry positions over (:1314-1325), alignment walked front to back.  A mismatching
  // diagonal column yields (None, None) WITHOUT consuming an original position.
  const long long NONE = (long long)0x8000000000000000ull;
  int xi = 0, yj = 0, cur = 0, out = 0;
  for (int o = n_ops - 1; o >= 0; --o) {
    unsigned char p = ops[o];
    if (p == 0) {
      if (x[xi] == y[yj]) {
        A.o_gs[pdst + out] = ogs[cur];
        A.o_ge[pdst + out] = oge[cur];
        ++cur;
      } else {
        A.o_gs[pdst + out] = NONE;
        A.o_ge[pdst + out] = NONE;
      }
      ++out; ++xi; ++yj;
    } else if (p == 1) {
      A.o_gs[pdst + out] = NONE;
      A.o_ge[pdst + out] = NONE;
      ++out; ++xi;
    } else {
      ++cur; ++yj;
    }
  }
  // replace_invalid_gene_positions (:1669-1691): prev_end is the end of the last entry that
  // was valid BEFORE repair; the look-ahead only sees entries that are still unrepaired.
  long long prev_end = 0;
  const long long rl = a.read_len ? a.read_len[r] : 0;
  for (int q = 0; q < N; ++q) {
    long long sv = A.o_gs[pdst + q], ev = A.o_ge[pdst + q];
    if (ev != NONE) prev_end = ev;
    if (sv == NONE && ev == NONE) {
      long long nxt = NONE;
      for (int w = q + 1; w < N; ++w)
        if (A.o_gs[pdst + w] != NONE) { nxt = A.o_gs[pdst + w]; break; }
      A.o_gs[pdst + q] = prev_end;
      A.o_ge[pdst + q] = (nxt != NONE) ? nxt : rl - 1;
    }
  }
}


// ---- fast path: M <= 64 original genes, N <= 128 corrected genes.  One wave per read, no
// workgroup barriers: lane j owns column j of the DP matrix and the wave computes one row per
// step — F[i-1,j] stays in the lane's own register, F[i-1,j-1] arrives from lane j-1 by a DPP
// shift, the dependency along the row is a prefix maximum (DPP scan).  Pointers are packed
// 2 bits per cell (16 rows per LDS word per lane).

#define NWF_WPB 1  // reads per workgroup (see GF_WPB: most reads take the shortcut, some fill a matrix:
                   // 0.94 ms with four, 0.87 with two, 0.71 with one)
struct NwfLds {  // one read's staging
  int x[NWF_MAX_N];
  unsigned int opw[(NWF_MAX_N + NWF_MAX_M) / 16 + 1];  // alignment ops, 2 bits each
  long long gs[NWF_MAX_N];
  long long ge[NWF_MAX_N];
  long long ogs[NWF_MAX_M];  // positions of the original genes
  long long oge[NWF_MAX_M];
};

__device__ __forceinline__ void nw_fast_read(const NwArgs& A, long long gi, int lane, NwfLds& S) {
  int (*s_x)[NWF_MAX_N] = &S.x;
  unsigned int (*s_opw)[(NWF_MAX_N + NWF_MAX_M) / 16 + 1] = &S.opw;
  long long (*s_gs)[NWF_MAX_N] = &S.gs;
  long long (*s_ge)[NWF_MAX_N] = &S.ge;
  long long (*s_ogs)[NWF_MAX_M] = &S.ogs;
  long long (*s_oge)[NWF_MAX_M] = &S.oge;
  const CorrArgs& a = A.a;
  const int wv = 0;
  // The kernel is bound by its chain of dependent global loads (one wave per read, ~15 us per
  // wave at full occupancy), not by the fill: one record load, then every per-gene load of the
  // read in one batch (the original positions included: the carry-over below reads them from
  // LDS), then only stores.
  const NwRec q = A.rec[gi];
  // wave-uniform by construction; tell the compiler so that loop control stays scalar
  const int N = __builtin_amdgcn_readfirstlane(q.N);
  if (N == 0) return;  // original genes kept, or a read for k_corr_nw
  const int M = __builtin_amdgcn_readfirstlane(q.M);
  const long long r = q.r, t0 = q.t0, dst = q.dst, pdst = q.pdst;
  int* X = s_x[wv];
  unsigned int* OPW = s_opw[wv];
  long long* GS = s_gs[wv];
  long long* GE = s_ge[wv];
  long long* OGS = s_ogs[wv];
  long long* OGE = s_oge[wv];
  const int x0 = lane < N ? a.tmp_tok[dst + lane] : -2;            // corrected genes 0..63
  const int x1 = lane + 64 < N ? a.tmp_tok[dst + lane + 64] : -2;  // and 64..127, one per lane
  const int yj = lane < M ? a.tokens[t0 + lane] : -1;
  const long long *pgs, *pge;
  pos_base(a, q.poff, pgs, pge);
  const long long ogs = lane < M ? pgs[lane] : 0;
  const long long oge = lane < M ? pge[lane] : 0;
  const long long rl = a.read_len ? a.read_len[r] : 0;  // (with the batch: not a third dependent round trip)
  if (lane < N) X[lane] = x0;
  if (lane + 64 < N) X[lane + 64] = x1;
  OGS[lane] = ogs;
  OGE[lane] = oge;
  // ---- shortcut: equally long gene lists that differ in at most two places.
  // With the reference's scores (match +1, mismatch 0, gap -1, and borders F[i,-1] = -i,
  // F[-1,j] = -j that make the first gap of a LEADING run free) an alignment of two lists of
  // the same length N with p >= 1 gaps in each scores at most (N - p) - 2p + 1 <= N - 2, the
  // pure diagonal N - m for m mismatching places.  m <= 1: the diagonal is the only optimum.
  // m == 2 (mismatches at a < b): N - 2 is reached only by "one free leading gap, N - 1
  // matches, one gap of the other kind somewhere": y[0] skipped, x[i] == y[i+1] up to the gap
  // that skips x[j], plain matches after it — which needs j >= b (no mismatch may follow the
  // gap) and therefore x[i] == y[i+1] for all i < b; or the mirror image with x[i+1] == y[i].
  // If neither holds the diagonal is again the only optimum.  A unique optimum is what the
  // traceback returns whatever the tie order, so the matrix is not needed: columns are
  // (x[q], y[q]), a mismatching column gives (None, None) and does not consume an original
  // position (:1314-1325).  (Tandem gene arrays do produce the tie: found by tools/fuzz_sweep.py,
  // kept as tests/golden/data/nw_tie_case.json.xz.)
  const long long NONE = (long long)0x8000000000000000ull;
  bool diagonal = false;
  if (N == M) {
    const unsigned long long mm = __ballot(lane < N && x0 != yj);
    const int m = __popcll(mm);
    diagonal = m <= 1;
    const int x_next = __shfl_down(x0, 1, 64), y_next = __shfl_down(yj, 1, 64);  // every lane shuffles
    const unsigned long long eq_a = __ballot(lane < N - 1 && x0 == y_next);  // x[i] == y[i+1]
    const unsigned long long eq_b = __ballot(lane < N - 1 && x_next == yj);  // x[i+1] == y[i]
    if (m == 2) {
      const int b = 63 - __clzll((long long)mm);               // the later mismatch, b >= 1
      const unsigned long long upto_b = (1ull << b) - 1ull;    // places 0 .. b-1
      diagonal = (eq_a & upto_b) != upto_b && (eq_b & upto_b) != upto_b;
    } else if (m == 3 || m == 4) {
      // Three or four mismatches: the diagonal scores N - m >= N - 4, any alignment with two or more gaps per
      // list at most N - 5, so only the alignments with ONE gap in each list can reach it.  Such an alignment
      // runs on the diagonal up to its first gap at u, one place off it (x[i] against y[i+1], or the mirror
      // image) up to its second gap at l, and on the diagonal again; it scores its matches - 2, + 1 when the
      // first gap is a leading one (u = 0: the first gap of a leading run is free).  With the match indicators
      // as bit masks (D diagonal, S shifted) its matches are prefD(u) - prefS(u) + prefS(l) + sufD(l): the best
      // over u <= l is a prefix maximum over the lanes.  If even the best such alignment stays BELOW N - m the
      // diagonal is the unique optimum (a tie would not do: the traceback prefers gaps).  Checked exhaustively
      // against the reference's alignment in tests/test_nw_shortcut_cpu.py.
      const unsigned long long dmask = ~mm & (N == 64 ? ~0ull : ((1ull << N) - 1ull));
      const unsigned long long below = (1ull << lane) - 1ull;
      const int prefD = __popcll(dmask & below);
      const int sufD = lane >= 63 ? 0 : __popcll(dmask >> (lane + 1));
      const int ID = (int)0x80000000 / 2;
      int best = ID;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int prefS = __popcll((side == 0 ? eq_a : eq_b) & below);
        int g = lane < N ? prefD - prefS + (lane == 0 ? 1 : 0) : ID;
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x111, 0xf, 0xf, false));  // row_shr:1
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x112, 0xf, 0xf, false));  // row_shr:2
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x114, 0xf, 0xf, false));  // row_shr:4
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x118, 0xf, 0xf, false));  // row_shr:8
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x142, 0xa, 0xf, false));  // row_bcast:15
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x143, 0xc, 0xf, false));  // row_bcast:31
        int h = lane < N ? g + prefS + sufD - 2 : ID;
        for (int d = 32; d > 0; d >>= 1) h = max(h, __shfl_xor(h, d, 64));
        best = max(best, h);
      }
      diagonal = best < N - m;
    }
    diagonal = diagonal && A.shortcuts != 0;
    if (diagonal && lane < N) {
      const bool match = ((mm >> lane) & 1ull) == 0ull;
      const int cur = __popcll(~mm & ((1ull << lane) - 1ull));  // matches before this column
      GS[lane] = match ? OGS[cur] : NONE;
      GE[lane] = match ? OGE[cur] : NONE;
    }
  }
  // ---- second shortcut: the OFFSET-DIAGONAL CERTIFICATE, for a corrected list that is the original one with an end
  // trimmed off and a few genes replaced (one re-threaded read in four of the cleaning sweep, and every one of them
  // filled a matrix: ~3 000 instructions).  If every gene of x occurs in y at most once, and where it does at i + s for
  // ONE offset s in [0, M - N], and at least two genes match, then: nothing off diagonal s scores, a detour from it
  // costs two gaps, so every optimal alignment runs along diagonal s from the first to the last match; the ties that
  // remain (where the s leading and M - N - s trailing gaps sit among the unmatched genes at either end) never move a
  // matched column.  What the reference carries over (:1314-1325) is then: matched x[q] -> original position number
  // s + (matches before q) — a gap column and a match consume an original position, a mismatching column does not —
  // unmatched -> (None, None).  Checked against the reference's alignment exhaustively on short lists and on random
  // trimmed / substituted / repeated ones in tests/test_nw_shortcut_cpu.py.
  if (!diagonal && A.shortcuts != 0 && N <= M && N >= 2) {  // (M <= 64 here: lane j holds y[j])
    int s_off = 0x7fffffff;
    bool ok = true;
    unsigned long long matched = 0ull;
    for (int i = 0; i < N && ok; ++i) {  // wave-uniform: one gene of x against all of y per step
      const int g = __builtin_amdgcn_readlane(x0, i);
      const unsigned long long at = __ballot(lane < M && yj == g);
      if (at != 0ull) {
        const int si = __ffsll((long long)at) - 1 - i;
        ok = (at & (at - 1ull)) == 0ull && (s_off == 0x7fffffff || si == s_off);
        s_off = si;
        matched |= 1ull << i;
      }
    }
    ok = ok && s_off != 0x7fffffff && s_off >= 0 && s_off <= M - N && __popcll(matched) >= 2;
    if (ok) {
      diagonal = true;
      if (lane < N) {
        const bool match = ((matched >> lane) & 1ull) != 0ull;
        const int cur = s_off + __popcll(matched & ((1ull << lane) - 1ull));
        GS[lane] = match ? OGS[cur] : NONE;
        GE[lane] = match ? OGE[cur] : NONE;
      }
    }
  }
  if (!diagonal) {
    // ---- fill, one matrix ROW per step (N steps instead of the N + M - 1 anti-diagonals of a
    // systolic sweep, which also idles half the lanes while it ramps up and down).  Lane j owns
    // column j and keeps F[i-1, j].  With c_j = max(F[i-1,j-1] + match, F[i-1,j] - 1) the row is
    //   F[i, j] = max(c_j, F[i, j-1] - 1) = max_{k <= j} (c_k + k) - j   (F[i,-1] = -i enters as k = -1)
    // i.e. a prefix maximum over the lanes: six DPP steps.  The pointer follows from the three
    // candidates with the reference's tie order UP (0,-1) > LEFT (-1,0) > DIAG.  Pointers stay in
    // registers: 2 bits per cell, word b of lane j = rows 16b .. 16b+15 of column j.
    int Fp = -lane;  // F[-1, j] = -j
    unsigned int ptrs[NWF_MAX_N / 16];
  #pragma unroll
    for (int blk = 0; blk < NWF_MAX_N / 16; ++blk) {
      unsigned int acc = 0;
      const int iend = N < blk * 16 + 16 ? N : blk * 16 + 16;
      for (int i = blk * 16; i < iend; ++i) {
        const int xi = blk < 4 ? __builtin_amdgcn_readlane(x0, i) : __builtin_amdgcn_readlane(x1, i - 64);
        // F[i-1, j-1]; lane 0 takes the border F[i-1, -1] = -(i-1), F[-1,-1] = 0
        const int fd = __builtin_amdgcn_update_dpp(i == 0 ? 0 : 1 - i, Fp, 0x138, 0xf, 0xf, false);
        const int s_d = fd + (xi == yj ? 1 : 0);
        const int s_l = Fp - 1;  // from F[i-1, j]: pointer LEFT = (-1, 0)
        const int c = s_d > s_l ? s_d : s_l;
        int g = c + lane;
        const int ID = (int)0x80000000;
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x111, 0xf, 0xf, false));  // row_shr:1
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x112, 0xf, 0xf, false));  // row_shr:2
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x114, 0xf, 0xf, false));  // row_shr:4
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x118, 0xf, 0xf, false));  // row_shr:8
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x142, 0xa, 0xf, false));  // row_bcast:15
        g = max(g, __builtin_amdgcn_update_dpp(ID, g, 0x143, 0xc, 0xf, false));  // row_bcast:31
        const int Fc = max(g, -i - 1) - lane;  // F[i, j]
        // F[i, j-1] - 1: pointer UP = (0, -1); lane 0 takes the border F[i, -1] = -i
        const int s_u = __builtin_amdgcn_update_dpp(-i, Fc, 0x138, 0xf, 0xf, false) - 1;
        const unsigned int ptr = s_u >= c ? 2u : (s_l >= s_d ? 1u : 0u);
        acc |= ptr << ((i & 15) * 2);
        Fp = Fc;
      }
      ptrs[blk] = acc;
    }
    // ---- traceback on the scalar unit: i, j and the ops are wave-uniform, a pointer is one
    // v_readlane away (no LDS round trip per step).  Ops are collected back to front, 16 per word.
    int n_ops = 0;
    {
      int i = N - 1, j = M - 1;
      unsigned int pack = 0;
  #pragma unroll
      for (int blk = NWF_MAX_N / 16 - 1; blk >= 0; --blk) {
        while (i >= blk * 16 && j >= 0) {
          const unsigned int w = (unsigned int)__builtin_amdgcn_readlane((int)ptrs[blk], j);
          const unsigned int p = (w >> ((i & 15) * 2)) & 3u;
          pack |= p << ((n_ops & 15) * 2);
          if ((n_ops & 15) == 15) {
            if (lane == 0) OPW[n_ops >> 4] = pack;
            pack = 0;
          }
          ++n_ops;
          if (p == 0) { --i; --j; }
          else if (p == 1) --i;
          else --j;
        }
      }
      while (i >= 0) {  // leading corrected genes: LEFT
        pack |= 1u << ((n_ops & 15) * 2);
        if ((n_ops & 15) == 15) {
          if (lane == 0) OPW[n_ops >> 4] = pack;
          pack = 0;
        }
        ++n_ops;
        --i;
      }
      while (j >= 0) {  // leading original genes: UP
        pack |= 2u << ((n_ops & 15) * 2);
        if ((n_ops & 15) == 15) {
          if (lane == 0) OPW[n_ops >> 4] = pack;
          pack = 0;
        }
        ++n_ops;
        --j;
      }
      if ((n_ops & 15) != 0 && lane == 0) OPW[n_ops >> 4] = pack;
    }
    wave_sync();
    // ---- positions, in parallel over the alignment columns (front to back)
    int base_x = 0, base_y = 0, base_cur = 0;
    for (int c0 = 0; c0 < n_ops; c0 += 64) {
      const int f = c0 + lane;
      const bool in = f < n_ops;
      const int g = in ? n_ops - 1 - f : 0;
      const unsigned int op = in ? (OPW[g >> 4] >> ((g & 15) * 2)) & 3u : 3u;
      const bool isx = in && (op == 0 || op == 1), isy = in && (op == 0 || op == 2);
      const unsigned long long lt = (1ull << lane) - 1ull;
      const unsigned long long bx = __ballot(isx), by = __ballot(isy);
      const int xi = base_x + __popcll(bx & lt), yy = base_y + __popcll(by & lt);
      const int ysel = __shfl(yj, yy < 64 ? yy : 0, 64);  // all lanes take part in the shuffle
      const bool match = in && op == 0 && X[xi < NWF_MAX_N ? xi : 0] == ysel;
      const bool inc = in && (op == 2 || match);
      const unsigned long long bc = __ballot(inc);
      const int cur = base_cur + __popcll(bc & lt);
      if (isx) {
        GS[xi] = match ? OGS[cur < NWF_MAX_M ? cur : 0] : NONE;
        GE[xi] = match ? OGE[cur < NWF_MAX_M ? cur : 0] : NONE;
      }
      base_x += __popcll(bx);
      base_y += __popcll(by);
      base_cur += __popcll(bc);
    }
  }
  wave_sync();
  // ---- replace_invalid_gene_positions, each lane repairs its own entries
  for (int q = lane; q < N; q += 64) {
    long long sv = GS[q], ev = GE[q];
    if (sv == NONE && ev == NONE) {
      long long prev_end = 0;
      for (int w = q - 1; w >= 0; --w)
        if (GE[w] != NONE) { prev_end = GE[w]; break; }
      long long nxt = NONE;
      for (int w = q + 1; w < N; ++w)
        if (GS[w] != NONE) { nxt = GS[w]; break; }
      sv = prev_end;
      ev = (nxt != NONE) ? nxt : rl - 1;
    }
    A.o_gs[pdst + q] = sv;
    A.o_ge[pdst + q] = ev;
  }
}

// (Sixteen lanes per read for the reads whose alignment is provably the diagonal — three reads in four — were built and
// measured in round 5: 0.27 ms for them plus 0.56 ms for the others against 0.575 ms for everybody here.  The pass IS
// the reads that fill a matrix, ~3 000 instructions each; the diagonal ones ride along for nothing.)
__global__ __launch_bounds__(64 * NWF_WPB) void k_corr_nw_fast(NwArgs A) {
  __shared__ NwfLds s_lds;
  if ((long long)blockIdx.x < A.n_gapped) nw_fast_read(A, (long long)blockIdx.x, (int)threadIdx.x, s_lds);
}

// live nodes of the graph the reads were corrected against (with PackArgs::dead_kept: an upper bound for the nodes of
// the graph the corrected reads will make)
__global__ __launch_bounds__(256) void k_count_alive(const unsigned char* __restrict__ alive, long long n,
                                                     unsigned long long* out) {
  __shared__ unsigned int s_part[4];
  unsigned int c = 0;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      const unsigned int w = *reinterpret_cast<const unsigned int*>(alive + i);  // (flags are 0 / 1 bytes)
      c += __popc(w & 0x01010101u);
    } else {
      for (long long j = i; j < n; ++j) c += alive[j] ? 1u : 0u;
    }
  }
  for (int d = 32; d > 0; d >>= 1) c += (unsigned int)__shfl_xor((int)c, d, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0 && (s_part[0] | s_part[1] | s_part[2] | s_part[3]))
    atomicAdd(out, (unsigned long long)s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// ---- compaction into the corrected CSR: a read is kept when len(list_of_genes) > 0 (:1130)
struct PackArgs {
  CorrArgs a;
  const long long* new_idx;    // exscan(new_len > 0)
  const long long* new_off;    // exscan(new_len)
  const unsigned char* final_cls;
  int* o_tok;
  long long* o_off;
  int* o_orig;
  unsigned char* o_changed;
  long long* o_src;          // per corrected read: token index of its first gene in the current read set (-1: re-threaded)
  const long long* pos_new;  // per read: pool index of a re-threaded read's new positions
  long long* o_posoff;       // per corrected read: pool index of its positions
  long long* o_rl;
  long long out_reads, out_tokens;  // the corrected CSR's last offset: o_off[out_reads] = out_tokens
  unsigned long long* dead_kept;    // dead windows of the reads that keep their original genes: 16 partial sums, 16 words apart
};

// One wave packs 64 consecutive reads.  Lane l owns read l's record: where its genes come from (the read itself,
// a slice of it, or the temp area of a re-threaded read), where they go, how many, and its entry of the corrected
// CSR — all of it loaded and stored coalesced, one read per lane.  Then the 64 lanes copy the reads' genes one read
// at a time (source / destination / length broadcast with v_readlane), four reads in flight.  (A wave per four
// reads issued ~20 small vector-memory instructions per read and was bound by their issue, not by bytes: 0.6 ms for
// 0.56 GB.)  Gene positions stay where they are: the corrected read only records where its positions begin
// (CorrArgs).
#define PACK_READS 64
__global__ __launch_bounds__(256) void k_corr_pack(PackArgs A) {
  const CorrArgs& a = A.a;
  const int lane = threadIdx.x & 63;
  const long long r = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PACK_READS + lane;
  if (blockIdx.x == 0 && threadIdx.x == 0) A.o_off[A.out_reads] = A.out_tokens;
  long long dst = 0, src = 0;  // src: token index; bit 62 set = in the temp area
  int n = 0;
  unsigned int dead = 0;  // windows of removed nodes this read brings back (see amg_correct_reads: the next build's table)
  if (r < a.n_reads && a.new_len[r] > 0) {
    dst = A.new_off[r];
    n = (int)a.new_len[r];
    const unsigned char fc = A.final_cls[r];
    if (fc == RC_KEEP_ORIG) {
      const long long nw = a.read_off[r + 1] - a.read_off[r] - a.k + 1;
      dead = nw <= 64 ? (unsigned int)(nw - __popcll(a.lmask[r])) : (unsigned int)nw;
    }
    long long poff = 0;
    if (fc == RC_GAPPED) {  // re-threaded read: genes staged in the temp area, positions written to the pool by
      src = a.tmp_off[r] | (1ll << 62);  // the carry-over kernels
      if (a.have_pos) poff = A.pos_new[r];
    } else {  // untouched read, kept original, or a slice [start : end + k] of it (:1277-1285)
      const long long cut = fc == RC_TRIM ? a.r_start[r] : 0;
      src = a.read_off[r] + cut;
      if (a.have_pos) poff = (a.pos_off ? a.pos_off[r] : a.read_off[r]) + cut;
    }
    const long long q = A.new_idx[r];
    A.o_off[q] = dst;
    A.o_orig[q] = (int)r;
    A.o_changed[q] = (fc == RC_TRIM || fc == RC_GAPPED) ? 1 : 0;
    A.o_src[q] = fc == RC_GAPPED ? -1ll : src;
    if (a.have_pos) A.o_posoff[q] = poff;
    if (a.read_len) A.o_rl[q] = a.read_len[r];
  }
  if (__ballot(dead != 0u) != 0ull) {
    for (int d = 32; d > 0; d >>= 1) dead += (unsigned int)__shfl_xor((int)dead, d, 64);
    if (lane == 0) atomicAdd(A.dead_kept + 16 * (blockIdx.x & 15), (unsigned long long)dead);
  }
  if (__ballot(n > 0) == 0ull) return;
  for (int j0 = 0; j0 < PACK_READS; j0 += 4) {
    const int* sp[4];
    long long d[4];
    int nn[4], vt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nn[j] = __builtin_amdgcn_readlane(n, j0 + j);
      const long long sj = bcast_i64(src, j0 + j);
      d[j] = bcast_i64(dst, j0 + j);
      sp[j] = ((sj >> 62) & 1 ? a.tmp_tok : a.tokens) + (sj & ~(1ll << 62));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane < nn[j]) vt[j] = sp[j][lane];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane < nn[j]) A.o_tok[d[j] + lane] = vt[j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      for (int i = 64 + lane; i < nn[j]; i += 64)  // reads longer than one wave
        A.o_tok[d[j] + i] = sp[j][i];
  }
}

__global__ void k_scatter_gapped(const unsigned int* __restrict__ flag, const long long* __restrict__ pos,
                                 long long n_reads, int* __restrict__ out, const long long* __restrict__ read_off,
                                 const int* __restrict__ r_start, const int* __restrict__ r_end,
                                 const long long* __restrict__ tmp_off, const unsigned long long* __restrict__ lmask,
                                 GapRec* __restrict__ rec) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads || !flag[r]) return;
  const long long gi = pos[r];
  out[gi] = (int)r;
  GapRec q;
  q.r = (int)r;
  q.t0 = read_off[r];
  const long long len = read_off[r + 1] - q.t0;
  q.L0 = (int)(len > 0x7fffffff ? 0x7fffffff : len);
  q.start = r_start[r];
  q.end = r_end[r];
  q.dst = tmp_off[r];
  q.mask = lmask[r];
  q.pad = 0;
  rec[gi] = q;
}

// global NW scratch size of gapped read gi (0 when it fits the LDS path)
__global__ void k_nw_sizes(const int* __restrict__ gapped, long long n_gapped,
                           const long long* __restrict__ read_off, const unsigned int* __restrict__ new_len,
                           const long long* __restrict__ tmp_off, const long long* __restrict__ new_off,
                           const unsigned char* __restrict__ final_cls, long long* __restrict__ size,
                           int allow_fast, unsigned long long* n_general, NwRec* __restrict__ rec,
                           const long long* __restrict__ pos_off, long long* __restrict__ plen) {
  long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= n_gapped) return;
  long long r = gapped[gi];
  long long N = new_len[r], M = read_off[r + 1] - read_off[r];
  {
    // everything k_corr_nw_fast needs to start, in one 32-byte record (instead of a chain of
    // dependent per-read loads at the head of a latency-bound kernel)
    NwRec q;
    q.r = (int)r;
    q.M = (int)(M > 0x7fffffff ? 0x7fffffff : M);
    q.N = (final_cls[r] != RC_KEEP_ORIG && allow_fast && nw_fast_ok(N, M)) ? (int)N : 0;  // 0: not for the fast kernel
    q.pad = 0;
    q.t0 = read_off[r];
    q.dst = tmp_off[r];
    q.pdst = 0;  // k_nw_place
    q.poff = pos_off ? pos_off[r] : read_off[r];
    rec[gi] = q;
    plen[gi] = final_cls[r] != RC_KEEP_ORIG ? N : 0;  // new positions of this read
  }
  bool small = (N <= NW_LDS_N && M <= NW_LDS_N && N * M <= NW_LDS_CELLS);
  long long bytes = 0;
  if (!small && final_cls[r] != RC_KEEP_ORIG)
    bytes = ((N * M + N + M + 15) & ~15ll) + ((3 * (N + 1) * 4 + 15) & ~15ll);
  size[gi] = bytes;
  if (final_cls[r] != RC_KEEP_ORIG && !(allow_fast && nw_fast_ok(N, M))) atomicAdd(n_general, 1ull);
}


// where the new positions of gapped read gi go (pool of produced positions, after `base`)
__global__ void k_nw_place(long long n_gapped, const long long* __restrict__ poffs, long long base, long long n0,
                           NwRec* __restrict__ rec, long long* __restrict__ pos_new) {
  long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= n_gapped) return;
  const long long at = base + poffs[gi];
  rec[gi].pdst = at;
  pos_new[rec[gi].r] = n0 + at;
}

// positions of the corrected set, gathered from the pools (only when the host asks for them)
__global__ __launch_bounds__(256) void k_gather_positions(CorrArgs a, const long long* __restrict__ c_off,
                                                          const long long* __restrict__ c_posoff, long long c_reads,
                                                          long long* __restrict__ o_gs, long long* __restrict__ o_ge) {
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= c_reads) return;
  const int lane = threadIdx.x & 63;
  const long long a0 = c_off[q], n = c_off[q + 1] - a0;
  const long long *gs, *ge;
  pos_base(a, c_posoff[q], gs, ge);
  for (long long i = lane; i < n; i += 64) {
    o_gs[a0 + i] = gs[i];
    o_ge[a0 + i] = ge[i];
  }
}

// device allocation that keeps its first `used` bytes when it has to grow
static int grow_keep(amg_ctx* c, DevBuf& b, size_t need, size_t used) {
  if (need <= b.cap && !b.borrowed) return AMG_OK;
  DevBuf nb;
  AMGCHK(nb.ensure(need + need / 2));
  if (used && b.p) HIPCHK(hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  b.release();
  b = nb;
  return AMG_OK;
}

static void fill_pos_args(amg_ctx* c, CorrArgs& a) {
  a.p0s = c->have_pos ? c->gene_start.as<long long>() : nullptr;
  a.p0e = c->have_pos ? c->gene_end.as<long long>() : nullptr;
  a.p1s = c->pos1_s.as<long long>();
  a.p1e = c->pos1_e.as<long long>();
  a.n0 = c->pos_n0;
  a.pos_off = (c->have_pos && !c->pos_identity) ? c->pos_off.as<long long>() : nullptr;
}

extern "C" int amg_correct_reads(amg_ctx* c, int64_t* n_out_reads, int64_t* n_out_tokens) {
  NEED_BUILT(c);
  hipStream_t st = c->stream;
  const long long R = c->n_reads;
  stages_reset(c);
  c->have_corrected = false;
  // per-read scratch arrays (they must survive the scans, so each group has its own buffer)
  size_t per_read = (size_t)(R + 2);
  AMGCHK(c->s0.ensure(per_read * (1 + 1) + 64));                 // cls, final_cls
  AMGCHK(c->s1.ensure(per_read * sizeof(int) * 2 + per_read * sizeof(long long) * 2 + 64));  // r_start, r_end, nw sizes/offsets
  AMGCHK(c->s2.ensure(per_read * sizeof(unsigned int) * 3));     // bound, new_len, keep/flag
  AMGCHK(c->s3.ensure(per_read * sizeof(long long) * 3));        // tmp_off, new_idx, new_off
  unsigned char* cls = c->s0.as<unsigned char>();
  unsigned char* final_cls = cls + per_read;
  int* r_start = c->s1.as<int>();
  int* r_end = r_start + per_read;
  unsigned int* bound = c->s2.as<unsigned int>();
  unsigned int* new_len = bound + per_read;
  unsigned int* flag = new_len + per_read;
  long long* tmp_off = c->s3.as<long long>();
  long long* new_idx = tmp_off + per_read;
  long long* new_off = new_idx + per_read;

  CorrArgs a;
  a.tokens = c->tokens.as<int>();
  a.read_off = c->read_off.as<long long>();
  a.tok_node = c->tok_node.as<int>();
  a.tok_dir = c->tok_dir.as<signed char>();
  a.read_fix = c->read_fix.as<unsigned char>();
  fill_pos_args(c, a);
  a.read_len = c->have_read_len ? c->read_len.as<long long>() : nullptr;
  a.n_reads = R;
  a.k = c->k;
  a.flip = c->two_v - 1;
  a.have_pos = c->have_pos ? 1 : 0;
  a.cls = cls;
  a.cls_final = final_cls;
  a.r_start = r_start;
  a.r_end = r_end;
  a.bound = bound;
  a.tmp_off = tmp_off;
  a.new_len = new_len;
  a.tmp_tok = nullptr;

  stage_begin(c, "correct_classify");
  unsigned long long* mx = c->status.as<unsigned long long>() + ST_MISC;
  AMGCHK(c->gm_mask.ensure(per_read * sizeof(unsigned long long)));
  AMGCHK(c->gm_ctr.ensure((16 + 256 + 256 + 16) * sizeof(unsigned long long)));
  unsigned long long* n_runs_d = c->gm_ctr.as<unsigned long long>() + 16;  // [16 x 16 words]; [0..16) belong to the memo
  unsigned long long* dead_kept_d = n_runs_d + 256;                         // [16 x 16 words] (k_corr_pack), then live nodes
  {
    ClearList cl;
    cl.add(bound, per_read * sizeof(unsigned int) * 3);
    cl.add(mx, sizeof(unsigned long long));
    cl.add(n_runs_d, (256 + 256 + 16) * sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
  }
  a.gflag = flag;
  a.max_bound = mx;
  a.lmask = c->gm_mask.as<unsigned long long>();
  a.n_runs = n_runs_d;
  if (R > 0) hipLaunchKernelGGL(k_corr_classify, dim3(nblk(R, 4 * CLS_READS)), dim3(256), 0, st, a);  // also new_len, flag, max
  // (for the next build's table: how many nodes are alive; rides along, read back with the pack step's words)
  const bool count_alive = c->n_nodes >= (1ll << 20);
  if (count_alive)
    hipLaunchKernelGGL(k_count_alive, dim3(128), dim3(256), 0, st, c->node_alive.as<unsigned char>(), c->n_nodes,
                       dead_kept_d + 256);
  // where the re-threaded reads' genes are staged, and the list of gapped reads: two scans, one launch
  AMGCHK(prim_exscan_u32_pair(c, bound, tmp_off, flag, new_idx, (size_t)R));
  long long tmp_total = 0;
  unsigned long long max_bound = 0;

  long long n_gapped = 0, total_runs = 0;
  {
    FetchList l;
    l.add(tmp_off + R);
    l.add(mx);
    l.add(new_idx + R);
    for (int i = 0; i < 16; ++i) l.add(n_runs_d + 16 * i);
    unsigned long long v[3 + 16];
    AMGCHK(fetch(c, l, v));
    tmp_total = (long long)v[0];
    max_bound = v[1];
    n_gapped = (long long)v[2];
    for (int i = 0; i < 16; ++i) total_runs += (long long)v[3 + i];
  }
  stage_end(c);

  AMGCHK(c->c_tokens_buf.ensure((size_t)(tmp_total + 4) * sizeof(int)));  // temp tokens live here first
  DevBuf& tmpTok = c->s4;
  AMGCHK(tmpTok.ensure((size_t)(tmp_total + 4) * sizeof(int)));
  a.tmp_tok = tmpTok.as<int>();

  if (n_gapped > 0) {
    stage_begin(c, "live_adjacency");
    AMGCHK(ensure_live_adj(c));
    stage_end(c);
    stage_begin(c, "correct_gapped");
    // gapped read list, path pool, candidate scratch: their own allocations
    DevBuf& glist = c->c_orig;  // free until the pack step
    AMGCHK(glist.ensure((size_t)(n_gapped + 1) * sizeof(int)));
    AMGCHK(c->gap_rec.ensure((size_t)(n_gapped + 1) * sizeof(GapRec)));
    hipLaunchKernelGGL(k_scatter_gapped, dim3(nblk(R, 256)), dim3(256), 0, st, flag, new_idx, R,
                       glist.as<int>(), a.read_off, r_start, r_end, tmp_off, a.lmask, c->gap_rec.as<GapRec>());
    // ---- path memo: every distinct (start, direction, end) question of the None runs is answered once
    const int* gq = nullptr;
    {
      const char* nm = getenv("AMG_NO_GAP_MEMO");  // A/B switch
      if (total_runs > 0 && !(nm && nm[0] == '1')) {
        const uint64_t qslots = pow2_at_least((uint64_t)total_runs * 2 + 16);
        AMGCHK(c->gm_tab.ensure((size_t)qslots * sizeof(unsigned long long)));
        AMGCHK(c->gm_res.ensure((size_t)qslots * sizeof(int4)));
        AMGCHK(c->gm_list.ensure((size_t)(total_runs + 1) * sizeof(int)));
        AMGCHK(c->gm_q.ensure((size_t)(n_gapped + 1) * GF_MAXGAP * sizeof(int)));
        {
          ClearList cl;
          cl.add(c->gm_tab.p, (size_t)qslots * sizeof(unsigned long long));
          cl.add(c->gm_ctr.p, 16 * sizeof(unsigned long long));
          cl.add(c->status.as<unsigned long long>() + ST_OVERFLOW, sizeof(unsigned long long));
          AMGCHK(clear_many(c, cl));
        }
        hipLaunchKernelGGL(k_gap_queries, dim3(nblk(n_gapped, 256)), dim3(256), 0, st, c->gap_rec.as<GapRec>(), n_gapped,
                           c->k, a.tok_node, a.tok_dir, c->gm_tab.as<unsigned long long>(), (unsigned int)(qslots - 1),
                           c->gm_ctr.as<unsigned long long>(), c->gm_list.as<int>(), c->gm_q.as<int>(),
                           c->status.as<unsigned long long>());
        unsigned long long v[2] = {0, 0};
        {
          FetchList l;
          l.add(c->gm_ctr.p);
          l.add(c->status.as<unsigned long long>() + ST_OVERFLOW);
          AMGCHK(fetch(c, l, v));
        }
        if (v[1]) return amg_fail(AMG_E_HIP, "correct_reads: path memo table full");
        const long long n_queries = (long long)v[0];
        // answers average ~20 ints; one that does not find room sends its reads to the general kernel
        const unsigned long long qcap = (unsigned long long)n_queries * (GM_INLINE + 32ull) + 4096ull;
        if (qcap <= 0x7fffffffull) {  // (pool offsets are ints)
        AMGCHK(c->gm_pool.ensure((size_t)qcap * sizeof(int)));
        AMGCHK(c->gm_gene.ensure((size_t)qcap * sizeof(int)));
        if (n_queries > 0)
          hipLaunchKernelGGL(k_gap_dfs, dim3((unsigned int)n_queries), dim3(64), 0, st, make_view(c), c->gm_list.as<int>(),
                             n_queries, c->gm_tab.as<unsigned long long>(), c->gm_ctr.as<unsigned long long>() + 1, qcap,
                             c->gm_pool.as<int>(), c->gm_res.as<int4>(), c->gm_gene.as<int>());
        gq = c->gm_q.as<int>();
        }
      }
    }
    const unsigned int threads_total = 64u * 2048u;
    unsigned int cand_stride = (unsigned int)(2 * max_bound + (max_bound + 3) / 4 + c->k + 16);
    DevBuf& cand = c->c_gstart;  // free until the pack step
    AMGCHK(cand.ensure((size_t)threads_total * cand_stride * sizeof(int)));
    // the general kernel only sees what the fast kernel hands over: start small, grow on demand
    unsigned long long pool_cap = 1ull << 22;
    AMGCHK(c->c_changed.ensure((size_t)n_gapped + 64));  // free until the pack step
    unsigned char* need_slow = c->c_changed.as<unsigned char>();
    for (int attempt = 0;; ++attempt) {
      DevBuf& pool = c->c_gend;  // free until the pack step
      AMGCHK(pool.ensure((size_t)pool_cap * sizeof(int)));
      unsigned long long* used = c->status.as<unsigned long long>() + ST_COMPACT_A;
      ClearList gcl;
      gcl.add(c->status.p, ST_WORDS * sizeof(unsigned long long));
      GapArgs G;
      G.a = a;
      G.g = make_view(c);
      G.rec = c->gap_rec.as<GapRec>();
      G.gapped_reads = glist.as<int>();
      G.n_gapped = n_gapped;
      G.pool = pool.as<int>();
      G.pool_cap = pool_cap;
      G.pool_used = used;
      G.status = c->status.as<unsigned long long>();
      G.cand = cand.as<int>();
      G.cand_stride = cand_stride;
      G.final_cls = final_cls;
      G.need_slow = need_slow;
      G.gq = gq;
      G.qres = c->gm_res.as<int4>();
      G.qpool = c->gm_pool.as<int>();
      if (attempt == 0) {
        const char* nf = getenv("AMG_NO_FAST_GAPPED");  // debugging / A-B switch
        const bool use_fast = !(nf && nf[0] == '1');
        gcl.add(need_slow, ((size_t)n_gapped + 4) & ~(size_t)3, use_fast ? 0u : 0x01010101u);  // (the buffer has 64 spare bytes)
        AMGCHK(clear_many(c, gcl));
        const char* nl = getenv("AMG_NO_LEAN_GAPPED");  // A/B + test switch: every read through the wave-per-read kernel
        if (use_fast && gq && !(nl && nl[0] == '1')) {
          // sixteen lanes per read where every question has one answer; the others are flagged for the wave-per-read
          // kernel (no count comes back to the host)
          AMGCHK(c->gm_fail.ensure((size_t)n_gapped + 64));
          hipLaunchKernelGGL(k_corr_gapped_lean, dim3(nblk(n_gapped, GL_THREADS / GL_GROUP)), dim3(GL_THREADS), 0, st, G,
                             c->gm_gene.as<int>(), c->gm_fail.as<unsigned char>());
          hipLaunchKernelGGL(k_corr_gapped_fast, dim3(nblk(n_gapped, LEAN_CHUNK)), dim3(64 * GF_WPB), 0, st, G,
                             c->gm_fail.as<unsigned char>());
        } else if (use_fast) {
          hipLaunchKernelGGL(k_corr_gapped_fast, dim3((unsigned int)n_gapped), dim3(64 * GF_WPB), 0, st, G,
                             (const unsigned char*)nullptr);
        }
      } else {
        AMGCHK(clear_many(c, gcl));  // (the status words alone)
      }
      unsigned int blocks = (unsigned int)((n_gapped + 63) / 64);
      if (blocks > 2048u) blocks = 2048u;
      hipLaunchKernelGGL(k_corr_gapped, dim3(blocks), dim3(64), 0, st, G);
      unsigned long long hs[ST_WORDS];
      AMGCHK(fetch_status(c, hs));
      if (!hs[ST_OVERFLOW]) break;
      if (attempt >= 8) return amg_fail(AMG_E_OVERFLOW, "correct_reads: path pool overflow");
      pool_cap = hs[ST_COMPACT_A] * 2 + (1ull << 20);
    }
    stage_end(c);

  }

  // ---- shape of the corrected set (before the position carry-over, which writes its reads'
  // positions straight into the output arrays)
  stage_begin(c, "correct_pack");
  // which reads stay and where their genes go: both scans in one launch, straight from the lengths
  AMGCHK(prim_exscan_keep_and_len(c, new_len, new_idx, new_off, (size_t)R));
  long long out_reads = 0, out_tokens = 0;
  FetchList shape;
  shape.add(new_idx + R);
  shape.add(new_off + R);
  const bool carry = n_gapped > 0 && c->have_pos;
  long long big_total = 0, pos_total = 0;
  unsigned long long n_general = 0;
  long long* nw_off = nullptr;
  long long *plen = nullptr, *poffs = nullptr, *pos_new = nullptr;
  const char* nfn = getenv("AMG_NO_FAST_NW");
  const int allow_fast = !(nfn && nfn[0] == '1');
  if (carry) {
    // global scratch only for reads too large for the register-resident kernel
    long long* nw_size = reinterpret_cast<long long*>(
        ((uintptr_t)(c->s1.as<int>() + 2 * per_read) + 15) & ~(uintptr_t)15);
    nw_off = nw_size + per_read;
    // per gapped read: number of new positions and their place in the pool; per read: pool index
    AMGCHK(c->s5.ensure((size_t)(2 * (n_gapped + 2) + R + 2) * sizeof(long long)));
    plen = c->s5.as<long long>();
    poffs = plen + (n_gapped + 2);
    pos_new = poffs + (n_gapped + 2);
    unsigned long long* n_general_d = c->status.as<unsigned long long>() + ST_MISC;
    {
      ClearList cl;
      cl.add(nw_size, (size_t)(n_gapped + 1) * sizeof(long long));
      cl.add(plen, (size_t)(n_gapped + 1) * sizeof(long long));
      cl.add(n_general_d, sizeof(unsigned long long));
      AMGCHK(clear_many(c, cl));
    }
    AMGCHK(c->nw_rec.ensure((size_t)(n_gapped + 1) * sizeof(NwRec)));
    hipLaunchKernelGGL(k_nw_sizes, dim3(nblk(n_gapped, 256)), dim3(256), 0, st, c->c_orig.as<int>(),
                       n_gapped, a.read_off, new_len, tmp_off, new_off, final_cls, nw_size, allow_fast,
                       n_general_d, c->nw_rec.as<NwRec>(), a.pos_off, plen);
    AMGCHK(prim_exscan_i64_pair(c, nw_size, nw_off, plen, poffs, (size_t)n_gapped));
    shape.add(nw_off + n_gapped);
    shape.add(poffs + n_gapped);
    shape.add(n_general_d);
  }
  {
    unsigned long long v[5] = {0, 0, 0, 0, 0};
    AMGCHK(fetch(c, shape, v));
    out_reads = (long long)v[0];
    out_tokens = (long long)v[1];
    big_total = (long long)v[2];
    pos_total = (long long)v[3];
    n_general = v[4];
  }
  stage_end(c);
  // from here on the scratch roles of the output buffers (gapped read list, path pool, candidate
  // scratch) are over
  AMGCHK(c->c_tokens_buf.ensure((size_t)(out_tokens + 64) * sizeof(int)));
  AMGCHK(c->c_read_off.ensure((size_t)(out_reads + 2) * sizeof(long long)));
  AMGCHK(c->c_orig.ensure((size_t)(out_reads + 2) * sizeof(int)));
  AMGCHK(c->c_changed.ensure((size_t)(out_reads + 2)));
  AMGCHK(c->c_src.ensure((size_t)(out_reads + 2) * sizeof(long long)));
  if (c->have_pos) AMGCHK(c->c_pos_off.ensure((size_t)(out_reads + 2) * sizeof(long long)));
  if (c->have_read_len) AMGCHK(c->c_read_len.ensure((size_t)(out_reads + 2) * sizeof(long long)));

  if (carry) {
    stage_begin(c, "correct_positions");
    AMGCHK(c->nw_big.ensure((size_t)big_total + 64));
    // the pool of produced positions grows by what this correction adds (earlier entries stay:
    // reads corrected before keep pointing at them)
    const size_t used = (size_t)c->pos1_used * sizeof(long long);
    const size_t need = (size_t)(c->pos1_used + pos_total + 64) * sizeof(long long);
    AMGCHK(grow_keep(c, c->pos1_s, need, used));
    AMGCHK(grow_keep(c, c->pos1_e, need, used));
    fill_pos_args(c, a);
    hipLaunchKernelGGL(k_nw_place, dim3(nblk(n_gapped, 256)), dim3(256), 0, st, n_gapped, poffs,
                       (long long)c->pos1_used, (long long)c->pos_n0, c->nw_rec.as<NwRec>(), pos_new);
    NwArgs W;
    W.a = a;
    W.rec = c->nw_rec.as<NwRec>();
    W.o_gs = c->pos1_s.as<long long>();
    W.o_ge = c->pos1_e.as<long long>();
    W.gapped_reads = nullptr;  // the records carry the read ids
    W.n_gapped = n_gapped;
    W.final_cls = final_cls;
    W.big_off = nw_off;
    W.big_buf = c->nw_big.as<unsigned char>();
    W.allow_fast = allow_fast;
    {
      const char* ns = getenv("AMG_NW_NO_SHORTCUT");
      W.shortcuts = !(ns && ns[0] == '1');
    }
    if (W.allow_fast)
      hipLaunchKernelGGL(k_corr_nw_fast, dim3((unsigned int)n_gapped), dim3(64 * NWF_WPB), 0, st, W);
    if (n_general > 0)  // reads too long for the register-resident kernel
      hipLaunchKernelGGL(k_corr_nw, dim3((unsigned int)n_gapped), dim3(64), 0, st, W);
    stage_end(c);
  }

  // ---- pack
  stage_begin(c, "correct_pack");
  PackArgs Pk;
  Pk.a = a;
  Pk.new_idx = new_idx;
  Pk.new_off = new_off;
  Pk.final_cls = final_cls;
  Pk.o_tok = c->c_tokens_buf.as<int>();
  Pk.o_off = c->c_read_off.as<long long>();
  Pk.o_orig = c->c_orig.as<int>();
  Pk.o_changed = c->c_changed.as<unsigned char>();
  Pk.o_src = c->c_src.as<long long>();
  Pk.pos_new = pos_new;
  Pk.o_posoff = c->have_pos ? c->c_pos_off.as<long long>() : nullptr;
  Pk.o_rl = c->have_read_len ? c->c_read_len.as<long long>() : nullptr;
  Pk.out_reads = out_reads;
  Pk.out_tokens = out_tokens;
  Pk.dead_kept = dead_kept_d;
  if (R > 0)
    hipLaunchKernelGGL(k_corr_pack, dim3(nblk(R, 4 * PACK_READS)), dim3(256), 0, st, Pk);
  else
    HIPCHK(hipMemcpyAsync(c->c_read_off.as<long long>() + out_reads, &out_tokens, sizeof(long long),
                          hipMemcpyHostToDevice, st));
  {
    // the call's last synchronisation; with it an upper bound for the nodes of the graph these reads will make (same k):
    // every window of a corrected read is a live node of this graph — untouched reads, slices, re-threaded paths —
    // except the dead windows of the reads that fell back to their original genes
    FetchList l;
    for (int i = 0; i < 16; ++i) l.add(dead_kept_d + 16 * i);
    l.add(dead_kept_d + 256);
    unsigned long long v[17];
    AMGCHK(fetch(c, l, v));
    unsigned long long bound = count_alive ? v[16] : (unsigned long long)c->n_nodes;
    unsigned long long dead_kept = 0;
    for (int i = 0; i < 16; ++i) dead_kept += v[i];
    bound += dead_kept;
    // the graph these reads will make is this graph's live part (amg_derive.hip) when no read was re-threaded or kept
    // its genes around a dead window, and no edge died on its own: reads were dropped or cut to their live windows
    c->c_derivable = n_gapped == 0 && dead_kept == 0 && !c->edge_own_deaths;
    c->c_node_bound = (int64_t)bound;
    if (const char* e = getenv("AMG_TEST_NODE_BOUND")) c->c_node_bound = atoll(e);  // test hook: a bound that does not hold
    c->c_node_bound_k = c->k;
  }
  stage_end(c);
  c->c_reads = out_reads;
  c->c_tokens = out_tokens;
  c->c_pos1_used = c->pos1_used + (carry ? pos_total : 0);  // becomes current with amg_adopt_corrected
  c->have_corrected = true;
  if (n_out_reads) *n_out_reads = out_reads;
  if (n_out_tokens) *n_out_tokens = out_tokens;
  return AMG_OK;
}

extern "C" int amg_get_corrected(amg_ctx* c, int32_t* tokens, int64_t* read_offsets, int32_t* orig_read,
                                 uint8_t* changed, int64_t* gene_start, int64_t* gene_end) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (!c->have_corrected) return amg_fail(AMG_E_STATE, "amg_correct_reads first");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  auto get = [&](void* dst, const DevBuf& src, size_t bytes) -> int {
    if (!dst || !bytes) return AMG_OK;
    HIPCHK(hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, st));
    return AMG_OK;
  };
  AMGCHK(get(tokens, c->c_tokens_buf, (size_t)c->c_tokens * sizeof(int32_t)));
  AMGCHK(get(read_offsets, c->c_read_off, (size_t)(c->c_reads + 1) * sizeof(int64_t)));
  AMGCHK(get(orig_read, c->c_orig, (size_t)c->c_reads * sizeof(int32_t)));
  AMGCHK(get(changed, c->c_changed, (size_t)c->c_reads));
  if (c->have_pos && (gene_start || gene_end) && c->c_tokens > 0) {
    AMGCHK(c->c_gstart.ensure((size_t)(c->c_tokens + 64) * sizeof(long long)));
    AMGCHK(c->c_gend.ensure((size_t)(c->c_tokens + 64) * sizeof(long long)));
    CorrArgs a;
    memset(&a, 0, sizeof(a));
    fill_pos_args(c, a);
    hipLaunchKernelGGL(k_gather_positions, dim3(nblk(c->c_reads, 4)), dim3(256), 0, st, a,
                       c->c_read_off.as<long long>(), c->c_pos_off.as<long long>(), (long long)c->c_reads,
                       c->c_gstart.as<long long>(), c->c_gend.as<long long>());
    AMGCHK(get(gene_start, c->c_gstart, (size_t)c->c_tokens * sizeof(int64_t)));
    AMGCHK(get(gene_end, c->c_gend, (size_t)c->c_tokens * sizeof(int64_t)));
  }
  HIPCHK(hipStreamSynchronize(st));
  return AMG_OK;
}

// ---- 32-bit positions at the boundary.  Read coordinates fit 32 bits; the position arrays are four fifths of what a
// cleaning sweep moves over PCIe (16 of 20 bytes per gene).  amg_set_positions32 takes them as int32 (widened on the
// device into the engine's own arrays); amg_get_corrected32 hands back, for every corrected read, WHERE its positions
// are — a slice of the caller's own arrays for a read that was left alone or only trimmed, new values (int32, laid end
// to end) only for the reads whose positions the carry-over produced.
__global__ void k_widen_pos(const int* __restrict__ s32, const int* __restrict__ e32, long long n,
                            long long* __restrict__ s64, long long* __restrict__ e64) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    s64[i] = (long long)s32[i];
    e64[i] = (long long)e32[i];
  }
}

extern "C" int amg_set_positions32(amg_ctx* c, const int32_t* gene_start, const int32_t* gene_end,
                                   const int64_t* read_len, int on_device) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  if (!gene_start || !gene_end) return amg_fail(AMG_E_ARG, "null positions");
  if (on_device != 0 && on_device != 1) return amg_fail(AMG_E_ARG, "amg_set_positions32: on_device is 0 or 1 (the arrays are widened, never borrowed)");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const long long T = c->n_tokens;
  const int* d_s = gene_start;
  const int* d_e = gene_end;
  if (!on_device) {  // staged in the buffers the read-back of a correction uses (free until then)
    AMGCHK(c->c_gstart.ensure((size_t)(T + 64) * sizeof(long long)));
    AMGCHK(c->c_gend.ensure((size_t)(T + 64) * sizeof(long long)));
    if (T > 0) {
      HIPCHK(hipMemcpyAsync(c->c_gstart.p, gene_start, (size_t)T * sizeof(int), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(c->c_gend.p, gene_end, (size_t)T * sizeof(int), hipMemcpyHostToDevice, st));
    }
    d_s = c->c_gstart.as<int>();
    d_e = c->c_gend.as<int>();
  }
  c->gene_start.unborrow();
  c->gene_end.unborrow();
  AMGCHK(c->gene_start.ensure((size_t)(T + 64) * sizeof(long long)));
  AMGCHK(c->gene_end.ensure((size_t)(T + 64) * sizeof(long long)));
  if (T > 0)
    hipLaunchKernelGGL(k_widen_pos, dim3(nblk(T, 1024) < 4096u ? nblk(T, 1024) : 4096u), dim3(256), 0, st, d_s, d_e, T,
                       c->gene_start.as<long long>(), c->gene_end.as<long long>());
  c->have_pos = true;
  c->pos_identity = true;
  c->pos0_own = false;
  c->pos_n0 = T;
  c->pos1_used = c->c_pos1_used = 0;
  c->have_corrected = false;
  c->have_read_len = false;
  if (read_len) {
    c->read_len.unborrow();
    AMGCHK(c->read_len.ensure((size_t)c->n_reads * sizeof(int64_t) + 64));
    if (c->n_reads > 0)
      HIPCHK(hipMemcpyAsync(c->read_len.p, read_len, (size_t)c->n_reads * sizeof(int64_t),
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    c->have_read_len = true;
  }
  HIPCHK(hipStreamSynchronize(st));
  return AMG_OK;
}

// per corrected read: number of positions the carry-over produced for it (0: its positions are a slice of the caller's)
// (own: indices below it are the CALLER's arrays — 0 once amg_adopt_corrected has compacted the pools into arrays of
// the engine's own, after which every read's positions travel)
__global__ void k_new_pos_len(const long long* __restrict__ c_off, const long long* __restrict__ c_posoff, long long own,
                              long long c_reads, long long* __restrict__ len) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < c_reads) len[q] = c_posoff[q] >= own ? c_off[q + 1] - c_off[q] : 0;
}

__global__ __launch_bounds__(256) void k_gather_new_positions32(CorrArgs a, const long long* __restrict__ c_off,
                                                                const long long* __restrict__ c_posoff, long long c_reads,
                                                                const long long* __restrict__ new_off,
                                                                long long* __restrict__ pos_src, int* __restrict__ o_gs,
                                                                int* __restrict__ o_ge, unsigned long long* too_wide,
                                                                long long own) {
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= c_reads) return;
  const int lane = threadIdx.x & 63;
  const long long off = c_posoff[q];
  if (off < own) {
    if (lane == 0) pos_src[q] = off;
    return;
  }
  const long long n = c_off[q + 1] - c_off[q], at = new_off[q];
  if (lane == 0) pos_src[q] = -1 - at;
  const long long *gs, *ge;
  pos_base(a, off, gs, ge);
  bool wide = false;
  for (long long i = lane; i < n; i += 64) {
    const long long s = gs[i], e = ge[i];
    wide = wide || s != (long long)(int)s || e != (long long)(int)e;
    o_gs[at + i] = (int)s;
    o_ge[at + i] = (int)e;
  }
  if (wide) *too_wide = 1ull;
}

extern "C" int amg_get_corrected32(amg_ctx* c, int32_t* tokens, int64_t* read_offsets, int32_t* orig_read,
                                   uint8_t* changed, int64_t* pos_src, int32_t* new_start, int32_t* new_end,
                                   int64_t* n_new) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (!c->have_corrected) return amg_fail(AMG_E_STATE, "amg_correct_reads first");
  if (!pos_src || !n_new) return amg_fail(AMG_E_ARG, "amg_get_corrected32: pos_src and n_new are required");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const long long R = c->c_reads;
  *n_new = 0;
  auto get = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!dst || !bytes) return AMG_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    return AMG_OK;
  };
  AMGCHK(get(tokens, c->c_tokens_buf.p, (size_t)c->c_tokens * sizeof(int32_t)));
  AMGCHK(get(read_offsets, c->c_read_off.p, (size_t)(R + 1) * sizeof(int64_t)));
  AMGCHK(get(orig_read, c->c_orig.p, (size_t)R * sizeof(int32_t)));
  AMGCHK(get(changed, c->c_changed.p, (size_t)R));
  if (!c->have_pos || R == 0) {
    HIPCHK(hipStreamSynchronize(st));
    return c->have_pos ? AMG_OK : amg_fail(AMG_E_STATE, "amg_get_corrected32: no positions were set");
  }
  // lengths -> offsets of the new positions (s1: lengths + their prefix, s2: pos_src), then one gather
  AMGCHK(c->s1.ensure((size_t)(2 * R + 4) * sizeof(long long)));
  AMGCHK(c->s2.ensure((size_t)(R + 2) * sizeof(long long)));
  AMGCHK(c->c_gstart.ensure((size_t)(c->c_tokens + 64) * sizeof(long long)));
  AMGCHK(c->c_gend.ensure((size_t)(c->c_tokens + 64) * sizeof(long long)));
  long long* len = c->s1.as<long long>();
  long long* off = len + (R + 2);
  CorrArgs a;
  memset(&a, 0, sizeof(a));
  fill_pos_args(c, a);
  unsigned long long* flag = c->status.as<unsigned long long>() + ST_MISC;
  HIPCHK(hipMemsetAsync(flag, 0, sizeof(unsigned long long), st));
  HIPCHK(hipMemsetAsync(len + R, 0, sizeof(long long), st));
  const long long own = c->pos0_own ? 0 : a.n0;
  hipLaunchKernelGGL(k_new_pos_len, dim3(nblk(R, 256)), dim3(256), 0, st, c->c_read_off.as<long long>(),
                     c->c_pos_off.as<long long>(), own, R, len);
  AMGCHK(prim_exscan_i64(c, len, off, (size_t)R + 1));
  hipLaunchKernelGGL(k_gather_new_positions32, dim3(nblk(R, 4)), dim3(256), 0, st, a, c->c_read_off.as<long long>(),
                     c->c_pos_off.as<long long>(), R, off, c->s2.as<long long>(), c->c_gstart.as<int>(),
                     c->c_gend.as<int>(), flag, own);
  unsigned long long h[2] = {0, 0};
  {
    FetchList l;
    l.add(off + R);
    l.add(flag);
    AMGCHK(fetch(c, l, h));
  }
  if (h[1]) return amg_fail(AMG_E_ARG, "amg_get_corrected32: a position does not fit 32 bits (amg_get_corrected returns 64-bit positions)");
  *n_new = (int64_t)h[0];
  AMGCHK(get(pos_src, c->s2.p, (size_t)R * sizeof(int64_t)));
  AMGCHK(get(new_start, c->c_gstart.p, (size_t)h[0] * sizeof(int32_t)));
  AMGCHK(get(new_end, c->c_gend.p, (size_t)h[0] * sizeof(int32_t)));
  HIPCHK(hipStreamSynchronize(st));
  return AMG_OK;
}

// every corrected read's positions laid end to end as 32-bit values (gathered on the device: half the bytes of
// amg_get_corrected's position arrays over PCIe, and those arrays are four fifths of what a correction hands back)
__global__ __launch_bounds__(256) void k_gather_positions32(CorrArgs a, const long long* __restrict__ c_off,
                                                            const long long* __restrict__ c_posoff, long long c_reads,
                                                            int* __restrict__ o_gs, int* __restrict__ o_ge,
                                                            unsigned long long* too_wide) {
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= c_reads) return;
  const int lane = threadIdx.x & 63;
  const long long a0 = c_off[q], n = c_off[q + 1] - a0;
  const long long *gs, *ge;
  pos_base(a, c_posoff[q], gs, ge);
  bool wide = false;
  for (long long i = lane; i < n; i += 64) {
    const long long s = gs[i], e = ge[i];
    wide = wide || s != (long long)(int)s || e != (long long)(int)e;
    o_gs[a0 + i] = (int)s;
    o_ge[a0 + i] = (int)e;
  }
  if (wide) *too_wide = 1ull;
}

extern "C" int amg_get_corrected_positions32(amg_ctx* c, int32_t* gene_start, int32_t* gene_end) {
  if (!c || !gene_start || !gene_end) return amg_fail(AMG_E_ARG, "null argument");
  if (!c->have_corrected) return amg_fail(AMG_E_STATE, "amg_correct_reads first");
  if (!c->have_pos) return amg_fail(AMG_E_STATE, "no gene positions were set");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const long long R = c->c_reads, T = c->c_tokens;
  if (T == 0) return AMG_OK;
  AMGCHK(c->c_gstart.ensure((size_t)(T + 64) * sizeof(long long)));
  AMGCHK(c->c_gend.ensure((size_t)(T + 64) * sizeof(long long)));
  unsigned long long* flag = c->status.as<unsigned long long>() + ST_MISC;
  HIPCHK(hipMemsetAsync(flag, 0, sizeof(unsigned long long), st));
  CorrArgs a;
  memset(&a, 0, sizeof(a));
  fill_pos_args(c, a);
  hipLaunchKernelGGL(k_gather_positions32, dim3(nblk(R, 4)), dim3(256), 0, st, a, c->c_read_off.as<long long>(),
                     c->c_pos_off.as<long long>(), R, c->c_gstart.as<int>(), c->c_gend.as<int>(), flag);
  unsigned long long wide = 0;
  {
    FetchList l;
    l.add(flag);
    AMGCHK(fetch(c, l, &wide));
  }
  if (wide) return amg_fail(AMG_E_ARG, "amg_get_corrected_positions32: a position does not fit 32 bits (amg_get_corrected returns 64-bit positions)");
  HIPCHK(hipMemcpyAsync(gene_start, c->c_gstart.p, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(gene_end, c->c_gend.p, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return AMG_OK;
}

extern "C" int amg_adopt_corrected(amg_ctx* c) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (!c->have_corrected) return amg_fail(AMG_E_STATE, "amg_correct_reads first");
  // borrowed genes / offsets / lengths go back to their owner: the corrected set lives in our own
  // allocations.  The position arrays stay where they are (borrowed or not): corrected reads point
  // into them and into the pool of produced positions.
  // The pool of produced positions only grows while corrections follow each other without a new
  // amg_set_positions.  Once it holds more than twice the live genes the corrected set's positions are gathered
  // into flat arrays of our own, which become the new "caller's arrays" (borrowed ones go back to their owner
  // here, earlier than the contract promises), and the pool starts empty again.
  bool compacted = false;
  if (c->have_pos && c->c_tokens > 0) {
    long long slack = 1ll << 20;
    if (const char* e = getenv("AMG_POS_COMPACT_MIN")) slack = atoll(e);  // test hook
    if (c->c_pos1_used > 2 * c->c_tokens + slack) {
      HIPCHK(hipSetDevice(c->device));
      AMGCHK(c->c_gstart.ensure((size_t)(c->c_tokens + 64) * sizeof(long long)));
      AMGCHK(c->c_gend.ensure((size_t)(c->c_tokens + 64) * sizeof(long long)));
      CorrArgs a;
      memset(&a, 0, sizeof(a));
      fill_pos_args(c, a);
      hipLaunchKernelGGL(k_gather_positions, dim3(nblk(c->c_reads, 4)), dim3(256), 0, c->stream, a,
                         c->c_read_off.as<long long>(), c->c_pos_off.as<long long>(), (long long)c->c_reads,
                         c->c_gstart.as<long long>(), c->c_gend.as<long long>());
      HIPCHK(hipStreamSynchronize(c->stream));  // the borrowed arrays are read for the last time
      c->gene_start.unborrow();
      c->gene_end.unborrow();
      std::swap(c->gene_start, c->c_gstart);
      std::swap(c->gene_end, c->c_gend);
      compacted = true;
      c->pos0_own = true;  // pool 0 is no longer what the caller handed over (amg_get_corrected32)
    }
  }
  for (DevBuf* b : {&c->tokens, &c->read_off, &c->read_len}) b->unborrow();
  std::swap(c->tokens, c->c_tokens_buf);
  std::swap(c->read_off, c->c_read_off);
  std::swap(c->rd_src, c->c_src);
  c->derive_ready = c->c_derivable;  // (the graph the reads were corrected against is still in place: amg_build may reuse it)
  c->dist_candidate = c->dist_mode;  // (a rank of a merged build: the ranks decide together, amg_dist.hip S_DV_*)
  c->c_derivable = false;
  if (c->have_pos && compacted) {
    c->pos_identity = true;
    c->pos_n0 = c->c_tokens;
    c->pos1_used = c->c_pos1_used = 0;
  } else if (c->have_pos) {
    std::swap(c->pos_off, c->c_pos_off);
    c->pos_identity = false;
    c->pos1_used = c->c_pos1_used;
  }
  if (c->have_read_len) std::swap(c->read_len, c->c_read_len);
  c->n_reads = c->c_reads;
  c->n_tokens = c->c_tokens;
  c->have_corrected = false;
  c->built = false;
  c->match_valid = false;
  // the next build's node table: not larger than the correction's bound asks for (a graph of uncorrected reads is
  // mostly error nodes that do not come back: 5.4 M nodes before, 0.5 M after on BASELINE config 3).  A build with
  // another k, or a bound that does not hold, costs what any undersized table costs: one repeated pass.
  if (c->c_node_bound > 0 && c->c_node_bound_k == c->k && (c->node_hint == 0 || c->c_node_bound < c->node_hint))
    c->node_hint = c->c_node_bound > 256 ? c->c_node_bound : 256;
  c->c_node_bound = 0;
  return AMG_OK;
}

// The corrected set of `src` becomes the read set of `dst`, device to device: what the reference does between
// correct_reads and the next GeneMerGraph(...) (graph_utils.py:147-150, :165) without the reads leaving the GPU.
// Positions are gathered into flat arrays of dst's own (pool 0, identity offsets); src keeps its corrected set.
extern "C" int amg_set_reads_from_corrected(amg_ctx* dst, amg_ctx* src) {
  if (!dst || !src) return amg_fail(AMG_E_ARG, "null ctx");
  if (!src->have_corrected) return amg_fail(AMG_E_STATE, "amg_correct_reads on the source ctx first");
  if (dst == src) return amg_adopt_corrected(src);
  if (dst->device != src->device) return amg_fail(AMG_E_ARG, "amg_set_reads_from_corrected: one device");
  HIPCHK(hipSetDevice(dst->device));
  HIPCHK(hipStreamSynchronize(src->stream));
  hipStream_t st = dst->stream;
  const long long R = src->c_reads, T = src->c_tokens;
  AMGCHK(dst->tokens.ensure((size_t)T * sizeof(int32_t) + 64));
  AMGCHK(dst->read_off.ensure((size_t)(R + 1) * sizeof(int64_t) + 64));
  if (T > 0) HIPCHK(hipMemcpyAsync(dst->tokens.p, src->c_tokens_buf.p, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(dst->read_off.p, src->c_read_off.p, (size_t)(R + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
  dst->n_reads = R;
  dst->n_tokens = T;
  dst->two_v = src->two_v;
  dst->have_pos = dst->have_read_len = false;
  if (src->have_pos) {
    AMGCHK(dst->gene_start.ensure((size_t)(T + 64) * sizeof(long long)));
    AMGCHK(dst->gene_end.ensure((size_t)(T + 64) * sizeof(long long)));
    if (R > 0 && T > 0) {
      CorrArgs a;
      memset(&a, 0, sizeof(a));
      fill_pos_args(src, a);
      hipLaunchKernelGGL(k_gather_positions, dim3(nblk(R, 4)), dim3(256), 0, st, a, src->c_read_off.as<long long>(),
                         src->c_pos_off.as<long long>(), R, dst->gene_start.as<long long>(),
                         dst->gene_end.as<long long>());
    }
    dst->have_pos = true;
    dst->pos0_own = true;  // gathered here: not arrays any caller holds
    dst->pos_identity = true;
    dst->pos_n0 = T;
    dst->pos1_used = dst->c_pos1_used = 0;
  }
  if (src->have_read_len) {
    AMGCHK(dst->read_len.ensure((size_t)(R + 1) * sizeof(long long) + 64));
    if (R > 0)
      HIPCHK(hipMemcpyAsync(dst->read_len.p, src->c_read_len.p, (size_t)R * sizeof(long long), hipMemcpyDeviceToDevice, st));
    dst->have_read_len = true;
  }
  HIPCHK(hipStreamSynchronize(st));
  dst->built = false;
  dst->derive_ready = false;
  dst->have_corrected = false;
  dst->match_valid = false;
  // (amg_adopt_corrected: the same bound — for a build at the gene-mer size it was made for; dst has no graph of its own)
  dst->node_hint = 0;
  dst->hint_bound = src->c_node_bound;
  dst->hint_bound_k = src->c_node_bound_k;
  dst->cnt_hint_reset = true;
  return AMG_OK;
}
