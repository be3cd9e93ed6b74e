// amg_derive.hip — the rebuild that reuses the previous build (graph_utils.py:145-166: every cleaning iteration builds
// three graphs, each from the reads the previous one corrected).
//
// When a correction re-threads no read — every corrected read is a read of the graph's read set as it was, or a slice
// [first live window .. last live window] of it, and reads whose windows all died are gone; what a correction after tip
// clipping looks like — the graph GeneMerGraph.__init__ (construct_graph.py:31-102) builds from the corrected reads IS the
// graph at hand restricted to its live nodes:
//   * every window of the new reads is a live node's window, and every window of a live node is still there (only dead
//     windows were cut off), so nodes, coverages and per-read node lists are the old ones;
//   * reads keep their order and their windows' order, so first-seen order — node ids, edge ids, list orders — is the
//     old order among the survivors: new id = number of live nodes before;
//   * an adjacency of two kept windows is kept, an edge class with a dead end has lost every adjacency: the classes are
//     the old classes with both ends alive, with their counts.
// No table pass, no counting, no ranking: a scan over the nodes' alive bytes, the per-window node ids gathered through
// it into the new reads' layout, node and edge-class arrays squeezed, first-seen token indices moved to the new reads'
// coordinates, then the ordinary edge emission.  amg_correct_reads says whether its output qualifies, amg_adopt_corrected
// arms the shortcut, the next amg_build at the same k takes it — and checks what it assumes on the way (a kept window
// that is not a live node, a first occurrence outside the kept slices): anything odd and the ordinary build runs instead.
// AMG_NO_DERIVE=1: A/B + test switch.
#include "amg_device.h"

static inline unsigned int nblk(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

// new token index of the old token t (the first token of a kept window): the new read whose slice holds it
__device__ __forceinline__ long long dv_new_token(long long t, const long long* __restrict__ src,
                                                  const long long* __restrict__ read_off, long long n_reads, int k,
                                                  bool* ok) {
  long long lo = 0, hi = n_reads;  // last read with src <= t
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (src[mid] <= t) lo = mid + 1; else hi = mid;
  }
  const long long i = lo - 1;
  if (i < 0) {
    *ok = false;
    return 0;
  }
  const long long q = t - src[i], len = read_off[i + 1] - read_off[i];
  if (q + k > len) *ok = false;  // not a window of that slice
  return read_off[i] + q;
}

// per-window node ids and directions of the new reads: read r is the old tokens src[r] .. of its length.
// One wave takes 64 consecutive reads: lane l owns read l's record (offset, length, source: loaded coalesced), then the
// 64 lanes walk the reads' windows four reads at a time (records broadcast with v_readlane) — a wave per read was a
// handful of one-row loads each and bound by their latency (0.36 ms for 0.6 GB)
#define DV_READS 64
__device__ __forceinline__ long long dv_bcast(long long v, int lane) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)v, lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__global__ __launch_bounds__(256) void k_dv_windows(const long long* __restrict__ read_off, const long long* __restrict__ src,
                                                    long long n_reads, int k, const int* __restrict__ old_node,
                                                    const signed char* __restrict__ old_dir,
                                                    const long long* __restrict__ new_id, int* __restrict__ tok_node,
                                                    signed char* __restrict__ tok_dir, unsigned long long* bad) {
  const int lane = threadIdx.x & 63;
  const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * DV_READS;
  if (r0 >= n_reads) return;
  long long o = 0, s = 0;
  int n = 0;
  if (r0 + lane < n_reads) {
    o = read_off[r0 + lane];
    n = (int)(read_off[r0 + lane + 1] - o);
    s = src[r0 + lane];
  }
  for (int j0 = 0; j0 < DV_READS; j0 += 4) {
    long long oo[4], ss[4];
    int nn[4], most = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nn[j] = __builtin_amdgcn_readlane(n, j0 + j);
      oo[j] = dv_bcast(o, j0 + j);
      ss[j] = dv_bcast(s, j0 + j);
      most = nn[j] > most ? nn[j] : most;
    }
    for (int q = lane; q < most; q += 64) {
      int old[4];
      signed char d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        old[j] = -1;
        d[j] = 0;
        if (q + k <= nn[j]) {
          old[j] = old_node[ss[j] + q];
          d[j] = old_dir[ss[j] + q];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (q >= nn[j]) continue;
        int v = -1;
        if (q + k <= nn[j]) {
          if (old[j] >= 0) v = (int)new_id[old[j]];
          else *bad = 1ull;  // a kept window that is not a live node: not the case this shortcut is for
        }
        tok_node[oo[j] + q] = v;
        tok_dir[oo[j] + q] = v >= 0 ? d[j] : (signed char)0;
      }
    }
  }
}

// [own_lo, own_hi): the tokens (in the coordinates of first-seen values) of the read set this ctx holds — everything on
// one GPU; on a rank of a merged build its shard, and first-seen values outside it are left to their rank (0 here)
__global__ void k_dv_nodes(const unsigned char* __restrict__ alive, const long long* __restrict__ new_id, long long n_old, int k,
                           const int* __restrict__ tok_in, const unsigned int* __restrict__ cov_in,
                           const long long* __restrict__ first_in, const long long* __restrict__ src,
                           const long long* __restrict__ read_off, long long n_reads, long long own_lo, long long own_hi,
                           int* __restrict__ tok_out, unsigned int* __restrict__ cov_out, long long* __restrict__ first_out,
                           unsigned char* __restrict__ alive_out, unsigned long long* bad) {
  long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_old || !alive[n]) return;
  const long long i = new_id[n];
  for (int x = 0; x < k; ++x) tok_out[i * k + x] = tok_in[n * k + x];
  cov_out[i] = cov_in[n];
  alive_out[i] = 1;
  const long long f = first_in[n], tg = f >> 1;
  if (tg < own_lo || tg >= own_hi) {
    first_out[i] = 0;
    return;
  }
  bool ok = true;
  const long long t = dv_new_token(tg - own_lo, src, read_off, n_reads, k, &ok);
  if (!ok) *bad = 2ull;
  first_out[i] = (t << 1) | (f & 1ll);
}

// an edge class stays when both its nodes do
__global__ void k_dv_pair_keep(const unsigned long long* __restrict__ pkey, long long n_pairs,
                               const unsigned char* __restrict__ alive, unsigned int* __restrict__ keep) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const unsigned long long key = pkey[p];
  const unsigned int lo = (unsigned int)((key >> 32) & 0x7fffffffull), hi = (unsigned int)(key & 0xffffffffull) - 1u;
  keep[p] = (alive[lo] && alive[hi]) ? 1u : 0u;
}

__global__ void k_dv_pairs(const unsigned long long* __restrict__ pkey, const unsigned long long* __restrict__ pfirst,
                           const unsigned int* __restrict__ pcnt, long long n_pairs, const unsigned int* __restrict__ keep,
                           const long long* __restrict__ pos, const long long* __restrict__ new_id,
                           const long long* __restrict__ src, const long long* __restrict__ read_off, long long n_reads, int k,
                           long long own_lo, long long own_hi,
                           unsigned long long* __restrict__ okey, unsigned long long* __restrict__ ofirst,
                           unsigned int* __restrict__ ocnt, unsigned long long* bad) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs || !keep[p]) return;
  const unsigned long long key = pkey[p], f = pfirst[p];
  const unsigned int lo = (unsigned int)((key >> 32) & 0x7fffffffull), hi = (unsigned int)(key & 0xffffffffull) - 1u;
  // (new ids ascend with the old ones: the smaller node stays the smaller one, the orientation bits stay true)
  const unsigned long long nlo = (unsigned long long)new_id[lo], nhi = (unsigned long long)new_id[hi];
  const long long q = pos[p];
  okey[q] = (key & (1ull << 63)) | (nlo << 32) | (nhi + 1ull);
  ocnt[q] = pcnt[p];
  const long long tg = (long long)(f >> 3);
  if (tg < own_lo || tg >= own_hi) {
    ofirst[q] = 0ull;
    return;
  }
  bool ok = true;
  // first-seen of a class = the token of the adjacency's FIRST window: a kept window next to a kept window
  const long long t = dv_new_token(tg - own_lo, src, read_off, n_reads, k, &ok);
  if (!ok) *bad = 3ull;
  ofirst[q] = ((unsigned long long)t << 3) | (f & 7ull);
}

// merged builds: where, among the squeezed node / class arrays (both in first-seen order), the entries first seen on
// every rank's shard begin — bounds[r] for nodes, bounds[world + 1 + r] for classes; base[r] = rank r's first token
__global__ void k_dv_bounds(const long long* __restrict__ base, int world, const long long* __restrict__ node_first,
                            long long n_nodes, const long long* __restrict__ new_id, const unsigned long long* __restrict__ pfirst,
                            long long n_pairs, const long long* __restrict__ ppos, long long* __restrict__ bounds) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > world) return;
  {
    long long lo = 0, hi = n_nodes;
    const long long want = r == world ? 0x7fffffffffffffffll : (base[r] << 1);
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (node_first[mid] < want) lo = mid + 1; else hi = mid;
    }
    bounds[r] = new_id[lo];
  }
  {
    long long lo = 0, hi = n_pairs;
    const unsigned long long want = r == world ? ~0ull : ((unsigned long long)base[r] << 3);
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (pfirst[mid] < want) lo = mid + 1; else hi = mid;
    }
    bounds[world + 1 + r] = ppos[lo];
  }
}

// The graph at hand restricted to its live nodes, in the layout of the current reads, into the alt_* buffers (nothing
// the ordinary build reads is touched).  own_lo / own_tokens: this ctx's shard in first-seen coordinates (0 and all
// tokens on one GPU); d_bases != null (merged builds): the ranks' first tokens [world + 1] on the device, and
// bounds_host receives k_dv_bounds' 2 * (world + 1) values.  *ok = false: not the case this shortcut is for.
int derive_local(amg_ctx* c, int k, long long own_lo, long long own_tokens, const long long* d_bases, int world,
                 long long* bounds_host, long long* D2_out, long long* P2_out, bool* ok) {
  *ok = false;
  hipStream_t st = c->stream;
  const long long D = c->n_nodes, P = c->n_pairs, R = c->n_reads, T = c->n_tokens;
  stage_begin(c, "derive");
  unsigned long long* bad = c->status.as<unsigned long long>() + ST_MISC;
  // ---- new ids of the live nodes, flags of the classes that stay
  AMGCHK(c->s2.ensure((size_t)(D + 2) * sizeof(long long)));
  AMGCHK(c->s3.ensure((size_t)(P + 2) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(P + 2) * sizeof(long long)));
  long long* new_id = c->s2.as<long long>();
  unsigned int* keep = c->s3.as<unsigned int>();
  long long* ppos = c->s4.as<long long>();
  {
    ClearList cl;
    cl.add(c->status.p, ST_WORDS * sizeof(unsigned long long));
    cl.add(keep + P, sizeof(unsigned int));
    AMGCHK(clear_many(c, cl));
  }
  AMGCHK(prim_exscan_bytes_set(c, c->node_alive.as<unsigned char>(), new_id, (size_t)D));
  if (P > 0)
    hipLaunchKernelGGL(k_dv_pair_keep, dim3(nblk(P, 256)), dim3(256), 0, st, c->pair_key.as<unsigned long long>(), P,
                       c->node_alive.as<unsigned char>(), keep);
  AMGCHK(prim_exscan_u32_to_i64(c, keep, ppos, (size_t)P + 1));
  // ---- per-window ids of the new reads (the largest piece: started before the host needs the counts)
  AMGCHK(c->alt_tok_node.ensure((size_t)(T + 8) * sizeof(int)));
  AMGCHK(c->alt_tok_dir.ensure((size_t)(T + 8)));
  if (R > 0)
    hipLaunchKernelGGL(k_dv_windows, dim3(nblk(R, 4 * DV_READS)), dim3(256), 0, st, c->read_off.as<long long>(),
                       c->rd_src.as<long long>(), R, k, c->tok_node.as<int>(), c->tok_dir.as<signed char>(), new_id,
                       c->alt_tok_node.as<int>(), c->alt_tok_dir.as<signed char>(), bad);
  long long* d_bounds = nullptr;
  if (d_bases) {
    AMGCHK(c->s1.ensure((size_t)(2 * (world + 1)) * sizeof(long long)));
    d_bounds = c->s1.as<long long>();
    hipLaunchKernelGGL(k_dv_bounds, dim3(nblk(world + 1, 64)), dim3(64), 0, st, d_bases, world, c->node_first.as<long long>(), D,
                       new_id, c->pair_first.as<unsigned long long>(), P, ppos, d_bounds);
  }
  unsigned long long v[2 + FETCH_MAX - 2];
  {
    FetchList l;
    l.add(new_id + D);
    l.add(ppos + P);
    if (d_bounds) l.add_words(d_bounds, 2 * (world + 1));
    if (l.overflow) return amg_fail(AMG_E_ARG, "derive: world too large");
    AMGCHK(fetch(c, l, v));
  }
  const long long D2 = (long long)v[0], P2 = (long long)v[1];
  if (bounds_host)
    for (int i = 0; i < 2 * (world + 1); ++i) bounds_host[i] = (long long)v[2 + i];
  // ---- node and edge-class arrays squeezed, first-seen values in the new reads' token coordinates
  AMGCHK(c->alt_ntok.ensure((size_t)(D2 * k + 1) * sizeof(int)));
  AMGCHK(c->alt_ncov.ensure((size_t)(D2 + 1) * sizeof(unsigned int)));
  AMGCHK(c->alt_nfirst.ensure((size_t)(D2 + 1) * sizeof(long long)));
  AMGCHK(c->alt_nalive.ensure((size_t)(D2 + 1)));
  AMGCHK(c->alt_pkey.ensure((size_t)(P2 + 2) * sizeof(unsigned long long)));
  AMGCHK(c->alt_pfirst.ensure((size_t)(P2 + 2) * sizeof(unsigned long long)));
  AMGCHK(c->alt_pcnt.ensure((size_t)(P2 + 2) * sizeof(unsigned int)));
  if (D > 0)
    hipLaunchKernelGGL(k_dv_nodes, dim3(nblk(D, 256)), dim3(256), 0, st, c->node_alive.as<unsigned char>(), new_id, D, k,
                       c->node_tokens.as<int>(), c->node_cov.as<unsigned int>(), c->node_first.as<long long>(),
                       c->rd_src.as<long long>(), c->read_off.as<long long>(), R, own_lo, own_lo + own_tokens,
                       c->alt_ntok.as<int>(), c->alt_ncov.as<unsigned int>(), c->alt_nfirst.as<long long>(),
                       c->alt_nalive.as<unsigned char>(), bad);
  if (P > 0)
    hipLaunchKernelGGL(k_dv_pairs, dim3(nblk(P, 256)), dim3(256), 0, st, c->pair_key.as<unsigned long long>(),
                       c->pair_first.as<unsigned long long>(), c->pair_cnt.as<unsigned int>(), P, keep, ppos, new_id,
                       c->rd_src.as<long long>(), c->read_off.as<long long>(), R, k, own_lo, own_lo + own_tokens,
                       c->alt_pkey.as<unsigned long long>(), c->alt_pfirst.as<unsigned long long>(),
                       c->alt_pcnt.as<unsigned int>(), bad);
  stage_end(c);
  // window / short-read counts of the new reads and their read-end bitmap (whatever builds next from these reads
  // finds what a build leaves); the verdict of the checks rides on its status read-back
  AMGCHK(bs_read_stats(c, k));
  unsigned long long hs[ST_WORDS];
  AMGCHK(fetch_status(c, hs));
  if (hs[ST_MISC] || hs[ST_BADINPUT]) return AMG_OK;  // not the case this is for (or malformed reads: the build says so)
  c->n_windows = (int64_t)hs[ST_N_WINDOWS];
  c->n_short = (int64_t)hs[ST_N_SHORT];
  *D2_out = D2;
  *P2_out = P2;
  *ok = true;
  return AMG_OK;
}

// the alt_* buffers become the graph (derive_local said ok), edges are emitted
int derive_commit(amg_ctx* c, long long D2, long long P2) {
  std::swap(c->tok_node, c->alt_tok_node);
  std::swap(c->tok_dir, c->alt_tok_dir);
  std::swap(c->node_tokens, c->alt_ntok);
  std::swap(c->node_cov, c->alt_ncov);
  std::swap(c->node_first, c->alt_nfirst);
  std::swap(c->node_alive, c->alt_nalive);
  std::swap(c->pair_key, c->alt_pkey);
  std::swap(c->pair_first, c->alt_pfirst);
  std::swap(c->pair_cnt, c->alt_pcnt);
  AMGCHK(c->node_comp.ensure((size_t)(D2 + 1) * sizeof(int)));
  c->n_nodes = c->n_local_nodes = D2;
  c->n_pairs = c->n_local_pairs = P2;
  c->comp_from_claims = false;  // (labels of the graph as it is now, made when somebody asks)
  c->filtered_build = false;
  AMGCHK(bs_finish_from_pairs(c));
  c->built = true;
  c->derived = true;
  c->node_hint = D2 > 256 ? D2 : 256;
  return AMG_OK;
}

// AMG_OK with *done = true: the graph of the current reads is in place (c->built); *done = false: the ordinary build
// is to run (nothing it reads has been touched)
int derive_from_previous(amg_ctx* c, int k, bool* done) {
  *done = false;
  if (c->n_nodes <= 0 || c->n_reads <= 0 || c->n_tokens <= 0) return AMG_OK;
  long long D2 = 0, P2 = 0;
  bool ok = false;
  AMGCHK(derive_local(c, k, 0, 0x3fffffffffffffffll, nullptr, 1, nullptr, &D2, &P2, &ok));
  if (!ok) return AMG_OK;
  AMGCHK(derive_commit(c, D2, P2));
  *done = true;
  return AMG_OK;
}
