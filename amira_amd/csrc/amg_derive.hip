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

// per-window node ids and directions of the new reads: read r is the old tokens src[r] .. of its length
__global__ __launch_bounds__(256) void k_dv_windows(const long long* __restrict__ read_off, const long long* __restrict__ src,
                                                    long long n_reads, int k, const int* __restrict__ old_node,
                                                    const signed char* __restrict__ old_dir,
                                                    const long long* __restrict__ new_id, int* __restrict__ tok_node,
                                                    signed char* __restrict__ tok_dir, unsigned long long* bad) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_reads) return;
  const int lane = threadIdx.x & 63;
  const long long o = read_off[r], n = read_off[r + 1] - o, s = src[r];
  for (long long j = lane; j < n; j += 64) {
    int v = -1;
    signed char d = 0;
    if (j + k <= n) {
      const int old = old_node[s + j];
      if (old >= 0) {
        v = (int)new_id[old];
        d = old_dir[s + j];
      } else {
        *bad = 1ull;  // a kept window that is not a live node: not the case this shortcut is for
      }
    }
    tok_node[o + j] = v;
    tok_dir[o + j] = d;
  }
}

__global__ void k_dv_nodes(const unsigned char* __restrict__ alive, const long long* __restrict__ new_id, long long n_old, int k,
                           const int* __restrict__ tok_in, const unsigned int* __restrict__ cov_in,
                           const long long* __restrict__ first_in, const long long* __restrict__ src,
                           const long long* __restrict__ read_off, long long n_reads, int* __restrict__ tok_out,
                           unsigned int* __restrict__ cov_out, long long* __restrict__ first_out,
                           unsigned char* __restrict__ alive_out, unsigned long long* bad) {
  long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_old || !alive[n]) return;
  const long long i = new_id[n];
  for (int x = 0; x < k; ++x) tok_out[i * k + x] = tok_in[n * k + x];
  cov_out[i] = cov_in[n];
  alive_out[i] = 1;
  const long long f = first_in[n];
  bool ok = true;
  const long long t = dv_new_token(f >> 1, src, read_off, n_reads, k, &ok);
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
  bool ok = true;
  // first-seen of a class = the token of the adjacency's FIRST window: a kept window next to a kept window
  const long long t = dv_new_token((long long)(f >> 3), src, read_off, n_reads, k, &ok);
  if (!ok) *bad = 3ull;
  ofirst[q] = ((unsigned long long)t << 3) | (f & 7ull);
  ocnt[q] = pcnt[p];
}

// AMG_OK with *done = true: the graph of the current reads is in place (c->built); *done = false: the ordinary build
// is to run (nothing it reads has been touched)
int derive_from_previous(amg_ctx* c, int k, bool* done) {
  *done = false;
  hipStream_t st = c->stream;
  const long long D = c->n_nodes, P = c->n_pairs, R = c->n_reads, T = c->n_tokens;
  if (D <= 0 || R <= 0 || T <= 0) return AMG_OK;
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
  hipLaunchKernelGGL(k_dv_windows, dim3(nblk(R, 4)), dim3(256), 0, st, c->read_off.as<long long>(), c->rd_src.as<long long>(),
                     R, k, c->tok_node.as<int>(), c->tok_dir.as<signed char>(), new_id, c->alt_tok_node.as<int>(),
                     c->alt_tok_dir.as<signed char>(), bad);
  unsigned long long v[2] = {0, 0};
  {
    FetchList l;
    l.add(new_id + D);
    l.add(ppos + P);
    AMGCHK(fetch(c, l, v));
  }
  const long long D2 = (long long)v[0], P2 = (long long)v[1];
  // ---- node and edge-class arrays squeezed, first-seen values in the new reads' token coordinates
  AMGCHK(c->alt_ntok.ensure((size_t)(D2 * k + 1) * sizeof(int)));
  AMGCHK(c->alt_ncov.ensure((size_t)(D2 + 1) * sizeof(unsigned int)));
  AMGCHK(c->alt_nfirst.ensure((size_t)(D2 + 1) * sizeof(long long)));
  AMGCHK(c->alt_nalive.ensure((size_t)(D2 + 1)));
  AMGCHK(c->alt_pkey.ensure((size_t)(P2 + 2) * sizeof(unsigned long long)));
  AMGCHK(c->alt_pfirst.ensure((size_t)(P2 + 2) * sizeof(unsigned long long)));
  AMGCHK(c->alt_pcnt.ensure((size_t)(P2 + 2) * sizeof(unsigned int)));
  hipLaunchKernelGGL(k_dv_nodes, dim3(nblk(D, 256)), dim3(256), 0, st, c->node_alive.as<unsigned char>(), new_id, D, k,
                     c->node_tokens.as<int>(), c->node_cov.as<unsigned int>(), c->node_first.as<long long>(),
                     c->rd_src.as<long long>(), c->read_off.as<long long>(), R, c->alt_ntok.as<int>(),
                     c->alt_ncov.as<unsigned int>(), c->alt_nfirst.as<long long>(), c->alt_nalive.as<unsigned char>(), bad);
  if (P > 0)
    hipLaunchKernelGGL(k_dv_pairs, dim3(nblk(P, 256)), dim3(256), 0, st, c->pair_key.as<unsigned long long>(),
                       c->pair_first.as<unsigned long long>(), c->pair_cnt.as<unsigned int>(), P, keep, ppos, new_id,
                       c->rd_src.as<long long>(), c->read_off.as<long long>(), R, k, c->alt_pkey.as<unsigned long long>(),
                       c->alt_pfirst.as<unsigned long long>(), c->alt_pcnt.as<unsigned int>(), bad);
  stage_end(c);
  // window / short-read counts of the new reads and their read-end bitmap (whatever builds next from these reads
  // finds what a build leaves); the verdict of the checks rides on its status read-back
  AMGCHK(bs_read_stats(c, k));
  unsigned long long hs[ST_WORDS];
  AMGCHK(fetch_status(c, hs));
  if (hs[ST_MISC] || hs[ST_BADINPUT]) return AMG_OK;  // not the case this is for (or malformed reads: the build says so)
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
  c->n_windows = (int64_t)hs[ST_N_WINDOWS];
  c->n_short = (int64_t)hs[ST_N_SHORT];
  c->n_nodes = c->n_local_nodes = D2;
  c->n_pairs = c->n_local_pairs = P2;
  c->comp_from_claims = false;  // (labels of the graph as it is now, made when somebody asks)
  c->filtered_build = false;
  AMGCHK(bs_finish_from_pairs(c));
  c->built = true;
  c->derived = true;
  c->node_hint = D2 > 256 ? D2 : 256;
  *done = true;
  return AMG_OK;
}
