// amg_dist.hip — read-sharded build with a key-owner table merge.
//
// Every rank holds a contiguous shard of the reads.  The single-graph result
// (graph_utils.py:105-124 at cores = 1, i.e. GeneMerGraph over all reads) is obtained in
// phases; the exchanges between them are done by the CALLER with RCCL (torch.distributed
// all_to_all_single / all_gather_into_tensor on the device buffers passed here), so the same
// phases also run in a single process with a loop-back exchange (tests on one GPU):
//
//   amg_dist_nodes_local   local windows -> local node table; records bucketed by owner
//   amg_dist_nodes_pack    records in destination order                     --> all-to-all
//   amg_dist_nodes_reduce  owner side: equal keys reduced (sum count, min first-seen)
//   amg_dist_nodes_owned   owned records                                    --> all-gather
//   amg_dist_nodes_global  all records: global ids = rank of first-seen; local slots -> ids
//   amg_dist_edges_local / _pack / _reduce / _owned / _global   same for the edge classes,
//                          keyed by GLOBAL node ids; then edges, components, adjacency
//
// After amg_dist_edges_global every rank holds the global node / edge tables and its own
// reads' node ids: filter / clip run identically everywhere, correct_reads on local reads.
// first-seen values carry GLOBAL token indices (token_base + local index), so minima over
// ranks reproduce the single-process insertion order exactly.
#include "amg_device.h"
#include "amg_x.h"

#define NEED_CTX(c)                                              \
  do {                                                           \
    if (!(c)) return amg_fail(AMG_E_ARG, "null ctx");            \
    HIPCHK(hipSetDevice((c)->device));                           \
  } while (0)

static inline unsigned int nblk(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

// node record: {u64 key, u64 first, u32 count, u32 k, i32 tok[k]} padded to 8 bytes
static inline size_t node_rec_bytes(int k) { return (size_t)((24 + 4 * k + 7) & ~7); }
#define EDGE_REC_BYTES 24  // {u64 key, u64 first, u32 count, u32 pad}

extern "C" int amg_dist_record_bytes(int32_t k, int64_t* node_bytes, int64_t* edge_bytes) {
  if (k < 1 || k > AMG_MAX_K) return amg_fail(AMG_E_ARG, "bad k");
  if (node_bytes) *node_bytes = (int64_t)node_rec_bytes(k);
  if (edge_bytes) *edge_bytes = EDGE_REC_BYTES;
  return AMG_OK;
}

extern "C" int amg_dist_set_filter(amg_ctx* c, uint32_t min_node_cov, uint32_t min_edge_cov) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  c->dist_min_node = min_node_cov < 1 ? 1 : min_node_cov;
  c->dist_min_edge = min_edge_cov < 1 ? 1 : min_edge_cov;
  return AMG_OK;
}

__device__ __forceinline__ unsigned int owner_of(unsigned long long key, unsigned int world) {
  return (unsigned int)(mix64(key ^ 0x5851F42D4C957F2Dull) % world);
}

// ------------------------------------------------------------------ phase: local nodes
// destination of every local node (compaction list: first / slot)
__global__ void k_dist_node_dest(const unsigned int* __restrict__ slots, long long n,
                                 const Slot* __restrict__ tab, unsigned int world,
                                 unsigned int* __restrict__ dest, unsigned int* __restrict__ idx,
                                 unsigned long long* __restrict__ counts) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int d = owner_of(tab[slots[i]].key, world);
  dest[i] = d;
  idx[i] = (unsigned int)i;
  (void)counts;  // per-destination counts come from the sorted array (k_dest_counts): millions of
                 // atomics on `world` addresses would serialise
}

// counts[d] = number of entries equal to d in the ascending array dest_sorted[0..n)
__global__ void k_dest_counts(const unsigned int* __restrict__ dest_sorted, long long n, unsigned int world,
                              unsigned long long* __restrict__ counts) {
  unsigned int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= world) return;
  auto lower = [&](unsigned int v) {
    long long lo = 0, hi = n;
    while (lo < hi) {
      long long mid = (lo + hi) >> 1;
      if (dest_sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  counts[d] = (unsigned long long)(lower(d + 1) - lower(d));
}

__global__ void k_dist_node_pack(const unsigned int* __restrict__ order, long long n,
                                 const unsigned int* __restrict__ slots,
                                 const unsigned long long* __restrict__ firsts,
                                 const Slot* __restrict__ tab, const unsigned int* __restrict__ lcnt,
                                 const int* __restrict__ tokens, int k,
                                 int two_v, long long tok_base, unsigned char* __restrict__ out,
                                 int rec_bytes) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  unsigned int i = order[j];
  const Slot* s = tab + slots[i];
  unsigned long long first = firsts[i];
  unsigned char* rec = out + (size_t)j * rec_bytes;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(rec);
  q[0] = s->key;
  q[1] = first;
  unsigned int* u = reinterpret_cast<unsigned int*>(rec + 16);
  u[0] = lcnt[s->id];  // s->id is still the LOCAL first-seen rank here
  u[1] = (unsigned int)k;
  int* tk = reinterpret_cast<int*>(rec + 24);
  long long t = (long long)(first >> 1) - tok_base;
  int dir = (first & 1ull) ? -1 : 1;
  const int flip = two_v - 1;
  for (int x = 0; x < k; ++x) tk[x] = dir > 0 ? tokens[t + x] : flip - tokens[t + k - 1 - x];
}

// per-destination record counts of a phase (dist_cnt[0 .. world))
static int fetch_counts(amg_ctx* c, unsigned long long* h, int world) {
  if (world <= FETCH_MAX) {
    FetchList l;
    l.add_words(c->dist_cnt.p, world);
    return fetch(c, l, h);
  }
  HIPCHK(hipMemcpyAsync(h, c->dist_cnt.p, (size_t)world * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// ------------------------------------------------------------------ exact local tables
// When the shard qualifies (bx_fits) the LOCAL passes are those of the single-GPU exact-key
// build (amg_build_x.hip: 16-byte slots, claim ids, dense per-claim arrays); the records that
// travel keep the format below — the key is the same 64-bit fingerprint of the tuple, so ranks
// on either path merge with each other.
// fingerprint of a canonical tuple given as tokens: same value as canon_fingerprint()
__device__ __forceinline__ unsigned long long tuple_fingerprint(const int* tok, int k, unsigned long long seed) {
  unsigned long long h = seed;
  for (int j = 0; j < k; ++j) {
    h = (h ^ (unsigned long long)(unsigned int)tok[j]) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
  }
  h = mix64(h);
  return h ? h : 1ull;
}

__global__ void k_xd_node_keys(const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                               long long n, int k, int bits, int two, unsigned long long seed, unsigned int world,
                               unsigned long long* __restrict__ keys, unsigned int* __restrict__ dest,
                               unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Slot16 s = tab[slot_by_claim[i]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;  // one-word keys: no tag there
  int tok[AMG_MAX_K];
  for (int j = 0; j < k; ++j) tok[j] = x_unpack(s.w1, tag, bits, j);
  const unsigned long long key = tuple_fingerprint(tok, k, seed);
  keys[i] = key;
  dest[i] = owner_of(key, world);
  idx[i] = (unsigned int)i;
}

__global__ void k_xd_node_pack(const unsigned int* __restrict__ order, long long n,
                               const unsigned long long* __restrict__ keys,
                               const unsigned int* __restrict__ first2,
                               long long tok_base, const unsigned int* __restrict__ lcnt,
                               const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                               int k, int bits, int two, unsigned char* __restrict__ out, int rec_bytes) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int c = order[j];
  const unsigned long long first = ((unsigned long long)tok_base << 1) + (unsigned long long)(unsigned int)~x_first_inv(first2, c);
  unsigned char* rec = out + (size_t)j * rec_bytes;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(rec);
  q[0] = keys[c];
  q[1] = first;
  unsigned int* u = reinterpret_cast<unsigned int*>(rec + 16);
  u[0] = lcnt[c];
  u[1] = (unsigned int)k;
  const Slot16 s = tab[slot_by_claim[c]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;
  int* tk = reinterpret_cast<int*>(rec + 24);
  for (int x = 0; x < k; ++x) tk[x] = x_unpack(s.w1, tag, bits, x);
}

// local claim -> global node id (looked up by key); -2 when the node fell to the fused filter
__global__ void k_xd_claims_to_global(const unsigned long long* __restrict__ keys, long long n,
                                      const Slot* __restrict__ gtab, unsigned long long gmask,
                                      int allow_missing, int* __restrict__ final_of_claim,
                                      unsigned long long* status) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = keys[i];
  unsigned long long s = (key >> 20) & gmask;
  for (unsigned int probes = 0; probes < (1u << 20); ++probes) {
    const unsigned long long cur = gtab[s].key;
    if (cur == key) {
      final_of_claim[i] = gtab[s].id;
      return;
    }
    if (cur == 0ull) break;
    s = (s + 1) & gmask;
  }
  final_of_claim[i] = -2;
  if (!allow_missing) status[ST_OVERFLOW] = 5;  // local key missing from the global table
}

__global__ void k_xd_edge_dest(const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                               long long n, unsigned int world, unsigned int* __restrict__ dest,
                               unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dest[i] = owner_of(etab[slot_by_claim[i]].w1, world);
  idx[i] = (unsigned int)i;
}

__global__ void k_xd_edge_pack(const unsigned int* __restrict__ order, long long n,
                               const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                               const unsigned int* __restrict__ first2,
                               long long tok_base, const unsigned int* __restrict__ lcnt,
                               unsigned char* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int c = order[j];
  unsigned long long* q = reinterpret_cast<unsigned long long*>(out + (size_t)j * EDGE_REC_BYTES);
  q[0] = etab[slot_by_claim[c]].w1;
  q[1] = ((unsigned long long)tok_base << 3) + (unsigned long long)(unsigned int)~x_first_inv(first2, c);
  q[2] = (unsigned long long)lcnt[c];
}

// per-destination send counts of n records whose destinations are in dest[]: sorts (dest, idx)
// into (dest_sorted, order) and fills send_counts
static int dest_counts(amg_ctx* c, long long n, int world, unsigned int* dest, unsigned int* idx,
                       unsigned int* dest_sorted, unsigned int* order, int64_t* send_counts) {
  hipStream_t st = c->stream;
  HIPCHK(hipMemsetAsync(c->dist_cnt.p, 0, (size_t)(world + 1) * sizeof(unsigned long long), st));
  if (n > 0) {
    AMGCHK(prim_sort_u32_u32(c, dest, dest_sorted, idx, order, (size_t)n, ilog2_ceil((uint64_t)world + 1) + 1));
    hipLaunchKernelGGL(k_dest_counts, dim3(nblk(world, 64)), dim3(64), 0, st, dest_sorted, n, (unsigned int)world,
                       c->dist_cnt.as<unsigned long long>());
  }
  std::vector<unsigned long long> h(world);
  AMGCHK(fetch_counts(c, h.data(), world));
  for (int i = 0; i < world; ++i) send_counts[i] = (int64_t)h[i];
  return AMG_OK;
}

static int nodes_local_x(amg_ctx* c, int k, int world, int64_t* send_counts) {
  hipStream_t st = c->stream;
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bx_nodes_upsert(c, k, &which);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 1 || attempt >= 8) return r;
    ++c->retries;
    if (c->node_slots >= (1ll << 30)) return amg_fail(AMG_E_OVERFLOW, "node table at maximum size");
    c->node_slots = c->node_slots * 4 > (1ll << 30) ? (1ll << 30) : c->node_slots * 4;
  }
  const long long n = c->n_local_nodes, T = c->n_tokens;
  // local occurrence counts per claim, straight from the per-window claims
  AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  AMGCHK(count_ids(c, c->tok_slot.as<int>(), T, nullptr, n, c->dist_lcnt.as<unsigned int>(), 2));
  AMGCHK(c->dist_a.ensure((size_t)(n + 1) * sizeof(unsigned int) * 4 + 64));
  unsigned int* dest = c->dist_a.as<unsigned int>();
  unsigned int* idx = dest + (n + 1);
  unsigned int* dest_sorted = idx + (n + 1);
  unsigned int* order = dest_sorted + (n + 1);
  AMGCHK(c->dist_cnt.ensure((size_t)(world + 1) * sizeof(unsigned long long)));
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));  // keys per claim
  if (n > 0)
    hipLaunchKernelGGL(k_xd_node_keys, dim3(nblk(n, 256)), dim3(256), 0, st, c->node_tab.as<Slot16>(),
                       c->x_slot.as<unsigned int>(), n, k, c->x_bits, (long long)k * c->x_bits > 63 ? 1 : 0, c->seed,
                       (unsigned int)world,
                       c->dist_first.as<unsigned long long>(), dest, idx);
  return dest_counts(c, n, world, dest, idx, dest_sorted, order, send_counts);
}

static int edges_local_x(amg_ctx* c, int world, int64_t* send_counts) {
  hipStream_t st = c->stream;
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bx_edges_upsert(c, &which);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 2 || attempt >= 8) return r;
    ++c->retries;
    c->edge_slots *= 4;
  }
  const long long n = c->n_local_pairs, T = c->n_tokens;
  AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  AMGCHK(count_ids(c, c->tok_pair.as<int>(), T, nullptr, n, c->dist_lcnt.as<unsigned int>(), 1));
  AMGCHK(c->dist_a.ensure((size_t)(n + 1) * sizeof(unsigned int) * 4 + 64));
  unsigned int* dest = c->dist_a.as<unsigned int>();
  unsigned int* idx = dest + (n + 1);
  unsigned int* dest_sorted = idx + (n + 1);
  unsigned int* order = dest_sorted + (n + 1);
  if (n > 0)
    hipLaunchKernelGGL(k_xd_edge_dest, dim3(nblk(n, 256)), dim3(256), 0, st, c->edge_tab.as<Slot16>(),
                       c->x_eslot.as<unsigned int>(), n, (unsigned int)world, dest, idx);
  return dest_counts(c, n, world, dest, idx, dest_sorted, order, send_counts);
}

extern "C" int amg_dist_nodes_local(amg_ctx* c, int32_t k, int64_t token_base, int64_t token_total,
                                    int32_t world, int64_t* send_counts) {
  NEED_CTX(c);
  if (k < 1 || k > AMG_MAX_K) return amg_fail(AMG_E_ARG, "k must be in [1, %d]", AMG_MAX_K);
  if (world < 1 || !send_counts) return amg_fail(AMG_E_ARG, "bad world / send_counts");
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  hipStream_t st = c->stream;
  stages_reset(c);
  c->built = false;
  c->have_corrected = false;
  c->match_valid = false;
  c->k = k;
  c->retries = 0;
  c->tok_base = token_base;
  c->tok_total = token_total;
  c->world = world;
  c->dist_mode = true;
  // merge keys and key owners are fingerprints of this seed: every rank must use the SAME one,
  // whatever collision retries an earlier single-GPU build on this ctx went through
  c->seed = kAmgSeed0;
  c->count_inline = false;  // local occurrence counts come from bs_count_by_slot, not per-window atomics
  bs_size_tables(c);
  c->exact_keys = false;
  c->dist_x = bx_fits(c, k);
  if (c->dist_x) return nodes_local_x(c, k, world, send_counts);
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bs_nodes_pass(c, k, &which);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 1 || attempt >= 8) return r;
    ++c->retries;
    if (c->node_slots >= (1ll << 30)) return amg_fail(AMG_E_OVERFLOW, "node table at maximum size");
    c->node_slots = c->node_slots * 4 > (1ll << 30) ? (1ll << 30) : c->node_slots * 4;
  }
  // compaction list lives in s1 (first) / s3 (slot); destination order -> dist_order
  const long long n = c->n_local_nodes;
  {
    // local occurrence counts: rank the local nodes by first-seen (hot nodes get low ids),
    // count through LDS (tok_node is free scratch until the edge pass writes it)
    int first_bits = ilog2_ceil((uint64_t)(c->tok_total > 0 ? c->tok_total : 1) * 2 + 2) + 1;
    AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->s2.as<unsigned long long>(),
                             c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)n, first_bits));
    AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
    AMGCHK(bs_count_by_slot(c, c->tok_slot.as<int>(), c->tok_node.as<int>(), c->n_tokens,
                            c->node_tab.as<Slot>(), c->s4.as<unsigned int>(), n,
                            c->dist_lcnt.as<unsigned int>(), 0));
  }
  AMGCHK(c->dist_a.ensure((size_t)(n + 1) * sizeof(unsigned int) * 4 + 64));
  unsigned int* dest = c->dist_a.as<unsigned int>();
  unsigned int* idx = dest + (n + 1);
  unsigned int* dest_sorted = idx + (n + 1);
  unsigned int* order = dest_sorted + (n + 1);
  AMGCHK(c->dist_cnt.ensure((size_t)(world + 1) * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(c->dist_cnt.p, 0, (size_t)(world + 1) * sizeof(unsigned long long), st));
  // keep the compaction list: the sort below uses the generic scratch
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->dist_slot.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  HIPCHK(hipMemcpyAsync(c->dist_first.p, c->s1.p, (size_t)n * sizeof(unsigned long long),
                        hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->dist_slot.p, c->s3.p, (size_t)n * sizeof(unsigned int),
                        hipMemcpyDeviceToDevice, st));
  if (n > 0) {
    hipLaunchKernelGGL(k_dist_node_dest, dim3(nblk(n, 256)), dim3(256), 0, st,
                       c->dist_slot.as<unsigned int>(), n, c->node_tab.as<Slot>(), (unsigned int)world,
                       dest, idx, c->dist_cnt.as<unsigned long long>());
    AMGCHK(prim_sort_u32_u32(c, dest, dest_sorted, idx, order, (size_t)n, ilog2_ceil((uint64_t)world + 1) + 1));
    hipLaunchKernelGGL(k_dest_counts, dim3(nblk(world, 64)), dim3(64), 0, st, dest_sorted, n, (unsigned int)world,
                       c->dist_cnt.as<unsigned long long>());
  }
  std::vector<unsigned long long> h(world);
  AMGCHK(fetch_counts(c, h.data(), world));
  for (int i = 0; i < world; ++i) send_counts[i] = (int64_t)h[i];
  return AMG_OK;
}

extern "C" int amg_dist_nodes_pack(amg_ctx* c, void* send_buf) {
  NEED_CTX(c);
  const long long n = c->n_local_nodes;
  if (n == 0) return AMG_OK;
  if (!send_buf) return amg_fail(AMG_E_ARG, "null send buffer");
  unsigned int* order = c->dist_a.as<unsigned int>() + 3 * (n + 1);
  if (c->dist_x) {
    hipLaunchKernelGGL(k_xd_node_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                       c->dist_first.as<unsigned long long>(), c->x_first.as<unsigned int>(),
                       (long long)c->tok_base,
                       c->dist_lcnt.as<unsigned int>(), c->node_tab.as<Slot16>(), c->x_slot.as<unsigned int>(),
                       c->k, c->x_bits, (long long)c->k * c->x_bits > 63 ? 1 : 0,
                       reinterpret_cast<unsigned char*>(send_buf), (int)node_rec_bytes(c->k));
    AMGCHK(stream_wait(c));
    return AMG_OK;
  }
  hipLaunchKernelGGL(k_dist_node_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                     c->dist_slot.as<unsigned int>(), c->dist_first.as<unsigned long long>(),
                     c->node_tab.as<Slot>(), c->dist_lcnt.as<unsigned int>(), c->tokens.as<int>(), c->k,
                     c->two_v, (long long)c->tok_base,
                     reinterpret_cast<unsigned char*>(send_buf), (int)node_rec_bytes(c->k));
  AMGCHK(stream_wait(c));
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: owner-side reduce
// Received records are sorted by key; a run of equal keys (one record per contributing rank)
// collapses to: sum of counts, min first-seen, tokens of the min-first record.  Records of a
// run must carry the same canonical tuple, otherwise two tuples share a fingerprint.
__global__ void k_rec_keys(const unsigned char* __restrict__ recs, long long n, int rec_bytes,
                           unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = *reinterpret_cast<const unsigned long long*>(recs + (size_t)i * rec_bytes);
  idx[i] = (unsigned int)i;
}

// head[i] = 1 for the first record of a run of equal keys whose reduced coverage reaches
// min_cov (edge classes that are self-loops count twice)
__global__ void k_run_heads(const unsigned char* __restrict__ recs, int rec_bytes, int is_edge,
                            const unsigned long long* __restrict__ keys_sorted,
                            const unsigned int* __restrict__ idx_sorted, long long n,
                            unsigned int min_cov, unsigned int* __restrict__ head) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int h = (i == 0 || keys_sorted[i] != keys_sorted[i - 1]) ? 1u : 0u;
  if (h && min_cov > 1) {
    const unsigned long long key = keys_sorted[i];
    unsigned long long total = 0;
    for (long long j = i; j < n && keys_sorted[j] == key; ++j)
      total += *reinterpret_cast<const unsigned int*>(recs + (size_t)idx_sorted[j] * rec_bytes + 16);
    if (is_edge) {
      unsigned int lo = (unsigned int)((key >> 32) & 0x7fffffffull);
      unsigned int hi = (unsigned int)(key & 0xffffffffull) - 1u;
      if (lo == hi) total *= 2;
    }
    if (total < min_cov) h = 0;
  }
  head[i] = h;
}

__global__ void k_reduce_runs(const unsigned char* __restrict__ recs, int rec_bytes, int tok_words,
                              const unsigned long long* __restrict__ keys_sorted,
                              const unsigned int* __restrict__ idx_sorted,
                              const unsigned int* __restrict__ head, const long long* __restrict__ pos,
                              long long n, unsigned char* __restrict__ out, unsigned long long* status) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const unsigned long long key = keys_sorted[i];
  unsigned long long best_first = ~0ull, total = 0;
  long long best = -1;
  for (long long j = i; j < n && keys_sorted[j] == key; ++j) {
    const unsigned char* r = recs + (size_t)idx_sorted[j] * rec_bytes;
    unsigned long long f = *reinterpret_cast<const unsigned long long*>(r + 8);
    total += *reinterpret_cast<const unsigned int*>(r + 16);
    if (f < best_first) {
      best_first = f;
      best = j;
    }
  }
  const unsigned char* b = recs + (size_t)idx_sorted[best] * rec_bytes;
  // exact tuple check across the run (fingerprint collision between ranks)
  for (long long j = i; j < n && keys_sorted[j] == key; ++j) {
    const int* t1 = reinterpret_cast<const int*>(recs + (size_t)idx_sorted[j] * rec_bytes + 24);
    const int* t0 = reinterpret_cast<const int*>(b + 24);
    for (int x = 0; x < tok_words; ++x)
      if (t1[x] != t0[x]) status[ST_COLLISION] = 1;
  }
  unsigned char* o = out + (size_t)pos[i] * rec_bytes;
  for (int x = 0; x < rec_bytes; x += 8)
    *reinterpret_cast<unsigned long long*>(o + x) = *reinterpret_cast<const unsigned long long*>(b + x);
  *reinterpret_cast<unsigned long long*>(o + 8) = best_first;
  *reinterpret_cast<unsigned int*>(o + 16) = (unsigned int)total;
}

static int reduce_records(amg_ctx* c, const void* recv, long long n, int rec_bytes, int tok_words,
                          unsigned int min_cov, DevBuf& owned, int64_t* n_owned) {
  hipStream_t st = c->stream;
  *n_owned = 0;
  if (n == 0) return AMG_OK;
  AMGCHK(c->s1.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->s2.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->s3.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  AMGCHK(c->s5.ensure((size_t)(n + 2) * (sizeof(unsigned int) + sizeof(long long))));
  const unsigned char* recs = reinterpret_cast<const unsigned char*>(recv);
  hipLaunchKernelGGL(k_rec_keys, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rec_bytes,
                     c->s1.as<unsigned long long>(), c->s3.as<unsigned int>());
  AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->s2.as<unsigned long long>(),
                           c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)n, 64));
  long long* pos = c->s5.as<long long>();
  unsigned int* head = reinterpret_cast<unsigned int*>(pos + (n + 2));
  hipLaunchKernelGGL(k_run_heads, dim3(nblk(n, 256)), dim3(256), 0, st, recs, rec_bytes, tok_words == 0 ? 1 : 0,
                     c->s2.as<unsigned long long>(), c->s4.as<unsigned int>(), n, min_cov, head);
  HIPCHK(hipMemsetAsync(head + n, 0, sizeof(unsigned int), st));
  AMGCHK(prim_exscan_u32_to_i64(c, head, pos, (size_t)n + 1));
  long long total = 0;
  {
    FetchList l;
    l.add(pos + n);
    AMGCHK(fetch(c, l, reinterpret_cast<unsigned long long*>(&total)));
  }
  AMGCHK(owned.ensure((size_t)(total + 1) * rec_bytes));
  HIPCHK(hipMemsetAsync(c->status.as<unsigned long long>() + ST_COLLISION, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_reduce_runs, dim3(nblk(n, 256)), dim3(256), 0, st, recs, rec_bytes, tok_words,
                     c->s2.as<unsigned long long>(), c->s4.as<unsigned int>(), head, pos, n,
                     owned.as<unsigned char>(), c->status.as<unsigned long long>());
  unsigned long long coll = 0;
  {
    FetchList l;
    l.add(c->status.as<unsigned long long>() + ST_COLLISION);
    AMGCHK(fetch(c, l, &coll));
  }
  if (coll) return amg_fail(AMG_E_OVERFLOW, "fingerprint collision across ranks: rebuild with another seed");
  *n_owned = total;
  return AMG_OK;
}

extern "C" int amg_dist_nodes_reduce(amg_ctx* c, const void* recv_buf, int64_t n_recv, int64_t* n_owned) {
  NEED_CTX(c);
  if (!n_owned || (n_recv > 0 && !recv_buf)) return amg_fail(AMG_E_ARG, "bad arguments");
  int r = reduce_records(c, recv_buf, n_recv, (int)node_rec_bytes(c->k), c->k, c->dist_min_node,
                         c->dist_owned, n_owned);
  c->n_owned = *n_owned;
  return r;
}

extern "C" int amg_dist_nodes_owned(amg_ctx* c, void* out) {
  NEED_CTX(c);
  if (c->n_owned > 0) {
    if (!out) return amg_fail(AMG_E_ARG, "null out");
    HIPCHK(hipMemcpyAsync(out, c->dist_owned.p, (size_t)c->n_owned * node_rec_bytes(c->k),
                          hipMemcpyDeviceToDevice, c->stream));
    AMGCHK(stream_wait(c));
  }
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: global node ids
__global__ void k_rec_firsts(const unsigned char* __restrict__ recs, long long n, int rec_bytes,
                             unsigned long long* __restrict__ firsts, unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  firsts[i] = *reinterpret_cast<const unsigned long long*>(recs + (size_t)i * rec_bytes + 8);
  idx[i] = (unsigned int)i;
}

// node arrays in global id order + key -> id table
__global__ void k_global_nodes(const unsigned char* __restrict__ recs, int rec_bytes, int k,
                               const unsigned int* __restrict__ idx_sorted, long long n,
                               Slot* __restrict__ gtab, unsigned long long gmask,
                               int* __restrict__ node_tokens, unsigned int* __restrict__ node_cov,
                               long long* __restrict__ node_first, unsigned char* __restrict__ node_alive,
                               unsigned long long* status) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* r = recs + (size_t)idx_sorted[i] * rec_bytes;
  const unsigned long long key = *reinterpret_cast<const unsigned long long*>(r);
  node_first[i] = (long long)*reinterpret_cast<const unsigned long long*>(r + 8);
  node_cov[i] = *reinterpret_cast<const unsigned int*>(r + 16);
  node_alive[i] = 1;
  const int* tk = reinterpret_cast<const int*>(r + 24);
  for (int x = 0; x < k; ++x) node_tokens[i * k + x] = tk[x];
  // insert key -> id (keys are unique after the owner-side reduce)
  unsigned long long s = (key >> 20) & gmask;
  for (unsigned int probes = 0;; ++probes) {
    unsigned long long cur = atomicCAS(&gtab[s].key, 0ull, key);
    if (cur == 0ull) {
      gtab[s].id = (int)i;
      return;
    }
    if (cur == key || probes > 1u << 20) {
      status[ST_OVERFLOW] = 4;  // duplicate key after reduce: cannot happen
      return;
    }
    s = (s + 1) & gmask;
  }
}

// local table slot -> global node id (looked up by key)
__global__ void k_local_to_global(Slot* __restrict__ ltab, unsigned long long n_slots,
                                  const Slot* __restrict__ gtab, unsigned long long gmask,
                                  const int* __restrict__ node_tokens, int k, int packed, int allow_missing,
                                  unsigned long long* status) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const unsigned long long key = ltab[i].key;
  if (key == 0ull) return;
  unsigned long long s = (key >> 20) & gmask;
  for (unsigned int probes = 0; probes < (1u << 20); ++probes) {
    unsigned long long cur = gtab[s].key;
    if (cur == key) {
      const int gid = gtab[s].id;
      if (packed)
        slot_pack(ltab + i, gid, node_tokens + (long long)gid * k, k);
      else
        ltab[i].id = gid;
      return;
    }
    if (cur == 0ull) break;
    s = (s + 1) & gmask;
  }
  if (allow_missing) {
    // the node did not reach the fused coverage threshold: its windows become None (-2)
    if (packed) {
      int none[AMG_MAX_K] = {0};
      slot_pack(ltab + i, -2, none, k);
    } else {
      ltab[i].id = -2;
    }
    return;
  }
  status[ST_OVERFLOW] = 5;  // local key missing from the global table
}

extern "C" int amg_dist_nodes_global(amg_ctx* c, const void* all_records, int64_t n_total) {
  NEED_CTX(c);
  hipStream_t st = c->stream;
  const long long n = n_total;
  const int rb = (int)node_rec_bytes(c->k);
  c->packed_nodes = (c->two_v <= 65536 && c->k <= AMG_PACK_MAX_K);
  c->n_nodes = n;
  AMGCHK(bs_alloc_nodes(c, n));
  uint64_t gslots = pow2_at_least((uint64_t)n * 2 + 16);
  AMGCHK(c->dist_gtab.ensure((size_t)gslots * sizeof(Slot)));
  HIPCHK(hipMemsetAsync(c->dist_gtab.p, 0, (size_t)gslots * sizeof(Slot), st));
  HIPCHK(hipMemsetAsync(c->status.as<unsigned long long>() + ST_OVERFLOW, 0, sizeof(unsigned long long), st));
  if (n > 0) {
    if (!all_records) return amg_fail(AMG_E_ARG, "null records");
    AMGCHK(c->s1.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
    AMGCHK(c->s2.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
    AMGCHK(c->s3.ensure((size_t)(n + 1) * sizeof(unsigned int)));
    AMGCHK(c->s4.ensure((size_t)(n + 1) * sizeof(unsigned int)));
    const unsigned char* recs = reinterpret_cast<const unsigned char*>(all_records);
    hipLaunchKernelGGL(k_rec_firsts, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rb,
                       c->s1.as<unsigned long long>(), c->s3.as<unsigned int>());
    int first_bits = ilog2_ceil((uint64_t)(c->tok_total > 0 ? c->tok_total : 1) * 2 + 2) + 1;
    AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->s2.as<unsigned long long>(),
                             c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)n, first_bits));
    hipLaunchKernelGGL(k_global_nodes, dim3(nblk(n, 256)), dim3(256), 0, st, recs, rb, c->k,
                       c->s4.as<unsigned int>(), n, c->dist_gtab.as<Slot>(), (unsigned long long)(gslots - 1),
                       c->node_tokens.as<int>(), c->node_cov.as<unsigned int>(),
                       c->node_first.as<long long>(), c->node_alive.as<unsigned char>(),
                       c->status.as<unsigned long long>());
  }
  if (c->dist_x) {
    c->packed_nodes = false;
    if (c->n_local_nodes > 0)
      hipLaunchKernelGGL(k_xd_claims_to_global, dim3(nblk(c->n_local_nodes, 256)), dim3(256), 0, st,
                         c->dist_first.as<unsigned long long>(), (long long)c->n_local_nodes,
                         c->dist_gtab.as<Slot>(), (unsigned long long)(gslots - 1),
                         c->dist_min_node > 1 ? 1 : 0, c->x_final.as<int>(), c->status.as<unsigned long long>());
  } else
  hipLaunchKernelGGL(k_local_to_global, dim3(nblk(c->node_slots, 256)), dim3(256), 0, st,
                     c->node_tab.as<Slot>(), (unsigned long long)c->node_slots, c->dist_gtab.as<Slot>(),
                     (unsigned long long)(gslots - 1), c->node_tokens.as<int>(), c->k,
                     c->packed_nodes ? 1 : 0, c->dist_min_node > 1 ? 1 : 0,
                     c->status.as<unsigned long long>());
  unsigned long long ov = 0;
  {
    FetchList l;
    l.add(c->status.as<unsigned long long>() + ST_OVERFLOW);
    AMGCHK(fetch(c, l, &ov));
  }
  if (ov) return amg_fail(AMG_E_DIST, "global node table inconsistent (code %llu)", ov);
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: edges
__global__ void k_dist_edge_dest(const unsigned int* __restrict__ slots, long long n,
                                 const Slot* __restrict__ tab, unsigned int world,
                                 unsigned int* __restrict__ dest, unsigned int* __restrict__ idx,
                                 unsigned long long* __restrict__ counts) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int d = owner_of(tab[slots[i]].key, world);
  dest[i] = d;
  idx[i] = (unsigned int)i;
  (void)counts;
}

__global__ void k_dist_edge_pack(const unsigned int* __restrict__ order, long long n,
                                 const unsigned int* __restrict__ slots,
                                 const unsigned long long* __restrict__ firsts,
                                 const Slot* __restrict__ tab, const unsigned int* __restrict__ lcnt,
                                 unsigned char* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  unsigned int i = order[j];
  const Slot* s = tab + slots[i];
  unsigned long long* q = reinterpret_cast<unsigned long long*>(out + (size_t)j * EDGE_REC_BYTES);
  q[0] = s->key;
  q[1] = firsts[i];
  q[2] = (unsigned long long)lcnt[s->id];
}

extern "C" int amg_dist_edges_local(amg_ctx* c, int32_t world, int64_t* send_counts) {
  NEED_CTX(c);
  if (world < 1 || !send_counts) return amg_fail(AMG_E_ARG, "bad world / send_counts");
  hipStream_t st = c->stream;
  if (c->dist_x) return edges_local_x(c, world, send_counts);
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bs_edges_pass(c, &which);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 2 || attempt >= 8) {
      if (which == 3) return amg_fail(AMG_E_OVERFLOW, "fingerprint collision: rebuild with another seed");
      return r;
    }
    ++c->retries;
    c->edge_slots *= 4;
  }
  const long long n = c->n_local_pairs;
  {
    int efirst_bits = ilog2_ceil((uint64_t)(c->tok_total > 0 ? c->tok_total : 1) * 8 + 8) + 1;
    AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->s2.as<unsigned long long>(),
                             c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)n, efirst_bits));
    AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
    AMGCHK(bs_count_by_slot(c, c->tok_pair.as<int>(), c->tok_pair.as<int>(), c->n_tokens,
                            c->edge_tab.as<Slot>(), c->s4.as<unsigned int>(), n,
                            c->dist_lcnt.as<unsigned int>(), 1));
  }
  AMGCHK(c->dist_a.ensure((size_t)(n + 1) * sizeof(unsigned int) * 4 + 64));
  unsigned int* dest = c->dist_a.as<unsigned int>();
  unsigned int* idx = dest + (n + 1);
  unsigned int* dest_sorted = idx + (n + 1);
  unsigned int* order = dest_sorted + (n + 1);
  HIPCHK(hipMemsetAsync(c->dist_cnt.p, 0, (size_t)(world + 1) * sizeof(unsigned long long), st));
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->dist_slot.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  HIPCHK(hipMemcpyAsync(c->dist_first.p, c->s1.p, (size_t)n * sizeof(unsigned long long),
                        hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->dist_slot.p, c->s3.p, (size_t)n * sizeof(unsigned int),
                        hipMemcpyDeviceToDevice, st));
  if (n > 0) {
    hipLaunchKernelGGL(k_dist_edge_dest, dim3(nblk(n, 256)), dim3(256), 0, st,
                       c->dist_slot.as<unsigned int>(), n, c->edge_tab.as<Slot>(), (unsigned int)world,
                       dest, idx, c->dist_cnt.as<unsigned long long>());
    AMGCHK(prim_sort_u32_u32(c, dest, dest_sorted, idx, order, (size_t)n, ilog2_ceil((uint64_t)world + 1) + 1));
    hipLaunchKernelGGL(k_dest_counts, dim3(nblk(world, 64)), dim3(64), 0, st, dest_sorted, n, (unsigned int)world,
                       c->dist_cnt.as<unsigned long long>());
  }
  std::vector<unsigned long long> h(world);
  AMGCHK(fetch_counts(c, h.data(), world));
  for (int i = 0; i < world; ++i) send_counts[i] = (int64_t)h[i];
  return AMG_OK;
}

extern "C" int amg_dist_edges_pack(amg_ctx* c, void* send_buf) {
  NEED_CTX(c);
  const long long n = c->n_local_pairs;
  if (n == 0) return AMG_OK;
  if (!send_buf) return amg_fail(AMG_E_ARG, "null send buffer");
  unsigned int* order = c->dist_a.as<unsigned int>() + 3 * (n + 1);
  if (c->dist_x) {
    hipLaunchKernelGGL(k_xd_edge_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                       c->edge_tab.as<Slot16>(), c->x_eslot.as<unsigned int>(), c->x_efirst.as<unsigned int>(),
                       (long long)c->tok_base,
                       c->dist_lcnt.as<unsigned int>(), reinterpret_cast<unsigned char*>(send_buf));
    AMGCHK(stream_wait(c));
    return AMG_OK;
  }
  hipLaunchKernelGGL(k_dist_edge_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                     c->dist_slot.as<unsigned int>(), c->dist_first.as<unsigned long long>(),
                     c->edge_tab.as<Slot>(), c->dist_lcnt.as<unsigned int>(),
                     reinterpret_cast<unsigned char*>(send_buf));
  AMGCHK(stream_wait(c));
  return AMG_OK;
}

extern "C" int amg_dist_edges_reduce(amg_ctx* c, const void* recv_buf, int64_t n_recv, int64_t* n_owned) {
  NEED_CTX(c);
  if (!n_owned || (n_recv > 0 && !recv_buf)) return amg_fail(AMG_E_ARG, "bad arguments");
  int r = reduce_records(c, recv_buf, n_recv, EDGE_REC_BYTES, 0, c->dist_min_edge, c->dist_owned, n_owned);
  c->n_owned = *n_owned;
  return r;
}

extern "C" int amg_dist_edges_owned(amg_ctx* c, void* out) {
  NEED_CTX(c);
  if (c->n_owned > 0) {
    if (!out) return amg_fail(AMG_E_ARG, "null out");
    HIPCHK(hipMemcpyAsync(out, c->dist_owned.p, (size_t)c->n_owned * EDGE_REC_BYTES,
                          hipMemcpyDeviceToDevice, c->stream));
    AMGCHK(stream_wait(c));
  }
  return AMG_OK;
}

__global__ __launch_bounds__(256) void k_flag_dead_reads(const int* __restrict__ tok_node,
                                                          const long long* __restrict__ read_off,
                                                          long long n_reads, unsigned char* __restrict__ read_fix) {
  long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_reads) return;
  const int lane = threadIdx.x & 63;
  bool hit = false;
  for (long long t = read_off[r] + lane; t < read_off[r + 1]; t += 64) hit = hit || (tok_node[t] == -2);
  if (__any(hit) && lane == 0) read_fix[r] = 1;
}

__global__ void k_global_pairs(const unsigned char* __restrict__ recs, const unsigned int* __restrict__ idx_sorted,
                               long long n, unsigned long long* __restrict__ pkey,
                               unsigned int* __restrict__ pcnt) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long* q =
      reinterpret_cast<const unsigned long long*>(recs + (size_t)idx_sorted[i] * EDGE_REC_BYTES);
  pkey[i] = q[0];
  pcnt[i] = (unsigned int)q[2];
}

extern "C" int amg_dist_edges_global(amg_ctx* c, const void* all_records, int64_t n_total) {
  NEED_CTX(c);
  hipStream_t st = c->stream;
  const long long n = n_total;
  c->n_pairs = n;
  AMGCHK(bs_alloc_pairs(c, n));
  if (n > 0) {
    if (!all_records) return amg_fail(AMG_E_ARG, "null records");
    AMGCHK(c->s1.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
    AMGCHK(c->s3.ensure((size_t)(n + 1) * sizeof(unsigned int)));
    AMGCHK(c->s4.ensure((size_t)(n + 1) * sizeof(unsigned int)));
    const unsigned char* recs = reinterpret_cast<const unsigned char*>(all_records);
    hipLaunchKernelGGL(k_rec_firsts, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, EDGE_REC_BYTES,
                       c->s1.as<unsigned long long>(), c->s3.as<unsigned int>());
    int efirst_bits = ilog2_ceil((uint64_t)(c->tok_total > 0 ? c->tok_total : 1) * 8 + 8) + 1;
    AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->pair_first.as<unsigned long long>(),
                             c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)n, efirst_bits));
    hipLaunchKernelGGL(k_global_pairs, dim3(nblk(n, 256)), dim3(256), 0, st, recs, c->s4.as<unsigned int>(), n,
                       c->pair_key.as<unsigned long long>(), c->pair_cnt.as<unsigned int>());
  }
  AMGCHK(bs_finish_from_pairs(c));
  if (c->dist_min_node > 1 && c->n_reads > 0) {
    // fused filter: reads that lost a node join _readsToCorrect (remove_node_from_reads :442-461)
    hipLaunchKernelGGL(k_flag_dead_reads, dim3(nblk(c->n_reads, 4)), dim3(256), 0, st, c->tok_node.as<int>(),
                       c->read_off.as<long long>(), c->n_reads, c->read_fix.as<unsigned char>());
    AMGCHK(stream_wait(c));
  }
  c->dist_min_node = c->dist_min_edge = 1;
  c->built = true;
  c->node_hint = c->n_local_nodes > 256 ? c->n_local_nodes : 256;
  return AMG_OK;
}
