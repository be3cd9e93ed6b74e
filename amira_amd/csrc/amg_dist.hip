// amg_dist.hip — read-sharded build with a key-owner table merge.
//
// Every rank holds a contiguous shard of the reads.  The single-graph result
// (graph_utils.py:105-124 at cores = 1, i.e. GeneMerGraph over all reads) is obtained in
// phases; the exchanges between them are done by the CALLER with RCCL (torch.distributed
// all_to_all_single / all_gather_into_tensor on the device buffers passed here), so the same
// phases also run in a single process with a loop-back exchange (tests on one GPU):
//
//   amg_dist_nodes_local   local windows -> local node table; one record per local node,
//                          bucketed by owner = hash(key) mod world
//   amg_dist_nodes_pack    records in destination order                     --> all-to-all
//   amg_dist_nodes_reduce  owner side: equal keys reduced through a hash table (sum count, min
//                          first-seen); the owner's survivors                --> all-gather
//                          and one reply per received record (its key's global first-seen, or
//                          "dropped by the fused filter")                    --> all-to-all back
//   amg_dist_nodes_global  global node id = rank of first-seen: a token position opens at most one
//                          window, so the first-seen token indices of the nodes are distinct and the
//                          rank is a prefix popcount over a bitmap of the GLOBAL token space — no sort,
//                          no key -> id table; every rank fills the node arrays from the gathered
//                          records and maps its own local nodes through the replies
//   amg_dist_edges_local / _pack / _reduce / _global   the same for the edge classes, keyed by
//                          GLOBAL node ids (no replies: nothing per adjacency needs the class id);
//                          then edges, components, adjacency
//
// After amg_dist_edges_global every rank holds the global node / edge tables and its own
// reads' node ids: filter / clip run identically everywhere, correct_reads on local reads.
// first-seen values carry GLOBAL token indices (token_base + local index), so minima over
// ranks reproduce the single-process insertion order exactly.
//
// No phase ends with a host synchronisation of its own: whatever the host needs (counts that size
// the caller's buffers, status words) comes back through fetch(), everything else is ordered by
// the ctx's stream — the caller issues its collectives on that same stream (amira_amd/dist.py).
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

// node record: {u64 key, u64 first, u32 count, u32 k, i32 tok[k]} padded to 8 bytes; key != 0
static inline size_t node_rec_bytes(int k) { return (size_t)((24 + 4 * k + 7) & ~7); }
#define EDGE_REC_BYTES 24  // {u64 key, u64 first, u32 count, u32 pad}; key != 0
#define REPLY_DROPPED (~0ull)

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
__global__ void k_dist_dest(const unsigned int* __restrict__ slots, long long n, const Slot* __restrict__ tab,
                            unsigned int world, unsigned int* __restrict__ dest, unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dest[i] = owner_of(tab[slots[i]].key, world);
  idx[i] = (unsigned int)i;
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

// order == nullptr: the records leave in local order (one destination: nothing was sorted)
__global__ void k_dist_node_pack(const unsigned int* __restrict__ order, long long n,
                                 const unsigned int* __restrict__ slots,
                                 const unsigned long long* __restrict__ firsts,
                                 const Slot* __restrict__ tab, const unsigned int* __restrict__ lcnt,
                                 const int* __restrict__ tokens, int k,
                                 int two_v, long long tok_base, unsigned char* __restrict__ out,
                                 int rec_bytes) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  unsigned int i = order ? order[j] : (unsigned int)j;
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
// travel keep the format above — the key is the same 64-bit fingerprint of the tuple, so ranks
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

// (claim ids nobody took — shard counters leave holes — have first-seen 0: they get destination `world`, which sorts
// behind every rank and is never sent; `bucket`: destinations are wanted, i.e. world > 1 or there are holes)
__global__ void k_xd_node_keys(const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                               const unsigned int* __restrict__ first2,
                               long long n, int k, int bits, int two, unsigned long long seed, unsigned int world, int bucket,
                               unsigned long long* __restrict__ keys, unsigned int* __restrict__ dest,
                               unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (x_first_inv(first2, i) == 0u) {
    keys[i] = 0ull;
    if (bucket) {
      dest[i] = world;
      idx[i] = (unsigned int)i;
    }
    return;
  }
  const Slot16 s = tab[slot_by_claim[i]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;  // one-word keys: no tag there
  int tok[AMG_MAX_K];
  for (int j = 0; j < k; ++j) tok[j] = x_unpack(s.w1, tag, bits, j);
  const unsigned long long key = tuple_fingerprint(tok, k, seed);
  keys[i] = key;
  if (bucket) {
    dest[i] = world > 1 ? owner_of(key, world) : 0u;
    idx[i] = (unsigned int)i;
  }
}

__global__ void k_xd_node_pack(const unsigned int* __restrict__ order, long long n,
                               const unsigned long long* __restrict__ keys,
                               const unsigned int* __restrict__ first2,
                               long long tok_base, const unsigned int* __restrict__ lcnt,
                               const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                               int k, int bits, int two, unsigned char* __restrict__ out, int rec_bytes) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
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

__global__ void k_xd_edge_dest(const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                               const unsigned int* __restrict__ first2, long long n, unsigned int world,
                               unsigned int* __restrict__ dest, unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dest[i] = x_first_inv(first2, i) == 0u ? world : (world > 1 ? owner_of(etab[slot_by_claim[i]].w1, world) : 0u);
  idx[i] = (unsigned int)i;
}

__global__ void k_xd_edge_pack(const unsigned int* __restrict__ order, long long n,
                               const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                               const unsigned int* __restrict__ first2,
                               long long tok_base, const unsigned int* __restrict__ lcnt,
                               unsigned char* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(out + (size_t)j * EDGE_REC_BYTES);
  q[0] = etab[slot_by_claim[c]].w1;
  q[1] = ((unsigned long long)tok_base << 3) + (unsigned long long)(unsigned int)~x_first_inv(first2, c);
  q[2] = (unsigned long long)lcnt[c];
}

// the four arrays of a bucketing (n + 1 words each) inside dist_a
struct Bucketing {
  unsigned int *dest, *idx, *dest_sorted, *order;
};
static int bucketing(amg_ctx* c, long long n, Bucketing* b) {
  AMGCHK(c->dist_a.ensure((size_t)(n + 1) * sizeof(unsigned int) * 4 + 64));
  b->dest = c->dist_a.as<unsigned int>();
  b->idx = b->dest + (n + 1);
  b->dest_sorted = b->idx + (n + 1);
  b->order = b->dest_sorted + (n + 1);
  return AMG_OK;
}
// order in which the local records leave (nullptr: local order — one destination and no unclaimed ids in between);
// the bucketing ran over c->dist_nspace claim ids
static const unsigned int* send_order(const amg_ctx* c) {
  return c->dist_sorted ? c->dist_a.as<unsigned int>() + 3 * (c->dist_nspace + 1) : nullptr;
}

// per-destination send counts of n records whose destinations are in dest[]: sorts (dest, idx)
// into (dest_sorted, order) and fills send_counts.  One destination: nothing to sort, the records leave in local
// order (send_order() == nullptr) and no count has to come back from the device.
// n: ids bucketed (claim ids in use, holes included), n_real: the records among them
static int dest_counts(amg_ctx* c, long long n, long long n_real, int world, const Bucketing& b, int64_t* send_counts) {
  hipStream_t st = c->stream;
  c->dist_nspace = n;
  c->dist_sorted = world > 1 || n != n_real;
  if (world == 1) {
    send_counts[0] = n_real;
    if (n == n_real) return AMG_OK;
    // one destination, but unclaimed ids in between: the sort moves them behind the records
    if (n > 0)
      AMGCHK(prim_sort_u32_u32(c, b.dest, b.dest_sorted, b.idx, b.order, (size_t)n, 2));
    return AMG_OK;
  }
  AMGCHK(c->dist_cnt.ensure((size_t)(world + 1) * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(c->dist_cnt.p, 0, (size_t)(world + 1) * sizeof(unsigned long long), st));
  if (n > 0) {
    AMGCHK(prim_sort_u32_u32(c, b.dest, b.dest_sorted, b.idx, b.order, (size_t)n, ilog2_ceil((uint64_t)world + 1) + 1));
    hipLaunchKernelGGL(k_dest_counts, dim3(nblk(world, 64)), dim3(64), 0, st, b.dest_sorted, n, (unsigned int)world,
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
    int r = bx_nodes_upsert(c, k, &which, !getenv("AMG_DIST_ONE_COUNTER"), false);  // claims from the shard counters (A/B switch)
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 1 || attempt >= 8) return r;
    ++c->retries;
    if (c->node_slots >= (1ll << 30)) return amg_fail(AMG_E_OVERFLOW, "node table at maximum size");
    c->node_slots = c->node_slots * 4 > (1ll << 30) ? (1ll << 30) : c->node_slots * 4;
  }
  // claim ids in use lie below n (shard counters: with ids nobody took in between, first-seen 0)
  const long long n = c->x_nspace, T = c->n_tokens;
  // local occurrence counts per claim, straight from the per-window claims
  stage_begin(c, "node_count");
  AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  AMGCHK(count_ids(c, c->tok_slot.as<int>(), T, nullptr, n, c->dist_lcnt.as<unsigned int>(), 4));
  stage_end(c);
  stage_begin(c, "merge_node_bucket");
  Bucketing b;
  AMGCHK(bucketing(c, n, &b));
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));  // keys per claim
  if (n > 0)
    hipLaunchKernelGGL(k_xd_node_keys, dim3(nblk(n, 256)), dim3(256), 0, st, c->node_tab.as<Slot16>(),
                       c->x_slot.as<unsigned int>(), c->x_first.as<unsigned int>(), n, k, c->x_bits,
                       (long long)k * c->x_bits > 63 ? 1 : 0, c->seed, (unsigned int)world,
                       (world > 1 || n != c->n_local_nodes) ? 1 : 0, c->dist_first.as<unsigned long long>(), b.dest, b.idx);
  const int r = dest_counts(c, n, c->n_local_nodes, world, b, send_counts);
  stage_end(c);
  return r;
}

static int edges_local_x(amg_ctx* c, int world, int64_t* send_counts) {
  hipStream_t st = c->stream;
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bx_edges_upsert(c, &which, false, !getenv("AMG_DIST_ONE_COUNTER"), false);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 2 || attempt >= 8) return r;
    ++c->retries;
    c->edge_slots *= 4;
  }
  const long long n = c->x_espace, T = c->n_tokens;  // (claim ids in use lie below n: see nodes_local_x)
  stage_begin(c, "edge_count");
  AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  AMGCHK(count_ids(c, c->tok_pair.as<int>(), T, nullptr, n, c->dist_lcnt.as<unsigned int>(), 5));
  stage_end(c);
  stage_begin(c, "merge_edge_bucket");
  Bucketing b;
  AMGCHK(bucketing(c, n, &b));
  if (n > 0 && (world > 1 || n != c->n_local_pairs))
    hipLaunchKernelGGL(k_xd_edge_dest, dim3(nblk(n, 256)), dim3(256), 0, st, c->edge_tab.as<Slot16>(),
                       c->x_eslot.as<unsigned int>(), c->x_efirst.as<unsigned int>(), n, (unsigned int)world, b.dest, b.idx);
  const int r = dest_counts(c, n, c->n_local_pairs, world, b, send_counts);
  stage_end(c);
  return r;
}

extern "C" int amg_dist_nodes_local(amg_ctx* c, int32_t k, int64_t token_base, int64_t token_total,
                                    int32_t world, int32_t attempt, int64_t* send_counts) {
  NEED_CTX(c);
  if (k < 1 || k > AMG_MAX_K) return amg_fail(AMG_E_ARG, "k must be in [1, %d]", AMG_MAX_K);
  if (world < 1 || !send_counts) return amg_fail(AMG_E_ARG, "bad world / send_counts");
  if (attempt < 0) return amg_fail(AMG_E_ARG, "bad attempt");
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
  // whatever collision retries an earlier single-GPU build on this ctx went through.  `attempt`
  // is the caller's collective retry counter (a cross-rank fingerprint collision makes every
  // rank come back with attempt + 1: amira_amd/dist.py)
  c->seed = kAmgSeed0;
  for (int a = 0; a < attempt; ++a) c->seed = c->seed * 6364136223846793005ull + 1442695040888963407ull;
  c->count_inline = false;  // local occurrence counts come from bs_count_by_slot, not per-window atomics
  bs_size_tables(c);
  c->exact_keys = false;
  c->dist_x = bx_tuple_fits(c, k);  // (the records carry the tuple: the slots must hold it)
  if (c->dist_x) return nodes_local_x(c, k, world, send_counts);
  for (int tries = 0;; ++tries) {
    int which = 0;
    int r = bs_nodes_pass(c, k, &which);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 1 || tries >= 8) return r;
    ++c->retries;
    if (c->node_slots >= (1ll << 30)) return amg_fail(AMG_E_OVERFLOW, "node table at maximum size");
    c->node_slots = c->node_slots * 4 > (1ll << 30) ? (1ll << 30) : c->node_slots * 4;
  }
  // compaction list lives in s1 (first) / s3 (slot); destination order -> dist_a
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
  Bucketing b;
  AMGCHK(bucketing(c, n, &b));
  // keep the compaction list: the sort below uses the generic scratch
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->dist_slot.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  HIPCHK(hipMemcpyAsync(c->dist_first.p, c->s1.p, (size_t)n * sizeof(unsigned long long),
                        hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->dist_slot.p, c->s3.p, (size_t)n * sizeof(unsigned int),
                        hipMemcpyDeviceToDevice, st));
  if (n > 0 && world > 1)
    hipLaunchKernelGGL(k_dist_dest, dim3(nblk(n, 256)), dim3(256), 0, st,
                       c->dist_slot.as<unsigned int>(), n, c->node_tab.as<Slot>(), (unsigned int)world, b.dest, b.idx);
  return dest_counts(c, n, n, world, b, send_counts);
}

extern "C" int amg_dist_nodes_pack(amg_ctx* c, void* send_buf) {
  NEED_CTX(c);
  const long long n = c->n_local_nodes;
  if (n == 0) return AMG_OK;
  if (!send_buf) return amg_fail(AMG_E_ARG, "null send buffer");
  const unsigned int* order = send_order(c);
  stage_begin(c, "merge_node_pack");
  if (c->dist_x)
    hipLaunchKernelGGL(k_xd_node_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                       c->dist_first.as<unsigned long long>(), c->x_first.as<unsigned int>(),
                       (long long)c->tok_base,
                       c->dist_lcnt.as<unsigned int>(), c->node_tab.as<Slot16>(), c->x_slot.as<unsigned int>(),
                       c->k, c->x_bits, (long long)c->k * c->x_bits > 63 ? 1 : 0,
                       reinterpret_cast<unsigned char*>(send_buf), (int)node_rec_bytes(c->k));
  else
    hipLaunchKernelGGL(k_dist_node_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                       c->dist_slot.as<unsigned int>(), c->dist_first.as<unsigned long long>(),
                       c->node_tab.as<Slot>(), c->dist_lcnt.as<unsigned int>(), c->tokens.as<int>(), c->k,
                       c->two_v, (long long)c->tok_base,
                       reinterpret_cast<unsigned char*>(send_buf), (int)node_rec_bytes(c->k));
  stage_end(c);
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: owner-side reduce
// Records of one key arrive from every rank that saw it.  They meet in an open-addressing table
// keyed by the record key: count = sum of the local counts, first-seen = the minimum (atomicMax
// of the complement); the record that carries the minimum is the key's representative (first-seen
// values of different ranks differ: they are global token indices) and is what the owner hands
// on.  Every other record's canonical tuple must equal the representative's, otherwise two tuples
// share a fingerprint.  Records that all come from ONE rank are distinct keys already: no table.
__device__ __forceinline__ bool edge_key_self_loop(unsigned long long key) {
  const unsigned int lo = (unsigned int)((key >> 32) & 0x7fffffffull);
  const unsigned int hi = (unsigned int)(key & 0xffffffffull) - 1u;
  return lo == hi;
}

__global__ void k_own_upsert(const unsigned char* __restrict__ recs, long long n, int rec_bytes, Slot* tab,
                             unsigned long long mask, unsigned int* __restrict__ recslot,
                             unsigned long long* status) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned char* r = recs + (size_t)j * rec_bytes;
  const unsigned long long key = *reinterpret_cast<const unsigned long long*>(r);
  const unsigned long long first = *reinterpret_cast<const unsigned long long*>(r + 8);
  const unsigned int cnt = *reinterpret_cast<const unsigned int*>(r + 16);
  const long long slot = table_upsert(tab, mask, key, mix64(key), first, 1u << 16, false, status + ST_OVERFLOW);
  if (slot < 0) {
    status[ST_OVERFLOW] = 6;
    recslot[j] = 0u;
    return;
  }
  atomicAdd(&tab[slot].count, cnt);
  recslot[j] = (unsigned int)slot;
}

// flag[j] = record j is the representative of a key that reaches min_cov (edge classes that are
// self-loops count twice)
template <bool MULTI>
__global__ void k_own_flag(const unsigned char* __restrict__ recs, long long n, int rec_bytes, int is_edge,
                           unsigned int min_cov, Slot* tab, const unsigned int* __restrict__ recslot,
                           unsigned int* __restrict__ flag) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned char* r = recs + (size_t)j * rec_bytes;
  const unsigned long long key = *reinterpret_cast<const unsigned long long*>(r);
  const unsigned long long first = *reinterpret_cast<const unsigned long long*>(r + 8);
  unsigned long long total = *reinterpret_cast<const unsigned int*>(r + 16);
  bool rep = true;
  if (MULTI) {
    Slot* s = tab + recslot[j];
    total = s->count;
    rep = ~s->first_inv == first;
    if (rep) s->id = (int)j;  // one writer per slot
  }
  if (is_edge && edge_key_self_loop(key)) total *= 2;
  flag[j] = (rep && total >= min_cov) ? 1u : 0u;
}

template <bool MULTI>
__global__ void k_own_emit(const unsigned char* __restrict__ recs, long long n, int rec_bytes, int tok_words,
                           int is_edge, unsigned int min_cov, const Slot* __restrict__ tab,
                           const unsigned int* __restrict__ recslot, const unsigned int* __restrict__ flag,
                           const long long* __restrict__ pos, unsigned char* __restrict__ owned,
                           unsigned long long* __restrict__ replies, unsigned long long* status) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned char* r = recs + (size_t)j * rec_bytes;
  const unsigned long long key = *reinterpret_cast<const unsigned long long*>(r);
  unsigned long long gfirst = *reinterpret_cast<const unsigned long long*>(r + 8);
  unsigned long long total = *reinterpret_cast<const unsigned int*>(r + 16);
  if (MULTI) {
    const Slot* s = tab + recslot[j];
    total = s->count;
    gfirst = ~s->first_inv;
    if (tok_words > 0 && s->id != (int)j) {  // exact tuple check against the representative
      const int* t0 = reinterpret_cast<const int*>(recs + (size_t)s->id * rec_bytes + 24);
      const int* t1 = reinterpret_cast<const int*>(r + 24);
      for (int x = 0; x < tok_words; ++x)
        if (t1[x] != t0[x]) status[ST_COLLISION] = 1;
    }
  }
  const unsigned long long cov = (is_edge && edge_key_self_loop(key)) ? total * 2 : total;
  if (replies) replies[j] = cov >= min_cov ? gfirst : REPLY_DROPPED;
  if (flag[j]) {
    unsigned char* o = owned + (size_t)pos[j] * rec_bytes;
    for (int x = 0; x < rec_bytes; x += 8)
      *reinterpret_cast<unsigned long long*>(o + x) = *reinterpret_cast<const unsigned long long*>(r + x);
    *reinterpret_cast<unsigned int*>(o + 16) = (unsigned int)total;
  }
}

// owned_out: room for n records; replies_out (nodes): n words
static int reduce_records(amg_ctx* c, const void* recv, long long n, int n_sources, int rec_bytes, int tok_words,
                          unsigned int min_cov, void* owned_out, unsigned long long* replies_out, int64_t* n_owned) {
  hipStream_t st = c->stream;
  *n_owned = 0;
  if (n == 0) return AMG_OK;
  if (!recv || !owned_out) return amg_fail(AMG_E_ARG, "null record buffer");
  const bool multi = n_sources > 1;
  const int is_edge = tok_words == 0 ? 1 : 0;
  AMGCHK(c->s3.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(n + 2) * sizeof(unsigned int)));
  AMGCHK(c->s5.ensure((size_t)(n + 2) * sizeof(long long)));
  const unsigned char* recs = reinterpret_cast<const unsigned char*>(recv);
  unsigned int* recslot = c->s3.as<unsigned int>();
  unsigned int* flag = c->s4.as<unsigned int>();
  long long* pos = c->s5.as<long long>();
  unsigned long long* status = c->status.as<unsigned long long>();
  uint64_t slots = 0;
  {
    ClearList cl;
    cl.add(flag + n, sizeof(unsigned int));
    cl.add(status + ST_OVERFLOW, sizeof(unsigned long long));
    cl.add(status + ST_COLLISION, sizeof(unsigned long long));
    if (multi) {
      slots = pow2_at_least((uint64_t)n * 2 + 16);
      AMGCHK(c->dist_gtab.ensure((size_t)slots * sizeof(Slot)));
      cl.add(c->dist_gtab.p, (size_t)slots * sizeof(Slot));
    }
    AMGCHK(clear_many(c, cl));
  }
  Slot* tab = c->dist_gtab.as<Slot>();
  if (multi) {
    hipLaunchKernelGGL(k_own_upsert, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rec_bytes, tab,
                       (unsigned long long)(slots - 1), recslot, status);
    hipLaunchKernelGGL(k_own_flag<true>, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rec_bytes, is_edge, min_cov,
                       tab, recslot, flag);
  } else {
    hipLaunchKernelGGL(k_own_flag<false>, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rec_bytes, is_edge, min_cov,
                       tab, recslot, flag);
  }
  AMGCHK(prim_exscan_u32_to_i64(c, flag, pos, (size_t)n + 1));
  if (multi)
    hipLaunchKernelGGL(k_own_emit<true>, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rec_bytes, tok_words, is_edge,
                       min_cov, tab, recslot, flag, pos, reinterpret_cast<unsigned char*>(owned_out), replies_out, status);
  else
    hipLaunchKernelGGL(k_own_emit<false>, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, rec_bytes, tok_words, is_edge,
                       min_cov, tab, recslot, flag, pos, reinterpret_cast<unsigned char*>(owned_out), replies_out, status);
  unsigned long long h[3] = {0, 0, 0};
  {
    FetchList l;
    l.add(pos + n);
    l.add(status + ST_COLLISION);
    l.add(status + ST_OVERFLOW);
    AMGCHK(fetch(c, l, h));
  }
  if (h[2]) return amg_fail(AMG_E_DIST, "owner table full (code %llu)", h[2]);
  if (h[1]) return amg_fail(AMG_E_COLLISION, "fingerprint collision across ranks: the merged build is repeated with the next seed");
  *n_owned = (int64_t)h[0];
  return AMG_OK;
}

extern "C" int amg_dist_nodes_reduce(amg_ctx* c, const void* recv_buf, int64_t n_recv, int32_t n_sources,
                                     void* owned_out, void* replies_out, int64_t* n_owned) {
  NEED_CTX(c);
  if (!n_owned || n_recv < 0 || (n_recv > 0 && !replies_out)) return amg_fail(AMG_E_ARG, "bad arguments");
  stage_begin(c, "merge_node_reduce");
  const int r = reduce_records(c, recv_buf, n_recv, n_sources, (int)node_rec_bytes(c->k), c->k, c->dist_min_node,
                               owned_out, reinterpret_cast<unsigned long long*>(replies_out), n_owned);
  stage_end(c);
  c->n_owned = *n_owned;
  return r;
}

// ------------------------------------------------------------------ phase: global ids
// rank of a first-seen value among all of them = number of set bits before its token in a bitmap
// over the GLOBAL token space (one bit per record).  Many records: one byte per token first, set
// with plain stores and folded into the bitmap words by the pass that counts them (scattered
// atomicOr runs at the memory-side atomic rate); few records (a filtered graph over a long token
// stream): atomicOr on the words directly, nothing token-sized but the words to clear.
__global__ void k_d_setflags(const unsigned char* __restrict__ recs, long long n_slots, int rec_bytes, int shift,
                             unsigned char* __restrict__ flags, unsigned int* __restrict__ bits) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const unsigned char* r = recs + (size_t)i * rec_bytes;
  if (*reinterpret_cast<const unsigned long long*>(r) == 0ull) return;  // padding of the all-gather
  const unsigned long long t = *reinterpret_cast<const unsigned long long*>(r + 8) >> shift;
  if (flags)
    flags[t] = 1;
  else
    atomicOr(bits + (t >> 5), 1u << (t & 31));
}

__device__ __forceinline__ long long d_rank_of(unsigned long long t, const unsigned int* __restrict__ bits,
                                               const long long* __restrict__ prefix) {
  const unsigned int w = bits[t >> 5];
  return prefix[t >> 5] + (long long)__popc(w & ((1u << (t & 31)) - 1u));
}

// bitmap (s1) + exclusive prefix of the word popcounts (s5; s5[words] = number of set bits) over the
// global token space from the first-seen values of n_slots gathered records, n_total of them real
static int d_rank_bitmap(amg_ctx* c, const unsigned char* recs, long long n_slots, long long n_total, int rec_bytes,
                         int shift) {
  hipStream_t st = c->stream;
  const long long words = ((c->tok_total > 0 ? c->tok_total : 1) >> 5) + 2;
  AMGCHK(c->s1.ensure((size_t)words * sizeof(unsigned int)));
  AMGCHK(c->s5.ensure((size_t)(words + 2) * sizeof(long long)));
  const bool bytes = n_total * 64 > c->tok_total;
  unsigned char* flags = nullptr;
  ClearList cl;
  if (bytes) {
    AMGCHK(c->s0.ensure((size_t)words * 32 + 64));
    flags = c->s0.as<unsigned char>();
    cl.add(flags, (size_t)words * 32);
  } else {
    cl.add(c->s1.p, (size_t)words * sizeof(unsigned int));
  }
  AMGCHK(clear_many(c, cl));
  if (n_slots > 0)
    hipLaunchKernelGGL(k_d_setflags, dim3(nblk(n_slots, 256)), dim3(256), 0, st, recs, n_slots, rec_bytes, shift,
                       flags, c->s1.as<unsigned int>());
  // the scan folds the flag bytes into the bitmap words (or takes the words as they are) and counts them in one launch
  if (flags) return prim_exscan_flag_words(c, flags, c->s1.as<unsigned int>(), c->s5.as<long long>(), (size_t)words);
  return prim_exscan_bits_popc(c, c->s1.as<unsigned int>(), c->s5.as<long long>(), (size_t)words);
}

// node arrays in global id order, straight from the gathered records
__global__ void k_global_nodes(const unsigned char* __restrict__ recs, long long n_slots, int rec_bytes, int k,
                               const unsigned int* __restrict__ bits, const long long* __restrict__ prefix,
                               int* __restrict__ node_tokens, unsigned int* __restrict__ node_cov,
                               long long* __restrict__ node_first, unsigned char* __restrict__ node_alive) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_slots) return;
  const unsigned char* r = recs + (size_t)j * rec_bytes;
  if (*reinterpret_cast<const unsigned long long*>(r) == 0ull) return;
  const unsigned long long first = *reinterpret_cast<const unsigned long long*>(r + 8);
  const long long i = d_rank_of(first >> 1, bits, prefix);
  node_first[i] = (long long)first;
  node_cov[i] = *reinterpret_cast<const unsigned int*>(r + 16);
  node_alive[i] = 1;
  const int* tk = reinterpret_cast<const int*>(r + 24);
  for (int x = 0; x < k; ++x) node_tokens[i * k + x] = tk[x];
}

// local node (record j of what this rank sent) -> global node id through its owner's reply; -2
// when the node fell to the fused filter (its windows then read None)
__global__ void k_replies_to_claims(const unsigned long long* __restrict__ replies, long long n,
                                    const unsigned int* __restrict__ order, const unsigned int* __restrict__ bits,
                                    const long long* __restrict__ prefix, int* __restrict__ final_of_claim) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long g = replies[j];
  final_of_claim[order ? order[j] : (unsigned int)j] = g == REPLY_DROPPED ? -2 : (int)d_rank_of(g >> 1, bits, prefix);
}

__global__ void k_replies_to_slots(const unsigned long long* __restrict__ replies, long long n,
                                   const unsigned int* __restrict__ order, const unsigned int* __restrict__ bits,
                                   const long long* __restrict__ prefix, const unsigned int* __restrict__ slots,
                                   Slot* __restrict__ ltab, const int* __restrict__ node_tokens, int k, int packed) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long g = replies[j];
  Slot* s = ltab + slots[order ? order[j] : (unsigned int)j];
  const int gid = g == REPLY_DROPPED ? -2 : (int)d_rank_of(g >> 1, bits, prefix);
  if (!packed) {
    s->id = gid;
  } else if (gid >= 0) {
    slot_pack(s, gid, node_tokens + (long long)gid * k, k);
  } else {
    int none[AMG_MAX_K] = {0};
    slot_pack(s, -2, none, k);
  }
}

// all_records: n_slots record slots as the all-gather delivered them (equal-size contributions:
// the unused tail of a rank's part is zero), n_total of them real; my_replies: one word per record
// this rank sent, in the order it sent them
extern "C" int amg_dist_nodes_global(amg_ctx* c, const void* all_records, int64_t n_slots, int64_t n_total,
                                     const void* my_replies) {
  NEED_CTX(c);
  hipStream_t st = c->stream;
  const long long n = n_total;
  const int rb = (int)node_rec_bytes(c->k);
  if (n_slots < n || n < 0) return amg_fail(AMG_E_ARG, "bad record counts");
  if (n > 0 && !all_records) return amg_fail(AMG_E_ARG, "null records");
  if (c->n_local_nodes > 0 && !my_replies) return amg_fail(AMG_E_ARG, "null replies");
  stage_begin(c, "merge_node_global");
  c->packed_nodes = !c->dist_x && (c->two_v <= 65536 && c->k <= AMG_PACK_MAX_K);
  c->n_nodes = n;
  AMGCHK(bs_alloc_nodes(c, n));
  const unsigned char* recs = reinterpret_cast<const unsigned char*>(all_records);
  AMGCHK(d_rank_bitmap(c, recs, n_slots, n, rb, 1));
  const unsigned int* bits = c->s1.as<unsigned int>();
  const long long* prefix = c->s5.as<long long>();
  if (n_slots > 0)
    hipLaunchKernelGGL(k_global_nodes, dim3(nblk(n_slots, 256)), dim3(256), 0, st, recs, (long long)n_slots, rb, c->k,
                       bits, prefix, c->node_tokens.as<int>(), c->node_cov.as<unsigned int>(),
                       c->node_first.as<long long>(), c->node_alive.as<unsigned char>());
  const long long nl = c->n_local_nodes;
  const unsigned long long* rep = reinterpret_cast<const unsigned long long*>(my_replies);
  if (nl > 0 && c->dist_x)
    hipLaunchKernelGGL(k_replies_to_claims, dim3(nblk(nl, 256)), dim3(256), 0, st, rep, nl, send_order(c), bits,
                       prefix, c->x_final.as<int>());
  else if (nl > 0)
    hipLaunchKernelGGL(k_replies_to_slots, dim3(nblk(nl, 256)), dim3(256), 0, st, rep, nl, send_order(c), bits,
                       prefix, c->dist_slot.as<unsigned int>(), c->node_tab.as<Slot>(), c->node_tokens.as<int>(), c->k,
                       c->packed_nodes ? 1 : 0);
  // distinct first-seen values <=> as many bits as records
  const long long words = ((c->tok_total > 0 ? c->tok_total : 1) >> 5) + 2;
  unsigned long long set = 0;
  {
    FetchList l;
    l.add(prefix + words);
    AMGCHK(fetch(c, l, &set));
  }
  stage_end(c);
  if ((long long)set != n)
    return amg_fail(AMG_E_DIST, "global node table inconsistent: %lld records, %llu distinct first-seen positions", n, set);
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: edges
__global__ void k_dist_edge_pack(const unsigned int* __restrict__ order, long long n,
                                 const unsigned int* __restrict__ slots,
                                 const unsigned long long* __restrict__ firsts,
                                 const Slot* __restrict__ tab, const unsigned int* __restrict__ lcnt,
                                 unsigned char* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  unsigned int i = order ? order[j] : (unsigned int)j;
  const Slot* s = tab + slots[i];
  unsigned long long* q = reinterpret_cast<unsigned long long*>(out + (size_t)j * EDGE_REC_BYTES);
  q[0] = s->key;
  q[1] = firsts[i];
  q[2] = (unsigned long long)lcnt[s->id];
}

extern "C" int amg_dist_edges_local(amg_ctx* c, int32_t world, int64_t* send_counts) {
  NEED_CTX(c);
  if (world < 1 || world != c->world || !send_counts) return amg_fail(AMG_E_ARG, "bad world / send_counts");
  hipStream_t st = c->stream;
  if (c->dist_x) return edges_local_x(c, world, send_counts);
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bs_edges_pass(c, &which);
    if (r == AMG_OK) break;
    if (r != AMG_E_OVERFLOW || which != 2 || attempt >= 8) {
      if (which == 3)
        return amg_fail(AMG_E_COLLISION, "fingerprint collision: the merged build is repeated with the next seed");
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
  Bucketing b;
  AMGCHK(bucketing(c, n, &b));
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->dist_slot.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  HIPCHK(hipMemcpyAsync(c->dist_first.p, c->s1.p, (size_t)n * sizeof(unsigned long long),
                        hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->dist_slot.p, c->s3.p, (size_t)n * sizeof(unsigned int),
                        hipMemcpyDeviceToDevice, st));
  if (n > 0 && world > 1)
    hipLaunchKernelGGL(k_dist_dest, dim3(nblk(n, 256)), dim3(256), 0, st,
                       c->dist_slot.as<unsigned int>(), n, c->edge_tab.as<Slot>(), (unsigned int)world, b.dest, b.idx);
  return dest_counts(c, n, n, world, b, send_counts);
}

extern "C" int amg_dist_edges_pack(amg_ctx* c, void* send_buf) {
  NEED_CTX(c);
  const long long n = c->n_local_pairs;
  if (n == 0) return AMG_OK;
  if (!send_buf) return amg_fail(AMG_E_ARG, "null send buffer");
  const unsigned int* order = send_order(c);
  stage_begin(c, "merge_edge_pack");
  if (c->dist_x)
    hipLaunchKernelGGL(k_xd_edge_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                       c->edge_tab.as<Slot16>(), c->x_eslot.as<unsigned int>(), c->x_efirst.as<unsigned int>(),
                       (long long)c->tok_base,
                       c->dist_lcnt.as<unsigned int>(), reinterpret_cast<unsigned char*>(send_buf));
  else
    hipLaunchKernelGGL(k_dist_edge_pack, dim3(nblk(n, 256)), dim3(256), 0, c->stream, order, n,
                       c->dist_slot.as<unsigned int>(), c->dist_first.as<unsigned long long>(),
                       c->edge_tab.as<Slot>(), c->dist_lcnt.as<unsigned int>(),
                       reinterpret_cast<unsigned char*>(send_buf));
  stage_end(c);
  return AMG_OK;
}

extern "C" int amg_dist_edges_reduce(amg_ctx* c, const void* recv_buf, int64_t n_recv, int32_t n_sources,
                                     void* owned_out, int64_t* n_owned) {
  NEED_CTX(c);
  if (!n_owned || n_recv < 0) return amg_fail(AMG_E_ARG, "bad arguments");
  stage_begin(c, "merge_edge_reduce");
  const int r = reduce_records(c, recv_buf, n_recv, n_sources, EDGE_REC_BYTES, 0, c->dist_min_edge, owned_out,
                               nullptr, n_owned);
  stage_end(c);
  c->n_owned = *n_owned;
  return r;
}

// edge classes in first-seen order (the input of the edge emission), straight from the gathered records
__global__ void k_global_pairs(const unsigned char* __restrict__ recs, long long n_slots,
                               const unsigned int* __restrict__ bits, const long long* __restrict__ prefix,
                               unsigned long long* __restrict__ pkey, unsigned long long* __restrict__ pfirst,
                               unsigned int* __restrict__ pcnt) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_slots) return;
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(recs + (size_t)j * EDGE_REC_BYTES);
  if (q[0] == 0ull) return;
  const long long i = d_rank_of(q[1] >> 3, bits, prefix);
  pkey[i] = q[0];
  pfirst[i] = q[1];
  pcnt[i] = (unsigned int)q[2];
}

extern "C" int amg_dist_edges_global(amg_ctx* c, const void* all_records, int64_t n_slots, int64_t n_total) {
  NEED_CTX(c);
  hipStream_t st = c->stream;
  const long long n = n_total;
  if (n_slots < n || n < 0) return amg_fail(AMG_E_ARG, "bad record counts");
  if (n > 0 && !all_records) return amg_fail(AMG_E_ARG, "null records");
  stage_begin(c, "merge_edge_global");
  c->n_pairs = n;
  AMGCHK(bs_alloc_pairs(c, n));
  const unsigned char* recs = reinterpret_cast<const unsigned char*>(all_records);
  AMGCHK(d_rank_bitmap(c, recs, n_slots, n, EDGE_REC_BYTES, 3));
  if (n_slots > 0)
    hipLaunchKernelGGL(k_global_pairs, dim3(nblk(n_slots, 256)), dim3(256), 0, st, recs, (long long)n_slots,
                       c->s1.as<unsigned int>(), c->s5.as<long long>(), c->pair_key.as<unsigned long long>(),
                       c->pair_first.as<unsigned long long>(), c->pair_cnt.as<unsigned int>());
  const long long words = ((c->tok_total > 0 ? c->tok_total : 1) >> 5) + 2;
  unsigned long long set = 0;
  {
    FetchList l;
    l.add(c->s5.as<long long>() + words);
    AMGCHK(fetch(c, l, &set));
  }
  stage_end(c);
  if ((long long)set != n)
    return amg_fail(AMG_E_DIST, "global edge table inconsistent: %lld records, %llu distinct first-seen positions", n, set);
  AMGCHK(bs_finish_from_pairs(c));
  if (c->dist_min_node > 1)
    // fused filter: reads that lost a node join _readsToCorrect (remove_node_from_reads :442-461)
    AMGCHK(bx_flag_dead_reads(c));
  c->dist_min_node = c->dist_min_edge = 1;
  c->built = true;
  c->node_hint = c->n_local_nodes > 256 ? c->n_local_nodes : 256;
  return AMG_OK;
}
