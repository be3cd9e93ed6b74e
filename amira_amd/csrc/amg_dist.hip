// amg_dist.hip — read-sharded build with a key-owner table merge, driven from HERE.
//
// Every rank holds a contiguous shard of the reads.  The single-graph result (graph_utils.py:105-124 at cores = 1,
// i.e. GeneMerGraph over all reads) is obtained, for the nodes and then for the edge classes, in six steps:
//
//   local    the ordinary table pass on the shard (exact keys + claim ids when the tuple fits, else 32-byte fingerprint
//            slots), occurrence counts per local key, one 24-byte record {merge key, local first-seen, count} per
//            local key, bucketed by owner = hash(key) mod world; the per-peer record counts are written into a COUNT
//            MESSAGE on the device (with the shard's token count, the retry counter and — in place of the count — a
//            negative code when a device phase of this rank failed)                       --> all-to-all of the messages
//   counts   ONE read-back: what I send, what I receive, every shard's token count (first-seen values become global
//            token indices from here on), everybody's verdict; records in destination order --> all-to-all of records
//   reduce   owner side: equal keys meet in an open-addressing table of 16-byte slots {key, ~min first-seen}, counts in
//            a dense array by slot; every received record is answered with {the key's global first-seen | "dropped by
//            the fused filter", its total count}                                          --> all-to-all back
//   hold     the rank whose own first-seen IS the global one HOLDS the key: holders are ranked by their local
//            first-seen (a bitmap over the LOCAL tokens) and emit {first-seen, total, tuple} in that order; the
//            number held (or a failure code) goes into a second device-built message      --> all-gather of the messages
//   hcounts  ONE read-back: held records of every rank, everybody's verdict               --> all-gather of held records
//   global   shards are contiguous read ranges, so the gathered buffer — rank 0's held records, then rank 1's, ... —
//            IS the table in global first-seen order: id = records of the ranks before + index.  No bitmap over the
//            global token space, no sort, no scatter: a coalesced unpack.  A local key finds its id by binary search
//            of its reply among the (ascending) first-seen values and checks its tuple against the holder's (two
//            gene-mers under one 64-bit merge key: every rank repeats the build with the next seed).
//
// Two host waits per kind, four per merged build.  The exchanges are `amg_xfer`s: amg_dist_merge performs them with
// RCCL (ncclSend / ncclRecv groups and ncclAllGather on the ctx's stream; librccl is opened at amg_dist_init, not
// linked), amg_dist_merge_local with device copies between the ctxs of one process (emulated ranks: tests, the scaling
// model), and amg_dist_merge_begin / _next hand them to the caller (tests between processes over gloo).
#include <dlfcn.h>

#include <chrono>
#include <string>
#include <vector>

#include <rccl/rccl.h>  // types and prototypes only: the entry points are resolved with dlsym

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

#define REC_BYTES 24    // {u64 merge key, u64 first-seen, u32 count, u32 pad}: what travels to the owners (both kinds)
#define REPLY_WORDS 2   // {u64 global first-seen | REPLY_DROPPED, u64 total count}: what comes back per record
// held records are arrays of 32-bit words (they are the bytes of the all-gathers: 20 + 24 bytes per class + node of a
// rebuilt graph at k = 5, where 8-byte fields and padding made 24 + 40):
//   edge class: {key lo, key hi, first-seen lo, first-seen hi, count}
//   node:       {first-seen lo, first-seen hi, count, tokens: two per word while every token fits 16 bits, else one}
#define HELD_EDGE_BYTES 20
static inline bool held_tok16(int two_v) { return two_v <= 65536; }
static inline size_t held_node_bytes(int k, int two_v) { return (size_t)(4 * (3 + (held_tok16(two_v) ? (k + 1) / 2 : k))); }
__device__ __forceinline__ void held_put_tokens(unsigned int* w, const int* tok, int k, bool t16) {
  if (!t16) {
    for (int x = 0; x < k; ++x) w[x] = (unsigned int)tok[x];
    return;
  }
  for (int x = 0; x < k; x += 2)
    w[x >> 1] = ((unsigned int)tok[x] & 0xffffu) | (x + 1 < k ? ((unsigned int)tok[x + 1] << 16) : 0u);
}
#define REPLY_DROPPED (~0ull)
#define CNT_WORDS 4     // count message per peer: {records | code < 0, tokens of my shard, attempt, kind}
#define HC_WORDS 2      // held-count message: {records held | code < 0, attempt}
#define CODE_ERROR (-1ll)
#define CODE_COLLISION (-2ll)
#define ST_DIST_BAD 13  // status word: a reply that no gathered record answers (never expected; reported, not retried)
#define MAX_ATTEMPTS 4

__device__ __forceinline__ unsigned int owner_of(unsigned long long key, unsigned int world) {
  return (unsigned int)(mix64(key ^ 0x5851F42D4C957F2Dull) % world);
}

// ------------------------------------------------------------------ state of a ctx's merges
enum { S_IDLE = 0, S_LOCAL, S_COUNTS, S_REDUCE, S_HOLD, S_HCOUNTS, S_GLOBAL, S_DV_LOCAL, S_DV_ASK, S_DV_FILL, S_N };
static const char* const kPhaseNames[2 * S_N] = {
    "", "nodes_local", "nodes_counts_pack", "nodes_reduce", "nodes_hold", "nodes_hcounts", "nodes_global",
    "derive_local", "derive_ask", "derive_fill",
    "", "edges_local", "edges_counts_pack", "edges_reduce", "edges_hold", "edges_hcounts", "edges_global", "", "", ""};

struct DistState {
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  bool always_exchange = false;  // test hook: world 1 sends its records through the transport all the same
  // one merge
  int k = 0, attempt = 0, state = S_IDLE, kind = 0;
  uint32_t mn = 1, me = 1;
  int fail_ret = 0;  // a host-side failure of this rank waiting for the next count exchange
  std::string fail_msg;
  std::vector<int64_t> send_counts, recv_counts, held_counts, tokens, one, words;
  std::vector<int64_t> x_send, x_recv;  // the counts an amg_xfer points at (element counts per peer)
  int64_t n_send = 0, n_recv = 0, n_held = 0, m_pad = 0, n_total = 0;
  int n_sources = 0;
  bool exchanged = false;
  DevBuf cnt_send, cnt_recv, hc_send, hc_recv, offs;
  DevBuf send, recv, rep_out, rep_in, held, held_pad, gathered;
  // the rebuild that reuses the previous merged graph (amg_derive.hip; S_DV_*)
  bool dv_ok = false;
  long long dv_D2 = 0, dv_P2 = 0, dv_mN = 0, dv_mP = 0;
  std::vector<long long> dv_bases, dv_bounds;
  const void* gathered_p = nullptr;
  const void* recv_p = nullptr;    // the records this rank owns the keys of (one rank: what it packed)
  void* rep_out_p = nullptr;       // the answers to them
  const void* rep_in_p = nullptr;  // the answers to what this rank sent (one rank: the same array)
  // statistics (amg_dist_stats) and per-phase times (amg_dist_merge_local with timing on)
  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double phase_ms[2 * S_N] = {0};
  bool time_phases = false;
  int phase_now = -1;
  std::chrono::steady_clock::time_point phase_t0;
};

static DistState* dm(amg_ctx* c) {
  if (!c->dist) c->dist = new DistState();
  return c->dist;
}

// ------------------------------------------------------------------ RCCL, opened on demand
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
};
static Rccl g_rccl;

static int rccl_open() {
  if (g_rccl.h) return AMG_OK;
  // (a process that has imported torch already holds its librccl under this soname: the same library is reused)
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return amg_fail(AMG_E_DIST, "librccl.so.1 not found: %s", dlerror());
#define RSYM(name)                                                                          \
  g_rccl.name = reinterpret_cast<decltype(g_rccl.name)>(dlsym(h, "nccl" #name));            \
  if (!g_rccl.name) return amg_fail(AMG_E_DIST, "librccl lacks nccl" #name)
  RSYM(GetUniqueId);
  RSYM(CommInitRank);
  RSYM(CommDestroy);
  RSYM(GetErrorString);
  RSYM(GroupStart);
  RSYM(GroupEnd);
  RSYM(Send);
  RSYM(Recv);
  RSYM(AllGather);
#undef RSYM
  g_rccl.h = h;
  return AMG_OK;
}
#define NCCLCHK(call)                                                                                        \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess) return amg_fail(AMG_E_DIST, "%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
  } while (0)

extern "C" int amg_dist_unique_id(void* out, int32_t bytes) {
  if (!out || bytes < (int32_t)sizeof(ncclUniqueId)) return amg_fail(AMG_E_ARG, "amg_dist_unique_id: room for %d bytes", (int)sizeof(ncclUniqueId));
  AMGCHK(rccl_open());
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return AMG_OK;
}

void dist_release(amg_ctx* c) {
  DistState* d = c->dist;
  if (!d) return;
  if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
  DevBuf* all[] = {&d->cnt_send, &d->cnt_recv, &d->hc_send, &d->hc_recv, &d->offs, &d->send, &d->recv, &d->rep_out,
                   &d->rep_in, &d->held, &d->held_pad, &d->gathered};
  for (DevBuf* b : all) b->release();
  delete d;
  c->dist = nullptr;
}

static int set_world(amg_ctx* c, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) return amg_fail(AMG_E_ARG, "bad rank %d / world %d", rank, world);
  DistState* d = dm(c);
  if (d->state != S_IDLE) return amg_fail(AMG_E_STATE, "a merged build is under way");
  if (d->comm) {
    (void)g_rccl.CommDestroy(d->comm);
    d->comm = nullptr;
  }
  d->rank = rank;
  d->world = world;
  const char* e = getenv("AMG_DIST_ALWAYS_EXCHANGE");  // test hook
  d->always_exchange = e && e[0] == '1';
  return AMG_OK;
}

extern "C" int amg_dist_init(amg_ctx* c, const void* unique_id, int32_t rank, int32_t world) {
  NEED_CTX(c);
  if (!unique_id) return amg_fail(AMG_E_ARG, "null unique id");
  AMGCHK(rccl_open());
  AMGCHK(set_world(c, rank, world));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  NCCLCHK(g_rccl.CommInitRank(&dm(c)->comm, world, id, rank));
  return AMG_OK;
}

extern "C" int amg_dist_init_external(amg_ctx* c, int32_t rank, int32_t world) {
  NEED_CTX(c);
  return set_world(c, rank, world);
}

extern "C" int amg_dist_finalize(amg_ctx* c) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  dist_release(c);
  return AMG_OK;
}

extern "C" int amg_copy_d2h(amg_ctx* c, const void* device_ptr, void* host_ptr, int64_t bytes) {
  NEED_CTX(c);
  if (bytes < 0 || (bytes > 0 && (!device_ptr || !host_ptr))) return amg_fail(AMG_E_ARG, "bad copy");
  if (bytes) HIPCHK(hipMemcpyAsync(host_ptr, device_ptr, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

extern "C" int amg_copy_h2d(amg_ctx* c, void* device_ptr, const void* host_ptr, int64_t bytes) {
  NEED_CTX(c);
  if (bytes < 0 || (bytes > 0 && (!device_ptr || !host_ptr))) return amg_fail(AMG_E_ARG, "bad copy");
  if (bytes) HIPCHK(hipMemcpyAsync(device_ptr, host_ptr, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// device words for the host: the pinned mailbox while they fit one list, a copy otherwise
struct WordRange {
  const void* p;
  int n;
};
static int fetch_ranges(amg_ctx* c, const WordRange* r, int n_ranges, long long* out) {
  int total = 0;
  for (int i = 0; i < n_ranges; ++i) total += r[i].n;
  if (total <= FETCH_MAX) {
    FetchList l;
    for (int i = 0; i < n_ranges; ++i) l.add_words(r[i].p, r[i].n);
    return fetch(c, l, reinterpret_cast<unsigned long long*>(out));
  }
  long long* o = out;
  for (int i = 0; i < n_ranges; ++i) {
    HIPCHK(hipMemcpyAsync(o, r[i].p, (size_t)r[i].n * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    o += r[i].n;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: local tables -> records by destination
// destination of every local node (compaction list: first / slot) — fingerprint path
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

// the count message of a phase, one CNT_WORDS block per peer.  code != 0: this rank's phase failed on the host;
// otherwise the device's own status words are looked at (a reply nobody answers, a tuple that is not its key's)
__global__ void k_cnt_msg(const unsigned long long* __restrict__ counts, long long single_count, int world,
                          long long n_tokens, int attempt, int kind, long long code,
                          const unsigned long long* __restrict__ status, long long* __restrict__ msg) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= world) return;
  if (code == 0 && status[ST_DIST_BAD]) code = CODE_ERROR;
  if (code == 0 && status[ST_COLLISION]) code = CODE_COLLISION;
  long long* m = msg + (size_t)p * CNT_WORDS;
  m[0] = code ? code : (counts ? (long long)counts[p] : single_count);
  m[1] = n_tokens;
  m[2] = attempt;
  m[3] = kind;
}

__global__ void k_hc_msg(const long long* __restrict__ n_held, int attempt, const unsigned long long* __restrict__ status,
                         long long* __restrict__ msg) {
  long long code = 0;
  if (status[ST_OVERFLOW] || status[ST_DIST_BAD]) code = CODE_ERROR;
  msg[0] = code ? code : *n_held;
  msg[1] = attempt;
}

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

// exact-key shards: merge key and destination per claim
// (claim ids nobody took — shard counters leave holes — have first-seen 0: they get destination `world`, which sorts
// behind every rank and is never sent; `bucket`: destinations are wanted, i.e. world > 1 or there are holes)
__global__ void k_xd_node_keys(const Slot16* __restrict__ tab, const unsigned int* __restrict__ slot_by_claim,
                               const unsigned int* __restrict__ first2,
                               long long n, int k, int bits, int two, unsigned long long seed, unsigned long long key_mask,
                               unsigned int world, int bucket,
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
  unsigned long long key = tuple_fingerprint(tok, k, seed);  // (the same value as the fingerprint shards' slot keys)
  if (key_mask != ~0ull) key = (key & key_mask) | 1ull;      // test hook, see nodes_local
  keys[i] = key;
  if (bucket) {
    dest[i] = world > 1 ? owner_of(key, world) : 0u;
    idx[i] = (unsigned int)i;
  }
}

__global__ void k_xd_edge_dest(const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                               const unsigned int* __restrict__ first2, long long n, unsigned int world,
                               unsigned int* __restrict__ dest, unsigned int* __restrict__ idx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dest[i] = x_first_inv(first2, i) == 0u ? world : (world > 1 ? owner_of(etab[slot_by_claim[i]].w1, world) : 0u);
  idx[i] = (unsigned int)i;
}

// records in destination order.  order == nullptr: local order (one destination and no unclaimed ids in between)
// exact-key shards: key per claim from `keys` (nodes) or the class slot (edges), first-seen = base + local value
__global__ void k_xd_pack(const unsigned int* __restrict__ order, long long n, const unsigned long long* __restrict__ keys,
                          const Slot16* __restrict__ etab, const unsigned int* __restrict__ slot_by_claim,
                          const unsigned int* __restrict__ first2, unsigned long long base,
                          const unsigned int* __restrict__ lcnt, unsigned long long* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
  unsigned long long* q = out + 3 * j;
  q[0] = keys ? keys[c] : etab[slot_by_claim[c]].w1;
  q[1] = base + (unsigned long long)(unsigned int)~x_first_inv(first2, c);
  q[2] = (unsigned long long)lcnt[c];
}

// fingerprint shards: the compaction list (firsts / slots) in destination order
__global__ void k_fd_pack(const unsigned int* __restrict__ order, long long n, const unsigned int* __restrict__ slots,
                          const unsigned long long* __restrict__ firsts, unsigned long long base,
                          const Slot* __restrict__ tab, const unsigned int* __restrict__ lcnt,
                          unsigned long long* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int i = order ? order[j] : (unsigned int)j;
  const Slot* s = tab + slots[i];
  unsigned long long* q = out + 3 * j;
  q[0] = s->key;
  q[1] = firsts[i] + base;
  q[2] = (unsigned long long)lcnt[s->id];  // s->id is still the LOCAL first-seen rank here
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
// order in which the local records leave (nullptr: local order); the bucketing ran over c->dist_nspace ids
static const unsigned int* send_order(const amg_ctx* c) {
  return c->dist_sorted ? c->dist_a.as<unsigned int>() + 3 * (c->dist_nspace + 1) : nullptr;
}

// Records by destination without a sort: a histogram of the destinations (LDS per tile, one atomic per tile and bin), the
// bins' first places, and a scatter in which every tile reserves its stretch of each bin with one atomic.  The order of
// the records INSIDE a destination is whatever the atomics make it — owners sum counts and minimise first-seen values,
// and replies come back in the order the records left.  (A library radix sort of 5.4 M (destination, index) pairs was
// ~0.25 ms of a first build's node phase.)
#define BK_MAX 256
#define BK_PER 8
__global__ __launch_bounds__(256) void k_bucket_hist(const unsigned int* __restrict__ dest, long long n, int bins,
                                                     unsigned long long* __restrict__ counts) {
  __shared__ unsigned int h[BK_MAX];
  for (int b = threadIdx.x; b < bins; b += 256) h[b] = 0u;
  __syncthreads();
  const long long i0 = (long long)blockIdx.x * (256 * BK_PER) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < BK_PER; ++j) {
    const long long i = i0 + (long long)j * 256;
    if (i < n) atomicAdd(&h[dest[i]], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 256)
    if (h[b]) atomicAdd(&counts[b], (unsigned long long)h[b]);
}
__global__ void k_bucket_starts(const unsigned long long* __restrict__ counts, int bins, unsigned long long* __restrict__ cursor) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned long long s = 0;
  for (int b = 0; b < bins; ++b) {
    cursor[b] = s;
    s += counts[b];
  }
}
__global__ __launch_bounds__(256) void k_bucket_scatter(const unsigned int* __restrict__ dest, long long n, int bins,
                                                        unsigned long long* __restrict__ cursor, unsigned int* __restrict__ order) {
  __shared__ unsigned int h[BK_MAX];
  __shared__ unsigned long long base[BK_MAX];
  for (int b = threadIdx.x; b < bins; b += 256) h[b] = 0u;
  __syncthreads();
  const long long i0 = (long long)blockIdx.x * (256 * BK_PER) + threadIdx.x;
  unsigned int d[BK_PER], rank[BK_PER];
#pragma unroll
  for (int j = 0; j < BK_PER; ++j) {
    const long long i = i0 + (long long)j * 256;
    d[j] = 0u;
    rank[j] = 0u;
    if (i < n) {
      d[j] = dest[i];
      rank[j] = atomicAdd(&h[d[j]], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 256)
    if (h[b]) base[b] = atomicAdd(&cursor[b], (unsigned long long)h[b]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BK_PER; ++j) {
    const long long i = i0 + (long long)j * 256;
    if (i < n) order[base[d[j]] + rank[j]] = (unsigned int)i;
  }
}

// n ids bucketed (claim ids in use, holes included: destination `world`), n_real records among them: `order` lists the
// ids by destination and the per-destination counts stay in dist_cnt ON THE DEVICE (one rank without holes: nothing to do)
static int dest_counts(amg_ctx* c, long long n, long long n_real, int world, const Bucketing& b) {
  hipStream_t st = c->stream;
  c->dist_nspace = n;
  c->dist_sorted = world > 1 || n != n_real;
  if (!c->dist_sorted) return AMG_OK;
  const int bins = world + 1;
  AMGCHK(c->dist_cnt.ensure((size_t)(2 * bins + 2) * sizeof(unsigned long long)));
  unsigned long long* counts = c->dist_cnt.as<unsigned long long>();
  unsigned long long* cursor = counts + bins + 1;
  HIPCHK(hipMemsetAsync(counts, 0, (size_t)(2 * bins + 2) * sizeof(unsigned long long), st));
  if (n <= 0) return AMG_OK;
  if (bins <= BK_MAX) {
    hipLaunchKernelGGL(k_bucket_hist, dim3(nblk(n, 256 * BK_PER)), dim3(256), 0, st, b.dest, n, bins, counts);
    hipLaunchKernelGGL(k_bucket_starts, dim3(1), dim3(1), 0, st, counts, bins, cursor);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(nblk(n, 256 * BK_PER)), dim3(256), 0, st, b.dest, n, bins, cursor, b.order);
    return AMG_OK;
  }
  AMGCHK(prim_sort_u32_u32(c, b.dest, b.dest_sorted, b.idx, b.order, (size_t)n, ilog2_ceil((uint64_t)world + 1) + 1));
  hipLaunchKernelGGL(k_dest_counts, dim3(nblk(world, 64)), dim3(64), 0, st, b.dest_sorted, n, (unsigned int)world, counts);
  return AMG_OK;
}

static int nodes_local_x(amg_ctx* c, int k, int world, unsigned long long key_mask) {
  hipStream_t st = c->stream;
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bx_nodes_upsert(c, k, &which, true, false);  // (claims from the shard counters)
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
                       (long long)k * c->x_bits > 63 ? 1 : 0, c->seed, key_mask, (unsigned int)world,
                       (world > 1 || n != c->n_local_nodes) ? 1 : 0, c->dist_first.as<unsigned long long>(), b.dest, b.idx);
  const int r = dest_counts(c, n, c->n_local_nodes, world, b);
  stage_end(c);
  return r;
}

static int edges_local_x(amg_ctx* c, int world) {
  hipStream_t st = c->stream;
  for (int attempt = 0;; ++attempt) {
    int which = 0;
    int r = bx_edges_upsert(c, &which, false, true, false);
    if (r == AMG_OK) break;
    if (r == AMG_E_OVERFLOW && which == 3)
      return amg_fail(AMG_E_COLLISION, "two gene-mers share a merge key: the merged build is repeated with the next seed");
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
  const int r = dest_counts(c, n, c->n_local_pairs, world, b);
  stage_end(c);
  return r;
}

// the node pass of the shard; first-seen values stay LOCAL here (the shard's token base is learnt in the count
// exchange that follows) and become global when the records are packed
static int nodes_local(amg_ctx* c, DistState* d) {
  const int k = d->k, world = d->world;
  stages_reset(c);
  c->built = false;
  c->derive_ready = false;
  c->derived = false;
  c->have_corrected = false;
  c->match_valid = false;
  c->k = k;
  c->retries = 0;
  c->tok_base = 0;
  c->tok_total = c->n_tokens;
  c->world = world;
  c->dist_mode = true;
  c->comp_from_claims = false;
  c->dist_min_node = d->mn;
  c->dist_min_edge = d->me;
  // merge keys and key owners are fingerprints of this seed: every rank must use the SAME one, whatever collision
  // retries an earlier single-GPU build on this ctx went through; `attempt` is the ranks' common retry counter
  c->seed = kAmgSeed0;
  for (int a = 0; a < d->attempt; ++a) c->seed = c->seed * 6364136223846793005ull + 1442695040888963407ull;
  c->count_inline = false;  // local occurrence counts come from the per-window claims, not per-window atomics
  bs_size_tables(c);
  c->exact_keys = false;
  // test hooks.  AMG_TEST_DIST_FAIL=r: rank r's node phase fails (its peers must be told).  AMG_TEST_DIST_WEAK_KEYS=n:
  // the first n attempts cut the merge keys to 10 bits, so that gene-mers share them and the build has to be repeated
  if (const char* e = getenv("AMG_TEST_DIST_FAIL"))
    if (atoi(e) == d->rank) return amg_fail(AMG_E_STATE, "told to fail (AMG_TEST_DIST_FAIL)");
  bool weak = false;
  if (const char* e = getenv("AMG_TEST_DIST_WEAK_KEYS")) weak = atoi(e) > d->attempt;
  c->weak_fp_builds = 0;
  c->dist_x = bx_tuple_fits(c, k);  // (the held records carry the tuple: the slots must hold it)
  if (c->dist_x) return nodes_local_x(c, k, world, weak ? 0x3ffull : ~0ull);
  c->weak_fp_builds = weak ? 1 : 0;
  hipStream_t st = c->stream;
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
    int first_bits = ilog2_ceil((uint64_t)(c->n_tokens > 0 ? c->n_tokens : 1) * 2 + 2) + 1;
    AMGCHK(prim_sort_u64_u32(c, c->s1.as<unsigned long long>(), c->s2.as<unsigned long long>(),
                             c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)n, first_bits));
    AMGCHK(c->dist_lcnt.ensure((size_t)(n + 2) * sizeof(unsigned int)));
    AMGCHK(bs_count_by_slot(c, c->tok_slot.as<int>(), c->tok_node.as<int>(), c->n_tokens,
                            c->node_tab.as<Slot>(), c->s4.as<unsigned int>(), n,
                            c->dist_lcnt.as<unsigned int>(), 0));
  }
  Bucketing b;
  AMGCHK(bucketing(c, n, &b));
  // keep the compaction list in first-seen order (s2 / s4: the sort's output): the later sorts use the generic scratch
  AMGCHK(c->dist_first.ensure((size_t)(n + 1) * sizeof(unsigned long long)));
  AMGCHK(c->dist_slot.ensure((size_t)(n + 1) * sizeof(unsigned int)));
  HIPCHK(hipMemcpyAsync(c->dist_first.p, c->s2.p, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->dist_slot.p, c->s4.p, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToDevice, st));
  if (n > 0 && world > 1)
    hipLaunchKernelGGL(k_dist_dest, dim3(nblk(n, 256)), dim3(256), 0, st,
                       c->dist_slot.as<unsigned int>(), n, c->node_tab.as<Slot>(), (unsigned int)world, b.dest, b.idx);
  return dest_counts(c, n, n, world, b);
}

static int edges_local(amg_ctx* c, DistState* d) {
  const int world = d->world;
  hipStream_t st = c->stream;
  if (c->dist_x) return edges_local_x(c, world);
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
  HIPCHK(hipMemcpyAsync(c->dist_first.p, c->s2.p, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(c->dist_slot.p, c->s4.p, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToDevice, st));
  if (n > 0 && world > 1)
    hipLaunchKernelGGL(k_dist_dest, dim3(nblk(n, 256)), dim3(256), 0, st,
                       c->dist_slot.as<unsigned int>(), n, c->edge_tab.as<Slot>(), (unsigned int)world, b.dest, b.idx);
  return dest_counts(c, n, n, world, b);
}

// ------------------------------------------------------------------ phase: owner-side reduce
// Records of one key arrive from every rank that saw it.  They meet in an open-addressing table keyed by the merge key,
// one 32-byte slot = one sector per key.  The record that CREATES a slot (one compare-and-swap on the key) leaves its
// first-seen and count there with plain stores in fields of its own; only the records that FIND their key pay atomics
// (atomicMax on the complement of first-seen, atomicAdd on the count) in the slot's shared fields — nine keys in ten of
// an uncorrected read set come in one record.  Records that all come from ONE rank are distinct keys already: no
// table.  Every record is answered with its key's global first-seen and total, or "dropped" when the total stays
// below the fused filter's threshold.
struct __attribute__((aligned(32))) OSlot {
  unsigned long long key;
  unsigned long long first_inv;  // others: ~min first-seen (0: nobody but the creator)
  unsigned long long cfirst;     // creator's first-seen
  unsigned int cnt;              // others' counts
  unsigned int ccnt;             // creator's count
};
static_assert(sizeof(OSlot) == 32, "owner slot = one sector");

__device__ __forceinline__ bool edge_key_self_loop(unsigned long long key) {
  const unsigned int lo = (unsigned int)((key >> 32) & 0x7fffffffull);
  const unsigned int hi = (unsigned int)(key & 0xffffffffull) - 1u;
  return lo == hi;
}

__global__ void k_own_upsert(const unsigned long long* __restrict__ recs, long long n, OSlot* tab, unsigned long long mask,
                             unsigned int* __restrict__ recslot, unsigned long long* status) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long key = recs[3 * j], fi = ~recs[3 * j + 1];
  const unsigned int c = (unsigned int)recs[3 * j + 2];
  unsigned long long idx = mix64(key) & mask;
  for (unsigned int probes = 0;; ++probes) {
    OSlot* s = tab + idx;
    unsigned long long cur = ld_u64(&s->key);
    if (cur == 0ull) {
      cur = atomicCAS(&s->key, 0ull, key);
      if (cur == 0ull) {
        s->cfirst = ~fi;
        s->ccnt = c;
        recslot[j] = (unsigned int)idx;
        return;
      }
    }
    if (cur == key) {
      if (ld_u64(&s->first_inv) < fi) atomicMax(&s->first_inv, fi);
      atomicAdd(&s->cnt, c);
      recslot[j] = (unsigned int)idx;
      return;
    }
    if (probes >= (1u << 16)) {
      status[ST_OVERFLOW] = 6;
      recslot[j] = 0u;
      return;
    }
    idx = (idx + 1) & mask;
  }
}

template <bool MULTI>
__global__ void k_own_reply(const unsigned long long* __restrict__ recs, long long n, int is_edge, unsigned int min_cov,
                            const OSlot* __restrict__ tab, const unsigned int* __restrict__ recslot,
                            unsigned long long* __restrict__ replies) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long key = recs[3 * j];
  unsigned long long gfirst = recs[3 * j + 1];
  unsigned long long total = (unsigned int)recs[3 * j + 2];
  if (MULTI) {
    const OSlot s = tab[recslot[j]];
    const unsigned long long others = ~s.first_inv;  // (nobody but the creator: ~0)
    gfirst = s.cfirst < others ? s.cfirst : others;
    total = (unsigned long long)s.ccnt + s.cnt;
  }
  // (edge classes that are self-loops count twice)
  const unsigned long long cov = (is_edge && edge_key_self_loop(key)) ? total * 2 : total;
  replies[2 * j] = cov >= min_cov ? gfirst : REPLY_DROPPED;
  replies[2 * j + 1] = total;
}

static int reduce_records(amg_ctx* c, DistState* d, int is_edge) {
  hipStream_t st = c->stream;
  const long long n = d->n_recv;
  if (n == 0) return AMG_OK;
  const bool multi = d->n_sources > 1;
  const unsigned int min_cov = is_edge ? d->me : d->mn;
  const unsigned long long* recs = static_cast<const unsigned long long*>(d->recv_p);
  unsigned long long* replies = static_cast<unsigned long long*>(d->rep_out_p);
  unsigned long long* status = c->status.as<unsigned long long>();
  if (!multi) {
    hipLaunchKernelGGL(k_own_reply<false>, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, is_edge, min_cov,
                       (const OSlot*)nullptr, (const unsigned int*)nullptr, replies);
    return AMG_OK;
  }
  const uint64_t slots = pow2_at_least((uint64_t)n * 2 + 16);
  {
    ClearList cl;
    cl.add(c->dist_gtab.p, (size_t)slots * sizeof(OSlot));
    AMGCHK(clear_many(c, cl));
  }
  hipLaunchKernelGGL(k_own_upsert, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, c->dist_gtab.as<OSlot>(),
                     (unsigned long long)(slots - 1), c->s3.as<unsigned int>(), status);
  hipLaunchKernelGGL(k_own_reply<true>, dim3(nblk(n, 256)), dim3(256), 0, st, recs, n, is_edge, min_cov,
                     c->dist_gtab.as<OSlot>(), c->s3.as<unsigned int>(), replies);
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: hold
__device__ __forceinline__ long long d_rank_of(unsigned long long t, const unsigned int* __restrict__ bits,
                                               const long long* __restrict__ prefix) {
  const unsigned int w = bits[t >> 5];
  return prefix[t >> 5] + (long long)__popc(w & ((1u << (t & 31)) - 1u));
}

// exact-key shards: one flag byte per LOCAL token at the first-seen position of every claim this rank holds
__global__ void k_xh_flags(const unsigned long long* __restrict__ replies, long long n, const unsigned int* __restrict__ order,
                           const unsigned int* __restrict__ first2, unsigned long long base, int shift,
                           unsigned char* __restrict__ flags) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long g = replies[2 * j];
  if (g == REPLY_DROPPED) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
  const unsigned int local = ~x_first_inv(first2, c);
  if (base + (unsigned long long)local == g) flags[local >> shift] = 1;
}

__global__ void k_xh_emit_nodes(const unsigned long long* __restrict__ replies, long long n,
                                const unsigned int* __restrict__ order, const unsigned int* __restrict__ first2,
                                unsigned long long base, const unsigned int* __restrict__ bits,
                                const long long* __restrict__ prefix, const Slot16* __restrict__ tab,
                                const unsigned int* __restrict__ slot_by_claim, int k, int xbits, int two,
                                unsigned int* __restrict__ out, int rec_words, int t16) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long g = replies[2 * j];
  if (g == REPLY_DROPPED) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
  const unsigned int local = ~x_first_inv(first2, c);
  if (base + (unsigned long long)local != g) return;
  unsigned int* w = out + (size_t)d_rank_of(local >> 1, bits, prefix) * rec_words;
  w[0] = (unsigned int)g;
  w[1] = (unsigned int)(g >> 32);
  w[2] = (unsigned int)replies[2 * j + 1];
  const Slot16 s = tab[slot_by_claim[c]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;
  int tok[AMG_MAX_K];
  for (int x = 0; x < k; ++x) tok[x] = x_unpack(s.w1, tag, xbits, x);
  held_put_tokens(w + 3, tok, k, t16 != 0);
}

__global__ void k_xh_emit_edges(const unsigned long long* __restrict__ replies, long long n,
                                const unsigned int* __restrict__ order, const unsigned int* __restrict__ first2,
                                unsigned long long base, const unsigned int* __restrict__ bits,
                                const long long* __restrict__ prefix, const unsigned long long* __restrict__ sent,
                                unsigned int* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long g = replies[2 * j];
  if (g == REPLY_DROPPED) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
  const unsigned int local = ~x_first_inv(first2, c);
  if (base + (unsigned long long)local != g) return;
  unsigned int* w = out + 5 * d_rank_of(local >> 3, bits, prefix);
  const unsigned long long key = sent[3 * j];
  w[0] = (unsigned int)key;
  w[1] = (unsigned int)(key >> 32);
  w[2] = (unsigned int)g;
  w[3] = (unsigned int)(g >> 32);
  w[4] = (unsigned int)replies[2 * j + 1];
}

// fingerprint shards: the compaction list is in local first-seen order already — a flag per entry, a scan
__global__ void k_fh_flags(const unsigned long long* __restrict__ replies, long long n, const unsigned int* __restrict__ order,
                           const unsigned long long* __restrict__ firsts, unsigned long long base,
                           unsigned int* __restrict__ flag) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int i = order ? order[j] : (unsigned int)j;
  const unsigned long long g = replies[2 * j];
  flag[i] = (g != REPLY_DROPPED && firsts[i] + base == g) ? 1u : 0u;
}

__global__ void k_fh_emit_nodes(const unsigned long long* __restrict__ replies, long long n,
                                const unsigned int* __restrict__ order, const unsigned long long* __restrict__ firsts,
                                const unsigned int* __restrict__ flag, const long long* __restrict__ pos,
                                const int* __restrict__ tokens, int k, int two_v, unsigned int* __restrict__ out,
                                int rec_words, int t16) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int i = order ? order[j] : (unsigned int)j;
  if (!flag[i]) return;
  unsigned int* w = out + (size_t)pos[i] * rec_words;
  const unsigned long long g = replies[2 * j];
  w[0] = (unsigned int)g;
  w[1] = (unsigned int)(g >> 32);
  w[2] = (unsigned int)replies[2 * j + 1];
  const unsigned long long first = firsts[i];  // local: the node pass ran with token base 0
  const long long t = (long long)(first >> 1);
  const int dir = (first & 1ull) ? -1 : 1;
  const int flip = two_v - 1;
  int tok[AMG_MAX_K];
  for (int x = 0; x < k; ++x) tok[x] = dir > 0 ? tokens[t + x] : flip - tokens[t + k - 1 - x];
  held_put_tokens(w + 3, tok, k, t16 != 0);
}

__global__ void k_fh_emit_edges(const unsigned long long* __restrict__ replies, long long n,
                                const unsigned int* __restrict__ order, const unsigned int* __restrict__ flag,
                                const long long* __restrict__ pos, const unsigned long long* __restrict__ sent,
                                unsigned int* __restrict__ out) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int i = order ? order[j] : (unsigned int)j;
  if (!flag[i]) return;
  unsigned int* w = out + 5 * pos[i];
  const unsigned long long key = sent[3 * j], g = replies[2 * j];
  w[0] = (unsigned int)key;
  w[1] = (unsigned int)(key >> 32);
  w[2] = (unsigned int)g;
  w[3] = (unsigned int)(g >> 32);
  w[4] = (unsigned int)replies[2 * j + 1];
}

// holders ranked, held records emitted in local first-seen order, the number held in d->hc_send
static int hold_records(amg_ctx* c, DistState* d, int is_edge) {
  hipStream_t st = c->stream;
  const long long n = d->n_send, T = c->n_tokens;
  const unsigned long long* rep = static_cast<const unsigned long long*>(d->rep_in_p);
  const unsigned int* order = send_order(c);
  const int shift = is_edge ? 3 : 1;
  const unsigned long long base = (unsigned long long)c->tok_base << shift;
  const int rb = is_edge ? HELD_EDGE_BYTES : (int)held_node_bytes(c->k, c->two_v);
  unsigned long long* status = c->status.as<unsigned long long>();
  const long long* n_held = nullptr;
  if (c->dist_x) {
    const long long words = (T >> 5) + 2;
    AMGCHK(c->s0.ensure((size_t)words * 32 + 64));
    AMGCHK(c->s1.ensure((size_t)words * sizeof(unsigned int)));
    AMGCHK(c->s5.ensure((size_t)(words + 2) * sizeof(long long)));
    c->rank_flags_clean = 0;
    {
      ClearList cl;
      cl.add(c->s0.p, (size_t)words * 32);
      AMGCHK(clear_many(c, cl));
    }
    const unsigned int* first2 = is_edge ? c->x_efirst.as<unsigned int>() : c->x_first.as<unsigned int>();
    if (n > 0)
      hipLaunchKernelGGL(k_xh_flags, dim3(nblk(n, 256)), dim3(256), 0, st, rep, n, order, first2, base, shift,
                         c->s0.as<unsigned char>());
    AMGCHK(prim_exscan_flag_words(c, c->s0.as<unsigned char>(), c->s1.as<unsigned int>(), c->s5.as<long long>(), (size_t)words));
    if (n > 0 && !is_edge)
      hipLaunchKernelGGL(k_xh_emit_nodes, dim3(nblk(n, 256)), dim3(256), 0, st, rep, n, order, first2, base,
                         c->s1.as<unsigned int>(), c->s5.as<long long>(), c->node_tab.as<Slot16>(),
                         c->x_slot.as<unsigned int>(), c->k, c->x_bits, (long long)c->k * c->x_bits > 63 ? 1 : 0,
                         d->held.as<unsigned int>(), rb / 4, held_tok16(c->two_v) ? 1 : 0);
    else if (n > 0)
      hipLaunchKernelGGL(k_xh_emit_edges, dim3(nblk(n, 256)), dim3(256), 0, st, rep, n, order, first2, base,
                         c->s1.as<unsigned int>(), c->s5.as<long long>(), d->send.as<unsigned long long>(),
                         d->held.as<unsigned int>());
    n_held = c->s5.as<long long>() + words;
  } else {
    AMGCHK(c->s4.ensure((size_t)(n + 2) * sizeof(unsigned int)));
    AMGCHK(c->s5.ensure((size_t)(n + 2) * sizeof(long long)));
    unsigned int* flag = c->s4.as<unsigned int>();
    long long* pos = c->s5.as<long long>();
    HIPCHK(hipMemsetAsync(flag + n, 0, sizeof(unsigned int), st));
    // (nodes: the list holds local first-seen values; edge classes were made after the token base was known)
    const unsigned long long add = is_edge ? 0ull : base;
    if (n > 0)
      hipLaunchKernelGGL(k_fh_flags, dim3(nblk(n, 256)), dim3(256), 0, st, rep, n, order,
                         c->dist_first.as<unsigned long long>(), add, flag);
    AMGCHK(prim_exscan_u32_to_i64(c, flag, pos, (size_t)n + 1));
    if (n > 0 && !is_edge)
      hipLaunchKernelGGL(k_fh_emit_nodes, dim3(nblk(n, 256)), dim3(256), 0, st, rep, n, order,
                         c->dist_first.as<unsigned long long>(), flag, pos, c->tokens.as<int>(), c->k, c->two_v,
                         d->held.as<unsigned int>(), rb / 4, held_tok16(c->two_v) ? 1 : 0);
    else if (n > 0)
      hipLaunchKernelGGL(k_fh_emit_edges, dim3(nblk(n, 256)), dim3(256), 0, st, rep, n, order, flag, pos,
                         d->send.as<unsigned long long>(), d->held.as<unsigned int>());
    n_held = pos + n;
  }
  hipLaunchKernelGGL(k_hc_msg, dim3(1), dim3(1), 0, st, n_held, d->attempt, status, d->hc_send.as<long long>());
  return AMG_OK;
}

// ------------------------------------------------------------------ phase: global ids
// off[r] = held records of the ranks before r (off[world] = all of them)
__global__ void k_offs(const long long* __restrict__ hc, int world, long long* __restrict__ off) {
  if (threadIdx.x || blockIdx.x) return;
  long long s = 0;
  for (int r = 0; r < world; ++r) {
    off[r] = s;
    s += hc[(size_t)r * HC_WORDS] > 0 ? hc[(size_t)r * HC_WORDS] : 0;
  }
  off[world] = s;
}

// node arrays in global id order: the gathered buffer (world parts of m record slots) unpacked
__global__ void k_global_nodes(const unsigned int* __restrict__ recs, long long m, int world,
                               const long long* __restrict__ off, int rec_words, int k, int t16, int* __restrict__ node_tokens,
                               unsigned int* __restrict__ node_cov, long long* __restrict__ node_first,
                               unsigned char* __restrict__ node_alive) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * world) return;
  const int r = (int)(i / m);
  const long long j = i - (long long)r * m;
  if (j >= off[r + 1] - off[r]) return;
  const long long id = off[r] + j;
  const unsigned int* w = recs + (size_t)i * rec_words;
  node_first[id] = (long long)((unsigned long long)w[0] | ((unsigned long long)w[1] << 32));
  node_cov[id] = w[2];
  node_alive[id] = 1;
  for (int x = 0; x < k; ++x)
    node_tokens[id * k + x] = t16 ? (int)((w[3 + (x >> 1)] >> ((x & 1) * 16)) & 0xffffu) : (int)w[3 + x];
}

__global__ void k_global_pairs(const unsigned int* __restrict__ recs, long long m, int world,
                               const long long* __restrict__ off, unsigned long long* __restrict__ pkey,
                               unsigned long long* __restrict__ pfirst, unsigned int* __restrict__ pcnt) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * world) return;
  const int r = (int)(i / m);
  const long long j = i - (long long)r * m;
  if (j >= off[r + 1] - off[r]) return;
  const long long id = off[r] + j;
  const unsigned int* w = recs + 5 * i;
  pkey[id] = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
  pfirst[id] = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
  pcnt[id] = w[4];
}

// id of the node whose first-seen value is g: the node arrays are in ascending first-seen order
__device__ __forceinline__ long long id_of_first(const long long* __restrict__ node_first, long long n, unsigned long long g) {
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((unsigned long long)node_first[mid] < g) lo = mid + 1; else hi = mid;
  }
  return (lo < n && (unsigned long long)node_first[lo] == g) ? lo : -1;
}

// local node (record j of what this rank sent) -> global node id through its owner's reply; -2 when the node fell to
// the fused filter (its windows then read None).  The tuple of the local key must be the holder's.
__global__ void k_map_claims(const unsigned long long* __restrict__ replies, long long n, const unsigned int* __restrict__ order,
                             const long long* __restrict__ node_first, long long n_nodes,
                             const int* __restrict__ node_tokens, const Slot16* __restrict__ tab,
                             const unsigned int* __restrict__ slot_by_claim, int k, int xbits, int two,
                             int* __restrict__ final_of_claim, unsigned long long* status) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned int c = order ? order[j] : (unsigned int)j;
  const unsigned long long g = replies[2 * j];
  if (g == REPLY_DROPPED) {
    final_of_claim[c] = -2;
    return;
  }
  const long long id = id_of_first(node_first, n_nodes, g);
  if (id < 0) {
    status[ST_DIST_BAD] = 1;
    final_of_claim[c] = -2;
    return;
  }
  const Slot16 s = tab[slot_by_claim[c]];
  const unsigned int tag = two ? (unsigned int)(s.w2 >> 32) : 0u;
  for (int x = 0; x < k; ++x)
    if (x_unpack(s.w1, tag, xbits, x) != node_tokens[id * k + x]) status[ST_COLLISION] = 1;
  final_of_claim[c] = (int)id;
}

// fingerprint shards: the id goes into the local slot (the edge pass verifies every window's tuple against it)
__global__ void k_map_slots(const unsigned long long* __restrict__ replies, long long n, const unsigned int* __restrict__ order,
                            const long long* __restrict__ node_first, long long n_nodes,
                            const unsigned int* __restrict__ slots, Slot* __restrict__ ltab,
                            const int* __restrict__ node_tokens, int k, int packed, unsigned long long* status) {
  long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long g = replies[2 * j];
  Slot* s = ltab + slots[order ? order[j] : (unsigned int)j];
  long long id = -2;
  if (g != REPLY_DROPPED) {
    id = id_of_first(node_first, n_nodes, g);
    if (id < 0) {
      status[ST_DIST_BAD] = 1;
      id = -2;
    }
  }
  if (!packed) {
    s->id = (int)id;
  } else if (id >= 0) {
    slot_pack(s, (int)id, node_tokens + id * k, k);
  } else {
    int none[AMG_MAX_K] = {0};
    slot_pack(s, -2, none, k);
  }
}

static int nodes_global(amg_ctx* c, DistState* d) {
  hipStream_t st = c->stream;
  const long long n = d->n_total, m = d->m_pad;
  const int rb = (int)held_node_bytes(c->k, c->two_v);
  stage_begin(c, "merge_node_global");
  c->packed_nodes = !c->dist_x && (c->two_v <= 65536 && c->k <= AMG_PACK_MAX_K);
  c->n_nodes = n;
  AMGCHK(bs_alloc_nodes(c, n));
  if (m > 0)
    hipLaunchKernelGGL(k_global_nodes, dim3(nblk(m * d->world, 256)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned int*>(d->gathered_p), m, d->world, d->offs.as<long long>(), rb / 4,
                       c->k, held_tok16(c->two_v) ? 1 : 0, c->node_tokens.as<int>(), c->node_cov.as<unsigned int>(),
                       c->node_first.as<long long>(), c->node_alive.as<unsigned char>());
  const long long nl = d->n_send;
  const unsigned long long* rep = static_cast<const unsigned long long*>(d->rep_in_p);
  unsigned long long* status = c->status.as<unsigned long long>();
  if (nl > 0 && c->dist_x)
    hipLaunchKernelGGL(k_map_claims, dim3(nblk(nl, 256)), dim3(256), 0, st, rep, nl, send_order(c),
                       c->node_first.as<long long>(), n, c->node_tokens.as<int>(), c->node_tab.as<Slot16>(),
                       c->x_slot.as<unsigned int>(), c->k, c->x_bits, (long long)c->k * c->x_bits > 63 ? 1 : 0,
                       c->x_final.as<int>(), status);
  else if (nl > 0)
    hipLaunchKernelGGL(k_map_slots, dim3(nblk(nl, 256)), dim3(256), 0, st, rep, nl, send_order(c),
                       c->node_first.as<long long>(), n, c->dist_slot.as<unsigned int>(), c->node_tab.as<Slot>(),
                       c->node_tokens.as<int>(), c->k, c->packed_nodes ? 1 : 0, status);
  stage_end(c);
  return AMG_OK;
}

static int edges_global(amg_ctx* c, DistState* d) {
  hipStream_t st = c->stream;
  const long long n = d->n_total, m = d->m_pad;
  stage_begin(c, "merge_edge_global");
  c->n_pairs = n;
  AMGCHK(bs_alloc_pairs(c, n));
  if (m > 0)
    hipLaunchKernelGGL(k_global_pairs, dim3(nblk(m * d->world, 256)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned int*>(d->gathered_p), m, d->world, d->offs.as<long long>(),
                       c->pair_key.as<unsigned long long>(), c->pair_first.as<unsigned long long>(),
                       c->pair_cnt.as<unsigned int>());
  stage_end(c);
  AMGCHK(bs_finish_from_pairs(c));
  if (c->dist_min_node > 1)
    // fused filter: reads that lost a node join _readsToCorrect (remove_node_from_reads :442-461)
    AMGCHK(bx_flag_dead_reads(c));
  c->dist_min_node = c->dist_min_edge = 1;
  c->built = true;
  c->node_hint = c->n_local_nodes > 256 ? c->n_local_nodes : 256;
  return AMG_OK;
}

// ------------------------------------------------------------------ the merged rebuild that reuses the previous one
// (amg_derive.hip: when NO rank re-threaded a read the graph every rank holds, restricted to its live nodes, is the
// graph of the corrected reads — every rank squeezes its copy; what only the rank that holds a first occurrence knows,
// its token index in the NEW reads, travels in one all-gather: two exchanges and one host wait instead of ten and four)
__global__ void k_dv_msg(long long ok, long long n_tokens, long long attempt, long long n_nodes, long long* __restrict__ msg) {
  msg[0] = ok;
  msg[1] = n_tokens;
  msg[2] = attempt;
  msg[3] = n_nodes;
}
// this rank's part of the all-gather: the first-seen values of the nodes, then of the classes, first seen on its shard
// (local token indices so far: the shard's new first token is added), each padded to the largest part
__global__ void k_dv_contrib(const long long* __restrict__ nfirst, long long n_n, long long add_n,
                             const unsigned long long* __restrict__ pfirst, long long n_p, unsigned long long add_p,
                             long long m_n, long long m_p, unsigned long long* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_n) out[i] = (unsigned long long)(nfirst[i] + add_n);
  else if (i >= m_n && i - m_n < n_p) out[i] = pfirst[i - m_n] + add_p;
}
__global__ void k_dv_fill(const unsigned long long* __restrict__ all, long long m_n, long long m_p, int world,
                          const long long* __restrict__ bounds, long long* __restrict__ nfirst,
                          unsigned long long* __restrict__ pfirst) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long m = m_n + m_p;
  if (i >= m * world) return;
  const int r = (int)(i / m);
  const long long j = i - (long long)r * m;
  if (j < m_n) {
    if (j < bounds[r + 1] - bounds[r]) nfirst[bounds[r] + j] = (long long)all[i];
  } else {
    const long long* pb = bounds + world + 1;
    const long long q = j - m_n;
    if (q < pb[r + 1] - pb[r]) pfirst[pb[r] + q] = all[i];
  }
}

// ------------------------------------------------------------------ the driver
static long long n_local_of(const amg_ctx* c, int kind) { return kind ? c->n_local_pairs : c->n_local_nodes; }

static void note_fail(DistState* d, int ret) {
  d->fail_ret = ret;
  d->fail_msg = g_amg_err;
}

// what every rank saw in a count exchange decides what every rank does: 0 go on, 1 repeat the build with the next
// seed (all failures were merge-key collisions), < 0 this rank's return code
static int verdict(DistState* d, const long long* codes, int stride, const long long* attempts) {
  bool any = false, all_collisions = true;
  std::string who;
  for (int r = 0; r < d->world; ++r) {
    const long long v = codes[(size_t)r * stride];
    if (attempts && v >= 0 && attempts[(size_t)r * stride] != d->attempt)
      return amg_fail(AMG_E_DIST, "rank %d is at attempt %lld of the merged build, this rank at %d", r,
                      attempts[(size_t)r * stride], d->attempt);
    if (v >= 0) continue;
    any = true;
    if (v != CODE_COLLISION) all_collisions = false;
    who += (who.empty() ? "" : ", ") + std::to_string(r);
  }
  if (!any) return 0;
  if (all_collisions) {
    if (d->attempt + 1 < MAX_ATTEMPTS) return 1;
    return amg_fail(AMG_E_COLLISION, "merge keys still collide after %d seeds", MAX_ATTEMPTS);
  }
  if (d->fail_ret && d->fail_ret != AMG_E_COLLISION) {  // this rank's own error, as it was reported
    g_amg_err = d->fail_msg;
    return d->fail_ret;
  }
  if (codes[(size_t)d->rank * stride] < 0 && codes[(size_t)d->rank * stride] != CODE_COLLISION)
    return amg_fail(AMG_E_DIST, "merged build inconsistent on this rank (a reply without a record, or the owner table full)");
  return amg_fail(AMG_E_DIST, "merged build abandoned: device phase failed on rank(s) %s", who.c_str());
}

static void restart(DistState* d) {
  ++d->attempt;
  d->kind = 0;
  d->state = S_LOCAL;
  d->fail_ret = 0;
}

static void xfer_a2a(DistState* d, amg_xfer* x, const void* send, void* recv, int elem_bytes, const std::vector<int64_t>& sc,
                     const std::vector<int64_t>& rc, int stat) {
  d->x_send = sc;
  d->x_recv = rc;
  x->kind = AMG_XFER_ALL_TO_ALL;
  x->elem_bytes = elem_bytes;
  x->send = send;
  x->recv = recv;
  x->send_counts = d->x_send.data();
  x->recv_counts = d->x_recv.data();
  x->count = 0;
  ++d->st[1];
  if (stat >= 0) {
    int64_t most = 0, sum = 0;
    for (int p = 0; p < d->world; ++p)
      if (p != d->rank) {
        most = sc[p] > most ? sc[p] : most;
        sum += sc[p];
      }
    d->st[stat] += most * elem_bytes;
    if (stat == 2) d->st[6] += sum * elem_bytes;
  }
}

static void xfer_ag(DistState* d, amg_xfer* x, const void* send, void* recv, int elem_bytes, int64_t count, bool stat) {
  x->kind = AMG_XFER_ALL_GATHER;
  x->elem_bytes = elem_bytes;
  x->send = send;
  x->recv = recv;
  x->send_counts = x->recv_counts = nullptr;
  x->count = count;
  ++d->st[1];
  if (stat) d->st[4] += count * elem_bytes;
}

// per-phase clock of the driver (amg_dist_phase_ms): the phase that ends is booked, `next` begins (-1: none)
static void phase_tick(amg_ctx* c, DistState* d, int next) {
  if (!d->time_phases) return;
  (void)hipStreamSynchronize(c->stream);
  const auto now = std::chrono::steady_clock::now();
  if (d->phase_now >= 0) d->phase_ms[d->phase_now] += std::chrono::duration<double, std::milli>(now - d->phase_t0).count();
  d->phase_now = next;
  d->phase_t0 = now;
}

// runs the machine up to its next exchange.  1: *x is to be performed, then call again; 0: the build is complete
static int advance(amg_ctx* c, amg_xfer* x) {
  DistState* d = dm(c);
  hipStream_t st = c->stream;
  const int W = d->world;
  const bool wire = W > 1 || d->always_exchange;
  unsigned long long* status = c->status.as<unsigned long long>();
  for (;;) {
    const int is_edge = d->kind;
    phase_tick(c, d, d->kind * S_N + d->state);
    switch (d->state) {
      case S_LOCAL: {
        const int r = is_edge ? edges_local(c, d) : nodes_local(c, d);
        if (r != AMG_OK) {
          if (!wire) {
            if (r == AMG_E_COLLISION && d->attempt + 1 < MAX_ATTEMPTS) {
              restart(d);
              continue;
            }
            d->state = S_IDLE;
            return r;
          }
          note_fail(d, r);
        }
        d->n_send = d->fail_ret ? 0 : n_local_of(c, is_edge);
        if (!wire) {
          d->send_counts.assign(1, d->n_send);
          d->recv_counts = d->send_counts;
          d->tokens.assign(1, c->n_tokens);
          d->state = S_COUNTS;
          continue;
        }
        AMGCHK(d->cnt_send.ensure((size_t)W * CNT_WORDS * sizeof(long long)));
        AMGCHK(d->cnt_recv.ensure((size_t)W * CNT_WORDS * sizeof(long long)));
        const long long code = !d->fail_ret ? 0 : (d->fail_ret == AMG_E_COLLISION ? CODE_COLLISION : CODE_ERROR);
        hipLaunchKernelGGL(k_cnt_msg, dim3(nblk(W, 64)), dim3(64), 0, st,
                           (W > 1 && !d->fail_ret) ? c->dist_cnt.as<unsigned long long>() : (const unsigned long long*)nullptr,
                           (long long)d->n_send, W, (long long)c->n_tokens, d->attempt, is_edge, code, status,
                           d->cnt_send.as<long long>());
        d->one.assign(W, 1);
        xfer_a2a(d, x, d->cnt_send.p, d->cnt_recv.p, CNT_WORDS * (int)sizeof(long long), d->one, d->one, -1);
        d->state = S_COUNTS;
        return 1;
      }
      case S_COUNTS: {
        if (wire) {
          d->words.assign((size_t)W * CNT_WORDS * 2, 0);
          const WordRange rr[2] = {{d->cnt_recv.p, W * CNT_WORDS}, {d->cnt_send.p, W * CNT_WORDS}};
          AMGCHK(fetch_ranges(c, rr, 2, reinterpret_cast<long long*>(d->words.data())));
          ++d->st[0];
          const long long* got = reinterpret_cast<const long long*>(d->words.data());
          const long long* sent = got + (size_t)W * CNT_WORDS;
          const int v = verdict(d, got, CNT_WORDS, got + 2);
          if (v == 1) {
            restart(d);
            continue;
          }
          if (v < 0) {
            d->state = S_IDLE;
            return v;
          }
          d->send_counts.resize(W);
          d->recv_counts.resize(W);
          d->tokens.resize(W);
          for (int p = 0; p < W; ++p) {
            d->recv_counts[p] = got[(size_t)p * CNT_WORDS];
            d->send_counts[p] = sent[(size_t)p * CNT_WORDS];
            d->tokens[p] = got[(size_t)p * CNT_WORDS + 1];
            if (got[(size_t)p * CNT_WORDS + 3] != is_edge) {
              d->state = S_IDLE;
              return amg_fail(AMG_E_DIST, "rank %d is in another phase of the merged build", p);
            }
          }
        }
        if (!is_edge) {  // first-seen values are global token indices from here on
          long long base = 0, total = 0;
          for (int p = 0; p < W; ++p) {
            if (p < d->rank) base += d->tokens[p];
            total += d->tokens[p];
          }
          if (total >= (1ll << 60)) return amg_fail(AMG_E_ARG, "too many tokens");
          c->tok_base = base;
          c->tok_total = total;
        }
        d->n_recv = 0;
        d->n_sources = 0;
        long long n_send = 0;
        for (int p = 0; p < W; ++p) {
          d->n_recv += d->recv_counts[p];
          n_send += d->send_counts[p];
          if (d->recv_counts[p] > 0) ++d->n_sources;
        }
        if (n_send != d->n_send) {
          d->state = S_IDLE;
          return amg_fail(AMG_E_DIST, "%lld local records but %lld destinations", (long long)d->n_send, n_send);
        }
        // every buffer up to the next count exchange is made here: nothing between two exchanges fails for want of memory
        const int hb = is_edge ? HELD_EDGE_BYTES : (int)held_node_bytes(c->k, c->two_v);
        AMGCHK(d->send.ensure((size_t)(d->n_send + 1) * REC_BYTES));
        AMGCHK(d->rep_in.ensure((size_t)(d->n_send + 1) * REPLY_WORDS * sizeof(long long)));
        AMGCHK(d->rep_out.ensure((size_t)(d->n_recv + 1) * REPLY_WORDS * sizeof(long long)));
        AMGCHK(d->held.ensure((size_t)(d->n_send + 1) * hb));
        AMGCHK(d->hc_send.ensure(HC_WORDS * sizeof(long long)));
        AMGCHK(d->hc_recv.ensure((size_t)W * HC_WORDS * sizeof(long long)));
        AMGCHK(d->offs.ensure((size_t)(W + 2) * sizeof(long long)));
        if (wire) AMGCHK(d->recv.ensure((size_t)(d->n_recv + 1) * REC_BYTES));
        if (d->n_sources > 1) {
          const uint64_t slots = pow2_at_least((uint64_t)d->n_recv * 2 + 16);
          AMGCHK(c->dist_gtab.ensure((size_t)slots * sizeof(OSlot)));
          AMGCHK(c->s3.ensure((size_t)(d->n_recv + 1) * sizeof(unsigned int)));
        }
        stage_begin(c, is_edge ? "merge_edge_pack" : "merge_node_pack");
        if (d->n_send > 0) {
          const unsigned int* order = send_order(c);
          const int shift = is_edge ? 3 : 1;
          const unsigned long long base = (unsigned long long)c->tok_base << shift;
          if (c->dist_x)
            hipLaunchKernelGGL(k_xd_pack, dim3(nblk(d->n_send, 256)), dim3(256), 0, st, order, (long long)d->n_send,
                               is_edge ? (const unsigned long long*)nullptr : c->dist_first.as<unsigned long long>(),
                               c->edge_tab.as<Slot16>(), c->x_eslot.as<unsigned int>(),
                               is_edge ? c->x_efirst.as<unsigned int>() : c->x_first.as<unsigned int>(), base,
                               c->dist_lcnt.as<unsigned int>(), d->send.as<unsigned long long>());
          else
            hipLaunchKernelGGL(k_fd_pack, dim3(nblk(d->n_send, 256)), dim3(256), 0, st, order, (long long)d->n_send,
                               c->dist_slot.as<unsigned int>(), c->dist_first.as<unsigned long long>(),
                               is_edge ? 0ull : base, is_edge ? c->edge_tab.as<Slot>() : c->node_tab.as<Slot>(),
                               c->dist_lcnt.as<unsigned int>(), d->send.as<unsigned long long>());
        }
        stage_end(c);
        d->state = S_REDUCE;
        if (!wire) continue;
        xfer_a2a(d, x, d->send.p, d->recv.p, REC_BYTES, d->send_counts, d->recv_counts, 2);
        return 1;
      }
      case S_REDUCE: {
        // (one rank, nothing on the wire: what was packed is what arrives, and the answers are read where they are written)
        d->recv_p = wire ? d->recv.p : d->send.p;
        d->rep_out_p = wire ? d->rep_out.p : d->rep_in.p;
        d->rep_in_p = d->rep_in.p;
        stage_begin(c, is_edge ? "merge_edge_reduce" : "merge_node_reduce");
        const int r = reduce_records(c, d, is_edge);
        stage_end(c);
        if (r != AMG_OK) return r;
        d->state = S_HOLD;
        if (!wire) continue;
        xfer_a2a(d, x, d->rep_out.p, d->rep_in.p, REPLY_WORDS * (int)sizeof(long long), d->recv_counts, d->send_counts, 3);
        return 1;
      }
      case S_HOLD: {
        stage_begin(c, is_edge ? "merge_edge_hold" : "merge_node_hold");
        const int r = hold_records(c, d, is_edge);
        stage_end(c);
        if (r != AMG_OK) {
          d->state = S_IDLE;
          return r;
        }
        d->state = S_HCOUNTS;
        if (!wire) continue;
        xfer_ag(d, x, d->hc_send.p, d->hc_recv.p, HC_WORDS * (int)sizeof(long long), 1, false);
        return 1;
      }
      case S_HCOUNTS: {
        d->words.assign((size_t)W * HC_WORDS, 0);
        const WordRange rr[1] = {{wire ? d->hc_recv.p : d->hc_send.p, W * HC_WORDS}};
        AMGCHK(fetch_ranges(c, rr, 1, reinterpret_cast<long long*>(d->words.data())));
        ++d->st[0];
        const long long* got = reinterpret_cast<const long long*>(d->words.data());
        const int v = verdict(d, got, HC_WORDS, got + 1);
        if (v == 1) {
          restart(d);
          continue;
        }
        if (v < 0) {
          d->state = S_IDLE;
          return v;
        }
        d->held_counts.resize(W);
        d->m_pad = 0;
        d->n_total = 0;
        for (int p = 0; p < W; ++p) {
          d->held_counts[p] = got[(size_t)p * HC_WORDS];
          d->n_total += d->held_counts[p];
          if (d->held_counts[p] > d->m_pad) d->m_pad = d->held_counts[p];
        }
        d->n_held = d->held_counts[d->rank];
        if (d->n_held > d->n_send) {
          d->state = S_IDLE;
          return amg_fail(AMG_E_DIST, "%lld records held of %lld sent", (long long)d->n_held, (long long)d->n_send);
        }
        if (d->n_total >= (is_edge ? (1ll << 30) : (1ll << 29)))
          return amg_fail(AMG_E_OVERFLOW, "merged graph beyond 2^%d %s", is_edge ? 30 : 29, is_edge ? "edge classes" : "nodes");
        hipLaunchKernelGGL(k_offs, dim3(1), dim3(1), 0, st, wire ? d->hc_recv.as<long long>() : d->hc_send.as<long long>(), W,
                           d->offs.as<long long>());
        d->state = S_GLOBAL;
        if (!wire || d->m_pad == 0) {
          d->gathered_p = d->held.p;
          continue;
        }
        const int hb = is_edge ? HELD_EDGE_BYTES : (int)held_node_bytes(c->k, c->two_v);
        // equal-size contributions of m record slots; what lies behind a rank's own records is never read
        const void* src = d->held.p;
        if (d->held.cap < (size_t)d->m_pad * hb) {
          AMGCHK(d->held_pad.ensure((size_t)d->m_pad * hb));
          HIPCHK(hipMemcpyAsync(d->held_pad.p, d->held.p, (size_t)d->n_held * hb, hipMemcpyDeviceToDevice, st));
          src = d->held_pad.p;
        }
        AMGCHK(d->gathered.ensure((size_t)W * (size_t)d->m_pad * hb));
        d->gathered_p = d->gathered.p;
        xfer_ag(d, x, src, d->gathered.p, hb, d->m_pad, true);
        return 1;
      }
      case S_GLOBAL: {
        const int r = is_edge ? edges_global(c, d) : nodes_global(c, d);
        if (r != AMG_OK) {
          d->state = S_IDLE;
          return r;
        }
        if (!is_edge) {
          d->kind = 1;
          d->state = S_LOCAL;
          continue;
        }
        d->state = S_IDLE;
        d->st[5] += d->attempt;
        return 0;
      }
      case S_DV_LOCAL: {
        // every rank squeezes its copy of the graph (when ITS correction allows it) and says so; only if all do is the
        // squeezed graph taken — otherwise the ordinary merged build runs, nothing it reads has been touched
        d->dv_ok = false;
        const bool mine = c->derive_ready && (int)d->tokens.size() == W;
        c->derive_ready = false;
        c->dist_candidate = false;
        if (mine) {
          d->dv_bases.assign(W + 1, 0);
          for (int p = 0; p < W; ++p) d->dv_bases[p + 1] = d->dv_bases[p] + d->tokens[p];
          AMGCHK(d->offs.ensure((size_t)(2 * W + 4) * sizeof(long long)));
          HIPCHK(hipMemcpyAsync(d->offs.p, d->dv_bases.data(), (size_t)(W + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
          d->dv_bounds.assign(2 * (W + 1), 0);
          AMGCHK(derive_local(c, d->k, d->dv_bases[d->rank], d->tokens[d->rank], d->offs.as<long long>(), W,
                              d->dv_bounds.data(), &d->dv_D2, &d->dv_P2, &d->dv_ok));
        }
        if (!wire) {
          if (!d->dv_ok) {
            d->state = S_LOCAL;
            continue;
          }
          d->tokens.assign(1, c->n_tokens);
          c->tok_base = 0;
          c->tok_total = c->n_tokens;
          AMGCHK(derive_commit(c, d->dv_D2, d->dv_P2));
          ++d->st[7];
          d->state = S_IDLE;
          return 0;
        }
        AMGCHK(d->cnt_send.ensure((size_t)W * CNT_WORDS * sizeof(long long)));
        AMGCHK(d->cnt_recv.ensure((size_t)W * CNT_WORDS * sizeof(long long)));
        hipLaunchKernelGGL(k_dv_msg, dim3(1), dim3(1), 0, st, d->dv_ok ? 1ll : 0ll, (long long)c->n_tokens, (long long)d->attempt,
                           d->dv_ok ? d->dv_D2 : -1ll, d->cnt_send.as<long long>());
        xfer_ag(d, x, d->cnt_send.p, d->cnt_recv.p, CNT_WORDS * (int)sizeof(long long), 1, false);
        d->state = S_DV_ASK;
        return 1;
      }
      case S_DV_ASK: {
        d->words.assign((size_t)W * CNT_WORDS, 0);
        const WordRange rr[1] = {{d->cnt_recv.p, W * CNT_WORDS}};
        AMGCHK(fetch_ranges(c, rr, 1, reinterpret_cast<long long*>(d->words.data())));
        ++d->st[0];
        const long long* got = reinterpret_cast<const long long*>(d->words.data());
        bool all_ok = true;
        for (int p = 0; p < W; ++p) all_ok = all_ok && got[(size_t)p * CNT_WORDS] == 1 && got[(size_t)p * CNT_WORDS + 3] == d->dv_D2;
        if (!all_ok) {  // somebody re-threaded a read (or disagrees about the graph): the ordinary merged build
          d->state = S_LOCAL;
          continue;
        }
        long long base = 0, total = 0;
        for (int p = 0; p < W; ++p) {
          d->tokens[p] = got[(size_t)p * CNT_WORDS + 1];
          if (p < d->rank) base += d->tokens[p];
          total += d->tokens[p];
        }
        c->tok_base = base;
        c->tok_total = total;
        const long long* nb = d->dv_bounds.data();
        const long long* pb = nb + W + 1;
        d->dv_mN = d->dv_mP = 0;
        for (int p = 0; p < W; ++p) {
          if (nb[p + 1] - nb[p] > d->dv_mN) d->dv_mN = nb[p + 1] - nb[p];
          if (pb[p + 1] - pb[p] > d->dv_mP) d->dv_mP = pb[p + 1] - pb[p];
        }
        const long long m = d->dv_mN + d->dv_mP;
        d->state = S_DV_FILL;
        if (m == 0) continue;
        AMGCHK(d->held.ensure((size_t)m * sizeof(unsigned long long)));
        AMGCHK(d->gathered.ensure((size_t)W * (size_t)m * sizeof(unsigned long long)));
        hipLaunchKernelGGL(k_dv_contrib, dim3(nblk(m, 256)), dim3(256), 0, st, c->alt_nfirst.as<long long>() + nb[d->rank],
                           nb[d->rank + 1] - nb[d->rank], (long long)(base << 1),
                           c->alt_pfirst.as<unsigned long long>() + pb[d->rank], pb[d->rank + 1] - pb[d->rank],
                           (unsigned long long)base << 3, d->dv_mN, d->dv_mP, d->held.as<unsigned long long>());
        xfer_ag(d, x, d->held.p, d->gathered.p, (int)sizeof(unsigned long long), m, true);
        return 1;
      }
      case S_DV_FILL: {
        const long long m = d->dv_mN + d->dv_mP;
        if (m > 0) {
          HIPCHK(hipMemcpyAsync(d->offs.p, d->dv_bounds.data(), (size_t)(2 * (W + 1)) * sizeof(long long), hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(k_dv_fill, dim3(nblk(m * W, 256)), dim3(256), 0, st, d->gathered.as<unsigned long long>(), d->dv_mN,
                             d->dv_mP, W, d->offs.as<long long>(), c->alt_nfirst.as<long long>(),
                             c->alt_pfirst.as<unsigned long long>());
        }
        AMGCHK(derive_commit(c, d->dv_D2, d->dv_P2));
        ++d->st[7];
        d->state = S_IDLE;
        return 0;
      }
      default:
        return amg_fail(AMG_E_STATE, "amg_dist_merge_begin first");
    }
  }
}

extern "C" int amg_dist_merge_begin(amg_ctx* c, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov) {
  NEED_CTX(c);
  if (k < 1 || k > AMG_MAX_K) return amg_fail(AMG_E_ARG, "k must be in [1, %d]", AMG_MAX_K);
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  DistState* d = dm(c);
  d->k = k;
  d->mn = min_node_cov < 1 ? 1 : min_node_cov;
  d->me = min_edge_cov < 1 ? 1 : min_edge_cov;
  d->attempt = 0;
  d->kind = 0;
  d->fail_ret = 0;
  // the reads are what a correction left of the reads of the merged graph still held (amg_adopt_corrected), same k, plain
  // build: the ranks first find out whether that graph's live part will do (S_DV_*; AMG_NO_DERIVE=1: A/B + test switch)
  const bool candidate = c->dist_candidate && c->dist_mode && c->world == d->world && k == c->k && d->mn == 1 && d->me == 1 &&
                         !getenv("AMG_NO_DERIVE");
  if (!candidate) c->dist_candidate = c->derive_ready = false;
  d->state = candidate ? S_DV_LOCAL : S_LOCAL;
  return AMG_OK;
}

extern "C" int amg_dist_merge_next(amg_ctx* c, amg_xfer* out) {
  NEED_CTX(c);
  if (!out) return amg_fail(AMG_E_ARG, "null xfer");
  DistState* d = dm(c);
  if (d->state == S_IDLE) return amg_fail(AMG_E_STATE, "amg_dist_merge_begin first");
  const int r = advance(c, out);
  if (r < 0) d->state = S_IDLE;
  phase_tick(c, d, -1);
  return r;
}

// ---- the exchanges over RCCL, on the ctx's stream
static int rccl_perform(amg_ctx* c, DistState* d, const amg_xfer& x) {
  hipStream_t st = c->stream;
  if (x.kind == AMG_XFER_ALL_GATHER) {
    NCCLCHK(g_rccl.AllGather(x.send, x.recv, (size_t)x.count * x.elem_bytes, ncclChar, d->comm, st));
    return AMG_OK;
  }
  const char* sp = static_cast<const char*>(x.send);
  char* rp = static_cast<char*>(x.recv);
  size_t so = 0, ro = 0;
  NCCLCHK(g_rccl.GroupStart());
  for (int p = 0; p < d->world; ++p) {
    const size_t sb = (size_t)x.send_counts[p] * x.elem_bytes, rb = (size_t)x.recv_counts[p] * x.elem_bytes;
    if (sb) NCCLCHK(g_rccl.Send(sp + so, sb, ncclChar, p, d->comm, st));
    if (rb) NCCLCHK(g_rccl.Recv(rp + ro, rb, ncclChar, p, d->comm, st));
    so += sb;
    ro += rb;
  }
  NCCLCHK(g_rccl.GroupEnd());
  return AMG_OK;
}

extern "C" int amg_dist_merge(amg_ctx* c, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov) {
  NEED_CTX(c);
  DistState* d = dm(c);
  if ((d->world > 1 || d->always_exchange) && !d->comm)
    return amg_fail(AMG_E_STATE, "amg_dist_init first (or drive the exchanges yourself: amg_dist_merge_begin / _next)");
  AMGCHK(amg_dist_merge_begin(c, k, min_node_cov, min_edge_cov));
  for (;;) {
    amg_xfer x;
    const int r = amg_dist_merge_next(c, &x);
    if (r <= 0) return r;
    const int e = rccl_perform(c, d, x);
    if (e != AMG_OK) {
      d->state = S_IDLE;
      return e;
    }
  }
}

// ---- emulated ranks: the ctxs of one process, one device; the exchanges are device copies
extern "C" int amg_dist_merge_local(amg_ctx* const* ctxs, int32_t world, int32_t k, uint32_t min_node_cov,
                                    uint32_t min_edge_cov) {
  if (!ctxs || world < 1) return amg_fail(AMG_E_ARG, "bad ctx list");
  for (int r = 0; r < world; ++r) {
    if (!ctxs[r]) return amg_fail(AMG_E_ARG, "null ctx");
    if (ctxs[r]->device != ctxs[0]->device) return amg_fail(AMG_E_ARG, "emulated ranks share one device");
    DistState* d = dm(ctxs[r]);
    if (d->world != world || d->rank != r || d->comm) AMGCHK(set_world(ctxs[r], r, world));
    AMGCHK(amg_dist_merge_begin(ctxs[r], k, min_node_cov, min_edge_cov));
  }
  HIPCHK(hipSetDevice(ctxs[0]->device));
  std::vector<amg_xfer> xs(world);
  auto abandon = [&](int ret) {
    const std::string msg = g_amg_err;
    for (int r = 0; r < world; ++r) dm(ctxs[r])->state = S_IDLE;
    g_amg_err = msg;
    return ret;
  };
  for (;;) {
    int pending = 0, done = 0, err = 0;
    std::string err_msg;
    for (int r = 0; r < world; ++r) {
      const int v = amg_dist_merge_next(ctxs[r], &xs[r]);
      if (v == 1) ++pending;
      else if (v == 0) ++done;
      else if (!err || err == AMG_E_DIST) {  // (a peer's "rank r failed" gives way to the failing rank's own message)
        err = v;
        err_msg = g_amg_err;
      }
    }
    if (err) {
      g_amg_err = err_msg;
      return abandon(err);
    }
    if (done == world) return AMG_OK;
    if (pending != world) return abandon(amg_fail(AMG_E_DIST, "emulated ranks fell out of step"));
    for (int r = 0; r < world; ++r) HIPCHK(hipStreamSynchronize(ctxs[r]->stream));
    hipStream_t st = ctxs[0]->stream;
    for (int r = 1; r < world; ++r)
      if (xs[r].kind != xs[0].kind || xs[r].elem_bytes != xs[0].elem_bytes)
        return abandon(amg_fail(AMG_E_DIST, "emulated ranks ask for different exchanges"));
    const size_t eb = (size_t)xs[0].elem_bytes;
    if (xs[0].kind == AMG_XFER_ALL_GATHER) {
      for (int dst = 0; dst < world; ++dst)
        for (int src = 0; src < world; ++src) {
          if (xs[src].count != xs[0].count) return abandon(amg_fail(AMG_E_DIST, "all-gather sizes differ"));
          if (xs[src].count)
            HIPCHK(hipMemcpyAsync(static_cast<char*>(xs[dst].recv) + (size_t)src * xs[0].count * eb, xs[src].send,
                                  (size_t)xs[0].count * eb, hipMemcpyDeviceToDevice, st));
        }
    } else {
      for (int dst = 0; dst < world; ++dst) {
        size_t ro = 0;
        for (int src = 0; src < world; ++src) {
          size_t so = 0;
          for (int p = 0; p < dst; ++p) so += (size_t)xs[src].send_counts[p] * eb;
          const size_t bytes = (size_t)xs[src].send_counts[dst] * eb;
          if (xs[dst].recv_counts[src] != xs[src].send_counts[dst])
            return abandon(amg_fail(AMG_E_DIST, "all-to-all counts of ranks %d and %d disagree", src, dst));
          if (bytes)
            HIPCHK(hipMemcpyAsync(static_cast<char*>(xs[dst].recv) + ro, static_cast<const char*>(xs[src].send) + so, bytes,
                                  hipMemcpyDeviceToDevice, st));
          ro += bytes;
        }
      }
    }
    HIPCHK(hipStreamSynchronize(st));
  }
}

// out[0] host waits on exchanged counts, [1] exchanges, [2] most bytes of records to ONE peer, [3] the same of replies,
// [4] bytes contributed to the all-gathers of held records, [5] repeated builds (merge-key collisions), [6] bytes of
// records sent to all peers, [7] builds made from the previous merged graph's live part; since the last reset
extern "C" int amg_dist_stats(amg_ctx* c, int64_t* out, int32_t reset) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  DistState* d = dm(c);
  if (out)
    for (int i = 0; i < 8; ++i) out[i] = d->st[i];
  if (reset)
    for (int i = 0; i < 8; ++i) d->st[i] = 0;
  return AMG_OK;
}

// synchronised wall time per phase of the driver since the last reset (on = 1 starts the measurement, which
// synchronises the stream around every phase): names[i] points at static strings; returns the number of phases
extern "C" int amg_dist_phase_ms(amg_ctx* c, int32_t on, const char** names, double* ms, int32_t cap) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  DistState* d = dm(c);
  int n = 0;
  for (int i = 0; i < 2 * S_N; ++i) {
    if (!kPhaseNames[i][0]) continue;
    if (n < cap) {
      if (names) names[n] = kPhaseNames[i];
      if (ms) ms[n] = d->phase_ms[i];
    }
    ++n;
    d->phase_ms[i] = 0.0;
  }
  d->time_phases = on != 0;
  return n < cap ? n : cap;
}
