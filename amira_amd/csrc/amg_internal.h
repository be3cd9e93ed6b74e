// amg_internal.h — shared declarations of libamg.so (MI355X / gfx950 only).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/amg.h"

// ------------------------------------------------------------------ error plumbing
extern thread_local std::string g_amg_err;
int amg_fail(int code, const char* fmt, ...);

#define HIPCHK(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return amg_fail(AMG_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
                      hipGetErrorString(e_));                                          \
  } while (0)
#define AMGCHK(call)                 \
  do {                               \
    int r_ = (call);                 \
    if (r_ != AMG_OK) return r_;     \
  } while (0)

// ------------------------------------------------------------------ device buffers
// grow-only device allocation reused across builds (hipMalloc is ~ms for GB sizes)
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // a BORROWED view of caller memory (amg_set_reads / amg_set_positions with on_device = 2): the
  // buffer's own allocation is parked meanwhile and comes back with the first ensure / unborrow;
  // borrowed memory is never written, resized or freed here
  void* own_p = nullptr;
  size_t own_cap = 0;
  bool borrowed = false;
  void borrow(const void* ptr, size_t bytes) {
    if (!borrowed) {
      own_p = p;
      own_cap = cap;
    }
    p = const_cast<void*>(ptr);
    cap = bytes;
    borrowed = true;
  }
  void unborrow() {
    if (!borrowed) return;
    p = own_p;
    cap = own_cap;
    own_p = nullptr;
    own_cap = 0;
    borrowed = false;
  }
  int ensure(size_t bytes) {
    unborrow();
    if (bytes <= cap) return AMG_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return amg_fail(AMG_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return AMG_OK;
  }
  void release() {
    unborrow();
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// ------------------------------------------------------------------ hash-table slot
// One 32-byte slot = one 32-B sector: a probe touches exactly one sector.
//   key      : 64-bit fingerprint of the canonical k-tuple (nodes) or the exact packed
//              (lo, hi, sign) key (edges); 0 = empty.  Claimed by atomicCAS.
//   first_inv: ~first_seen, maximised by atomicMax  (zero-initialised == "never seen").
//              nodes: first_seen = (token index << 1) | (direction == -1)
//              edges: first_seen = (token index << 3) | orientation bits
//   count    : occurrences (atomicAdd)
//   id       : dense id, written by the ranking kernel
struct __attribute__((aligned(32))) Slot {
  unsigned long long key;
  unsigned long long first_inv;
  unsigned int count;
  int id;
  unsigned long long pad;
};
static_assert(sizeof(Slot) == 32, "slot must be one 32-byte sector");

// device-side status words (index into ctx->d_status)
enum {
  ST_NODE_INSERTS = 0,   // number of node slots claimed
  ST_PAIR_INSERTS = 1,   // number of edge-class slots claimed
  ST_OVERFLOW = 2,       // a probe sequence exceeded its limit
  ST_PALINDROME = 3,     // window equal to its reverse complement
  ST_COLLISION = 4,      // fingerprint collision found by the exact verify
  ST_N_WINDOWS = 5,
  ST_N_SHORT = 6,
  ST_COMPACT_A = 7,      // scratch counters for compaction kernels
  ST_COMPACT_B = 8,
  ST_MISC = 9,
  ST_BADINPUT = 10,      // malformed CSR: 1 = read offsets, 2 = a token outside [0, two_v)
  ST_COV_SUM = 11,       // tip clipping: sum of the live nodes' coverages, [12] = number of live nodes
  ST_WORDS = 16
};

struct StageTime {
  const char* name;
  hipEvent_t a, b;
  float ms;
};

static const uint64_t kAmgSeed0 = 0x9E3779B97F4A7C15ull;  // initial fingerprint seed (amg_build re-seeds on a collision)

struct DistState;  // amg_dist.hip: communicator, buffers and progress of the ctx's merged builds
struct BubbleState;  // amg_bubbles.hip: what amg_junction_paths found, until the caller has fetched it

struct amg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;

  // ---- current read set (device)
  DevBuf tokens;      // int32[n_tokens + pad]
  DevBuf read_off;    // int64[n_reads + 1]
  DevBuf gene_start;  // int64[n_tokens]   (optional)
  DevBuf gene_end;    // int64[n_tokens]
  DevBuf read_len;    // int64[n_reads]
  bool have_pos = false, have_read_len = false;
  // where the positions of the CURRENT reads' genes are (amg_passes.hip, CorrArgs): offsets into
  // gene_start / gene_end as handed to amg_set_positions (pos_n0 entries) or, from pos_n0 on, into
  // the pool of positions the carry-over kernels produced
  DevBuf pos_off, c_pos_off;   // int64[n_reads] (pos_identity: the read's token offset, array unused)
  DevBuf pos1_s, pos1_e;       // int64[pos1_used]
  bool pos_identity = true;
  int64_t pos_n0 = 0, pos1_used = 0, c_pos1_used = 0;
  int64_t n_reads = 0, n_tokens = 0;
  int32_t two_v = 0;

  // ---- build products
  bool built = false;
  int32_t k = 0;
  uint64_t seed = kAmgSeed0;
  int64_t n_windows = 0, n_short = 0;
  int64_t n_nodes = 0, n_pairs = 0, n_edges = 0, n_components = 0;
  int64_t node_slots = 0, edge_slots = 0, retries = 0;
  int64_t node_hint = 0;  // distinct-node estimate carried between builds
  int64_t c_node_bound = 0;  // amg_correct_reads: upper bound for the nodes of the graph the corrected reads make (at gene-mer size c_node_bound_k)
  int c_node_bound_k = 0;
  int64_t hint_bound = 0;    // amg_set_reads_from_corrected: the source's bound for THESE reads, applied by a build at hint_bound_k
  int hint_bound_k = 0;
  bool filtered_build = false;  // bx_nodes_upsert is running for amg_build_filtered (head launch: see there)
  int64_t n_local_nodes = 0, n_local_pairs = 0;  // before a multi-GPU merge
  int64_t tok_base = 0;   // global index of this shard's first token (0 on a single GPU)
  int64_t tok_total = 0;  // tokens over all shards

  DevBuf node_tab;   // Slot[node_slots]
  DevBuf edge_tab;   // Slot[edge_slots]
  DevBuf tok_slot;   // int32[n_tokens]  slot index | LAST flag, -1 = no window
  DevBuf tok_node;   // int32[n_tokens]  node id, -1 no window, -2 removed
  DevBuf tok_dir;    // int8 [n_tokens]
  DevBuf tok_pair;   // int32[n_tokens]  edge-class slot (then id) of the adjacency t -> t+1
  bool packed_nodes = false;  // node slots rewritten as packed {id, tuple} records (amg_device.h)
  int weak_fp_builds = 0;     // test hook, see amg_build
  bool dist_mode = false;     // building a shard of a merged (multi-GPU) graph
  bool count_inline = false;  // true: count by one global atomic per window (merge path)
  // nodes (id order)
  DevBuf node_tokens;  // int32[n_nodes * k]
  DevBuf node_cov;     // uint32[n_nodes]
  DevBuf node_first;   // int64[n_nodes]  first_seen value ((tok << 1) | dirbit)
  DevBuf node_comp;    // int32[n_nodes]
  DevBuf node_alive;   // uint8[n_nodes]
  // edges (id order)
  DevBuf edge_src, edge_tgt;    // int32[n_edges]
  DevBuf edge_sdir, edge_tdir;  // int8[n_edges]
  DevBuf edge_cov;              // uint32[n_edges]
  DevBuf edge_alive;            // uint8[n_edges]
  // edge classes in first-seen order (input of the edge emission)
  DevBuf pair_key, pair_first;  // uint64[n_pairs]
  DevBuf pair_cnt;              // uint32[n_pairs]
  // adjacency CSR, row 2n = forward list of node n, 2n+1 = backward list
  DevBuf adj_off;   // int64[2 n_nodes + 1]
  DevBuf adj_edge;  // int32[n_edges]
  // live adjacency (only alive edges, targets inline) — rebuilt lazily after removals
  DevBuf ladj_off;  // int64[2 n_nodes + 1]
  DevBuf ladj;      // int2[n_live_edges]  {target node, target direction}
  DevBuf ladj_rows; // int4[2 n_nodes]  {offset, live count, first target, first direction}
  DevBuf ladj_pos, ladj_keys;  // scratch of the live-adjacency build (callers hold s0..s5)
  DevBuf hub_bits;             // bitmaps over the edge ids for adjacency rows beyond HUGE_ROW (huge_row_in_order)
  bool ladj_valid = false;
  bool ladj_stale = false;  // the live lists are those of the graph before some NODES died (ensure_live_adj patches them)
  bool comp_valid = false, adj_valid = false;  // component ids / full edge lists of the built graph are made on demand
  bool pristine = false;  // nothing has been removed since the build, and its component labels are those of the graph as it is
  // reads
  DevBuf read_fix;  // uint8[n_reads]  read is in _readsToCorrect

  // ---- corrected read set (output of amg_correct_reads)
  bool have_corrected = false;
  bool pos0_own = false;  // pool 0 of the positions is the engine's own compaction, not the caller's arrays any more
  int64_t c_reads = 0, c_tokens = 0;
  DevBuf c_tokens_buf, c_read_off, c_orig, c_changed, c_gstart, c_gend, c_read_len;

  // ---- the rebuild that reuses the previous build (amg_derive.hip)
  DevBuf c_src, rd_src;          // int64[reads]: token index, in the read set the graph was built from, of every corrected /
                                 // current read's first gene
  bool c_derivable = false;      // amg_correct_reads: the corrected set only drops and trims reads of the graph's read set
  bool derive_ready = false;     // the current reads are such a set of the graph still held (amg_adopt_corrected)
  bool derived = false;          // the graph at hand was made that way (amg_counts)
  bool dist_candidate = false;   // the same on a rank of a merged build, whatever THIS rank's correction did (every rank asks)
  bool edge_own_deaths = false;  // since the build an edge was removed with both its nodes alive
  DevBuf alt_tok_node, alt_tok_dir, alt_ntok, alt_ncov, alt_nfirst, alt_nalive, alt_pkey, alt_pfirst, alt_pcnt;

  // ---- exact-key build (amg_build_x.hip): arrays indexed by CLAIM id (order of slot creation)
  bool exact_keys = false;   // this build used the exact-key path
  int x_bits = 0;            // bits per token in the packed tuple
  bool x_fp = false;         // the tuple does not fit the slot: the key is its 94-bit fingerprint (amg_build_x.hip, x_fp94)
  int64_t x_nspace = 0, x_espace = 0;  // claim ids in use are below these (== n_nodes / n_pairs unless the
                                       // claims came from the shard counters: XShard in amg_x.h)
  int64_t x_max_claims = 0, x_max_eclaims = 0;  // capacity of the per-claim arrays (second half of x_first / x_efirst starts there)
  bool dist_x = false;       // this merged build keeps its LOCAL tables in the exact-key layout
  DevBuf x_first, x_slot;    // uint32[claims]  ~first_seen of a node claim, its table slot
  DevBuf x_final;            // int32 [claims]  claim id -> node id
  DevBuf x_efirst, x_eslot;  // the same for edge-class claims
  DevBuf x_ecnt;             // uint32[edge claims] occurrences
  DevBuf x_ncnt;             // uint32[node claims] occurrences (plain build: scattered into node_cov)
  int64_t rank_flags_clean = 0;  // words of the ranking bitmap's flag bytes (s0) a read-back's filler has zeroed already
  DevBuf x_ftag;             // int32 [node claims] x_final | AMG_SINGLE_BIT on nodes of coverage 1 (edge pass with lone classes)
  DevBuf f_ctrs;             // fused table pass: per-shard claim counters
  DevBuf x_efinal;           // int32 [edge claims] claim id -> edge-class id
  DevBuf x_first_all;        // uint32[claims] filtered build: ~first_seen of EVERY claim (k_x_drop_claims zeroes x_first)
  bool comp_from_claims = false;  // filtered build: component ids come from the claims (bx_components_from_claims)

  // ---- multi-GPU merge (amg_dist.hip)
  int world = 1;
  uint32_t dist_min_node = 1, dist_min_edge = 1;  // fused filter of the next merged build
  int64_t dist_nspace = 0;   // ids the current bucketing ran over (local records + claim ids nobody took)
  bool dist_sorted = false;  // the local records leave in sorted order (send_order)
  DevBuf dist_a, dist_cnt, dist_first, dist_slot, dist_gtab, dist_lcnt;
  DistState* dist = nullptr;
  BubbleState* bub = nullptr;  // amg_bubbles.hip

  // ---- K6 result cache (two-call protocol of amg_match_patterns)
  bool match_valid = false;
  int64_t match_total = 0, match_npat = 0;
  DevBuf match_read, match_pos, match_off, match_wave;  // match_wave: hits per wave of k_match and their prefix

  // ---- scratch
  DevBuf status;       // unsigned long long[ST_WORDS]
  // mailbox for the few words the host reads back between launches (fetch(), amg_api.hip): pinned host memory
  // the device writes, a ticket last; the host spins on the ticket instead of sleeping in hipStreamSynchronize
  unsigned long long* mail_host = nullptr;
  unsigned long long* mail_dev = nullptr;
  unsigned long long mail_ticket = 0;
  DevBuf sort_tmp;     // rocPRIM temp storage
  DevBuf scan_state;   // amg_scan.hip: ticket counter + one status word per tile
  unsigned long long scan_tickets = 0;  // tiles all scans so far have used
  unsigned int scan_epoch = 0;
  DevBuf s0, s1, s2, s3, s4, s5;  // general scratch arrays
  DevBuf nw_big;       // global scratch of the general position carry-over kernel (long reads)
  DevBuf gap_rec;      // per gapped read: the record k_corr_gapped_fast starts from
  // path memo of the re-threading (amg_passes.hip: k_gap_queries / k_gap_dfs)
  DevBuf gm_mask;      // uint64[n_reads]   live-window mask per read (k_corr_classify)
  DevBuf gm_tab;       // uint64[slots]     question table: (start node, direction, end node)
  DevBuf gm_res;       // int4  [slots]     per question {pool offset, ints, paths}
  DevBuf gm_list;      // int32 [questions] occupied slots
  DevBuf gm_q;         // int32 [gapped reads x GF_MAXGAP] question slot per None run
  DevBuf gm_pool;      // int32 path records
  DevBuf gm_gene;      // int32 per pool entry of a one-answer question: last gene of the path node (k_corr_gapped_lean)
  DevBuf gm_fail;      // uint8 [gapped reads] 1: the sixteen-lanes-per-read kernel left the read to the wave-per-read one
  DevBuf gm_ctr;       // uint64[4]         {questions listed, pool ints used}
  DevBuf nw_rec;       // per gapped read: the record k_corr_nw_fast starts from
  DevBuf bnd_bits;     // uint32[(n_tokens >> 5) + pad]: bit t set when a read ends at token t
  DevBuf cnt_state;    // counting sweeps: per-sweep left-over counts and done flags + the hints
  bool cnt_hint_reset = false;
  DevBuf cnt_list;     // k_count_ids: what the first sweep of a count found beyond its range, while that is little
  int cnt_sweeps[2] = {4, 4};  // sweeps the last node / edge-class count made use of (count_ids launches no more)

  std::vector<StageTime> stages;
  bool timing = true;
};

// ------------------------------------------------------------------ primitives (amg_prims.hip)
int prim_sort_u64_u32(amg_ctx* c, const unsigned long long* kin, unsigned long long* kout,
                      const unsigned int* vin, unsigned int* vout, size_t n, int end_bit);
int prim_sort_u32_u32(amg_ctx* c, const unsigned int* kin, unsigned int* kout,
                      const unsigned int* vin, unsigned int* vout, size_t n, int end_bit);
int prim_exscan_i64(amg_ctx* c, const long long* in, long long* out, size_t n);
int prim_exscan_u32_to_i64(amg_ctx* c, const unsigned int* in, long long* out, size_t n);
int prim_exscan_u32_pair(amg_ctx* c, const unsigned int* in_a, long long* out_a, const unsigned int* in_b, long long* out_b,
                         size_t n);
int prim_exscan_i64_pair(amg_ctx* c, const long long* in_a, long long* out_a, const long long* in_b, long long* out_b,
                         size_t n);
int prim_exscan_bytes_set(amg_ctx* c, const unsigned char* in, long long* out, size_t n);
int prim_exscan_apply_kill(amg_ctx* c, unsigned char* kill, unsigned char* alive, long long* out, size_t n);
int prim_exscan_keep_and_len(amg_ctx* c, const unsigned int* len, long long* out_keep, long long* out_off, size_t n);
int prim_exscan_flag_words(amg_ctx* c, const unsigned char* flags, unsigned int* bits, long long* out, size_t n_words);
int prim_exscan_bits_popc(amg_ctx* c, const unsigned int* bits, long long* out, size_t n_words);
int prim_exscan_pair_width(amg_ctx* c, const unsigned long long* pkey, long long* out, size_t n_pairs);

// one launch that zeroes up to 8 device ranges (a hipMemsetAsync is a kernel launch of its own: ~5 us
// each, and a build issued ~40 of them); sizes are rounded up to 4 bytes — pad the allocations
#define CLEAR_MAX 8
struct ClearList {
  void* p[CLEAR_MAX];
  unsigned long long bytes[CLEAR_MAX];
  int n = 0;
  bool overflow = false;  // more than CLEAR_MAX ranges were added: clear_many refuses the list
  unsigned int fill[CLEAR_MAX];  // 32-bit word the range is filled with (0: cleared)
  void add(void* ptr, size_t b, unsigned int value = 0u) {
    if (b == 0) return;
    if (n >= CLEAR_MAX) {
      overflow = true;
      return;
    }
    p[n] = ptr;
    bytes[n] = (unsigned long long)((b + 3) & ~(size_t)3);
    fill[n] = value;
    ++n;
  }
};
int clear_many(amg_ctx* c, const ClearList& l);

// Device words the host needs NOW (counts that size the next allocation, status flags): everything queued on the
// stream before the call has finished when fetch() returns, like hipMemcpyAsync + hipStreamSynchronize, at a
// fraction of the latency (the stream stays idle ~10 us per read-back instead of ~40).
#define FETCH_MAX 96
struct FetchList {
  const unsigned long long* p[FETCH_MAX];
  int n = 0;
  bool overflow = false;  // more than FETCH_MAX words were asked for: fetch refuses the list
  void add(const void* q) {
    if (n < FETCH_MAX)
      p[n++] = static_cast<const unsigned long long*>(q);
    else
      overflow = true;
  }
  void add_words(const void* q, int words) { for (int i = 0; i < words; ++i) add(static_cast<const unsigned long long*>(q) + i); }
};
// filler: ranges to zero AFTER the read-back kernel and before the host waits for it — work that does not depend on the
// words read, queued so that the stream is not idle while the host turns the answer into the next launches
int fetch(amg_ctx* c, const FetchList& l, unsigned long long* out, const ClearList* filler = nullptr);
static_assert(ST_WORDS <= FETCH_MAX, "fetch_status reads the status words in one list");
int fetch_status(amg_ctx* c, unsigned long long* out /*[ST_WORDS]*/, const ClearList* filler = nullptr);
int stream_wait(amg_ctx* c);  // hipStreamSynchronize at the latency of fetch()

// ------------------------------------------------------------------ stage timing
void stage_begin(amg_ctx* c, const char* name);
void stage_end(amg_ctx* c);
void stages_reset(amg_ctx* c);

// ------------------------------------------------------------------ walkers' view of the live graph (amg_passes.hip)
struct GView {
  // live adjacency: row 2n = forward list of node n, row 2n+1 = backward list, only ALIVE
  // edges, in list order, with the target inline (.x = target node, .y = target direction)
  const int2* lent;
  const int4* lrows;  // per row {offset, live count, first target, first direction}: one 16-B load
                      // tells a walker everything about a row with <= 1 live edge (most rows)
  const unsigned char* n_alive;
  const unsigned int* n_cov;
  const int* n_tok;
  const long long* n_first;
  const int* n_comp;
  int k, flip;
};
int ensure_live_adj(amg_ctx* c);  // the live lists below exist and are up to date
GView make_view(amg_ctx* c);
void bubbles_release(amg_ctx* c);

void dist_release(amg_ctx* c);  // amg_dist.hip
// amg_derive.hip
int derive_from_previous(amg_ctx* c, int k, bool* done);
int derive_local(amg_ctx* c, int k, long long own_lo, long long own_tokens, const long long* d_bases, int world,
                 long long* bounds_host, long long* D2_out, long long* P2_out, bool* ok);
int derive_commit(amg_ctx* c, long long D2, long long P2);

// build stages (amg_build.hip), shared with the multi-GPU path (amg_dist.hip)
uint64_t pow2_at_least(uint64_t x);
uint64_t slots_for(uint64_t n_keys);
void bs_size_tables(amg_ctx* c);
int bs_read_stats(amg_ctx* c, int k, const ClearList* also = nullptr);
int bs_nodes_pass(amg_ctx* c, int k, int* which);
int bs_alloc_nodes(amg_ctx* c, long long D);
int bs_nodes_rank_local(amg_ctx* c);
int bs_edges_pass(amg_ctx* c, int* which);
int bs_alloc_pairs(amg_ctx* c, long long P);
int bs_pairs_from_local(amg_ctx* c);
int bs_finish_from_pairs(amg_ctx* c);
int ensure_components(amg_ctx* c);
int ensure_adjacency(amg_ctx* c);
bool bx_applicable(const amg_ctx* c, int k);
bool bx_fits(const amg_ctx* c, int k);
bool bx_tuple_fits(const amg_ctx* c, int k);
int bx_nodes(amg_ctx* c, int k, int* which);
int bx_nodes_upsert(amg_ctx* c, int k, int* which, bool sharded = false, bool rank_follows = true);
int bx_nodes_rank(amg_ctx* c);
int bx_edges(amg_ctx* c, int* which, unsigned int min_edge_cov = 0);
int bx_nodes_filtered(amg_ctx* c, int k, unsigned int min_cov, int* which);
int bx_flag_dead_reads(amg_ctx* c);
int bx_edges_upsert(amg_ctx* c, int* which, bool lone = false, bool sharded = false, bool rank_follows = true);
int bx_edges_rank(amg_ctx* c, unsigned int min_edge_cov = 0, bool nodes_counted = false);
int bx_node_count(amg_ctx* c, bool tag);
int count_ids(amg_ctx* c, int* ids, long long n, const Slot* gather_tab, long long n_ids,
              unsigned int* out, int kind, const int* remap = nullptr);
// counts of remap[claim] over per-window node claims (claim | AMG_LAST_FLAG, -1 none); the array is
// rewritten to the remapped ids
int count_ids_remap(amg_ctx* c, int* claims, long long n, const int* remap, long long n_ids, unsigned int* out,
                    int edges);
int bx_bits(const amg_ctx* c, int k);
int bx_pairs_rank(amg_ctx* c, const int* final_of_claim, int* efinal);
int bx_components_from_claims(amg_ctx* c);
int bs_count_by_slot(amg_ctx* c, const int* slots, int* ids_scratch, long long n, Slot* tab,
                     const unsigned int* slot_sorted, long long n_ids, unsigned int* out, int kind);

static inline int ilog2_ceil(uint64_t x) {
  int b = 0;
  while ((1ull << b) < x && b < 63) ++b;
  return b;
}
