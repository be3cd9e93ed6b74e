/*
 * amg.h — C ABI of libamg.so, the MI355X (gfx950) gene-mer de Bruijn graph engine.
 *
 * This is the drop-in boundary for the read -> gene-mer-graph hot path of
 * Danderson123/Amira v0.11.0.  The reference has no FFI of its own (it is pure
 * Python); every entry point below replaces the Python call named beside it
 * (file:line under the reference tree).  The binding a maintainer would add is the
 * ctypes stub shown in INTEGRATION.md (amira_amd/_ffi.py is that stub).
 *
 * Conventions
 *   - handle based: one amg_ctx per device; a ctx is not thread-safe;
 *   - every function returns 0 on success, <0 on error (AMG_E_*); the message of the
 *     last error on the calling thread is amg_last_error();
 *   - the caller owns every buffer it passes; the library owns device memory behind
 *     the handle; output sizes come from amg_counts() (two-call pattern);
 *   - calls are synchronous with respect to the ctx's HIP stream when they return
 *     data to the host; amg_sync() drains the stream otherwise;
 *   - plain pointers and sizes only — no torch / C++ types cross this boundary.
 *
 * Data model (DESIGN.md "Data layout"):
 *   token  = V + rank(gene) for '+', V - 1 - rank(gene) for '-', rank = ascending order
 *            of sha256(pickle(name)) over the vocabulary (construct_gene.py:5-10,91-93).
 *            Integer order of tokens == order of the reference's signed 256-bit gene
 *            hashes, strand flip == two_v - 1 - token.
 *   reads  = CSR: tokens[int32, n_tokens], read_offsets[int64, n_reads + 1].
 *   window = k consecutive tokens of one read, identified by the index t of its first
 *            token.  All per-window outputs are token-indexed arrays of length n_tokens
 *            (entry t describes the window starting at token t, -1/0 where no window
 *            starts), so a read's node list is the slice
 *            [read_offsets[r], read_offsets[r+1] - k + 1).
 *   node id = rank of the node's first occurrence (== insertion order of the
 *            reference's GeneMerGraph._nodes dict, construct_graph.py:188-190).
 *   edge id = insertion order of the reference's GeneMerGraph._edges dict
 *            (construct_graph.py:268-277).
 */
#ifndef AMG_H
#define AMG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amg_ctx amg_ctx;

enum {
  AMG_OK = 0,
  AMG_E_HIP = -1,        /* a HIP runtime call failed                                  */
  AMG_E_ARG = -2,        /* bad argument                                               */
  AMG_E_STATE = -3,      /* call order (e.g. getters before amg_build)                 */
  AMG_E_PALINDROME = -4, /* gene-mer == its reverse complement; the reference asserts  */
                         /* (construct_gene_mer.py:23-25) -> AssertionError in Python  */
  AMG_E_OVERFLOW = -5,   /* internal table overflow that retries could not resolve     */
  AMG_E_NOMEM = -6,
  AMG_E_DIST = -7,       /* multi-GPU merge inconsistency                              */
  AMG_E_COLLISION = -8   /* merged build: two gene-mers shared a 64-bit merge key under every seed
                            tried (amg_dist_merge repeats the build on every rank by itself) */
};

#define AMG_MAX_K 16

typedef struct amg_counts_t {
  int64_t n_reads;        /* reads in the current read set                             */
  int64_t n_tokens;       /* genes over all reads                                      */
  int64_t n_windows;      /* gene-mers = sum max(0, len - k + 1)                       */
  int64_t n_short_reads;  /* reads with < k genes  (construct_graph.py:53-55)          */
  int64_t n_nodes;        /* distinct canonical gene-mers ever inserted                */
  int64_t n_edges;        /* directed edges ever inserted (reference _edges entries)   */
  int64_t n_pairs;        /* undirected edge classes (each gives 2 edges, 1 if loop)   */
  int64_t n_components;
  int64_t n_live_nodes;   /* after filter / removals                                   */
  int64_t n_live_edges;
  int64_t n_reads_to_correct;
  int64_t node_table_slots;
  int64_t edge_table_slots;  /* hashed slots (exact keys: one directly addressed slot per node id on top)   */
  int64_t build_retries;  /* table growth / fingerprint-collision rebuilds             */
  int32_t k;
  int32_t two_v;
  int32_t exact_keys;     /* 1: the build keyed nodes by the packed tuple itself (k * bits */
                          /* per token <= 94); 2: the same 16-byte slots keyed by a 94-bit */
                          /* fingerprint, every window verified against its key's first    */
                          /* occurrence in the token stream; 0: 32-byte slots, verified    */
                          /* 64-bit fingerprint (>= 2^29 tokens, the merge, AMG_KEY_MODE=fp) */
  int32_t derived;        /* 1: amg_build made this graph from the previous one's live part instead of from the */
                          /* reads (the reads were that graph's reads with some dropped or cut to their live    */
                          /* windows by amg_correct_reads, nothing re-threaded: the result is the same graph)   */
} amg_counts_t;

/* ---- lifetime ------------------------------------------------------------------ */
int amg_create(int device, amg_ctx** out);
int amg_destroy(amg_ctx* ctx);
const char* amg_last_error(void);
int amg_sync(amg_ctx* ctx);
/* the HIP stream every kernel of this ctx is launched on (for hipEvent timing) */
void* amg_stream(amg_ctx* ctx);

/* ---- input: replaces the readDict / gene_positions arguments of
 *      GeneMerGraph.__init__ (construct_graph.py:31) ------------------------------- */
/* on_device: 0 = host pointers (copied H2D); 1 = device pointers on ctx's device (copied D2D);
 * 2 = device pointers BORROWED without a copy: the caller keeps the memory alive and unchanged — reads
 * (tokens, read_offsets, read lengths) until the next amg_set_reads / amg_set_read_lengths or amg_adopt_corrected,
 * POSITIONS until the next amg_set_positions / amg_set_reads (see there: longer than in releases before round 2,
 * where amg_adopt_corrected let go of them too); borrowed memory is never written.
 * Stream contract for on_device != 0: the library reads the arrays on ITS OWN stream (amg_stream),
 * which is not ordered against the stream that produced them — the caller synchronises its
 * producer stream (or device) before the call.
 * Validation: host inputs are checked here; device inputs are checked by the first kernel of
 * amg_build (read_offsets[0] == 0, non-decreasing, read_offsets[n_reads] == number of tokens;
 * 0 <= token < two_v), which then fails with AMG_E_ARG before anything is indexed with them. */
int amg_set_reads(amg_ctx* ctx, const int32_t* tokens, const int64_t* read_offsets,
                  int64_t n_reads, int32_t two_v, int on_device);
/* optional: per-gene [start,end] and per-read sequence length, used only by
 * amg_correct_reads to carry gene positions (construct_graph.py:1311-1328,1669-1691).
 * read_len may be NULL when no read will need position inference.
 * Call it after amg_set_reads of the same read set.  BORROWED position arrays (on_device = 2) are
 * read until the next amg_set_positions / amg_set_reads, across amg_adopt_corrected: corrected
 * reads keep pointing at the positions of their unchanged genes instead of copying them. */
int amg_set_positions(amg_ctx* ctx, const int64_t* gene_start, const int64_t* gene_end,
                      const int64_t* read_len, int on_device);
/* the same from 32-bit position arrays (read coordinates fit; half the bytes over PCIe): widened on the device into
 * the engine's own arrays — on_device 0 (host) or 1 (device), never borrowed; read_len stays 64-bit */
int amg_set_positions32(amg_ctx* ctx, const int32_t* gene_start, const int32_t* gene_end,
                        const int64_t* read_len, int on_device);

/* per-read sequence length only (len(fastq[read]["sequence"]), construct_graph.py:1685);
 * may be called any time before amg_correct_reads */
int amg_set_read_lengths(amg_ctx* ctx, const int64_t* read_len, int on_device);

/* ---- build: GeneMerGraph.__init__ (construct_graph.py:31-102), i.e.
 *      Read.get_geneMers (construct_read.py:37-59), define_geneMer
 *      (construct_gene_mer.py:42-56), add_node (:196-212), add_node_to_read (:165-178),
 *      add_edge (:300-324), assign_component_ids (:920-927) --------------------------- */
int amg_build(amg_ctx* ctx, int32_t k);
/* amg_build followed by amg_filter(min_node_cov, min_edge_cov) — GeneMerGraph.__init__ + filter_graph
 * (construct_graph.py:31-102, :523-540), the opening of every cleaning iteration (graph_utils.py:147-149) — with the
 * filter applied on the way where the key layout allows it: nodes / edge classes below the thresholds are never
 * ranked, stored or joined by edges.  Live nodes, live edges, coverages, list orders, masked windows and the reads
 * queued for correction are those of the two separate calls, and so are the COMPONENT ids: the reference labels
 * components once, in __init__, on the graph of all nodes (construct_graph.py:101-102) and keeps the labels through
 * filter_graph; the one-pass build makes the same labels from the per-window claims of its node pass (on first use).
 * Node / edge ids number what is present, in first-seen order (with the two calls the filtered nodes keep their ids
 * with alive = 0). */
int amg_build_filtered(amg_ctx* ctx, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov);
/* amg_build of the SAME reads for n (<= 8) gene-mer sizes, graph i on ctxs[i] with k = ks[i] — choose_kmer_size's
 * builds for k = 3, 5, ..., 15 (graph_utils.py:258-296).  The reads are those of ctxs[0] (amg_set_reads there first) and
 * are on the device once: the other ctxs BORROW its device arrays, which stay valid until ctxs[0]'s reads change or it
 * is destroyed.  Positions are not shared (set them per ctx if a correction is to follow).  One device; the call is
 * synchronous.  Each graph is exactly what amg_build(ctxs[i], ks[i]) builds — and is built by it. */
int amg_build_multi(amg_ctx* const* ctxs, const int32_t* ks, int32_t n);
int amg_counts(amg_ctx* ctx, amg_counts_t* out);
/* amg_build leaves component ids (assign_component_ids, construct_graph.py:920-927) and the per-node
 * forward / backward edge lists (construct_node.py:79-101) to the first call that needs them
 * (amg_counts, amg_get_nodes with `component`, amg_get_node_adj, tip clipping, component filter);
 * amg_finalize computes both now, so that the ctx holds everything GeneMerGraph.__init__ leaves behind. */
int amg_finalize(amg_ctx* ctx);
/* reads / genes of the current read set, without touching the device (amg_counts recounts the
 * live flags): len(readDict), sum(len(genes)) */
int amg_sizes(amg_ctx* ctx, int64_t* n_reads, int64_t* n_tokens);
/* array sizes of the built graph (nodes / directed edges ever inserted) and its k, without touching
 * the device: what the read-back calls below need to size their buffers */
int amg_graph_sizes(amg_ctx* ctx, int64_t* n_nodes, int64_t* n_edges, int32_t* k);

/* ---- graph read-back (any pointer may be NULL to skip that column) ---------------- */
/* nodes in id order; canon_tokens is [n_nodes * k]; first_token = token index of the
 * node's first occurrence; first_dir = direction of that occurrence (+1/-1), i.e.
 * node.get_geneMer().get_geneMerDirection() (construct_node.py:16-18). */
int amg_get_nodes(amg_ctx* ctx, int32_t* canon_tokens, uint32_t* coverage,
                  int64_t* first_token, int8_t* first_dir, int32_t* component,
                  uint8_t* alive);
/* directed edges in id order: Edge(sourceNode, targetNode, sourceNodeDirection,
 * targetNodeDirection).edgeCoverage (construct_edge.py:31-37) */
int amg_get_edges(amg_ctx* ctx, int32_t* src, int32_t* tgt, int8_t* sdir, int8_t* tdir,
                  uint32_t* coverage, uint8_t* alive);
/* per-token node id (-1: no window starts here, -2: node removed => None in the
 * reference's _readNodes) and direction (+1/-1, 0 where no node)
 * — get_readNodes / get_readNodeDirections (construct_graph.py:117-123) */
int amg_get_read_nodes(amg_ctx* ctx, int32_t* tok_node, int8_t* tok_dir);
/* the same for a FEW reads: row r = the windows first_token[r] .. first_token[r] + (out_start[r + 1] - out_start[r]) - 1
 * (token indices of the read set), written to node_ids[out_start[r] ..) — out_start[0] = 0, out_start[n_rows] entries
 * in all; what read-path clustering asks for the reads of a gene's nodes instead of the whole per-window array */
int amg_get_read_nodes_rows(amg_ctx* ctx, const int64_t* first_token, const int64_t* out_start, int64_t n_rows,
                            int32_t* node_ids);
/* adjacency, 2 rows per node: row 2*n = forwardEdgeHashes, row 2*n+1 =
 * backwardEdgeHashes of node n, edge ids in list order (construct_node.py:79-101);
 * offsets[2*n_nodes + 1], edge_ids[n_edges].  Dead edges stay listed; test `alive`. */
int amg_get_node_adj(amg_ctx* ctx, int64_t* offsets, int32_t* edge_ids);
/* Node.listOfReads (construct_node.py:64-67): ordered, de-duplicated read indices.
 * Call with read_idx == NULL to get offsets (and thereby the total) first. */
int amg_get_node_reads(amg_ctx* ctx, int64_t* offsets, int32_t* read_idx);

/* ---- coverage filter and removals -------------------------------------------------- */
/* filter_graph(minNodeCoverage, minEdgeCoverage) (construct_graph.py:523-540) */
int amg_filter(amg_ctx* ctx, uint32_t min_node_cov, uint32_t min_edge_cov);
/* remove_node for each listed node (construct_graph.py:463-484) */
int amg_remove_nodes(amg_ctx* ctx, const int32_t* node_ids, int64_t n);
/* remove_edge for each listed DIRECTED edge (construct_graph.py:409-428): the edge leaves the graph and its source
 * node's forward / backward list; its reverse twin is an edge of its own */
int amg_remove_edges(amg_ctx* ctx, const int32_t* edge_ids, int64_t n);
/* remove_short_linear_paths(min_length) (construct_graph.py:679-720); protect[n] != 0
 * keeps node n (the AMR_nodes exemption); removed_ids may be NULL. */
int amg_remove_short_linear_paths(amg_ctx* ctx, int32_t min_length, const uint8_t* protect,
                                  int64_t* n_removed, int32_t* removed_ids);
/* remove_low_coverage_components(min) (construct_graph.py:950-958) */
int amg_remove_low_coverage_components(amg_ctx* ctx, uint32_t min_component_coverage);
/* get_reads_to_correct() as a 0/1 flag per read (construct_graph.py:148-150) */
int amg_get_reads_to_correct(amg_ctx* ctx, uint8_t* flags);

/* ---- per-read correction: correct_reads (construct_graph.py:1123-1396,1433-1480) --- */
/* Re-threads every marked read through the filtered graph on the device and leaves
 * the corrected read set in ctx.  Outputs: reads kept (marked reads whose nodes were
 * all removed, and reads without any window, are dropped) and their total genes. */
int amg_correct_reads(amg_ctx* ctx, int64_t* n_out_reads, int64_t* n_out_tokens);
/* corrected CSR + origin: orig_read[i] = index (in the current read set) of corrected
 * read i; changed[i] != 0 if read i was re-threaded (its gene list is a new list);
 * gene_start/gene_end NULL unless positions were set. */
int amg_get_corrected(amg_ctx* ctx, int32_t* tokens, int64_t* read_offsets,
                      int32_t* orig_read, uint8_t* changed, int64_t* gene_start,
                      int64_t* gene_end);
/* The same with 32-bit positions and no byte moved that the caller already has.  pos_src[i] >= 0: the positions of
 * corrected read i are the caller's own (the arrays handed to amg_set_positions / amg_set_positions32) from index
 * pos_src[i] on — a read left alone or only trimmed; pos_src[i] < 0: they are new_start / new_end from index
 * -1 - pos_src[i] on (what the position carry-over produced, construct_graph.py:1314-1328, 1669-1691).
 * new_start / new_end: room for the corrected genes (n_out_tokens); *n_new entries are written.  A position beyond
 * 32 bits is an error (amg_get_corrected returns 64-bit positions). */
int amg_get_corrected32(amg_ctx* ctx, int32_t* tokens, int64_t* read_offsets, int32_t* orig_read,
                        uint8_t* changed, int64_t* pos_src, int32_t* new_start, int32_t* new_end,
                        int64_t* n_new);
/* The positions of amg_get_corrected — every corrected read's, laid end to end like the genes — as 32-bit values, gathered
 * on the device: read coordinates fit, and 64-bit position arrays are four fifths of the bytes a correction hands back
 * (construct_graph.py:1123-1134 returns them per read).  gene_start / gene_end: room for n_out_tokens values each.  A
 * position beyond 32 bits is an error, AMG_E_ARG (amg_get_corrected returns 64-bit positions). */
int amg_get_corrected_positions32(amg_ctx* ctx, int32_t* gene_start, int32_t* gene_end);
/* the corrected read set becomes the current read set (device resident; the next
 * amg_build runs on it) — the rebuild of graph_utils.py:147-150,165 */
int amg_adopt_corrected(amg_ctx* ctx);
/* the corrected set of `src` (after amg_correct_reads there) becomes the read set of `dst` — genes, read offsets, read
 * lengths and the positions, gathered into flat arrays of dst's own — device to device: correct_reads followed by the
 * next GeneMerGraph(...) (graph_utils.py:147-150, :165) without the reads leaving the GPU.  src keeps its corrected
 * set; dst == src is amg_adopt_corrected.  Same device; synchronous. */
int amg_set_reads_from_corrected(amg_ctx* dst, amg_ctx* src);

/* ---- read-path clustering support: batched exact sub-list search
 *      (is_sublist / find_sublist_indices, construct_graph.py:1957-1966,2117-2123;
 *      Tree.find_all call sites path_finding_utils.py:244,290) ------------------------ */
/* patterns are CSR token lists; each is searched forward in every read's TOKENS
 * (which = 0) or NODE ids (which = 1).  Two-call: hit_read == NULL returns counts in
 * hit_offsets[n_pat + 1]; then (hit_read, hit_pos) get one entry per occurrence,
 * ordered by (pattern, read, position). */
int amg_match_patterns(amg_ctx* ctx, int which, const int32_t* pat, const int64_t* pat_offsets,
                       int64_t n_pat, int64_t* hit_offsets, int32_t* hit_read,
                       int32_t* hit_pos);

/* ---- bubble popping support: scaled MinHash of nucleotide segments, the
 *      sketches sourmash.MinHash(n=0, ksize, scaled).add_sequence(seq, force=True) builds at
 *      construct_graph.py:1567-1575 (per path, ksize 9, scaled 1) and :2148-2158 (per node, ksize 11,
 *      scaled 10).  sourmash is a third-party dependency of the reference (pyproject.toml:28); its
 *      published definition is implemented: upper-cased bases, windows holding a character outside ACGT
 *      skipped, canonical k-mer = min(k-mer, reverse complement), hash = first 64 bits of
 *      MurmurHash3_x64_128(seed 42), kept when <= round(2^64 / scaled) (all for scaled = 1).
 *      bases[seg_off[s] .. seg_off[s+1]) is segment s and belongs to sketch seg_set[s] (HOST arrays).
 *      Output: one (sketch id, hash) pair per kept k-mer occurrence, unordered; *n_out = their number
 *      (call with out_set = NULL for the count). ------------------------------------------------ */
int amg_minhash(amg_ctx* ctx, const uint8_t* bases, const int64_t* seg_off, const int32_t* seg_set,
                int64_t n_seg, int32_t ksize, uint64_t scaled, int32_t* out_set, uint64_t* out_hash,
                int64_t cap, int64_t* n_out);

/* ---- bubble popping on device ids (row f1, correct_low_coverage_paths construct_graph.py:2196-2250).
 *      amg_junction_paths: the junctions of the live graph (identify_potential_bubble_starts :2252-2265: a live node
 *      with more than one live edge in its forward list is junction (node, +1), in its backward list (node, -1); in
 *      node order, forward before backward) and every path get_all_paths_between_junctions_in_component (:2066-2098)
 *      would put into its set BEFORE it picks the smaller of the path and its mirror image: for every start junction,
 *      for every stop junction in list order, the paths new_find_paths_between_nodes(start, stop, max_distance) (:2292-
 *      2342) returns that arrive at the stop through its junction side, in the order it returns them, when there are
 *      at least two.  (One search per start instead of one per pair: amira_amd/csrc/amg_bubbles.hip.)  Paths never
 *      leave a component, so the caller splits the list by the component of the start.
 *      sizes[0] = junctions, [1] = paths, [2] = nodes over all paths, [3] = flags: bit 0 — a path ends at a junction
 *      over two nodes with more than one edge between them, where the reference fails (:1515-1523 on a list); bit 1 —
 *      a search was abandoned after 2^24 steps.  With a flag set the paths are not to be used.
 *      amg_get_junction_paths copies out: junction_node / junction_dir [junctions], path_start [paths] (index of the
 *      start junction), path_off [paths + 1], path_node / path_dir [sizes[2]]; any pointer may be NULL. ---------- */
int amg_junction_paths(amg_ctx* ctx, int32_t max_distance, int64_t* sizes /*[4]*/);
int amg_get_junction_paths(amg_ctx* ctx, int32_t* junction_node, int8_t* junction_dir, int32_t* path_start,
                           int64_t* path_off, int32_t* path_node, int8_t* path_dir);
/* the reads' nucleotide sequences (fastq_data[read]["sequence"]) resident on a device: seq[i] points at len[i] bytes
 * (HOST memory, copied during the call).  One handle serves every graph of a cleaning run. */
typedef struct amg_seqs amg_seqs;
int amg_seqs_create(int32_t device, const char* const* seq, const int64_t* len, int64_t n, amg_seqs** out);
int amg_seqs_destroy(amg_seqs* seqs);
/* The sketches bubble popping compares (:2148-2194, :1747-1786): a node's sketch is the scaled MinHash (ksize, scaled;
 * amg_minhash's definition) of sequence[start of the window's first gene : end of its last gene + 1] over every window
 * of the ctx's reads that sits on the node, a path's sketch the union over its nodes.  Paths are lists of node ids
 * (path_off [n_paths + 1], path_node); read row r of the ctx has sequence row_to_seq[r] of `seqs` (NULL: r).
 * Out: sketch_size[p] = hashes in path p's sketch; common[q] = hashes the sketches of paths pair_a[q] and pair_b[q]
 * share.  Needs the gene positions of the ctx's reads (amg_set_positions*) as handed over, i.e. a ctx whose reads
 * were set, not adopted from its own correction. */
int amg_path_sketch_overlaps(amg_ctx* ctx, const amg_seqs* seqs, const int32_t* row_to_seq, int32_t ksize,
                             uint64_t scaled, int64_t n_paths, const int64_t* path_off, const int32_t* path_node,
                             int64_t n_pairs, const int32_t* pair_a, const int32_t* pair_b, int64_t* sketch_size,
                             int64_t* common);

/* needleman_wunsch (construct_graph.py:1433-1480) of two short lists of interned genes on the host: match 1, mismatch 0,
 * gap -1, ties UP > LEFT > DIAG as the reference's max over (score, pointer) gives them.  ops (room for n + m), in
 * alignment order: 0 = (x gene, y gene), 1 = (x gene, "*"), 2 = ("*", y gene). */
int amg_nw_align(const int32_t* x, int32_t n, const int32_t* y, int32_t m, int8_t* ops, int32_t* n_ops);

/* ---- multi-GPU: read-sharded build with a key-owner table merge — the single-graph result of
 *      build_multiprocessed_graph + merge_graphs (graph_utils.py:94-124) at cores = 1.
 *      One process per GPU; every rank holds a contiguous shard of the reads in its ctx (amg_set_reads), rank r the
 *      reads after rank r - 1's.  amg_dist_merge replaces amg_build / amg_build_filtered: every rank ends with the
 *      graph GeneMerGraph would build from ALL reads (node / edge ids, coverages, list orders) and with the node ids
 *      of its own reads' windows; filters and clipping then run identically everywhere, amg_correct_reads on the
 *      rank's reads.  The library runs the device phases AND the exchanges between them (RCCL on the ctx's stream:
 *      grouped ncclSend / ncclRecv all-to-alls to the key owners and back, ncclAllGather of the held records;
 *      librccl is opened by amg_dist_init, libamg.so does not link it).  Collective: every rank calls with the same
 *      arguments.  A failing rank tells its peers in the next count exchange (nobody waits for a rank that has left);
 *      two gene-mers under one 64-bit merge key make every rank repeat the build with the next seed.
 *      Phases and record formats: amira_amd/csrc/amg_dist.hip, DESIGN.md section 6. ------------------------------ */
/* 128 bytes that name a communicator (ncclGetUniqueId): made on one rank, handed to every rank by the caller's own
 * means (a file, a socket, MPI, torch.distributed's store), then passed to amg_dist_init by all of them */
int amg_dist_unique_id(void* out, int32_t bytes);
int amg_dist_init(amg_ctx* ctx, const void* unique_id, int32_t rank, int32_t world);
/* min_node_cov / min_edge_cov > 1 fuse filter_graph (construct_graph.py:523-540) into the merge: owners answer "dropped"
 * for nodes / edge classes below the thresholds, so the (often 10x larger) set of low-coverage nodes is never
 * replicated.  The state then equals the merged build followed by amg_filter for everything the correction reads (live
 * nodes and edges, their coverages and list orders, masked windows, reads to correct); ids number the survivors only
 * and component ids are those of the filtered graph.  (1, 1): plain build. */
int amg_dist_merge(amg_ctx* ctx, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov);
/* releases the communicator and the merge buffers (amg_destroy does it too) */
int amg_dist_finalize(amg_ctx* ctx);

/* The same build with the exchanges left to the CALLER (another transport: the tests run ranks as processes over
 * gloo, staging through the host): amg_dist_init_external instead of amg_dist_init, then amg_dist_merge_begin and
 * amg_dist_merge_next until it returns 0.  A return of 1 means: perform *out on every rank, then call again.
 * Buffers are DEVICE pointers owned by the ctx; counts are host arrays valid until the next call. */
enum { AMG_XFER_ALL_TO_ALL = 1, AMG_XFER_ALL_GATHER = 2 };
typedef struct amg_xfer {
  int32_t kind;               /* AMG_XFER_*                                                                 */
  int32_t elem_bytes;         /* bytes per element                                                          */
  const void* send;           /* all-to-all: elements for rank 0, then rank 1, ...; all-gather: `count` elements */
  void* recv;                 /* all-to-all: elements from rank 0, then rank 1, ...; all-gather: world * count   */
  const int64_t* send_counts; /* all-to-all: elements for every rank [world]                                */
  const int64_t* recv_counts; /* all-to-all: elements from every rank [world]                               */
  int64_t count;              /* all-gather: elements every rank contributes                                */
} amg_xfer;
int amg_dist_init_external(amg_ctx* ctx, int32_t rank, int32_t world);
int amg_dist_merge_begin(amg_ctx* ctx, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov);
int amg_dist_merge_next(amg_ctx* ctx, amg_xfer* out);
/* `world` EMULATED ranks: the ctxs of one process on one device, rank r = ctxs[r], the exchanges as device copies —
 * the device phases are exactly those of the N-GPU run (tests on a one-GPU box, tools/scaling_model.py) */
int amg_dist_merge_local(amg_ctx* const* ctxs, int32_t world, int32_t k, uint32_t min_node_cov, uint32_t min_edge_cov);
/* bytes between a device pointer of the ctx's device and the host, ordered after the ctx's stream and synchronous
 * (what a host-staged transport needs around its collectives) */
int amg_copy_d2h(amg_ctx* ctx, const void* device_ptr, void* host_ptr, int64_t bytes);
int amg_copy_h2d(amg_ctx* ctx, void* device_ptr, const void* host_ptr, int64_t bytes);
/* counters of the ctx's merged builds since the last reset: out[0] host waits on exchanged counts, [1] exchanges,
 * [2] most bytes of records sent to ONE peer, [3] the same of replies, [4] bytes contributed to the all-gathers,
 * [5] builds repeated after a merge-key collision, [6] bytes of records sent to all peers, [7] builds made from the
 * previous merged graph's live part (no rank had re-threaded a read: two exchanges instead of ten) */
int amg_dist_stats(amg_ctx* ctx, int64_t* out, int32_t reset);
/* synchronised wall time per phase of the merge driver since the last call (on = 1: keep measuring — every phase is
 * bracketed by stream synchronisations —, 0: stop); names[i] are static strings; returns the number of phases */
int amg_dist_phase_ms(amg_ctx* ctx, int32_t on, const char** names, double* ms, int32_t cap);

/* ---- native front-end / write-back (host code, no GPU needed):
 *      gene-call JSON {"read": ["+geneA", "-geneB", ...]} as dumped / reloaded by the reference
 *      (__main__.py:464-496, result_utils.py:1260-1264) -> vocabulary ranked by the reference's
 *      gene hash (construct_gene.py:5-10) + CSR tokens; positions JSON {"read": [[s, e], ...]};
 *      corrected CSR -> JSON ------------------------------------------------------------ */
typedef struct amg_calls amg_calls;
int amg_calls_load_json(const char* path, amg_calls** out);
int amg_calls_counts(amg_calls* calls, int64_t* n_reads, int64_t* n_tokens, int64_t* n_genes,
                     int64_t* names_bytes, int64_t* ids_bytes);
/* tokens[n_tokens], read_offsets[n_reads + 1], NUL-separated gene names in RANK order
 * (names_bytes), NUL-separated read ids in file order (ids_bytes), 32-byte sha256 per gene */
int amg_calls_get(amg_calls* calls, int32_t* tokens, int64_t* read_offsets, char* gene_names,
                  char* read_ids, uint8_t* gene_hashes);
/* *out = 1 when some gene name of the file held a blank (names are stored with '_' in its place, construct_gene.py:54-56:
 * a caller that compares RAW names, as pre_processing.py:54 does, then has to look at the file itself) */
int amg_calls_has_blanks(amg_calls* calls, int32_t* out);
int amg_calls_load_positions_json(amg_calls* calls, const char* path, int64_t* gene_start,
                                  int64_t* gene_end);
int amg_calls_write_json(const char* path, const int32_t* tokens, const int64_t* read_offsets,
                         int64_t n_reads, const char* gene_names, int64_t n_genes,
                         const char* read_ids);
/* {"read": [[start, end], ...]} as json.dumps(gene_position_dict) writes it (result_utils.py:1260-1264, second file) */
int amg_calls_write_positions_json(const char* path, const int64_t* gene_start, const int64_t* gene_end,
                                   const int64_t* read_offsets, int64_t n_reads, const char* read_ids);
/* the same from 32-bit position arrays (what amg_get_corrected_positions32 hands back) */
int amg_calls_write_positions_json32(const char* path, const int32_t* gene_start, const int32_t* gene_end,
                                     const int64_t* read_offsets, int64_t n_reads, const char* read_ids);
/* pre_processing.py:44-63 (process_pandora_json keeps the genes of interest the reads contain, in the order the reads
 * first show them): first_index[i] = index of the first token whose gene has rank wanted_ranks[i] (either strand),
 * -1 when no token does */
int amg_calls_first_use(const int32_t* tokens, int64_t n_tokens, int32_t two_v, const int32_t* wanted_ranks,
                        int64_t n_wanted, int64_t* first_index);
int amg_calls_free(amg_calls* calls);
/* the loader and the writers keep their gigabyte-sized work buffers (file bytes, output text) for the next call
 * instead of faulting fresh pages in every time; this gives them back to the system (released_bytes may be NULL) */
int amg_calls_trim(int64_t* released_bytes);

/* ---- read-path clustering, block search (host code; construct_graph.py:2725-2749 get_full_paths and
 *      path_finding_utils.py:88-247 process_anchors / get_blocks_from_subtree / generate_contexts /
 *      generate_full_paths / cluster_*_adjacent_paths / update_full_blocks) --------------------------
 * For one gene of interest: the keys of `full_blocks` in the reference's insertion order, as tuples of DEVICE
 * NODE IDS (-2 = None).  seq / seq_off: per-window node ids of the reads that hold the gene, reads in the
 * iteration order of the reference's set of read names; anchors: node ids in the iteration order of the
 * reference's anchor set; anchor_rank[i]: rank of anchor i's node hash among the anchors; py_hash[id]:
 * Python's hash() of node id's 256-bit hash for every id in seq; none_hash: hash(None).  The reference's
 * containers are Python sets of tuples whose iteration order decides the order of the result: CPython's set
 * and tuple hash are reproduced (amg_pyset_script / amg_py_tuple_hash expose the emulation to the check that
 * amira_amd/clustering.py runs against the interpreter's own sets before relying on it). */
typedef struct amg_blocks amg_blocks;
int amg_cluster_full_blocks(const int32_t* seq, const int64_t* seq_off, int64_t n_reads,
                            const int32_t* anchors, const int32_t* anchor_rank, int32_t n_anchors,
                            const int64_t* py_hash, int64_t n_nodes, int64_t none_hash, amg_blocks** out);
/* the read loop of get_AMR_anchors (construct_graph.py:2644-2676) over the same seq / seq_off: per AMR node
 * out[4 i ..] = {stopped at an anchor occurrence, all(singletons), flags, True flags}; read_order: the reads of seq
 * in ascending order of their rows in the read set (NULL: seq is in that order already) */
int amg_cluster_anchor_stats(const int32_t* seq, const int64_t* seq_off, int64_t n_reads, const int64_t* read_order,
                             const int32_t* amr_ids, int32_t n_amr, int64_t n_nodes, int32_t* out);
int amg_cluster_blocks_sizes(const amg_blocks* blocks, int64_t* n_blocks, int64_t* n_ids);
int amg_cluster_blocks_get(const amg_blocks* blocks, int64_t* block_off, int32_t* block_ids);
int amg_cluster_blocks_free(amg_blocks* blocks);
int64_t amg_py_tuple_hash(const int64_t* item_hashes, int64_t n);
/* ops: n_ops (op, a, b) triples over n_sets sets — 0: sets[a].add(key b); 1: sets[a].update(sets[b]);
 * 2: sets[a] = set(); 3: sets[a] = {k for k in sets[b]}.  key_hash[k] = hash of key k.  Out: every set's
 * iteration order (out_off[n_sets + 1], out_keys). */
int amg_pyset_script(const int32_t* ops, int64_t n_ops, const int64_t* key_hash, int32_t n_sets,
                     int32_t* out_keys, int64_t* out_off);

/* ---- per-stage device time of the last call, for bench.py ------------------------- */
/* names[i] points at static strings; returns the number of stages (<= cap). */
int amg_last_timings(amg_ctx* ctx, const char** names, float* ms, int cap);
/* Stage timing brackets every stage with two HIP events (~5 us of stream idle each, ~90 per cleaning
 * sweep): on by default (AMG_TIMING=0 in the environment turns it off), switched per ctx here.  With it
 * off amg_last_timings returns 0 stages. */
int amg_set_timing(amg_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* AMG_H */
