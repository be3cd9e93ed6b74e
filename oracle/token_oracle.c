/*
 * token_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C, strictly sequential restatement of GeneMerGraph.__init__
 * (reference construct_graph.py:31-102) in TOKEN space, for checks at sizes the
 * pure-Python oracle cannot reach.  It walks reads in order and windows in order and
 * inserts into ordinary chained hash maps exactly as the reference's dicts would be
 * filled, so "first seen" is literal program order:
 *   - canonical orientation: lexicographic min of the window and its reverse complement
 *     (construct_gene_mer.py:15-39) with rc[j] = two_v - 1 - w[k-1-j];
 *   - node coverage += 1 per window (construct_graph.py:71,86,100), node id = insertion
 *     order (:188-190), first direction = direction of the first window (construct_node.py:6);
 *   - per adjacency the two directed edges E1 = (A,B,dA,dB), E2 = (B,A,-dB,-dA)
 *     (:246-262) fall into classes (src, tgt, dS*dT) (construct_edge.py:104-124); the first
 *     object of a class is kept with its directions, coverage += 1 per event (:81-82).
 * Checked against oracle/amira_oracle (which is pinned to the reference by goldens) in
 * tests/test_token_oracle.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t n_nodes, n_edges, n_windows, n_short;
  int32_t* node_tokens; /* n_nodes * k */
  uint32_t* node_cov;
  int8_t* node_first_dir;
  int32_t* edge_src; int32_t* edge_tgt; int8_t* edge_sdir; int8_t* edge_tdir; uint32_t* edge_cov;
} tok_graph;

typedef struct { int64_t cap, n; int64_t* head; int64_t* next; } chain;

static uint64_t mix(uint64_t x) { x ^= x >> 31; x *= 0x7fb5d329728ea185ull; x ^= x >> 27; x *= 0x81dadef4bc2dd44dull; x ^= x >> 33; return x; }

static void chain_init(chain* c, int64_t cap) {
  c->cap = 1; while (c->cap < cap * 2 + 16) c->cap <<= 1;
  c->head = (int64_t*)malloc(sizeof(int64_t) * c->cap);
  for (int64_t i = 0; i < c->cap; ++i) c->head[i] = -1;
  c->next = NULL; c->n = 0;
}

/* Builds the graph; tok_node / tok_dir are token-indexed like the engine's outputs
 * (-1 / 0 where no window starts).  Returns 0, or -4 on a palindromic window. */
int token_oracle_build(const int32_t* tokens, const int64_t* read_off, int64_t n_reads, int32_t k,
                       int32_t two_v, int32_t* tok_node, int8_t* tok_dir, tok_graph* out) {
  int64_t T = read_off[n_reads];
  memset(out, 0, sizeof(*out));
  int64_t cap_nodes = T > 0 ? T : 1;
  out->node_tokens = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap_nodes * k);
  out->node_cov = (uint32_t*)calloc((size_t)cap_nodes, sizeof(uint32_t));
  out->node_first_dir = (int8_t*)malloc((size_t)cap_nodes);
  int64_t cap_edges = 2 * cap_nodes + 2;
  out->edge_src = (int32_t*)malloc(sizeof(int32_t) * cap_edges);
  out->edge_tgt = (int32_t*)malloc(sizeof(int32_t) * cap_edges);
  out->edge_sdir = (int8_t*)malloc((size_t)cap_edges);
  out->edge_tdir = (int8_t*)malloc((size_t)cap_edges);
  out->edge_cov = (uint32_t*)calloc((size_t)cap_edges, sizeof(uint32_t));
  chain nodes, edges;
  chain_init(&nodes, cap_nodes);
  chain_init(&edges, cap_edges);
  nodes.next = (int64_t*)malloc(sizeof(int64_t) * cap_nodes);
  edges.next = (int64_t*)malloc(sizeof(int64_t) * cap_edges);
  for (int64_t t = 0; t < T; ++t) { tok_node[t] = -1; tok_dir[t] = 0; }
  int32_t canon[64];
  const int32_t flip = two_v - 1;
  for (int64_t r = 0; r < n_reads; ++r) {
    int64_t a = read_off[r], b = read_off[r + 1], n = (b - a) - k + 1;
    if (n <= 0) { out->n_short++; continue; }
    int64_t prev = -1; int prev_dir = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t* w = tokens + a + i;
      int dir = 0;
      for (int j = 0; j < k && !dir; ++j) { int32_t x = w[j], y = flip - w[k - 1 - j]; if (x != y) dir = x < y ? 1 : -1; }
      if (!dir) return -4;
      uint64_t h = 1469598103934665603ull;
      for (int j = 0; j < k; ++j) { canon[j] = dir > 0 ? w[j] : flip - w[k - 1 - j]; h = mix(h ^ (uint64_t)(uint32_t)canon[j]); }
      int64_t slot = (int64_t)(h & (uint64_t)(nodes.cap - 1)), id = nodes.head[slot];
      while (id >= 0 && memcmp(out->node_tokens + id * k, canon, sizeof(int32_t) * k) != 0) id = nodes.next[id];
      if (id < 0) {
        id = nodes.n++;
        memcpy(out->node_tokens + id * k, canon, sizeof(int32_t) * k);
        out->node_first_dir[id] = (int8_t)dir;
        nodes.next[id] = nodes.head[slot]; nodes.head[slot] = id;
      }
      out->node_cov[id]++; out->n_windows++;
      tok_node[a + i] = (int32_t)id; tok_dir[a + i] = (int8_t)dir;
      if (prev >= 0) {
        /* E1 = (prev, id, prev_dir, dir), E2 = (id, prev, -dir, -prev_dir) */
        int64_t es[2] = {prev, id}, et[2] = {id, prev};
        int ds[2] = {prev_dir, -dir}, dt[2] = {dir, -prev_dir};
        for (int q = 0; q < 2; ++q) {
          int sgn = ds[q] * dt[q];
          uint64_t eh = mix(mix((uint64_t)es[q] * 0x9E3779B97F4A7C15ull ^ (uint64_t)et[q]) ^ (uint64_t)(sgn + 2));
          int64_t eslot = (int64_t)(eh & (uint64_t)(edges.cap - 1)), e = edges.head[eslot];
          while (e >= 0 && !(out->edge_src[e] == es[q] && out->edge_tgt[e] == et[q] &&
                             out->edge_sdir[e] * out->edge_tdir[e] == sgn)) e = edges.next[e];
          if (e < 0) {
            e = edges.n++;
            out->edge_src[e] = (int32_t)es[q]; out->edge_tgt[e] = (int32_t)et[q];
            out->edge_sdir[e] = (int8_t)ds[q]; out->edge_tdir[e] = (int8_t)dt[q];
            edges.next[e] = edges.head[eslot]; edges.head[eslot] = e;
          }
          out->edge_cov[e]++;
        }
      }
      prev = id; prev_dir = dir;
    }
  }
  out->n_nodes = nodes.n; out->n_edges = edges.n;
  free(nodes.head); free(nodes.next); free(edges.head); free(edges.next);
  return 0;
}

void token_oracle_free(tok_graph* g) {
  free(g->node_tokens); free(g->node_cov); free(g->node_first_dir);
  free(g->edge_src); free(g->edge_tgt); free(g->edge_sdir); free(g->edge_tdir); free(g->edge_cov);
  memset(g, 0, sizeof(*g));
}
