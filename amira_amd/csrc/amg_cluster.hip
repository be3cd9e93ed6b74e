// amg_cluster.hip — host C++ (no kernels): the block search of the read-path clustering,
// path_finding_utils.py:88-247 of the reference as called from get_full_paths (construct_graph.py:2725-2749):
// for every ordered pair of anchors (a1, a2) the blocks a1 .. a2 on the reads, their upstream / downstream
// contexts, the greedy context clusters and the full blocks u + block + d that some read really holds.
//
// Everything runs on the DEVICE NODE IDS of the per-window array (-2 = a masked window, the reference's None).
// The one thing that cannot be replaced by arrays is the ORDER in which the reference's containers iterate: the
// contexts are Python sets of tuples of 256-bit node hashes, and the order in which such a set hands out tuples of
// equal length decides the order of the full blocks and, in the end, the numbering of the alleles.  That order is a
// pure function of the element hashes and of the sequence of set operations, so it is reproduced here exactly:
// PySet below is CPython's set (Objects/setobject.c, 3.8 - 3.12: open addressing, 9 linear probes, perturbation
// shift 5, growth 4x / 2x, set_merge's three cases), tuple_hash is CPython's tuple hash (xxHash-style, 3.8+) over
// the elements' own Python hashes, which the caller supplies (hash(int) and hash(None) are computed by Python).
// amira_amd/clustering.py checks this emulation against the running interpreter's own sets before it is used and
// keeps the pure-Python path otherwise.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>

#include "amg_internal.h"

namespace {

// ------------------------------------------------------------------ CPython's tuple hash
constexpr uint64_t XXPRIME_1 = 11400714785074694791ULL;
constexpr uint64_t XXPRIME_2 = 14029467366897019727ULL;
constexpr uint64_t XXPRIME_5 = 2870177450012600261ULL;

struct TupleHasher {
  uint64_t acc = XXPRIME_5;
  inline void push(uint64_t lane) {
    acc += lane * XXPRIME_2;
    acc = (acc << 31) | (acc >> 33);
    acc *= XXPRIME_1;
  }
  inline int64_t done(uint64_t len) const {
    uint64_t a = acc + (len ^ (XXPRIME_5 ^ 3527539ULL));
    if (a == (uint64_t)-1) return 1546275796;
    return (int64_t)a;
  }
};

// ------------------------------------------------------------------ CPython's set
constexpr int LINEAR_PROBES = 9;
constexpr int PERTURB_SHIFT = 5;

struct PySet {
  struct Entry {
    int32_t key;  // -1: unused
    int64_t hash;
  };
  std::vector<Entry> table;
  size_t mask = 7, fill = 0, used = 0;
  PySet() : table(8, Entry{-1, 0}) {}

  static void insert_clean(std::vector<Entry>& t, size_t mask, int32_t key, int64_t hash) {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      if (t[e].key < 0) {
        t[e] = Entry{key, hash};
        return;
      }
      if (i + LINEAR_PROBES <= mask) {
        for (int j = 0; j < LINEAR_PROBES; ++j) {
          ++e;
          if (t[e].key < 0) {
            t[e] = Entry{key, hash};
            return;
          }
        }
      }
      perturb >>= PERTURB_SHIFT;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }

  void resize(size_t minused) {
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    std::vector<Entry> nt(newsize, Entry{-1, 0});
    const size_t newmask = newsize - 1;
    for (size_t i = 0; i <= mask; ++i)
      if (table[i].key >= 0) insert_clean(nt, newmask, table[i].key, table[i].hash);
    table.swap(nt);
    mask = newmask;
    fill = used;
  }

  // set_add_entry (no deletions here: no dummy entries)
  void add(int32_t key, int64_t hash) {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
      do {
        if (table[e].key < 0) {  // unused
          ++fill;
          ++used;
          table[e] = Entry{key, hash};
          if (fill * 5 < mask * 3) return;
          resize(used > 50000 ? used * 2 : used * 4);
          return;
        }
        if (table[e].hash == hash && table[e].key == key) return;  // found_active (equal tuples are ONE interned key)
        ++e;
      } while (probes--);
      perturb >>= PERTURB_SHIFT;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }

  // set_merge: so.update(other), other a set
  void merge(const PySet& o) {
    if (&o == this || o.used == 0) return;
    if ((fill + o.used) * 5 >= mask * 3) resize((used + o.used) * 2);
    if (fill == 0 && mask == o.mask && o.fill == o.used) {
      table = o.table;
      fill = o.fill;
      used = o.used;
      return;
    }
    if (fill == 0) {
      fill = o.used;
      used = o.used;
      for (size_t i = 0; i <= o.mask; ++i)
        if (o.table[i].key >= 0) insert_clean(table, mask, o.table[i].key, o.table[i].hash);
      return;
    }
    for (size_t i = 0; i <= o.mask; ++i)
      if (o.table[i].key >= 0) add(o.table[i].key, o.table[i].hash);
  }

  template <class F>
  void each(F f) const {
    for (size_t i = 0; i <= mask; ++i)
      if (table[i].key >= 0) f(table[i].key, table[i].hash);
  }
};

// ------------------------------------------------------------------ interned tuples of node ids
struct Tuples {
  std::vector<int32_t> flat;
  std::vector<int64_t> off{0};
  std::vector<int64_t> hash;
  std::unordered_map<int64_t, std::vector<int32_t>> by_hash;
  const int64_t* py_hash = nullptr;
  int64_t none_hash = 0;

  inline uint64_t lane(int32_t id) const { return (uint64_t)(id >= 0 ? py_hash[id] : none_hash); }
  inline int len(int32_t t) const { return (int)(off[t + 1] - off[t]); }
  inline const int32_t* data(int32_t t) const { return flat.data() + off[t]; }

  int32_t intern(const int32_t* ids, int n) {
    TupleHasher h;
    for (int i = 0; i < n; ++i) h.push(lane(ids[i]));
    return intern_hashed(ids, n, h.done((uint64_t)n));
  }
  int32_t intern_hashed(const int32_t* ids, int n, int64_t hv) {
    auto& bucket = by_hash[hv];
    for (int32_t t : bucket)
      if (len(t) == n && (n == 0 || std::memcmp(data(t), ids, (size_t)n * sizeof(int32_t)) == 0)) return t;
    const int32_t t = (int32_t)hash.size();
    // (ids may point into flat itself: copy through a temporary when the vector might grow)
    if (ids >= flat.data() && ids < flat.data() + flat.size()) {
      std::vector<int32_t> tmp(ids, ids + n);
      flat.insert(flat.end(), tmp.begin(), tmp.end());
    } else {
      flat.insert(flat.end(), ids, ids + n);
    }
    off.push_back((int64_t)flat.size());
    hash.push_back(hv);
    bucket.push_back(t);
    return t;
  }
};

struct OrderedKeys {  // a Python dict used for its key order: insertion order, re-assignment keeps the place
  std::vector<int32_t> keys;
  std::unordered_map<int32_t, int32_t> index;
  int32_t put(int32_t k) {
    auto it = index.find(k);
    if (it != index.end()) return it->second;
    const int32_t i = (int32_t)keys.size();
    index.emplace(k, i);
    keys.push_back(k);
    return i;
  }
};

// path_finding_utils.py:11-64 (cluster_downstream_adjacent_paths / cluster_upstream_adjacent_paths): the keys of the
// returned dict, in its order
static std::vector<int32_t> greedy_clusters(const PySet& paths, Tuples& T, bool upstream) {
  std::vector<int32_t> ordered;
  paths.each([&](int32_t k, int64_t) { ordered.push_back(k); });
  std::stable_sort(ordered.begin(), ordered.end(), [&](int32_t a, int32_t b) { return T.len(a) > T.len(b); });
  std::vector<int32_t> reps;
  std::vector<PySet> members;
  auto fits = [&](int32_t p, int32_t c) {
    const int n = T.len(p), m = T.len(c);
    if (n == 0) return true;
    if (n > m) return false;
    const int32_t* pd = T.data(p);
    const int32_t* cd = T.data(c) + (upstream ? m - n : 0);
    return std::memcmp(pd, cd, (size_t)n * sizeof(int32_t)) == 0;
  };
  for (int32_t p : ordered) {
    int n_match = 0, which = -1;
    for (size_t g = 0; g < reps.size(); ++g)
      if (fits(p, reps[g])) {
        ++n_match;
        which = (int)g;
      }
    if (n_match == 0) {
      reps.push_back(p);
      members.emplace_back();
      members.back().add(p, T.hash[p]);
    } else if (n_match == 1) {
      members[which].add(p, T.hash[p]);
    }
  }
  OrderedKeys result;
  for (size_t g = 0; g < reps.size(); ++g) {
    int32_t best = -1;
    members[g].each([&](int32_t k, int64_t) {  // min(members, key=len): the first of the shortest in iteration order
      if (best < 0 || T.len(k) < T.len(best)) best = k;
    });
    result.put(best);
  }
  return result.keys;
}

struct Options {  // get_all_context_options (path_finding_utils.py:131-140) of one upstream / downstream list
  PySet set;
};

struct Result {
  std::vector<int64_t> off{0};
  std::vector<int32_t> ids;
};

struct Gene {
  const int32_t* seq;
  const int64_t* off;
  int64_t R;
  Tuples T;
  std::unordered_map<int32_t, std::unique_ptr<Options>> up_cache, down_cache;  // keyed by the interned list
  std::vector<std::vector<std::pair<int32_t, int32_t>>> occ;                  // per anchor: (read, position) ascending

  const Options& options(int32_t list_t, bool up) {
    auto& cache = up ? up_cache : down_cache;
    auto it = cache.find(list_t);
    if (it != cache.end()) return *it->second;
    auto o = std::make_unique<Options>();
    const int n = T.len(list_t);
    std::vector<int32_t> buf(T.data(list_t), T.data(list_t) + n);
    if (up) {  // {up[-i:] for i in range(1, len(up) + 1)}
      for (int i = 1; i <= n; ++i) {
        const int32_t t = T.intern(buf.data() + (n - i), i);
        o->set.add(t, T.hash[t]);
      }
    } else {  // {down[:i] for i in range(1, len(down) + 1)}: prefix hashes in one pass
      TupleHasher h;
      for (int i = 1; i <= n; ++i) {
        h.push(T.lane(buf[i - 1]));
        const int32_t t = T.intern_hashed(buf.data(), i, h.done((uint64_t)i));
        o->set.add(t, T.hash[t]);
      }
    }
    const int32_t e = T.intern(nullptr, 0);
    o->set.add(e, T.hash[e]);
    const Options& ref = *o;
    cache.emplace(list_t, std::move(o));
    return ref;
  }

  int32_t reversed_of(int32_t t) {
    const int n = T.len(t);
    std::vector<int32_t> buf(n);
    const int32_t* d = T.data(t);
    for (int i = 0; i < n; ++i) buf[i] = d[n - 1 - i];
    return T.intern(buf.data(), n);
  }

  // Tree.find_all(f) != []: f occurs in some read's node list, forwards or (reads with more than one distinct node
  // have a "<read>_reverse" entry; for the others both directions read the same) backwards.  Every candidate holds
  // anchor `a` (index ai), which sits at f[j].
  bool held_by_a_read(const int32_t* f, int n, int ai, int j) const {
    for (const auto& rp : occ[ai]) {
      const int64_t a = off[rp.first], b = off[rp.first + 1];
      const int64_t pos = a + rp.second;
      // forwards: f[0] at pos - j
      if (pos - j >= a && pos - j + n <= b && std::memcmp(seq + (pos - j), f, (size_t)n * sizeof(int32_t)) == 0) return true;
      // backwards: read[pos + j - x] == f[x]
      if (pos + j < b && pos + j - (n - 1) >= a) {
        bool ok = true;
        for (int x = 0; x < n && ok; ++x) ok = seq[pos + j - x] == f[x];
        if (ok) return true;
      }
    }
    return false;
  }
};

static void occurrences(const int32_t* s, int n, const int32_t* pat, int m, std::vector<int>& where) {
  where.clear();
  for (int i = 0; i + m <= n; ++i)
    if (s[i] == pat[0] && std::memcmp(s + i, pat, (size_t)m * sizeof(int32_t)) == 0) where.push_back(i);
}

}  // namespace

struct amg_blocks {
  Result res;
};

// ---- the emulation, driven op by op (tests; amira_amd/clustering.py's start-up check)
extern "C" int64_t amg_py_tuple_hash(const int64_t* item_hashes, int64_t n) {
  TupleHasher h;
  for (int64_t i = 0; i < n; ++i) h.push((uint64_t)item_hashes[i]);
  return h.done((uint64_t)n);
}

// ops: (op, a, b) triples on `n_sets` sets: 0 = sets[a].add(key b), 1 = sets[a].update(sets[b]), 2 = sets[a] = set(),
// 3 = sets[a] = {k for k in sets[b]} (a new set filled in b's iteration order).  key_hash[k] = the Python hash of key k.
// out_keys / out_off: the iteration order of every set at the end.
extern "C" int amg_pyset_script(const int32_t* ops, int64_t n_ops, const int64_t* key_hash, int32_t n_sets,
                                int32_t* out_keys, int64_t* out_off) {
  if (!ops || !key_hash || !out_keys || !out_off || n_sets < 1) return amg_fail(AMG_E_ARG, "amg_pyset_script: null argument");
  std::vector<PySet> sets((size_t)n_sets);
  for (int64_t i = 0; i < n_ops; ++i) {
    const int32_t op = ops[3 * i], a = ops[3 * i + 1], b = ops[3 * i + 2];
    if (a < 0 || a >= n_sets || ((op == 1 || op == 3) && (b < 0 || b >= n_sets))) return amg_fail(AMG_E_ARG, "amg_pyset_script: bad set");
    if (op == 0) {
      sets[a].add(b, key_hash[b]);
    } else if (op == 1) {
      sets[a].merge(sets[b]);
    } else if (op == 2) {
      sets[a] = PySet();
    } else if (op == 3) {
      PySet n;
      sets[b].each([&](int32_t k, int64_t h) { n.add(k, h); });
      sets[a] = std::move(n);
    } else {
      return amg_fail(AMG_E_ARG, "amg_pyset_script: bad op");
    }
  }
  int64_t at = 0;
  for (int32_t s = 0; s < n_sets; ++s) {
    out_off[s] = at;
    sets[s].each([&](int32_t k, int64_t) { out_keys[at++] = k; });
  }
  out_off[n_sets] = at;
  return AMG_OK;
}

// The block search of one gene.  seq / seq_off: the node ids of the windows of the reads that hold the gene (-2: None),
// the reads in the order in which the reference's set of read names iterates; anchors: device ids of the anchor
// nodes in the iteration order of the reference's anchor set; anchor_rank[i]: rank of anchor i's 256-bit hash among
// the anchors (the smaller end decides a block's canonical orientation, path_finding_utils.py:127-128);
// py_hash[id] = hash(node hash) for every node id that occurs in seq; none_hash = hash(None).
// Result: the keys of full_blocks (path_finding_utils.py:237-247) in insertion order.
extern "C" int amg_cluster_full_blocks(const int32_t* seq, const int64_t* seq_off, int64_t n_reads, const int32_t* anchors,
                                       const int32_t* anchor_rank, int32_t n_anchors, const int64_t* py_hash,
                                       int64_t n_nodes, int64_t none_hash, amg_blocks** out) {
  if (!out) return amg_fail(AMG_E_ARG, "amg_cluster_full_blocks: null out");
  *out = nullptr;
  if (n_reads < 0 || n_anchors < 0 || !seq_off || (n_anchors > 0 && (!anchors || !anchor_rank)) || !py_hash)
    return amg_fail(AMG_E_ARG, "amg_cluster_full_blocks: bad argument");
  auto res = std::make_unique<amg_blocks>();
  Gene G;
  G.seq = seq;
  G.off = seq_off;
  G.R = n_reads;
  G.T.py_hash = py_hash;
  G.T.none_hash = none_hash;
  const int64_t total = n_reads > 0 ? seq_off[n_reads] : 0;
  for (int64_t i = 0; i < total; ++i)
    if (seq[i] >= n_nodes || seq[i] < -2) return amg_fail(AMG_E_ARG, "amg_cluster_full_blocks: node id out of range");
  // where every anchor sits: one pass over the reads
  std::unordered_map<int32_t, int> anchor_index;
  for (int i = 0; i < n_anchors; ++i) anchor_index.emplace(anchors[i], i);
  G.occ.assign((size_t)n_anchors, {});
  {
    std::vector<int8_t> is_anchor((size_t)(n_nodes > 0 ? n_nodes : 1), 0);
    for (int i = 0; i < n_anchors; ++i)
      if (anchors[i] >= 0 && anchors[i] < n_nodes) is_anchor[anchors[i]] = 1;
    for (int64_t r = 0; r < n_reads; ++r)
      for (int64_t t = seq_off[r]; t < seq_off[r + 1]; ++t) {
        const int32_t v = seq[t];
        if (v >= 0 && is_anchor[v]) G.occ[anchor_index[v]].emplace_back((int32_t)r, (int32_t)(t - seq_off[r]));
      }
  }
  Tuples& T = G.T;
  OrderedKeys full_blocks;
  std::vector<int> where;
  std::vector<int32_t> buf;
  struct Entry {
    int32_t read, start, end;
  };
  struct Ctx {
    PySet up, down;
  };
  for (int i1 = 0; i1 < n_anchors; ++i1) {
    // get_suffixes_from_initial_tree: per read the LONGEST suffix that starts at an occurrence of a1 = its first
    std::vector<std::pair<int32_t, int32_t>> first_a1;  // (read, p)
    for (const auto& rp : G.occ[i1])
      if (first_a1.empty() || first_a1.back().first != rp.first) first_a1.push_back(rp);
    for (int i2 = 0; i2 < n_anchors; ++i2) {
      if (i2 == i1 || anchors[i1] == anchors[i2]) continue;
      const bool canonical_pair = anchor_rank[i1] < anchor_rank[i2];  // blocks run a1 .. a2: canonical as they are?
      // get_blocks_from_subtree: per read the longest block a1 .. a2 behind the first a1 (the last a2 after it)
      OrderedKeys ctx_keys;   // contexts: key order = first read that holds its block exactly once
      std::vector<Ctx> ctxs;
      std::vector<std::vector<Entry>> todo;
      std::unordered_map<int32_t, bool> duplicates;
      {
        const auto& o2 = G.occ[i2];
        size_t j = 0;
        for (const auto& rp : first_a1) {
          const int32_t r = rp.first, p = rp.second;
          while (j < o2.size() && o2[j].first < r) ++j;
          int32_t q = -1;
          size_t jj = j;
          while (jj < o2.size() && o2[jj].first == r) {
            if (o2[jj].second > p) q = o2[jj].second;
            ++jj;
          }
          if (q < 0) continue;
          const int32_t* on_read = seq + seq_off[r];
          const int n = (int)(seq_off[r + 1] - seq_off[r]);
          const int m = q - p + 1;
          occurrences(on_read, n, on_read + p, m, where);
          int32_t key;
          if (canonical_pair) {
            key = T.intern(on_read + p, m);
          } else {
            buf.resize(m);
            for (int x = 0; x < m; ++x) buf[x] = on_read[q - x];
            key = T.intern(buf.data(), m);
          }
          if (where.size() > 1) duplicates[key] = true;
          else if (!duplicates.count(key)) duplicates[key] = false;
          if (where.size() == 1) {
            const int32_t ci = ctx_keys.put(key);
            if ((size_t)ci == ctxs.size()) {
              ctxs.emplace_back();
              todo.emplace_back();
            }
            todo[ci].push_back(Entry{r, where[0], where[0] + m - 1});
          }
        }
      }
      // generate_contexts: canonical reads add their options; a read in the other orientation REPLACES the entry by
      // sets made from that read alone (path_finding_utils.py:143-162) — all blocks of this pair share one orientation
      for (size_t ci = 0; ci < ctxs.size(); ++ci) {
        const auto& entries = todo[ci];
        const size_t first = canonical_pair ? 0 : entries.size() - 1;
        for (size_t e = first; e < entries.size(); ++e) {
          const Entry& en = entries[e];
          const int32_t* on_read = seq + seq_off[en.read];
          const int n = (int)(seq_off[en.read + 1] - seq_off[en.read]);
          const int32_t up_t = T.intern(on_read, en.start);
          const int32_t down_t = T.intern(on_read + en.end + 1, n - en.end - 1);
          const Options& up = G.options(up_t, true);
          const Options& down = G.options(down_t, false);
          if (canonical_pair) {
            ctxs[ci].up.merge(up.set);
            ctxs[ci].down.merge(down.set);
          } else {
            PySet nu, nd;
            down.set.each([&](int32_t k, int64_t) {
              const int32_t t = G.reversed_of(k);
              nu.add(t, T.hash[t]);
            });
            up.set.each([&](int32_t k, int64_t) {
              const int32_t t = G.reversed_of(k);
              nd.add(t, T.hash[t]);
            });
            ctxs[ci].up = std::move(nu);
            ctxs[ci].down = std::move(nd);
          }
        }
      }
      // generate_full_paths
      for (size_t ci = 0; ci < ctxs.size(); ++ci) {
        const int32_t c = ctx_keys.keys[ci];
        if (duplicates[c]) continue;
        const std::vector<int32_t> ups = greedy_clusters(ctxs[ci].up, T, true);
        const std::vector<int32_t> downs = greedy_clusters(ctxs[ci].down, T, false);
        const int cn = T.len(c);
        for (int32_t u : ups)
          for (int32_t d : downs) {
            const int un = T.len(u), dn = T.len(d);
            buf.resize((size_t)(un + cn + dn));
            std::copy(T.data(u), T.data(u) + un, buf.begin());
            std::copy(T.data(c), T.data(c) + cn, buf.begin() + un);
            std::copy(T.data(d), T.data(d) + dn, buf.begin() + un + cn);
            // a1 sits at the start of the block as the reads run, i.e. at its canonical start or end
            const int j = canonical_pair ? un : un + cn - 1;
            if (G.held_by_a_read(buf.data(), (int)buf.size(), i1, j)) full_blocks.put(T.intern(buf.data(), (int)buf.size()));
          }
      }
    }
  }
  for (int32_t t : full_blocks.keys) {
    res->res.ids.insert(res->res.ids.end(), T.data(t), T.data(t) + T.len(t));
    res->res.off.push_back((int64_t)res->res.ids.size());
  }
  *out = res.release();
  return AMG_OK;
}

// What the read loop of get_AMR_anchors (construct_graph.py:2644-2676) finds out about every AMR node, from the node ids
// of the windows of the reads that hold the gene.  The loop walks a node's reads in read order and its positions on
// a read in ascending order, i.e. the node's occurrences in ascending token order; every occurrence is one of: the
// only window of its read (flag True, stop), a terminal window (flag True), an interior window with a non-AMR
// neighbour (anchor, stop), an interior window between AMR nodes (flag False).  read_order: the reads of seq in
// ascending order of their rows in the read set (seq itself follows the iteration order of a Python set).
// out[4 i ..]: {stopped at an anchor occurrence, all(singletons), number of flags, number of True flags} of amr_ids[i].
extern "C" int amg_cluster_anchor_stats(const int32_t* seq, const int64_t* seq_off, int64_t n_reads,
                                        const int64_t* read_order, const int32_t* amr_ids, int32_t n_amr,
                                        int64_t n_nodes, int32_t* out) {
  if (!seq_off || !out || (n_amr > 0 && !amr_ids) || n_reads < 0) return amg_fail(AMG_E_ARG, "amg_cluster_anchor_stats: bad argument");
  std::vector<int32_t> slot((size_t)(n_nodes > 0 ? n_nodes : 1), -1);
  for (int i = 0; i < n_amr; ++i) {
    if (amr_ids[i] < 0 || amr_ids[i] >= n_nodes) return amg_fail(AMG_E_ARG, "amg_cluster_anchor_stats: node id out of range");
    slot[amr_ids[i]] = i;
  }
  struct St {
    bool stopped = false, anchor = false, any = false, first_single = false;
    int32_t flags = 0, yes = 0;
  };
  std::vector<St> st((size_t)n_amr);
  auto amr = [&](int32_t v) { return v >= 0 && v < n_nodes && slot[v] >= 0; };
  for (int64_t x = 0; x < n_reads; ++x) {
    const int64_t r = read_order ? read_order[x] : x;
    if (r < 0 || r >= n_reads) return amg_fail(AMG_E_ARG, "amg_cluster_anchor_stats: bad read order");
    const int32_t* s = seq + seq_off[r];
    const int64_t n = seq_off[r + 1] - seq_off[r];
    for (int64_t i = 0; i < n; ++i) {
      if (!amr(s[i])) continue;
      St& a = st[slot[s[i]]];
      const int kind = n == 1 ? 3 : (i == 0 || i == n - 1) ? 1 : (!amr(s[i - 1]) || !amr(s[i + 1])) ? 2 : 0;
      if (!a.any) {
        a.any = true;
        a.first_single = kind == 3;
      }
      if (a.stopped) continue;
      if (kind >= 2) {
        a.stopped = true;
        if (kind == 2) a.anchor = true;
        else {
          ++a.flags;
          ++a.yes;
        }
      } else {
        ++a.flags;
        a.yes += kind == 1;
      }
    }
  }
  for (int i = 0; i < n_amr; ++i) {
    out[4 * i] = st[i].anchor ? 1 : 0;
    out[4 * i + 1] = (!st[i].any || st[i].first_single) ? 1 : 0;
    out[4 * i + 2] = st[i].flags;
    out[4 * i + 3] = st[i].yes;
  }
  return AMG_OK;
}

extern "C" int amg_cluster_blocks_sizes(const amg_blocks* b, int64_t* n_blocks, int64_t* n_ids) {
  if (!b || !n_blocks || !n_ids) return amg_fail(AMG_E_ARG, "amg_cluster_blocks_sizes: null argument");
  *n_blocks = (int64_t)b->res.off.size() - 1;
  *n_ids = (int64_t)b->res.ids.size();
  return AMG_OK;
}

extern "C" int amg_cluster_blocks_get(const amg_blocks* b, int64_t* block_off, int32_t* block_ids) {
  if (!b || !block_off) return amg_fail(AMG_E_ARG, "amg_cluster_blocks_get: null argument");
  std::copy(b->res.off.begin(), b->res.off.end(), block_off);
  if (block_ids) std::copy(b->res.ids.begin(), b->res.ids.end(), block_ids);
  return AMG_OK;
}

extern "C" int amg_cluster_blocks_free(amg_blocks* b) {
  delete b;
  return AMG_OK;
}
