// amg_calls.hip — native front-end / write-back, host code only.
//
// The hot path's input is the gene-call JSON that Amira itself dumps and reloads
// (gene_calls_with_gene_filtering.json / corrected_gene_calls.json: {"read": ["+geneA", ...]},
// reference __main__.py:464-496, result_utils.py:1260-1264) plus the matching gene-position
// JSON ({"read": [[start, end], ...]}).  Turning that into CSR tokens in Python costs ~10 s
// per 60 M genes; this does it natively:
//   * a small strict JSON reader for exactly these two shapes (strings with escapes, integers);
//   * gene parsing as construct_gene.py:49-65 (strand = first char, ' ' -> '_' in the name);
//   * the reference's gene hash: sha256(pickle.dumps(name)) with pickle protocol 4 framing
//     (80 04 95 <len8> 8c <n> <utf8> 94 2e, or 58 <len4> for names of 256+ bytes),
//     construct_gene.py:5-10 — computed once per DISTINCT name;
//   * ranks by ascending hash -> tokens (amira_amd/tokens.py: V + rank / V - 1 - rank).
// Checked bit for bit against the Python path in tests/test_calls_cpu.py (no GPU needed).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <unordered_map>
#include <vector>

#include "amg_internal.h"

// ------------------------------------------------------------------ SHA-256 (FIPS 180-4)
namespace {
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len = 0;
  size_t fill = 0;
  Sha256() {
    static const uint32_t init[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                     0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, init, sizeof(h));
  }
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
      w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const uint8_t* p, size_t n) {
    len += n;
    while (n) {
      size_t take = std::min(n, sizeof(buf) - fill);
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void final(uint8_t out[32]) {
    uint64_t bits = len * 8;
    uint8_t pad = 0x80;
    update(&pad, 1);
    uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
  }
};

// sha256(pickle.dumps(name)) for pickle.DEFAULT_PROTOCOL = 4 (CPython 3.8 - 3.13)
void gene_hash(const std::string& name, uint8_t out[32]) {
  std::string payload;
  if (name.size() < 256) {
    payload.push_back((char)0x8c);  // SHORT_BINUNICODE
    payload.push_back((char)name.size());
  } else {
    payload.push_back('X');  // BINUNICODE
    uint32_t n = (uint32_t)name.size();
    for (int i = 0; i < 4; ++i) payload.push_back((char)(n >> (8 * i)));
  }
  payload += name;
  payload.push_back((char)0x94);  // MEMOIZE
  payload.push_back('.');         // STOP
  std::string msg;
  msg.push_back((char)0x80);
  msg.push_back((char)0x04);
  msg.push_back((char)0x95);  // FRAME
  uint64_t fl = payload.size();
  for (int i = 0; i < 8; ++i) msg.push_back((char)(fl >> (8 * i)));
  msg += payload;
  Sha256 s;
  s.update(reinterpret_cast<const uint8_t*>(msg.data()), msg.size());
  s.final(out);
}

// ------------------------------------------------------------------ minimal JSON reader
struct Reader {
  const char* p;
  const char* end;
  std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool lit(char c) {
    ws();
    if (p < end && *p == c) { ++p; return true; }
    return false;
  }
  static void utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | cp >> 6)); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
      s.push_back((char)(0xE0 | cp >> 12)); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      s.push_back((char)(0xF0 | cp >> 18)); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
  bool hex4(uint32_t* v) {
    if (end - p < 4) return false;
    uint32_t x = 0;
    for (int i = 0; i < 4; ++i) {
      char c = p[i];
      x <<= 4;
      if (c >= '0' && c <= '9') x |= c - '0';
      else if (c >= 'a' && c <= 'f') x |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') x |= c - 'A' + 10;
      else return false;
    }
    p += 4;
    *v = x;
    return true;
  }
  bool str(std::string& out) {
    ws();
    if (p >= end || *p != '"') { err = "expected string"; return false; }
    ++p;
    out.clear();
    while (p < end && *p != '"') {
      if (*p != '\\') { out.push_back(*p++); continue; }
      if (++p >= end) break;
      char c = *p++;
      switch (c) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          uint32_t cp;
          if (!hex4(&cp)) { err = "bad \\u escape"; return false; }
          if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
            const char* save = p;
            p += 2;
            uint32_t lo;
            if (hex4(&lo) && lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            else p = save;
          }
          utf8(out, cp);
          break;
        }
        default: out.push_back(c);  // \" \\ \/
      }
    }
    if (p >= end) { err = "unterminated string"; return false; }
    ++p;
    return true;
  }
  // fast path: the string as a view into the file when it has no escapes (then *owned is
  // empty and [*b, *e) is the content); otherwise falls back to str() into *owned
  bool str_view(const char** b, const char** e, std::string* owned) {
    ws();
    if (p >= end || *p != '"') { err = "expected string"; return false; }
    const char* q = p + 1;
    while (q < end && *q != '"' && *q != '\\') ++q;
    if (q < end && *q == '"') {
      *b = p + 1;
      *e = q;
      p = q + 1;
      owned->clear();
      return true;
    }
    if (!str(*owned)) return false;
    *b = owned->data();
    *e = owned->data() + owned->size();
    return true;
  }
  bool integer(long long* v) {
    ws();
    const char* s = p;
    if (p < end && (*p == '-' || *p == '+')) ++p;
    if (p >= end || *p < '0' || *p > '9') { err = "expected integer"; return false; }
    long long x = 0;
    while (p < end && *p >= '0' && *p <= '9') x = x * 10 + (*p++ - '0');
    if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) {  // tolerate 12.0
      while (p < end && (*p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-' || (*p >= '0' && *p <= '9'))) ++p;
    }
    *v = (*s == '-') ? -x : x;
    return true;
  }
};

// name bytes -> first-seen id, open addressing over (offset, length) into one arena
struct Interner {
  std::vector<char> arena;
  std::vector<uint32_t> off, len;
  std::vector<int32_t> slots;
  size_t mask = 0;
  Interner() { slots.assign(1 << 16, -1); mask = slots.size() - 1; }
  static uint64_t hash(const char* b, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)b[i]) * 1099511628211ull;
    return h ^ (h >> 29);
  }
  void grow() {
    std::vector<int32_t> bigger(slots.size() * 4, -1);
    size_t m = bigger.size() - 1;
    for (size_t id = 0; id < off.size(); ++id) {
      size_t s = hash(&arena[off[id]], len[id]) & m;
      while (bigger[s] >= 0) s = (s + 1) & m;
      bigger[s] = (int32_t)id;
    }
    slots.swap(bigger);
    mask = m;
  }
  int32_t intern(const char* b, size_t n) { return intern_h(b, n, hash(b, n)); }
  int32_t intern_h(const char* b, size_t n, uint64_t h) {
    size_t s = h & mask;
    while (slots[s] >= 0) {
      int32_t id = slots[s];
      if (len[id] == n && memcmp(&arena[off[id]], b, n) == 0) return id;
      s = (s + 1) & mask;
    }
    int32_t id = (int32_t)off.size();
    off.push_back((uint32_t)arena.size());
    len.push_back((uint32_t)n);
    arena.insert(arena.end(), b, b + n);
    slots[s] = id;
    if (off.size() * 2 > slots.size()) grow();
    return id;
  }
  int32_t find(const char* b, size_t n) const {
    size_t s = hash(b, n) & mask;
    while (slots[s] >= 0) {
      int32_t id = slots[s];
      if (len[id] == n && memcmp(&arena[off[id]], b, n) == 0) return id;
      s = (s + 1) & mask;
    }
    return -1;
  }
  size_t size() const { return off.size(); }
  std::string name(size_t id) const { return std::string(&arena[off[id]], len[id]); }
};

bool slurp(const char* path, std::string& out, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path; return false; }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)n);
  size_t got = n ? fread(&out[0], 1, (size_t)n, f) : 0;
  fclose(f);
  if ((long)got != n) { err = "short read"; return false; }
  return true;
}
}  // namespace

struct amg_calls {
  Interner read_ids;  // read ids in file order (id = index), one arena instead of a string per read
  std::vector<int64_t> read_off{0};
  std::vector<int32_t> tokens;
  std::vector<std::string> names;  // rank order
  std::vector<uint8_t> hashes;     // 32 bytes per name, rank order
};

extern "C" int amg_calls_free(amg_calls* c) {
  delete c;
  return AMG_OK;
}

extern "C" int amg_calls_load_json(const char* path, amg_calls** out) {
  if (!path || !out) return amg_fail(AMG_E_ARG, "null argument");
  *out = nullptr;
  std::string text, err;
  if (!slurp(path, text, err)) return amg_fail(AMG_E_ARG, "%s", err.c_str());
  Reader r{text.data(), text.data() + text.size(), ""};
  amg_calls* c = new amg_calls();
  const bool timing = getenv("AMG_CALLS_TIMING") != nullptr;
  clock_t t_start = clock();
  Interner genes;               // name -> first-seen id
  std::vector<int32_t> gid;     // per gene occurrence: first-seen name id
  std::vector<int8_t> strand;   // per gene occurrence
  gid.reserve(text.size() / 8);
  strand.reserve(text.size() / 8);
  auto fail = [&](const char* what) {
    std::string m = std::string(what) + (r.err.empty() ? "" : (": " + r.err));
    delete c;
    return amg_fail(AMG_E_ARG, "%s: %s (offset %lld)", path, m.c_str(), (long long)(r.p - text.data()));
  };
  if (!r.lit('{')) return fail("expected an object of read -> gene list");
  std::string key, owned, fixed;
  if (!r.lit('}')) {
    do {
      const char *kb, *ke;
      if (!r.str_view(&kb, &ke, &key)) return fail("read id");
      {
        // json.load keeps the LAST value of a duplicated key at the FIRST key's position;
        // gene-call files never repeat a read id, so this is rejected rather than emulated
        const size_t before = c->read_ids.size();
        const int32_t rid = c->read_ids.intern(kb, (size_t)(ke - kb));
        if ((size_t)rid != before) return fail("duplicate read id");
      }
      if (!r.lit(':') || !r.lit('[')) return fail("expected ': ['");
      if (!r.lit(']')) {
        do {
          // one pass over the gene string: closing quote, escapes, blanks and the hash of the name
          // (construct_gene.py:49-65: strand = first char, name = rest with ' ' -> '_')
          r.ws();
          if (r.p >= r.end || *r.p != '"') { r.err = "expected string"; return fail("gene"); }
          const char* gb = r.p + 1;
          const char* q = gb;
          uint64_t hsh = 1469598103934665603ull;
          bool blank = true, has_space = false, plain = true;
          if (q < r.end && *q != '"' && *q != '\\') { blank = blank && *q == ' '; ++q; }  // strand char: not hashed
          while (q < r.end && *q != '"') {
            const char ch = *q;
            if (ch == '\\') { plain = false; break; }
            if (ch == ' ') has_space = true; else blank = false;
            hsh = (hsh ^ (unsigned char)(ch == ' ' ? '_' : ch)) * 1099511628211ull;
            ++q;
          }
          const char* ge;
          int32_t id;
          if (plain && q < r.end) {
            ge = q;
            r.p = q + 1;
            if (blank) return fail("Gene information is missing");
            if (*gb != '+' && *gb != '-') return fail("Strand information missing for a gene");
            if (ge - gb < 2) return fail("Gene name information missing for a gene");
            hsh ^= hsh >> 29;
            if (!has_space) {
              id = genes.intern_h(gb + 1, (size_t)(ge - gb - 1), hsh);
            } else {
              fixed.assign(gb + 1, ge);
              std::replace(fixed.begin(), fixed.end(), ' ', '_');
              id = genes.intern_h(fixed.data(), fixed.size(), hsh);
            }
          } else {  // escapes (or a truncated file): the general string reader
            if (!r.str(owned)) return fail("gene");
            gb = owned.data();
            ge = gb + owned.size();
            blank = true;
            has_space = false;
            for (const char* t = gb; t < ge; ++t) {
              if (*t == ' ') has_space = true; else blank = false;
            }
            if (blank) return fail("Gene information is missing");
            if (*gb != '+' && *gb != '-') return fail("Strand information missing for a gene");
            if (ge - gb < 2) return fail("Gene name information missing for a gene");
            fixed.assign(gb + 1, ge);
            std::replace(fixed.begin(), fixed.end(), ' ', '_');
            id = genes.intern(fixed.data(), fixed.size());
          }
          gid.push_back(id);
          strand.push_back(*gb == '+' ? 1 : -1);
        } while (r.lit(','));
        if (!r.lit(']')) return fail("expected ']'");
      }
      c->read_off.push_back((int64_t)gid.size());
    } while (r.lit(','));
    if (!r.lit('}')) return fail("expected '}'");
  }
  if (timing) fprintf(stderr, "parse %.3fs\n", (double)(clock() - t_start) / CLOCKS_PER_SEC);
  // ---- hash every distinct name once, rank by hash, tokens
  std::vector<std::string> names_seen(genes.off.size());
  for (size_t i = 0; i < names_seen.size(); ++i) names_seen[i] = genes.name(i);
  const size_t V = names_seen.size();
  std::vector<uint8_t> h(V * 32);
  for (size_t i = 0; i < V; ++i) gene_hash(names_seen[i], &h[i * 32]);
  std::vector<int32_t> order(V);
  for (size_t i = 0; i < V; ++i) order[i] = (int32_t)i;
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    int cmp = memcmp(&h[(size_t)a * 32], &h[(size_t)b * 32], 32);  // big-endian digest == integer order
    return cmp != 0 ? cmp < 0 : a < b;
  });
  std::vector<int32_t> rank(V);
  c->names.resize(V);
  c->hashes.resize(V * 32);
  for (size_t rnk = 0; rnk < V; ++rnk) {
    rank[order[rnk]] = (int32_t)rnk;
    c->names[rnk] = names_seen[order[rnk]];
    memcpy(&c->hashes[rnk * 32], &h[(size_t)order[rnk] * 32], 32);
  }
  const int32_t Vp = (int32_t)(V ? V : 1);
  c->tokens.resize(gid.size());
  for (size_t i = 0; i < gid.size(); ++i)
    c->tokens[i] = strand[i] > 0 ? Vp + rank[gid[i]] : Vp - 1 - rank[gid[i]];
  if (timing) fprintf(stderr, "total %.3fs\n", (double)(clock() - t_start) / CLOCKS_PER_SEC);
  *out = c;
  return AMG_OK;
}

extern "C" int amg_calls_counts(amg_calls* c, int64_t* n_reads, int64_t* n_tokens, int64_t* n_genes,
                                int64_t* names_bytes, int64_t* ids_bytes) {
  if (!c) return amg_fail(AMG_E_ARG, "null calls");
  if (n_reads) *n_reads = (int64_t)c->read_ids.size();
  if (n_tokens) *n_tokens = (int64_t)c->tokens.size();
  if (n_genes) *n_genes = (int64_t)c->names.size();
  int64_t nb = 0, ib = 0;
  for (auto& s : c->names) nb += (int64_t)s.size() + 1;
  for (size_t i = 0; i < c->read_ids.size(); ++i) ib += (int64_t)c->read_ids.len[i] + 1;
  if (names_bytes) *names_bytes = nb;
  if (ids_bytes) *ids_bytes = ib;
  return AMG_OK;
}

extern "C" int amg_calls_get(amg_calls* c, int32_t* tokens, int64_t* read_offsets, char* gene_names,
                             char* read_ids, uint8_t* gene_hashes) {
  if (!c) return amg_fail(AMG_E_ARG, "null calls");
  if (tokens && !c->tokens.empty()) memcpy(tokens, c->tokens.data(), c->tokens.size() * sizeof(int32_t));
  if (read_offsets) memcpy(read_offsets, c->read_off.data(), c->read_off.size() * sizeof(int64_t));
  if (gene_names) for (auto& s : c->names) { memcpy(gene_names, s.c_str(), s.size() + 1); gene_names += s.size() + 1; }
  if (read_ids)
    for (size_t i = 0; i < c->read_ids.size(); ++i) {
      const size_t n = c->read_ids.len[i];
      memcpy(read_ids, &c->read_ids.arena[c->read_ids.off[i]], n);
      read_ids[n] = 0;
      read_ids += n + 1;
    }
  if (gene_hashes && !c->hashes.empty()) memcpy(gene_hashes, c->hashes.data(), c->hashes.size());
  return AMG_OK;
}

// gene positions {"read": [[s, e], ...]} laid out in the read order of `c`; reads missing from
// the file, or with a different number of entries than genes, are an error
extern "C" int amg_calls_load_positions_json(amg_calls* c, const char* path, int64_t* gene_start,
                                             int64_t* gene_end) {
  if (!c || !path || !gene_start || !gene_end) return amg_fail(AMG_E_ARG, "null argument");
  std::string text, err;
  if (!slurp(path, text, err)) return amg_fail(AMG_E_ARG, "%s", err.c_str());
  Reader r{text.data(), text.data() + text.size(), ""};
  std::vector<char> seen(c->read_ids.size(), 0);
  auto fail = [&](const char* what) {
    return amg_fail(AMG_E_ARG, "%s: %s%s%s (offset %lld)", path, what, r.err.empty() ? "" : ": ",
                    r.err.c_str(), (long long)(r.p - text.data()));
  };
  if (!r.lit('{')) return fail("expected an object of read -> positions");
  std::string key;
  if (!r.lit('}')) {
    do {
      const char *kb, *ke;
      if (!r.str_view(&kb, &ke, &key)) return fail("read id");
      const int32_t rid = c->read_ids.find(kb, (size_t)(ke - kb));  // -1: a read that has no gene calls
      if (!r.lit(':') || !r.lit('[')) return fail("expected ': ['");
      int64_t at = rid < 0 ? -1 : c->read_off[rid];
      int64_t lim = rid < 0 ? -1 : c->read_off[rid + 1];
      if (!r.lit(']')) {
        do {
          long long s, e;
          if (!r.lit('[') || !r.integer(&s) || !r.lit(',') || !r.integer(&e) || !r.lit(']'))
            return fail("expected [start, end]");
          if (at >= 0) {
            if (at >= lim) return fail("more positions than genes for a read");
            gene_start[at] = s;
            gene_end[at] = e;
            ++at;
          }
        } while (r.lit(','));
        if (!r.lit(']')) return fail("expected ']'");
      }
      if (rid >= 0) {
        if (at != lim) return fail("fewer positions than genes for a read");
        seen[rid] = 1;
      }
    } while (r.lit(','));
    if (!r.lit('}')) return fail("expected '}'");
  }
  for (size_t i = 0; i < seen.size(); ++i)
    if (!seen[i] && c->read_off[i + 1] > c->read_off[i]) return amg_fail(AMG_E_ARG, "%s: no positions for read %s", path, c->read_ids.name(i).c_str());
  return AMG_OK;
}

static void json_string(FILE* f, const char* s) {
  fputc('"', f);
  for (const unsigned char* p = reinterpret_cast<const unsigned char*>(s); *p; ++p) {
    if (*p == '"' || *p == '\\') { fputc('\\', f); fputc(*p, f); }
    else if (*p < 0x20) fprintf(f, "\\u%04x", *p);
    else fputc(*p, f);
  }
  fputc('"', f);
}

// write-back: corrected CSR -> {"read": ["+gene", ...]} (json.dumps separators ', ' and ': ',
// ensure_ascii=False) — result_utils.py:1260-1264
extern "C" int amg_calls_write_json(const char* path, const int32_t* tokens, const int64_t* read_offsets,
                                    int64_t n_reads, const char* gene_names, int64_t n_genes,
                                    const char* read_ids) {
  if (!path || !read_offsets || !gene_names || !read_ids) return amg_fail(AMG_E_ARG, "null argument");
  std::vector<const char*> name(n_genes);
  const char* p = gene_names;
  for (int64_t i = 0; i < n_genes; ++i) { name[i] = p; p += strlen(p) + 1; }
  FILE* f = fopen(path, "wb");
  if (!f) return amg_fail(AMG_E_ARG, "cannot write %s", path);
  const int64_t V = n_genes ? n_genes : 1;
  fputc('{', f);
  const char* rid = read_ids;
  for (int64_t r = 0; r < n_reads; ++r) {
    if (r) fputs(", ", f);
    json_string(f, rid);
    rid += strlen(rid) + 1;
    fputs(": [", f);
    for (int64_t t = read_offsets[r]; t < read_offsets[r + 1]; ++t) {
      if (t > read_offsets[r]) fputs(", ", f);
      int32_t tok = tokens[t];
      std::string g = tok >= V ? std::string("+") + name[tok - V] : std::string("-") + name[V - 1 - tok];
      json_string(f, g.c_str());
    }
    fputc(']', f);
  }
  fputc('}', f);
  fclose(f);
  return AMG_OK;
}
