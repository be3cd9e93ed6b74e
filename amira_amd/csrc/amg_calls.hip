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
#include <chrono>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "amg_internal.h"

// ------------------------------------------------------------------ big host buffers, kept between calls
// A load or a write of a cfg-3-sized file fills one to two gigabytes of fresh heap (the file's bytes, the writers'
// texts); freed, glibc hands blocks of that size straight back to the kernel and the next call faults every page in
// again (every piece of a JSON -> sweep -> JSON round ran ~40 % slower inside the round than alone).  The 2 MB-aligned
// blocks of FileView and of the writers' texts go to a small cache instead and serve the next call that asks for about
// as much; amg_calls_trim() gives everything back to the system.
namespace {
const size_t kTwoMb = (size_t)2 << 20;
struct BigCache {
  static const int kSlots = 80;  // (two writers' 32 texts each + the file views)
  static const size_t kMaxBytes = (size_t)8 << 30;
  std::mutex mu;
  void* ptr[kSlots] = {};
  size_t bytes[kSlots] = {};
  size_t held = 0;
  // the smallest cached block of at least `want` bytes that is not more than twice as large; *got = its size
  void* take(size_t want, size_t* got) {
    std::lock_guard<std::mutex> g(mu);
    int best = -1;
    for (int i = 0; i < kSlots; ++i)
      if (ptr[i] && bytes[i] >= want && bytes[i] <= 2 * want + 4 * kTwoMb && (best < 0 || bytes[i] < bytes[best])) best = i;
    if (best < 0) return nullptr;
    void* p = ptr[best];
    *got = bytes[best];
    held -= bytes[best];
    ptr[best] = nullptr;
    bytes[best] = 0;
    return p;
  }
  bool give(void* p, size_t n) {  // false: no room, the caller frees the block
    std::lock_guard<std::mutex> g(mu);
    if (getenv("AMG_CALLS_NO_CACHE") || held + n > kMaxBytes) return false;
    for (int i = 0; i < kSlots; ++i)
      if (!ptr[i]) {
        ptr[i] = p;
        bytes[i] = n;
        held += n;
        return true;
      }
    return false;
  }
  size_t trim() {
    std::lock_guard<std::mutex> g(mu);
    const size_t was = held;
    for (int i = 0; i < kSlots; ++i) {
      if (ptr[i]) free(ptr[i]);
      ptr[i] = nullptr;
      bytes[i] = 0;
    }
    held = 0;
    return was;
  }
};
BigCache g_big;

// a 2 MB-aligned block of at least `want` bytes, advised as huge pages (a gigabyte of fresh heap is 250 k page faults
// otherwise, served under one lock while 32 threads fill it); *got = its size, which big_free wants back
void* big_alloc(size_t want, size_t* got) {
  const size_t whole = (want + kTwoMb - 1) & ~(kTwoMb - 1);
  if (void* p = g_big.take(whole, got)) return p;
  void* p = nullptr;
  if (posix_memalign(&p, kTwoMb, whole) != 0 || !p) return nullptr;
  madvise(p, whole, MADV_HUGEPAGE);  // (advice: ignored where transparent huge pages are off)
  *got = whole;
  return p;
}
void big_free(void* p, size_t size) {
  if (p && !g_big.give(p, size)) free(p);
}
}  // namespace

extern "C" int amg_calls_trim(int64_t* released_bytes) {
  const size_t was = g_big.trim();
  if (released_bytes) *released_bytes = (int64_t)was;
  return AMG_OK;
}

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
  void reserve(size_t n_items, size_t n_bytes) {  // room for n_items names of n_bytes in all without growing
    size_t want = slots.size();
    while (want < n_items * 2 + 2) want *= 4;
    if (want != slots.size()) {
      slots.assign(want, -1);
      mask = want - 1;
      for (size_t id = 0; id < off.size(); ++id) {
        size_t s = hash(&arena[off[id]], len[id]) & mask;
        while (slots[s] >= 0) s = (s + 1) & mask;
        slots[s] = (int32_t)id;
      }
    }
    arena.reserve(n_bytes);
    off.reserve(n_items);
    len.reserve(n_items);
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

int worker_count(size_t bytes);

// the file in memory, read by several threads at once (each its own stretch, pread): a gigabyte of gene calls is in
// memory in a fraction of a second, and — unlike a file mapping, whose first touch of every page is a fault served
// one at a time in sandboxed kernels — the parsers then run on ordinary memory
struct FileView {
  const char* data = nullptr;
  size_t size = 0;
  // (not a vector: no zero fill of a gigabyte that is about to be overwritten; a block of the big-buffer cache above)
  char* owned = nullptr;
  size_t owned_bytes = 0;
  FileView() = default;
  FileView(const FileView&) = delete;
  FileView& operator=(const FileView&) = delete;
  ~FileView() { big_free(owned, owned_bytes); }
  bool open(const char* path, std::string& err) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); err = std::string("cannot stat ") + path; return false; }
    size = (size_t)st.st_size;
    owned = static_cast<char*>(big_alloc(size + 1, &owned_bytes));
    if (!owned) { ::close(fd); err = "out of memory"; return false; }
    data = owned;
    const int workers = size ? worker_count(size) : 0;
    std::vector<int> bad((size_t)std::max(workers, 1), 0);
    std::vector<std::thread> th;
    for (int w = 0; w < workers; ++w)
      th.emplace_back([&, w] {
        size_t at = size / (size_t)workers * (size_t)w;
        const size_t end = w + 1 == workers ? size : size / (size_t)workers * (size_t)(w + 1);
        while (at < end) {
          const ssize_t n = pread(fd, owned + at, end - at, (off_t)at);
          if (n <= 0) { bad[(size_t)w] = 1; return; }
          at += (size_t)n;
        }
      });
    for (auto& t : th) t.join();
    ::close(fd);
    for (int b : bad)
      if (b) { err = std::string("short read of ") + path; return false; }
    return true;
  }
};

int worker_count(size_t bytes) {
  if (const char* e = getenv("AMG_CALLS_THREADS")) return std::max(1, atoi(e));
  unsigned hw = std::thread::hardware_concurrency();
  int n = (int)std::min<unsigned>(hw ? hw : 1, 32);
  const int by_size = (int)(bytes >> 22) + 1;  // at least 4 MB of text per thread
  return std::max(1, std::min(n, by_size));
}

// Cut the body of a {"key": [...], "key": [...]} object into `want` pieces at entry boundaries, found by their bytes as
// json.dumps writes them: `], "` — the bracket that closes one entry's list, the comma, the quote that opens the next
// key.  Inside a JSON string a quote is always escaped, so these four bytes can only MISLEAD when a string ends with
// `], ` right before its closing quote; every piece is then parsed on its own terms and must consume exactly its bytes —
// a cut in the wrong place fails that parse and the caller falls back to the single-threaded path.  A file written
// with other separators simply yields one piece.  Pieces: [begin, end) with begin at a key's opening quote.
std::vector<std::pair<const char*, const char*>> split_entries(const char* b, const char* e, int want) {
  std::vector<std::pair<const char*, const char*>> out;
  const char* at = b;
  const size_t total = (size_t)(e - b);
  for (int i = 1; i < want && at < e; ++i) {
    const char* nominal = b + total / want * i;
    if (nominal <= at) continue;
    const char* q = nominal;
    const char* cut = nullptr;
    while (q + 4 <= e) {
      q = static_cast<const char*>(memchr(q, ']', (size_t)(e - q - 3)));
      if (!q) break;
      if (q[1] == ',' && q[2] == ' ' && q[3] == '"') { cut = q; break; }
      ++q;
    }
    if (!cut) break;
    out.emplace_back(at, cut + 1);
    at = cut + 3;
  }
  out.emplace_back(at, e);
  return out;
}
}  // namespace

struct amg_calls {
  Interner read_ids;  // read ids in file order (id = index), one arena instead of a string per read
  std::vector<int64_t> read_off{0};
  std::vector<int32_t> tokens;
  std::vector<std::string> names;  // rank order
  std::vector<uint8_t> hashes;     // 32 bytes per name, rank order
  bool blanks = false;             // some gene name of the file held a blank
};

extern "C" int amg_calls_free(amg_calls* c) {
  delete c;
  return AMG_OK;
}

namespace {
// what one thread makes of its piece of the file
struct CallsPart {
  Interner ids;                 // read ids of the piece, in file order
  std::vector<int64_t> n_genes; // per read
  Interner genes;               // gene names, local first-seen ids
  std::vector<int32_t> gid;     // per gene occurrence: LOCAL name id
  std::vector<int8_t> strand;
  std::string error;            // non-empty: the piece did not parse
  long long error_at = 0;
  bool blanks = false;          // a gene name held a blank (stored with '_' in its place, construct_gene.py:54-56)
};

// entries `"read": ["+gene", ...]` separated by commas, up to the end of the reader's range (which must be reached)
bool parse_call_entries(Reader& r, CallsPart& part, const char* file_begin) {
  std::string key, owned, fixed;
  auto fail = [&](const char* what) {
    part.error = std::string(what) + (r.err.empty() ? "" : (": " + r.err));
    part.error_at = (long long)(r.p - file_begin);
    return false;
  };
  r.ws();
  if (r.p >= r.end) return true;
  do {
    const char *kb, *ke;
    if (!r.str_view(&kb, &ke, &key)) return fail("read id");
    {
      // json.load keeps the LAST value of a duplicated key at the FIRST key's position;
      // gene-call files never repeat a read id, so this is rejected rather than emulated
      const size_t before = part.ids.size();
      const int32_t rid = part.ids.intern(kb, (size_t)(ke - kb));
      if ((size_t)rid != before) return fail("duplicate read id");
    }
    if (!r.lit(':') || !r.lit('[')) return fail("expected ': ['");
    const size_t first_gene = part.gid.size();
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
            id = part.genes.intern_h(gb + 1, (size_t)(ge - gb - 1), hsh);
          } else {
            part.blanks = true;
            fixed.assign(gb + 1, ge);
            std::replace(fixed.begin(), fixed.end(), ' ', '_');
            id = part.genes.intern_h(fixed.data(), fixed.size(), hsh);
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
          if (has_space) part.blanks = true;
          fixed.assign(gb + 1, ge);
          std::replace(fixed.begin(), fixed.end(), ' ', '_');
          id = part.genes.intern(fixed.data(), fixed.size());
        }
        part.gid.push_back(id);
        part.strand.push_back(*gb == '+' ? 1 : -1);
      } while (r.lit(','));
      if (!r.lit(']')) return fail("expected ']'");
    }
    part.n_genes.push_back((int64_t)(part.gid.size() - first_gene));
  } while (r.lit(','));
  r.ws();
  if (r.p != r.end) return fail("expected ',' or the end of the object");
  return true;
}

template <class F>
void run_parts(size_t n, F f) {
  if (n <= 1) {
    if (n == 1) f(0);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(n);
  for (size_t i = 0; i < n; ++i) th.emplace_back([&f, i] { f(i); });
  for (auto& t : th) t.join();
}
}  // namespace

extern "C" int amg_calls_load_json(const char* path, amg_calls** out) {
  if (!path || !out) return amg_fail(AMG_E_ARG, "null argument");
  *out = nullptr;
  FileView file;
  std::string err;
  if (!file.open(path, err)) return amg_fail(AMG_E_ARG, "%s", err.c_str());
  const bool timing = getenv("AMG_CALLS_TIMING") != nullptr;
  clock_t t_start = clock();
  const auto w_start = std::chrono::steady_clock::now();
  auto wall = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - w_start).count(); };
  const char* fb = file.data;
  const char* fe = file.data + file.size;
  // the body between the outer braces
  Reader outer{fb, fe, ""};
  if (!outer.lit('{')) return amg_fail(AMG_E_ARG, "%s: expected an object of read -> gene list (offset %lld)", path, (long long)(outer.p - fb));
  const char* body_b = outer.p;
  const char* body_e = fe;
  while (body_e > body_b && (body_e[-1] == ' ' || body_e[-1] == '\n' || body_e[-1] == '\t' || body_e[-1] == '\r')) --body_e;
  if (body_e <= body_b || body_e[-1] != '}') return amg_fail(AMG_E_ARG, "%s: expected '}' (offset %lld)", path, (long long)(body_e - fb));
  --body_e;
  // ---- the pieces, one thread each; a piece that does not parse on its own terms sends everything down the
  // single-threaded path (whose error message, if the file is really malformed, names the true offset)
  std::vector<CallsPart> parts;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int want = attempt == 0 ? worker_count((size_t)(body_e - body_b)) : 1;
    auto pieces = split_entries(body_b, body_e, want);
    parts.clear();
    parts.resize(pieces.size());
    run_parts(pieces.size(), [&](size_t i) {
      Reader r{pieces[i].first, pieces[i].second, ""};
      parts[i].gid.reserve((size_t)(pieces[i].second - pieces[i].first) / 8);
      parts[i].strand.reserve((size_t)(pieces[i].second - pieces[i].first) / 8);
      parse_call_entries(r, parts[i], fb);
    });
    bool ok = true;
    for (auto& p : parts) ok = ok && p.error.empty();
    if (ok) break;
    if (pieces.size() == 1) {
      for (auto& p : parts)
        if (!p.error.empty()) return amg_fail(AMG_E_ARG, "%s: %s (offset %lld)", path, p.error.c_str(), p.error_at);
    }
  }
  if (timing) fprintf(stderr, "parse %.3fs cpu %.3fs wall, %zu piece(s)\n", (double)(clock() - t_start) / CLOCKS_PER_SEC, wall(), parts.size());
  amg_calls* c = new amg_calls();
  for (auto& p : parts) c->blanks = c->blanks || p.blanks;
  // ---- read ids in file order (one table for the whole file: a read id must not repeat), read offsets
  {
    size_t n_ids = 0, id_bytes = 0;
    for (auto& p : parts) {
      n_ids += p.ids.size();
      id_bytes += p.ids.arena.size();
    }
    c->read_ids.reserve(n_ids, id_bytes);
    c->read_off.reserve(n_ids + 1);
  }
  if (timing) fprintf(stderr, "  ids reserved %.3fs wall\n", wall());
  for (auto& p : parts) {
    for (size_t i = 0; i < p.ids.size(); ++i) {
      const size_t before = c->read_ids.size();
      const int32_t rid = c->read_ids.intern(&p.ids.arena[p.ids.off[i]], p.ids.len[i]);
      if ((size_t)rid != before) {
        const std::string name = p.ids.name(i);
        delete c;
        return amg_fail(AMG_E_ARG, "%s: duplicate read id %s", path, name.c_str());
      }
      c->read_off.push_back(c->read_off.back() + p.n_genes[i]);
    }
  }
  if (timing) fprintf(stderr, "  ids merged %.3fs wall\n", wall());
  // ---- gene names of all pieces -> one table; hash every distinct name once, rank by hash
  Interner genes;
  std::vector<std::vector<int32_t>> to_global(parts.size());
  for (size_t k = 0; k < parts.size(); ++k) {
    auto& p = parts[k];
    to_global[k].resize(p.genes.size());
    for (size_t i = 0; i < p.genes.size(); ++i)
      to_global[k][i] = genes.intern(&p.genes.arena[p.genes.off[i]], p.genes.len[i]);
  }
  std::vector<std::string> names_seen(genes.off.size());
  for (size_t i = 0; i < names_seen.size(); ++i) names_seen[i] = genes.name(i);
  const size_t V = names_seen.size();
  std::vector<uint8_t> h(V * 32);
  {
    const size_t workers = (size_t)std::max(1, std::min(worker_count(V * 4096), (int)((V + 255) / 256)));
    run_parts(workers, [&](size_t w) {
      for (size_t i = w; i < V; i += workers) gene_hash(names_seen[i], &h[i * 32]);
    });
  }
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
  if (timing) fprintf(stderr, "  genes ranked %.3fs wall\n", wall());
  // ---- tokens, every piece into its own stretch
  const int32_t Vp = (int32_t)(V ? V : 1);
  std::vector<size_t> base(parts.size() + 1, 0);
  for (size_t k = 0; k < parts.size(); ++k) base[k + 1] = base[k] + parts[k].gid.size();
  c->tokens.resize(base.back());
  run_parts(parts.size(), [&](size_t k) {
    const auto& p = parts[k];
    int32_t* dst = c->tokens.data() + base[k];
    const int32_t* g2 = to_global[k].data();
    for (size_t i = 0; i < p.gid.size(); ++i) {
      const int32_t r = rank[g2[p.gid[i]]];
      dst[i] = p.strand[i] > 0 ? Vp + r : Vp - 1 - r;
    }
  });
  if (timing) fprintf(stderr, "total %.3fs cpu %.3fs wall\n", (double)(clock() - t_start) / CLOCKS_PER_SEC, wall());
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

extern "C" int amg_calls_has_blanks(amg_calls* c, int32_t* out) {
  if (!c || !out) return amg_fail(AMG_E_ARG, "null argument");
  *out = c->blanks ? 1 : 0;
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

// process_pandora_json (pre_processing.py:44-63) keeps the genes of interest that occur in the reads, collected in a
// Python set in the order the reads first show them.  Here: for every wanted gene rank the index of the first token
// that carries it (either strand), -1 when none does — the stream cut into stretches, one thread each.
extern "C" int amg_calls_first_use(const int32_t* tokens, int64_t n_tokens, int32_t two_v, const int32_t* wanted_ranks,
                                   int64_t n_wanted, int64_t* first_index) {
  if ((n_tokens > 0 && !tokens) || (n_wanted > 0 && (!wanted_ranks || !first_index)) || two_v <= 0 || (two_v & 1))
    return amg_fail(AMG_E_ARG, "amg_calls_first_use: bad argument");
  const int32_t V = two_v / 2;
  std::vector<int32_t> slot((size_t)V, -1);  // gene rank -> index into wanted_ranks
  for (int64_t i = 0; i < n_wanted; ++i) {
    first_index[i] = -1;
    if (wanted_ranks[i] < 0 || wanted_ranks[i] >= V) return amg_fail(AMG_E_ARG, "amg_calls_first_use: rank outside [0, V)");
    if (slot[(size_t)wanted_ranks[i]] < 0) slot[(size_t)wanted_ranks[i]] = (int32_t)i;
  }
  if (n_wanted == 0 || n_tokens == 0) return AMG_OK;
  const size_t workers = (size_t)worker_count((size_t)n_tokens * sizeof(int32_t));
  std::vector<std::vector<int64_t>> part(workers, std::vector<int64_t>((size_t)n_wanted, -1));
  std::vector<int> bad(workers, 0);
  run_parts(workers, [&](size_t w) {
    const int64_t a = n_tokens * (int64_t)w / (int64_t)workers, b = n_tokens * (int64_t)(w + 1) / (int64_t)workers;
    int64_t* mine = part[w].data();
    const int32_t* sl = slot.data();
    for (int64_t t = a; t < b; ++t) {
      const int32_t tok = tokens[t];
      if ((uint32_t)tok >= (uint32_t)two_v) { bad[w] = 1; return; }
      const int32_t s = sl[tok >= V ? tok - V : V - 1 - tok];
      if (s >= 0 && mine[s] < 0) mine[s] = t;
    }
  });
  for (int b : bad)
    if (b) return amg_fail(AMG_E_ARG, "amg_calls_first_use: a token lies outside [0, two_v)");
  for (size_t w = 0; w < workers; ++w)
    for (int64_t i = 0; i < n_wanted; ++i)
      if (first_index[i] < 0 && part[w][(size_t)i] >= 0) first_index[i] = part[w][(size_t)i];
  for (int64_t i = 0; i < n_wanted; ++i)  // a rank listed twice answers both times
    if (first_index[i] < 0 && slot[(size_t)wanted_ranks[i]] != (int32_t)i) first_index[i] = first_index[slot[(size_t)wanted_ranks[i]]];
  return AMG_OK;
}

// gene positions {"read": [[s, e], ...]} laid out in the read order of `c`; reads missing from
// the file, or with a different number of entries than genes, are an error
namespace {
struct PosPart {
  std::string error;
  long long error_at = 0;
};

bool parse_position_entries(Reader& r, const amg_calls* c, int64_t* gene_start, int64_t* gene_end, char* seen,
                            PosPart& part, const char* file_begin) {
  std::string key;
  auto fail = [&](const char* what) {
    part.error = std::string(what) + (r.err.empty() ? "" : (": " + r.err));
    part.error_at = (long long)(r.p - file_begin);
    return false;
  };
  r.ws();
  if (r.p >= r.end) return true;
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
  r.ws();
  if (r.p != r.end) return fail("expected ',' or the end of the object");
  return true;
}
}  // namespace

extern "C" int amg_calls_load_positions_json(amg_calls* c, const char* path, int64_t* gene_start,
                                             int64_t* gene_end) {
  if (!c || !path || !gene_start || !gene_end) return amg_fail(AMG_E_ARG, "null argument");
  FileView file;
  std::string err;
  if (!file.open(path, err)) return amg_fail(AMG_E_ARG, "%s", err.c_str());
  const char* fb = file.data;
  const char* fe = file.data + file.size;
  Reader outer{fb, fe, ""};
  if (!outer.lit('{')) return amg_fail(AMG_E_ARG, "%s: expected an object of read -> positions (offset %lld)", path, (long long)(outer.p - fb));
  const char* body_b = outer.p;
  const char* body_e = fe;
  while (body_e > body_b && (body_e[-1] == ' ' || body_e[-1] == '\n' || body_e[-1] == '\t' || body_e[-1] == '\r')) --body_e;
  if (body_e <= body_b || body_e[-1] != '}') return amg_fail(AMG_E_ARG, "%s: expected '}' (offset %lld)", path, (long long)(body_e - fb));
  --body_e;
  std::vector<char> seen(c->read_ids.size(), 0);
  {  // the caller's arrays are fresh memory too: huge pages for their page-aligned middle, where the kernel offers them
    const size_t n_tok = (size_t)c->read_off.back() * sizeof(int64_t), two_mb = (size_t)2 << 20;
    for (int64_t* a : {gene_start, gene_end}) {
      const uintptr_t lo = ((uintptr_t)a + two_mb - 1) & ~(uintptr_t)(two_mb - 1), hi = ((uintptr_t)a + n_tok) & ~(uintptr_t)(two_mb - 1);
      if (hi > lo) madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    }
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int want = attempt == 0 ? worker_count((size_t)(body_e - body_b)) : 1;
    auto pieces = split_entries(body_b, body_e, want);
    std::vector<PosPart> parts(pieces.size());
    run_parts(pieces.size(), [&](size_t i) {
      Reader r{pieces[i].first, pieces[i].second, ""};
      parse_position_entries(r, c, gene_start, gene_end, seen.data(), parts[i], fb);
    });
    bool ok = true;
    for (auto& p : parts) ok = ok && p.error.empty();
    if (ok) break;
    if (pieces.size() == 1)
      return amg_fail(AMG_E_ARG, "%s: %s (offset %lld)", path, parts[0].error.c_str(), parts[0].error_at);
    std::fill(seen.begin(), seen.end(), 0);
  }
  for (size_t i = 0; i < seen.size(); ++i)
    if (!seen[i] && c->read_off[i + 1] > c->read_off[i]) return amg_fail(AMG_E_ARG, "%s: no positions for read %s", path, c->read_ids.name(i).c_str());
  return AMG_OK;
}

// output file written in pieces by many threads: the texts of a batch go to their offsets with pwrite, side by side
// (one thread's write() into the page cache moves 1.5 - 2 GB/s: it was two thirds of a writer's time)
struct PiecewiseFile {
  int fd = -1;
  off_t pos = 0;
  bool ok = true;
  // (no O_TRUNC: a file that is being replaced keeps its pages in the page cache and the new text is copied over them
  // — freeing a gigabyte of cached pages and faulting them in again cost as much as the write; finish() cuts the file
  // to its new length)
  explicit PiecewiseFile(const char* path) { fd = open(path, O_WRONLY | O_CREAT, 0666); }
  ~PiecewiseFile() { if (fd >= 0) close(fd); }
  static bool put_at(int fd, const char* p, size_t n, off_t at) {
    while (n) {
      const ssize_t w = pwrite(fd, p, n, at);
      if (w <= 0) return false;
      p += w;
      n -= (size_t)w;
      at += w;
    }
    return true;
  }
  void put(char ch) {
    ok = ok && put_at(fd, &ch, 1, pos);
    ++pos;
  }
  bool finish() {
    ok = ok && fd >= 0 && ftruncate(fd, pos) == 0;
    const bool closed = fd >= 0 && close(fd) == 0;
    fd = -1;
    return ok && closed;
  }
};

// A worker's text: ONE block sized by an upper bound, filled through a bare cursor (appending to a string checked its
// capacity once per character: the texts were two thirds of a writer's time), a block of the big-buffer cache above.
struct RawText {
  char* b = nullptr;
  size_t got = 0, n = 0;
  RawText() = default;
  RawText(const RawText&) = delete;
  RawText& operator=(const RawText&) = delete;
  ~RawText() { big_free(b, got); }
  bool room(size_t want) {
    n = 0;
    if (want <= got) return true;
    big_free(b, got);
    b = static_cast<char*>(big_alloc(want, &got));
    if (!b) got = 0;
    return b != nullptr;
  }
};

// the bytes of a JSON string for s[0..n): quotes, backslashes and control characters as json.dumps(ensure_ascii=False)
// writes them; at most 6 n + 2 bytes
static char* json_string_raw(char* q, const char* s, size_t n) {
  *q++ = '"';
  for (size_t i = 0; i < n; ++i) {
    const unsigned char ch = (unsigned char)s[i];
    if (ch == '"' || ch == '\\') { *q++ = '\\'; *q++ = (char)ch; }
    else if (ch < 0x20) { q += snprintf(q, 8, "\\u%04x", ch); }
    else *q++ = (char)ch;
  }
  *q++ = '"';
  return q;
}

static const char kDigits2[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
static inline char* put_i64(char* q, long long v) {
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  if (v < 0) *q++ = '-';
  char buf[24];
  int n = 0;
  while (u >= 100) {
    const unsigned r = (unsigned)(u % 100);
    u /= 100;
    buf[n++] = kDigits2[2 * r + 1];
    buf[n++] = kDigits2[2 * r];
  }
  if (u >= 10) {
    buf[n++] = kDigits2[2 * u + 1];
    buf[n++] = kDigits2[2 * u];
  } else {
    buf[n++] = (char)('0' + u);
  }
  while (n) *q++ = buf[--n];
  return q;
}

static bool put_raw(PiecewiseFile& f, const std::vector<RawText>& text) {
  std::vector<off_t> at(text.size() + 1, f.pos);
  for (size_t w = 0; w < text.size(); ++w) at[w + 1] = at[w] + (off_t)text[w].n;
  std::vector<char> good(text.size(), 1);
  const int fd = f.fd;
  run_parts(text.size(), [&](size_t w) { good[w] = text[w].n == 0 || PiecewiseFile::put_at(fd, text[w].b, text[w].n, at[w]); });
  for (char g : good) f.ok = f.ok && g;
  f.pos = at[text.size()];
  return f.ok;
}

// write-back: corrected CSR -> {"read": ["+gene", ...]} (json.dumps separators ', ' and ': ',
// ensure_ascii=False) — result_utils.py:1260-1264.  The text of a stretch of reads is made by one thread each, the
// stretches are written in order.  Every gene's two spellings ("+name" and "-name" as JSON strings) are made once; a
// read's text is a run of copies.
extern "C" int amg_calls_write_json(const char* path, const int32_t* tokens, const int64_t* read_offsets,
                                    int64_t n_reads, const char* gene_names, int64_t n_genes,
                                    const char* read_ids) {
  if (!path || !read_offsets || !gene_names || !read_ids) return amg_fail(AMG_E_ARG, "null argument");
  const bool timing = getenv("AMG_CALLS_TIMING") != nullptr;
  const auto w_start = std::chrono::steady_clock::now();
  auto wall = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - w_start).count(); };
  // spelled[2 g] / spelled[2 g + 1]: the JSON strings of "+name" / "-name" of gene rank g, in one arena
  std::vector<uint32_t> sp_off((size_t)2 * n_genes + 1, 0);
  std::string arena;
  {
    const char* p = gene_names;
    std::string tmp;
    for (int64_t i = 0; i < n_genes; ++i) {
      const size_t nl = strlen(p);
      for (int sgn = 0; sgn < 2; ++sgn) {
        tmp.assign(1, sgn == 0 ? '+' : '-');
        tmp.append(p, nl);
        const size_t at = arena.size();
        arena.resize(at + 6 * tmp.size() + 2);
        char* e = json_string_raw(&arena[at], tmp.data(), tmp.size());
        arena.resize((size_t)(e - arena.data()));
        sp_off[2 * i + sgn + 1] = (uint32_t)arena.size();
      }
      p += nl + 1;
    }
  }
  std::vector<const char*> rid((size_t)n_reads + 1);
  const char* p = read_ids;
  for (int64_t r = 0; r < n_reads; ++r) { rid[r] = p; p += strlen(p) + 1; }
  rid[n_reads] = p;
  PiecewiseFile f(path);
  if (f.fd < 0) return amg_fail(AMG_E_ARG, "cannot write %s", path);
  const int64_t V = n_genes ? n_genes : 1;
  const int64_t T = n_reads > 0 ? read_offsets[n_reads] : 0;
  const size_t workers = (size_t)std::max<int64_t>(1, std::min<int64_t>(worker_count((size_t)T * 10 + 1), n_reads));
  std::vector<RawText> text(workers);
  std::vector<char> fine(workers, 1);
  // batches of stretches, so that the text in memory stays bounded for very large files
  const int64_t per_batch = std::max<int64_t>((int64_t)workers * 65536, 1);
  f.put('{');
  double t_text = 0, t_put = 0;
  for (int64_t lo = 0; lo < n_reads && f.ok; lo += per_batch) {
    const int64_t hi = std::min(n_reads, lo + per_batch);
    const double t0 = wall();
    run_parts(workers, [&](size_t w) {
      RawText& o = text[w];
      const int64_t a = lo + (hi - lo) * (int64_t)w / (int64_t)workers, b = lo + (hi - lo) * (int64_t)(w + 1) / (int64_t)workers;
      // exact size of the genes' spellings, a bound for the read ids (escapes)
      size_t need = 16;
      for (int64_t t = read_offsets[a]; t < read_offsets[b]; ++t) {
        const int32_t tok = tokens[t];
        const int64_t g = tok >= V ? tok - V : V - 1 - tok;
        const size_t i = (size_t)2 * (size_t)g + (tok >= V ? 0 : 1);
        need += sp_off[i + 1] - sp_off[i] + 2;
      }
      need += (size_t)(rid[b] - rid[a]) * 6 + (size_t)(b - a) * 8;
      if (!o.room(need)) { fine[w] = 0; return; }
      char* q = o.b;
      for (int64_t r = a; r < b; ++r) {
        if (r) { *q++ = ','; *q++ = ' '; }
        q = json_string_raw(q, rid[r], (size_t)(rid[r + 1] - rid[r] - 1));
        *q++ = ':'; *q++ = ' '; *q++ = '[';
        for (int64_t t = read_offsets[r]; t < read_offsets[r + 1]; ++t) {
          if (t > read_offsets[r]) { *q++ = ','; *q++ = ' '; }
          const int32_t tok = tokens[t];
          const int64_t g = tok >= V ? tok - V : V - 1 - tok;
          const size_t i = (size_t)2 * (size_t)g + (tok >= V ? 0 : 1);
          const uint32_t len = sp_off[i + 1] - sp_off[i];
          memcpy(q, arena.data() + sp_off[i], len);
          q += len;
        }
        *q++ = ']';
      }
      o.n = (size_t)(q - o.b);
    });
    for (char g : fine) if (!g) return amg_fail(AMG_E_NOMEM, "no memory for the text of %s", path);
    const double t1 = wall();
    put_raw(f, text);
    t_text += t1 - t0;
    t_put += wall() - t1;
  }
  f.put('}');
  if (!f.finish()) return amg_fail(AMG_E_ARG, "short write to %s", path);
  if (timing) fprintf(stderr, "write calls: text %.3fs, file %.3fs, total %.3fs wall, %zu workers\n", t_text, t_put, wall(), workers);
  return AMG_OK;
}

// positions write-back: {"read": [[start, end], ...]} as json.dumps(gene_position_dict) writes it
// (result_utils.py:1260-1264, second file)
template <class Int>
static int write_positions_json(const char* path, const Int* gene_start, const Int* gene_end,
                                const int64_t* read_offsets, int64_t n_reads, const char* read_ids) {
  if (!path || !read_offsets || !read_ids || ((!gene_start || !gene_end) && n_reads > 0 && read_offsets[n_reads] > 0))
    return amg_fail(AMG_E_ARG, "null argument");
  const bool timing = getenv("AMG_CALLS_TIMING") != nullptr;
  const auto w_start = std::chrono::steady_clock::now();
  auto wall = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - w_start).count(); };
  std::vector<const char*> rid((size_t)n_reads + 1);
  const char* p = read_ids;
  for (int64_t r = 0; r < n_reads; ++r) { rid[r] = p; p += strlen(p) + 1; }
  rid[n_reads] = p;
  PiecewiseFile f(path);
  if (f.fd < 0) return amg_fail(AMG_E_ARG, "cannot write %s", path);
  const int64_t T = n_reads > 0 ? read_offsets[n_reads] : 0;
  const size_t workers = (size_t)std::max<int64_t>(1, std::min<int64_t>(worker_count((size_t)T * 16 + 1), n_reads));
  std::vector<RawText> text(workers);
  std::vector<char> fine(workers, 1);
  const int64_t per_batch = std::max<int64_t>((int64_t)workers * 65536, 1);
  f.put('{');
  double t_text = 0, t_put = 0;
  for (int64_t lo = 0; lo < n_reads && f.ok; lo += per_batch) {
    const int64_t hi = std::min(n_reads, lo + per_batch);
    const double t0 = wall();
    run_parts(workers, [&](size_t w) {
      RawText& o = text[w];
      const int64_t a = lo + (hi - lo) * (int64_t)w / (int64_t)workers, b = lo + (hi - lo) * (int64_t)(w + 1) / (int64_t)workers;
      // a bound from the longest number of the stretch
      unsigned long long most = 0;
      bool neg = false;
      for (int64_t t = read_offsets[a]; t < read_offsets[b]; ++t) {
        const long long s0 = gene_start[t], e0 = gene_end[t];
        neg = neg || s0 < 0 || e0 < 0;
        const unsigned long long us = s0 < 0 ? 0ull - (unsigned long long)s0 : (unsigned long long)s0;
        const unsigned long long ue = e0 < 0 ? 0ull - (unsigned long long)e0 : (unsigned long long)e0;
        most = us > most ? us : most;
        most = ue > most ? ue : most;
      }
      int digits = 1;
      while (most >= 10) { most /= 10; ++digits; }
      const size_t per_tok = (size_t)(2 * (digits + (neg ? 1 : 0)) + 6);
      const size_t need = (size_t)(read_offsets[b] - read_offsets[a]) * per_tok + (size_t)(rid[b] - rid[a]) * 6 + (size_t)(b - a) * 8 + 16;
      if (!o.room(need)) { fine[w] = 0; return; }
      char* q = o.b;
      for (int64_t r = a; r < b; ++r) {
        if (r) { *q++ = ','; *q++ = ' '; }
        q = json_string_raw(q, rid[r], (size_t)(rid[r + 1] - rid[r] - 1));
        *q++ = ':'; *q++ = ' '; *q++ = '[';
        for (int64_t t = read_offsets[r]; t < read_offsets[r + 1]; ++t) {
          if (t > read_offsets[r]) { *q++ = ','; *q++ = ' '; }
          *q++ = '[';
          q = put_i64(q, (long long)gene_start[t]);
          *q++ = ','; *q++ = ' ';
          q = put_i64(q, (long long)gene_end[t]);
          *q++ = ']';
        }
        *q++ = ']';
      }
      o.n = (size_t)(q - o.b);
    });
    for (char g : fine) if (!g) return amg_fail(AMG_E_NOMEM, "no memory for the text of %s", path);
    const double t1 = wall();
    put_raw(f, text);
    t_text += t1 - t0;
    t_put += wall() - t1;
  }
  f.put('}');
  if (!f.finish()) return amg_fail(AMG_E_ARG, "short write to %s", path);
  if (timing) fprintf(stderr, "write positions: text %.3fs, file %.3fs, total %.3fs wall, %zu workers\n", t_text, t_put, wall(), workers);
  return AMG_OK;
}

extern "C" int amg_calls_write_positions_json(const char* path, const int64_t* gene_start, const int64_t* gene_end,
                                              const int64_t* read_offsets, int64_t n_reads, const char* read_ids) {
  return write_positions_json<int64_t>(path, gene_start, gene_end, read_offsets, n_reads, read_ids);
}
extern "C" int amg_calls_write_positions_json32(const char* path, const int32_t* gene_start, const int32_t* gene_end,
                                                const int64_t* read_offsets, int64_t n_reads, const char* read_ids) {
  return write_positions_json<int32_t>(path, gene_start, gene_end, read_offsets, n_reads, read_ids);
}
