"""Scaled MinHash of DNA k-mers as sourmash computes it.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

The reference calls a third-party package that is not vendored: `sourmash>=4.8.4`
(pyproject.toml:28), call sites construct_graph.py:1568, 2151, 2461 (`MinHash(n=0, ksize=K,
scaled=S)`, `.add_sequence(seq, force=True)`, `.hashes`).  sourmash is absent from this image, so
its PUBLISHED algorithm is restated here (sourmash 4.8 `KmerMinHash`, DNA):
  * the sequence is upper-cased; every window of `ksize` bases that holds a character outside
    ACGT is skipped (`force=True`; without it sourmash raises);
  * a k-mer is hashed in canonical form: the smaller (bytewise) of the k-mer and its reverse
    complement;
  * hash = first 64 bits of MurmurHash3_x64_128(k-mer bytes, seed 42);
  * with `scaled = S` the sketch keeps every hash <= max_hash, max_hash = 2^64 - 1 for S = 1 and
    int(2^64 / S) (float division, truncated, as sourmash's Rust core does it) otherwise; `n = 0`: no size bound.
PARITY: pinned by reference-held test vectors only — tests/test_gene_mer_graph.py:5154-5155
(containments 0.9105839416058394 / 0.9091323161011159 on tests/test_1.fastq.gz) and :4528-4607
(assess_connectivity thresholds) — see tests/test_minhash_cpu.py.
"""
M64 = (1 << 64) - 1


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(k):
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & M64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & M64
    k ^= k >> 33
    return k


def murmurhash3_x64_128_h1(data, seed=42):
    """low 64-bit half (h1) of MurmurHash3_x64_128 — Austin Appleby's public-domain reference."""
    c1, c2 = 0x87C37B91114253D5, 0x4CF5AD432745937F
    h1 = h2 = seed & M64
    n = len(data)
    for i in range(0, n - n % 16, 16):
        k1 = int.from_bytes(data[i:i + 8], "little")
        k2 = int.from_bytes(data[i + 8:i + 16], "little")
        k1 = (k1 * c1) & M64
        k1 = _rotl(k1, 31)
        k1 = (k1 * c2) & M64
        h1 ^= k1
        h1 = _rotl(h1, 27)
        h1 = (h1 + h2) & M64
        h1 = (h1 * 5 + 0x52DCE729) & M64
        k2 = (k2 * c2) & M64
        k2 = _rotl(k2, 33)
        k2 = (k2 * c1) & M64
        h2 ^= k2
        h2 = _rotl(h2, 31)
        h2 = (h2 + h1) & M64
        h2 = (h2 * 5 + 0x38495AB5) & M64
    tail = data[n - n % 16:]
    k1 = k2 = 0
    t = len(tail)
    if t > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = (k2 * c2) & M64
        k2 = _rotl(k2, 33)
        k2 = (k2 * c1) & M64
        h2 ^= k2
    if t > 0:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = (k1 * c1) & M64
        k1 = _rotl(k1, 31)
        k1 = (k1 * c2) & M64
        h1 ^= k1
    h1 ^= n
    h2 ^= n
    h1 = (h1 + h2) & M64
    h2 = (h2 + h1) & M64
    h1 = _fmix(h1)
    h2 = _fmix(h2)
    h1 = (h1 + h2) & M64
    return h1


_COMP = bytes.maketrans(b"ACGT", b"TGCA")
_VALID = frozenset(b"ACGT")


def max_hash_for_scaled(scaled):
    if scaled == 0:
        return 0
    if scaled == 1:
        return M64
    # sourmash >= 4 (Rust core, max_hash_for_scaled): (u64::MAX as f64 / scaled as f64) as u64 — a truncation; the
    # older Python helper rounded: equal whenever the quotient is >= 2^53 (scaled <= 2048: the reference's 1 and 10)
    return min(int(float(2 ** 64) / float(scaled)), M64)


class MinHash:
    """the part of sourmash.MinHash the reference uses"""

    def __init__(self, n=0, ksize=21, scaled=0, **_):
        assert n == 0, "only scaled sketches are used by the reference"
        self.ksize, self.scaled = ksize, scaled
        self._max_hash = max_hash_for_scaled(scaled)
        self._hashes = set()

    def add_sequence(self, sequence, force=False):
        seq = (sequence.encode() if isinstance(sequence, str) else bytes(sequence)).upper()
        k = self.ksize
        if len(seq) < k:
            return
        rc = seq.translate(_COMP)[::-1]
        n = len(seq)
        bad = [i for i, c in enumerate(seq) if c not in _VALID]
        if bad and not force:
            raise ValueError("invalid DNA character in input k-mer")
        skip = set()
        for i in bad:
            skip.update(range(max(0, i - k + 1), i + 1))
        for i in range(n - k + 1):
            if i in skip:
                continue
            kmer = seq[i:i + k]
            krc = rc[n - k - i:n - i]
            h = murmurhash3_x64_128_h1(min(kmer, krc), 42)
            if h <= self._max_hash:
                self._hashes.add(h)

    @property
    def hashes(self):
        return {h: 1 for h in sorted(self._hashes)}

    def __len__(self):
        return len(self._hashes)

    def contained_by(self, other):
        """fraction of this sketch's hashes that the other one holds"""
        return len(self._hashes & other._hashes) / len(self._hashes) if self._hashes else 0.0
