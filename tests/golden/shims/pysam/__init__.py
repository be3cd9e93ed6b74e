"""Stand-in for the absent third-party package `pysam` (build container only), so that the
reference's modules import and its FASTQ reader (read_utils.py:24-29: name / sequence / quality of
every record) works when goldens are generated.  pysam's FastxFile takes the record name up to the
first whitespace."""
import gzip


class _Entry:
    __slots__ = ("name", "sequence", "quality", "comment")


class FastxFile:
    def __init__(self, path, *a, **k):
        opener = gzip.open if str(path).endswith(".gz") else open
        self._fh = opener(path, "rt")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._fh.close()

    def __iter__(self):
        fh = self._fh
        while True:
            head = fh.readline()
            if not head:
                return
            head = head.rstrip("\n")
            if not head:
                continue
            e = _Entry()
            parts = head[1:].split(None, 1)
            e.name = parts[0] if parts else ""
            e.comment = parts[1] if len(parts) > 1 else None
            e.sequence = fh.readline().rstrip("\n")
            if head[0] == "@":
                fh.readline()
                e.quality = fh.readline().rstrip("\n")
            else:
                e.quality = None
            yield e


class _Seg:
    class AlignedSegment:
        pass


libcalignedsegment = _Seg()
AlignedSegment = _Seg.AlignedSegment
