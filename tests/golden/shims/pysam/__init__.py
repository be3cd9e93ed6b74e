"""Import-only stub for pysam (absent here); only needed so reference modules import."""


class FastxFile:
    def __init__(self, *a, **k):
        raise NotImplementedError("pysam is not available in this container")


class _Seg:
    class AlignedSegment:
        pass


libcalignedsegment = _Seg()
AlignedSegment = _Seg.AlignedSegment
