"""Import-only stub for sourmash (absent here); bubble popping is not exercised."""


class MinHash:
    def __init__(self, *a, **k):
        raise NotImplementedError("sourmash is not available in this container")
