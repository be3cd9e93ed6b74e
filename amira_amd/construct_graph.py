"""GeneMerGraph — drop-in for amira/construct_graph.py (reference v0.11.0) on MI355X.

Same constructor, accessors and passes as the reference class; the work is done by
libamg.so (HIP, gfx950) through amira_amd.engine:

    __init__                          amg_set_reads / amg_set_positions / amg_build
    __init__(..., _filter=(n, m))     amg_build_filtered (= __init__ + filter_graph in one device pass)
    filter_graph                      amg_filter
    remove_node & friends             amg_remove_nodes
    remove_short_linear_paths         amg_remove_short_linear_paths
    remove_low_coverage_components    amg_remove_low_coverage_components
    correct_reads                     amg_correct_reads (+ amg_get_corrected)

The device keeps the graph as integer arrays (DESIGN.md "Data layout").  The reference's
object API (dicts keyed by 256-bit sha256 hashes, Node / Edge objects, per-read lists) is a
VIEW that is materialised lazily from those arrays the first time an accessor needs it and
dropped whenever a device pass changes the graph.  Hashes are computed on the host with the
reference's own hashlib/pickle recipe, once per distinct node / edge.

There is no CPU build path: constructing a graph without libamg.so or without a HIP device
raises.  Host-side methods below only do what the reference also does in Python on top of a
built graph (unitig gene strings, linear-path walks, anchor / block logic of the read-path
clustering).
"""
import ctypes as C
import os
import statistics
import sys
import weakref
from collections.abc import Mapping, Sequence
from itertools import product

import numpy as np

from . import _ffi
from . import clustering as _clustering
from .bubble_popping import BubblePopping
from .construct_edge import Edge
from .construct_gene import Gene, convert_int_strand_to_string, hashlib_hash
from .construct_gene_mer import GeneMer
from .construct_node import Node
from .construct_read import Read  # noqa: F401  (re-exported like the reference)
from .engine import Engine, _ENGINE_POOL, acquire_engine as _acquire_engine, release_engine as _release_engine  # noqa: F401
from .path_finding_utils import (
    Tree,
    construct_suffix_tree,
    filter_blocks,
    find_sublist_indices as _find_sublist_indices,
    get_suffixes_from_initial_tree,
    is_sublist as _is_sublist,
    process_anchors,
)
from .io import TokenizedPositions, TokenizedReads
from .tokens import tokenize

sys.setrecursionlimit(50000)  # construct_graph.py:27


class _LazyReadLists(Mapping):
    """dict-like {read id: per-window list} over the reads that have a window, whose lists are built
    from the device arrays the first time a read is looked up (read-path clustering touches a few
    thousand of millions).  The key list and the id -> row index are only made when asked for."""

    def __init__(self, graph, make):
        self._g, self._make, self._cache = weakref.proxy(graph), make, {}   # no cycle graph <-> view
        self._ids = None       # reads with >= 1 window, in read order (+ reads added by hand)
        self._extra = []       # reads added through add_node_to_read

    def _row(self, rid):
        r = self._g._known_rows.get(rid)      # reads the clustering has met already (no name -> row table of ALL reads)
        if r is None:
            r = self._g._read_index.get(rid)
        if r is None:
            return None
        off = self._g._read_off
        return r if off[r + 1] - off[r] >= self._g._kmerSize else None

    def __getitem__(self, rid):
        got = self._cache.get(rid)
        if got is None:
            r = self._row(rid)
            if r is None:
                raise KeyError(rid)
            got = self._cache[rid] = self._make(r)
        return got

    def __setitem__(self, rid, value):  # remove_node_from_reads / add_node_to_read style updates
        if rid not in self._cache and self._row(rid) is None:   # a read the build did not see (:165-178)
            self._extra.append(rid)
            if self._ids is not None:
                self._ids.append(rid)
        self._cache[rid] = value

    def _keys(self):
        if self._ids is None:
            g = self._g
            rows = np.flatnonzero(np.diff(g._read_off) >= g._kmerSize).tolist() if len(g._read_off) > 1 else []
            self._ids = [g._read_ids[r] for r in rows] + self._extra
        return self._ids

    def __iter__(self):
        return iter(self._keys())

    def __len__(self):
        return len(self._keys())

    def __contains__(self, rid):
        return rid in self._cache or self._row(rid) is not None


class _LazyHashes(Sequence):
    """node hashes of the rows of a token matrix, computed on first use (a list from then on)"""

    def __init__(self, token_rows, vocab):
        self._rows, self._vocab, self._made = token_rows, vocab, None

    def _list(self):
        if self._made is None:
            signed = self._vocab.signed_hash
            self._made = [hashlib_hash(tuple([signed(t) for t in row])) for row in self._rows.tolist()]
            self._rows = None
        return self._made

    def __len__(self):
        return len(self._rows) if self._made is None else len(self._made)

    def __getitem__(self, i):
        return self._list()[i]

    def __iter__(self):
        return iter(self._list())

    def __eq__(self, other):
        return self._list() == (other._list() if isinstance(other, _LazyHashes) else other)

    def __repr__(self):
        return repr(self._list())


def _graph_closed():
    raise RuntimeError("this Node's lists were never looked at while its GeneMerGraph was open, and the graph has been "
                       "closed (its device arrays are gone); read them before close() or keep the graph")


class _GraphNode(Node):
    """Node view whose read list is cut out of the device's node->reads CSR on first use."""

    def _lazy(self, maker):
        self._reads_maker = maker
        self._reads_list = None

    @property
    def listOfReads(self):
        if self._reads_list is None:
            self._reads_list = self._reads_maker() if getattr(self, "_reads_maker", None) else []
        return self._reads_list

    @listOfReads.setter
    def listOfReads(self, value):
        self._reads_list = value

    # the forward / backward edge-hash lists are made (and their edges hashed) the first time they are looked at:
    # read-path clustering asks for the neighbours of a few hundred nodes of tens of thousands
    def _lazy_edges(self, forward_maker, backward_maker):
        self._fw_maker, self._bw_maker = forward_maker, backward_maker
        self._fw = self._bw = None

    @property
    def forwardEdgeHashes(self):
        if self._fw is None:
            self._fw = self._fw_maker() if self._fw_maker else []   # (no maker: a node made by hand, add_node)
        return self._fw

    @forwardEdgeHashes.setter
    def forwardEdgeHashes(self, value):
        self._fw = value

    @property
    def backwardEdgeHashes(self):
        if self._bw is None:
            self._bw = self._bw_maker() if self._bw_maker else []
        return self._bw

    @backwardEdgeHashes.setter
    def backwardEdgeHashes(self, value):
        self._bw = value


class _LazyArrays(dict):
    """the device arrays of a view by name; an array that is big and rarely wanted is fetched by its maker the first
    time it is asked for (a maker returns {name: array} for everything it fetched)"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.makers = {}

    def __missing__(self, key):
        maker = self.makers.get(key)
        if maker is None:
            raise KeyError(key)
        self.update(maker())
        return dict.__getitem__(self, key)


class _HashToId:
    """{node hash: device node id} over the nodes of a view; hashes the view has not made yet are settled by making
    every node (see _View)."""

    __slots__ = ("_view",)

    def __init__(self, view):
        self._view = view

    def get(self, h, default=None):
        v = self._view
        i = v._id_of.get(h)
        if i is None and not v._nodes_complete:
            v.nodes   # (makes them all)
            i = v._id_of.get(h)
        return default if i is None else i

    def __getitem__(self, h):
        i = self.get(h)
        if i is None:
            raise KeyError(h)
        return i

    def __contains__(self, h):
        return self.get(h) is not None


class _View:
    """Reference-shaped object view of the device graph (see module docstring).  Node objects (a sha256 and a dozen
    small objects each) and Edge objects (two sha256 each) are made when somebody looks at them — read-path
    clustering touches the few thousand nodes its genes' reads run through, of tens of thousands: one by one through
    node_at / hash_at / node_by_hash and a node's edge lists, all of them — in the reference's insertion order — the
    first time the node / edge dict itself is asked for."""

    __slots__ = ("_nodes", "_nodes_complete", "_node_obj", "_make_node", "_id_of", "node_of_hash",
                 "_edges", "_edges_complete", "_edge_obj", "_make_edge", "_n_edges", "readNodes",
                 "readNodeDirections", "readNodePositions", "node_hash", "edge_hash", "alive", "arrays",
                 "_nh_table", "_hash_node")

    def dispose(self):
        """cut the view's own reference cycles (its makers close over it) so that it is freed the moment its graph lets
        go of it, not when the cyclic collector next walks the heap; called when the graph is closed"""
        self._make_node = self._make_edge = self.node_of_hash = self._hash_node = None
        if isinstance(self.arrays, _LazyArrays):
            self.arrays.makers.clear()   # (they hold the engine)
        self.readNodes = self.readNodeDirections = self.readNodePositions = None
        for node in self._nodes.values():   # (the nodes made so far)
            if isinstance(node, _GraphNode):
                # lists that were looked at stay as they are; the others can no longer be made: say so instead of
                # answering with an empty list
                node._reads_maker = node._fw_maker = node._bw_maker = _graph_closed
        self._node_obj, self._edge_obj = [], []

    # ---- nodes
    def node_at(self, i):
        """the node with device id i (None: removed)"""
        if self._nodes_complete:
            h = self.node_hash[i]
            return None if h is None else self._nodes[h]
        node = self._node_obj[i]
        return node if node is not None else self._make_node(i)

    def hash_at(self, i):
        h = self.node_hash[i]
        if h is None and not self._nodes_complete:
            h = self._hash_node(i)
        return h

    def ensure_hashes(self, ids):
        """node_hash[i] filled in for every id >= 0 of `ids` (an iterable of ints) — the hash alone (one sha256 of the
        node's tokens); the Node object with its genes is made when somebody asks for the node itself"""
        if self._nodes_complete:
            return
        nh, hash_node = self.node_hash, self._hash_node
        for i in ids:
            if i >= 0 and nh[i] is None:
                hash_node(i)

    @property
    def nodes(self):
        if not self._nodes_complete:
            ordered = {}
            for i in range(len(self.node_hash)):
                node = self.node_at(i)
                if node is not None:
                    ordered[self.node_hash[i]] = node
            self._nodes, self._nodes_complete = ordered, True
        return self._nodes

    def node_by_hash(self, h, default=None):
        got = self._nodes.get(h)
        if got is None and not self._nodes_complete:
            i = self._id_of.get(h)          # hashed already (ensure_hashes), object not made yet
            got = self.node_at(i) if i is not None else self.nodes.get(h)
        return default if got is None else got

    def node_hash_table(self, ids=None):
        """node hashes as an object array indexed by device node id, two None entries at the end (ids -2 and -1:
        a window without a node); with `ids` (an int array) only those entries are promised"""
        if not self._nodes_complete:
            if ids is None:
                self.nodes
            else:
                self.ensure_hashes(np.unique(ids).tolist())
        if self._nh_table is None:
            t = np.empty(len(self.node_hash) + 2, dtype=object)
            t[:len(self.node_hash)] = self.node_hash
            self._nh_table = t
        return self._nh_table

    # ---- edges
    def edge_hash_of(self, e):
        h = self.edge_hash[e]
        if h is None:
            edge = self._edge_obj[e] = self._make_edge(e)
            h = self.edge_hash[e] = edge.__hash__()
            self._edges[h] = edge
        return h

    @property
    def edges(self):
        if not self._edges_complete:
            alive = self.arrays["edges"]["alive"]
            ordered = {}
            for e in range(self._n_edges):
                if alive[e]:
                    h = self.edge_hash_of(e)
                    ordered[h] = self._edge_obj[e]
            self._edges, self._edges_complete = ordered, True
        return self._edges

    def edge_by_hash(self, h):
        got = self._edges.get(h)
        return got if got is not None else self.edges[h]


class _SubsetRows:
    """stands in for the {read: genes} subset handed to get_all_sublists where only its size is looked at"""
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


class GeneMerGraph(BubblePopping):
    # ------------------------------------------------------------------ build
    def __init__(self, readDict, kmerSize, gene_positions=None, device=None, _filter=None):
        """_filter = (minNodeCoverage, minEdgeCoverage): the graph comes out as GeneMerGraph(...).filter_graph(...)
        would leave it, with the filter applied during the build (amg_build_filtered; graph_utils.build_filtered_graph)"""
        self._init_fields(readDict, kmerSize, gene_positions, device)
        if isinstance(readDict, TokenizedReads):
            readDict = readDict.settled()   # (reads replaced by hand — bubble popping — are spelled into the arrays)
        on_device = readDict.device_source() if isinstance(readDict, TokenizedReads) else None
        if on_device is not None and on_device.engine().device == self._engine.device:
            # the output of a correct_reads that never left the GPU: taken over device to device, positions included
            # when they come from the same correction (amira_amd.io.DeviceCorrected)
            ids = readDict.read_ids
            self._vocab, self._read_off, self._read_ids = readDict.vocab, readDict.read_offsets, (ids if isinstance(ids, list) else list(ids))
            self._engine.set_reads_from_corrected(on_device.engine())
            same = isinstance(gene_positions, TokenizedPositions) and gene_positions.device_source() is on_device
            if not same:
                self._host_positions(gene_positions)
                self._upload_positions()
        else:
            if isinstance(readDict, TokenizedReads):
                self._vocab, toks, offs, self._read_ids = (readDict.vocab, readDict.tokens,
                                                           readDict.read_offsets, list(readDict.read_ids))
            else:
                self._vocab, toks, offs, self._read_ids = tokenize(readDict)
            self._read_off = offs
            self._tokens_val = toks
            self._engine.set_reads(toks, offs, self._vocab.two_v)
            self._host_positions(gene_positions)
            self._upload_positions()
        try:
            if _filter is None:
                self._engine.build(self._kmer_for_device())
            else:
                self._build_filter = (int(_filter[0]), int(_filter[1]))
                self._minNodeCoverage, self._minEdgeCoverage = _filter
                self._engine.build_filtered(self._kmer_for_device(), max(int(_filter[0]), 0), max(int(_filter[1]), 0))
        except _ffi.AmgError as err:
            if err.code == _ffi.E_PALINDROME:  # construct_gene_mer.py:23-25
                raise AssertionError("Gene-mer and reverse complement gene-mer are identical") from None
            raise
        self._note_short_reads()

    @classmethod
    def build_many(cls, readDict, kmerSizes, gene_positions=None, device=None):
        """[GeneMerGraph(readDict, k, gene_positions) for k in kmerSizes] — the seven graphs of choose_kmer_size
        (graph_utils.py:258-296) — with the reads tokenised and uploaded ONCE (amg_build_multi: every graph is built by
        the ordinary build on an engine of its own; the graphs after the first read the first one's device arrays and
        keep it alive)."""
        kmerSizes = list(kmerSizes)
        first = cls.__new__(cls)
        first._init_fields(readDict, kmerSizes[0], gene_positions, device)
        if isinstance(readDict, TokenizedReads):
            first._vocab, toks, offs, first._read_ids = (readDict.vocab, readDict.tokens, readDict.read_offsets,
                                                         list(readDict.read_ids))
        else:
            first._vocab, toks, offs, first._read_ids = tokenize(readDict)
        first._read_off, first._tokens_val = offs, toks
        first._engine.set_reads(toks, offs, first._vocab.two_v)
        first._host_positions(gene_positions)
        first._upload_positions()
        graphs = [first]
        for k in kmerSizes[1:]:
            g = cls.__new__(cls)
            g._init_fields(readDict, k, gene_positions, device)
            g._vocab, g._read_ids, g._read_off, g._tokens_val = first._vocab, first._read_ids, first._read_off, toks
            g._read_index_ = first._read_index_
            g._gs_val, g._ge_val = first._gs, first._ge
            g._positions_pending = g._gs_val is not None   # uploaded when a correction asks for them
            g._reads_owner = first
            first._borrowers += 1
            graphs.append(g)
        try:
            Engine.build_multi([g._engine for g in graphs], [g._kmer_for_device() for g in graphs])
        except _ffi.AmgError as err:
            for g in graphs:
                g.close()
            if err.code == _ffi.E_PALINDROME:
                raise AssertionError("Gene-mer and reverse complement gene-mer are identical") from None
            raise
        for g in graphs:
            g._note_short_reads()
        return graphs

    def _init_fields(self, readDict, kmerSize, gene_positions, device):
        self._reads = readDict
        self._kmerSize = kmerSize
        self._minNodeCoverage = 1
        self._minEdgeCoverage = 1
        self._genePositions = gene_positions
        self._view = None
        self._host_edits = False   # add_node / add_edge / remove_edge ... changed the host view
        self._lone_edges = set()   # directed edges removed one at a time (remove_edge) on the device
        self._known_rows = {}      # read name -> row for the reads the native clustering has looked at
        self._pass_log = []        # device passes applied since the build, in order (replayed by __setstate__)
        self._build_filter = None  # _filter of the build
        self._extra_to_correct = set()
        self._gene_cache = {}
        self._read_index_ = None
        self._tokens_val = self._gs_val = self._ge_val = None   # host copies: see the properties below
        self._positions_pending = False
        self._reads_owner = None   # build_many: the graph whose engine holds the reads this one's engine borrows
        self._borrowers = 0        # build_many: graphs that borrow this one's reads
        self._close_pending = False
        dev = int(os.environ.get("AMG_DEVICE", "0")) if device is None else int(device)
        self._engine = _acquire_engine(dev)

    def _kmer_for_device(self):
        # GeneMerGraph({}, 0) is legal in the reference: nothing to build
        return 1 if (self._kmerSize < 1 and int(self._read_off[-1]) == 0) else self._kmerSize

    # host copies of the genes and their positions: made at once when the caller handed dicts or host arrays over, on
    # first use when the reads came straight from another graph's correction on the device
    @property
    def _tokens(self):
        if self._tokens_val is None:
            self._tokens_val = self._reads.tokens
        return self._tokens_val

    def _positions_from_mapping(self):
        if self._gs_val is None and isinstance(self._genePositions, TokenizedPositions) \
                and self._genePositions.as_made() and self._genePositions._gs is not None:
            keep32 = self._genePositions.gene_start.dtype == np.int32 and self._genePositions.gene_end.dtype == np.int32
            self._gs_val = np.ascontiguousarray(self._genePositions.gene_start, np.int32 if keep32 else np.int64)
            self._ge_val = np.ascontiguousarray(self._genePositions.gene_end, np.int32 if keep32 else np.int64)

    @property
    def _gs(self):
        self._positions_from_mapping()
        return self._gs_val

    @property
    def _ge(self):
        self._positions_from_mapping()
        return self._ge_val

    def _host_positions(self, gene_positions):
        """flat int64 (start, end) per gene, aligned with the tokens"""
        offs = self._read_off
        if isinstance(gene_positions, TokenizedPositions):
            gene_positions = gene_positions.settled()   # (redirected / hand-set reads gathered into arrays of their own)
            # (int32 arrays stay int32: they cross PCIe as they are, amg_set_positions32)
            keep32 = gene_positions.gene_start.dtype == np.int32 and gene_positions.gene_end.dtype == np.int32
            self._gs_val = np.ascontiguousarray(gene_positions.gene_start, np.int32 if keep32 else np.int64)
            self._ge_val = np.ascontiguousarray(gene_positions.gene_end, np.int32 if keep32 else np.int64)
            assert len(self._gs_val) == int(offs[-1]) == len(self._ge_val), "positions do not match the gene calls"
        elif gene_positions:
            n = int(offs[-1])
            gs, ge = np.empty(n, np.int64), np.empty(n, np.int64)
            for r, rid in enumerate(self._read_ids):
                a, b = int(offs[r]), int(offs[r + 1])
                if b > a:
                    p = gene_positions[rid]
                    # Read.get_geneMers indexes positions[i] / positions[i + k - 1] (construct_read.py:48-52)
                    gs[a:b] = [x[0] for x in p[: b - a]]
                    ge[a:b] = [x[1] for x in p[: b - a]]
            self._gs_val, self._ge_val = gs, ge

    def _upload_positions(self):
        if self._gs_val is not None:
            self._engine.set_positions(self._gs_val, self._ge_val, None)
        self._positions_pending = False

    def _note_short_reads(self):
        offs = self._read_off
        short = np.flatnonzero(np.diff(offs) < self._kmerSize).tolist() if len(offs) > 1 else []
        self._shortReads = {self._read_ids[r]: self._reads[self._read_ids[r]] for r in short}   # :53-55

    @property
    def _read_index(self):
        if self._read_index_ is None:
            src = self._reads
            if isinstance(src, TokenizedReads) and len(src.read_ids) == len(self._read_ids):
                self._read_index_ = src._idx()   # the mapping's own name -> row table (same reads, same order)
            else:
                self._read_index_ = dict(zip(self._read_ids, range(len(self._read_ids))))
        return self._read_index_

    def close(self):
        """hand the device engine back (the graph can no longer be queried); also done when the object dies"""
        view, self._view = getattr(self, "_view", None), None
        if view is not None:
            view.dispose()
        owner, self._reads_owner = getattr(self, "_reads_owner", None), None
        if getattr(self, "_borrowers", 0) > 0:   # graphs of build_many still read this one's device arrays
            self._close_pending = True
            return
        engine, self._engine = getattr(self, "_engine", None), None
        _release_engine(engine)   # (pooled once no correct_reads output lives in its buffers any more)
        if owner is not None:
            owner._borrowers -= 1
            if owner._borrowers == 0 and getattr(owner, "_close_pending", False):
                owner.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ------------------------------------------------------------------ pickling (graph_utils.py:108-122: loky workers
    # hand whole GeneMerGraph objects back to the parent)
    def __getstate__(self):
        """What travels is what the graph was made FROM — the reads and positions (array-backed mappings fetch what
        still lives on the device), the gene-mer size, the filter of a filtered build — and the device passes
        applied since, in order.  The device arrays themselves stay behind: every pass is deterministic, so the
        receiving process builds the graph on its own GPU and replays the passes (__setstate__)."""
        if self._host_edits:
            raise TypeError("a GeneMerGraph edited by hand (add_node / add_edge / ...) lives in host objects of this "
                            "process only and cannot be pickled")
        for obj in (self._reads, self._genePositions):
            if isinstance(obj, TokenizedReads):
                obj.tokens          # noqa: B018  (fetches a DeviceCorrected)
            elif isinstance(obj, TokenizedPositions):
                obj.gene_start      # noqa: B018
        # the positions the graph was BUILT with: correct_reads replaces the positions of the reads it changes in the
        # caller's mapping (:1282-1284, :1328) while the graph keeps its reads, so the mapping as it is now may no
        # longer match them; the flat arrays made at construction do
        gs, ge = self._gs_val, self._ge_val
        if gs is None and isinstance(self._genePositions, TokenizedPositions):
            self._genePositions._from_device()
            gs, ge = self._genePositions._gs, self._genePositions._ge
        return {"reads": self._reads, "k": self._kmerSize, "positions": self._genePositions, "built_with": (gs, ge),
                "filter": self._build_filter, "passes": list(self._pass_log),
                "min_cov": (self._minNodeCoverage, self._minEdgeCoverage), "device": self._engine.device,
                "extra_to_correct": set(self._extra_to_correct), "lone_edges": set(self._lone_edges)}

    def __setstate__(self, state):
        gs, ge = state["built_with"]
        reads = state["reads"]
        built_with = None
        if gs is not None:
            ids = reads.read_ids if isinstance(reads, TokenizedReads) else list(reads)
            offs = reads.read_offsets if isinstance(reads, TokenizedReads) else None
            if offs is None:
                offs = np.zeros(len(ids) + 1, np.int64)
                np.cumsum([len(reads[r]) for r in ids], out=offs[1:])
            built_with = TokenizedPositions(ids, offs, gs, ge)
        self.__init__(reads, state["k"], built_with, device=state.get("device"), _filter=state["filter"])
        self._genePositions = state["positions"]
        eng = self._engine
        for what, arg in state["passes"]:
            if what == "filter":
                eng.filter(*arg)
            elif what == "remove_nodes":
                eng.remove_nodes(arg)
            elif what == "remove_edges":
                eng.remove_edges(arg)
            elif what == "remove_low_coverage_components":
                eng.remove_low_coverage_components(arg)
            elif what == "remove_short_linear_paths":
                eng.remove_short_linear_paths(*arg)
        self._pass_log = list(state["passes"])
        self._minNodeCoverage, self._minEdgeCoverage = state["min_cov"]
        self._extra_to_correct = set(state["extra_to_correct"])
        self._lone_edges = set(state.get("lone_edges", ()))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ view plumbing
    def _invalidate(self):
        self._view = None

    def _settle_leases(self):
        """What an earlier correct_reads of this graph left on the device (amira_amd.io.DeviceCorrected: the lazy
        mappings it handed back) is brought to the host before a pass that changes the graph: every such pass
        invalidates the engine's corrected set (amg_filter, amg_remove_nodes, amg_remove_short_linear_paths,
        amg_remove_low_coverage_components, amg_correct_reads), and the mappings must keep answering afterwards as
        the reference's dicts do."""
        for lease in list(self._engine._leases):
            lease.fetch()

    def _device_pass(self, what):
        """The incremental mutators of the reference class (add_node, add_edge, remove_edge,
        remove_node_from_reads, ... — used by its unit tests and its multi-process merge) edit the
        HOST view only; the device arrays no longer describe that graph, so device passes refuse it."""
        if self._host_edits:
            raise RuntimeError(f"{what}: this graph was edited through add_node / add_edge / remove_edge on the "
                               "host; build a GeneMerGraph from reads to run device passes")
        self._settle_leases()

    def _gene_obj(self, token):
        g = self._gene_cache.get(token)
        if g is None:
            g = self._gene_cache[token] = Gene(self._vocab.gene(token))
        return g

    def _hash_of_tokens(self, toks):
        return hashlib_hash(tuple([self._vocab.signed_hash(int(t)) for t in toks]))

    def _v(self):
        if self._view is not None:
            return self._view
        eng, vocab, k = self._engine, self._vocab, self._kmerSize
        nodes, edges = eng.nodes(), eng.edges()
        adj_off, adj_edge = eng.node_adj()
        v = _View()
        v._nh_table = None
        # the per-window node ids (240 MB per million reads), the window directions (60 MB) and the node -> reads lists
        # (computed on the device for ALL nodes, 4 bytes per window) are fetched when somebody asks for them: read-path
        # clustering asks the device for the windows of the reads it looks at instead (Engine.read_node_ids_of)
        arrays = v.arrays = _LazyArrays({"nodes": nodes, "edges": edges, "adj_off": adj_off, "adj_edge": adj_edge})
        arrays.makers["tok_node"] = lambda: {"tok_node": eng.read_node_ids()}
        arrays.makers["tok_dir"] = lambda: {"tok_dir": eng.read_dirs()}
        arrays.makers["node_reads_off"] = arrays.makers["node_reads"] = \
            lambda: dict(zip(("node_reads_off", "node_reads"), eng.node_reads()))
        D, E = len(nodes["coverage"]), len(edges["coverage"])
        v.alive = n_alive = nodes["alive"]
        v.node_hash = [None] * D
        v._node_obj = [None] * D
        v._nodes, v._nodes_complete, v._id_of = {}, False, {}
        v.node_of_hash = _HashToId(v)
        read_ids = self._read_ids
        n_tokens, n_cov, n_comp, n_fdir = nodes["tokens"], nodes["coverage"], nodes["component"], nodes["first_dir"]
        v.edge_hash = [None] * E
        v._edge_obj = [None] * E
        v._edges, v._edges_complete, v._n_edges = {}, False, E
        e_src, e_tgt, e_sdir, e_tdir, e_cov, e_alive = (edges["src"], edges["tgt"], edges["sdir"], edges["tdir"],
                                                         edges["coverage"], edges["alive"])

        def edge_list(lo, hi):
            return lambda: [v.edge_hash_of(e) for e in adj_edge[lo:hi].tolist() if e_alive[e]]

        # (nothing below may hold `self`: the view must not keep its graph alive)
        gene_cache, gene_name, flip, signed_hash = self._gene_cache, vocab.gene, vocab.flip, vocab.signed_hash

        def gene_obj(t):
            g = gene_cache.get(t)
            if g is None:
                g = gene_cache[t] = Gene(gene_name(t))
            return g

        def hash_node(i):
            if not n_alive[i]:
                return None
            h = v.node_hash[i] = hashlib_hash(tuple([signed_hash(t) for t in n_tokens[i].tolist()]))
            v._id_of[h] = i
            if v._nh_table is not None:
                v._nh_table[i] = h
            return h

        v._hash_node = hash_node

        def make_node(i):
            if not n_alive[i]:
                return None
            toks = n_tokens[i].tolist()
            canon = [gene_obj(t) for t in toks]
            rc = [gene_obj(flip(t)) for t in reversed(toks)]
            h = v.node_hash[i]
            if h is None:
                h = hashlib_hash(tuple([signed_hash(t) for t in toks]))
            node = _GraphNode(GeneMer._from_parts(canon, rc, int(n_fdir[i]), h))
            node.nodeCoverage = int(n_cov[i])
            node._component_ID = int(n_comp[i])
            node._lazy(lambda i=i: [read_ids[r] for r in
                                    arrays["node_reads"][arrays["node_reads_off"][i]:arrays["node_reads_off"][i + 1]].tolist()])
            node._amg_id = i
            lo, mid, hi = int(adj_off[2 * i]), int(adj_off[2 * i + 1]), int(adj_off[2 * i + 2])
            node._lazy_edges(edge_list(lo, mid), edge_list(mid, hi))
            v.node_hash[i] = h
            v._node_obj[i] = node
            v._nodes[h] = node
            v._id_of[h] = i
            if v._nh_table is not None:
                v._nh_table[i] = h
            return node

        v._make_node = make_node

        def make_edge(e):
            edge = Edge(v.node_at(int(e_src[e])), v.node_at(int(e_tgt[e])), int(e_sdir[e]), int(e_tdir[e]))
            edge.edgeCoverage = int(e_cov[e])
            edge._amg_id = e
            return edge

        v._make_edge = make_edge
        offs, nh = self._read_off, v.node_hash

        single_rows = [0]

        def window_ids(r):
            a, n = int(offs[r]), int(offs[r + 1] - offs[r]) - k + 1
            if n > 0 and single_rows[0] < 64 and not dict.__contains__(arrays, "tok_node"):
                # the first few reads somebody looks up are gathered on the device one by one (read-path clustering asks
                # for one read per allele); whoever keeps asking gets the whole per-window array, once
                single_rows[0] += 1
                return a, n, eng.read_node_ids_of([a], [n])[0].tolist()
            return a, n, arrays["tok_node"][a:a + n].tolist()   # (fetched on first use)

        def make_nodes(r):
            _, _, ids = window_ids(r)
            v.ensure_hashes(ids)
            return [nh[x] if x >= 0 else None for x in ids]

        def make_dirs(r):
            a, n, ids = window_ids(r)
            return [d if x >= 0 else None for d, x in zip(arrays["tok_dir"][a:a + n].tolist(), ids)]

        gs_all, ge_all = self._gs, self._ge

        def make_positions(r):
            a, n, ids = window_ids(r)
            if gs_all is None:
                return [None] * n
            s, e = gs_all[a:a + n].tolist(), ge_all[a + k - 1:a + k - 1 + n].tolist()
            return [(s[j], e[j]) if ids[j] >= 0 else None for j in range(n)]

        v.readNodes = _LazyReadLists(self, make_nodes)
        v.readNodeDirections = _LazyReadLists(self, make_dirs)
        v.readNodePositions = _LazyReadLists(self, make_positions)
        self._view = v
        return v

    def _node_id(self, node_or_hash):
        h = node_or_hash if isinstance(node_or_hash, int) else node_or_hash.__hash__()
        return self._v().node_of_hash[h]

    # ------------------------------------------------------------------ accessors (:104-163)
    def get_reads(self):
        return self._reads

    def get_short_read_annotations(self):
        return self._shortReads

    def get_gene_positions(self):
        return self._genePositions

    def get_short_read_gene_positions(self):
        return {r: self._genePositions[r] for r in self._shortReads}

    def get_readNodes(self):
        return self._v().readNodes

    def get_readNodeDirections(self):
        return self._v().readNodeDirections

    def get_readNodePositions(self):
        return self._v().readNodePositions

    def get_kmerSize(self):
        return self._kmerSize

    def get_minEdgeCoverage(self):
        return self._minEdgeCoverage

    def get_minNodeCoverage(self):
        return self._minNodeCoverage

    def set_minNodeCoverage(self, minNodeCoverage):
        self._minNodeCoverage = minNodeCoverage
        return self._minNodeCoverage

    def set_minEdgeCoverage(self, minEdgeCoverage):
        self._minEdgeCoverage = minEdgeCoverage
        return self._minEdgeCoverage

    def get_nodes(self):
        return self._v().nodes

    def get_edges(self):
        return self._v().edges

    def get_reads_to_correct(self):
        flags = self._engine.reads_to_correct()
        out = {self._read_ids[r] for r in np.nonzero(flags)[0].tolist()}
        out |= self._extra_to_correct
        return out

    def all_nodes(self):
        for h in self.get_nodes():
            yield self.get_nodes()[h]

    def get_reads_for_nodes(self, list_of_nodes):
        reads = set()
        for h in list_of_nodes:
            reads.update(self.get_node_by_hash(h).get_list_of_reads())
        return reads

    def get_nodes_containing_read(self, readId):
        v = self._v()
        found = [v.node_by_hash(h) for h in self.get_readNodes()[readId] if h is not None]
        return [n for n in found if n is not None]

    def get_node_by_hash(self, nodeHash):
        node = self._v().node_by_hash(nodeHash)
        if node is None:
            raise KeyError(nodeHash)
        return node

    def get_node(self, geneMer):
        node = self._v().node_by_hash(geneMer.__hash__())
        assert node is not None, "This gene-mer is not in the graph"
        return node

    def get_edge_by_hash(self, edgeHash):
        return self._v().edge_by_hash(edgeHash)

    def get_total_number_of_nodes(self):
        return self._engine.counts()["n_live_nodes"]

    def get_total_number_of_edges(self):
        return self._engine.counts()["n_live_edges"]

    def get_total_number_of_reads(self):
        return len(self._reads)

    def get_nodes_containing(self, geneOfInterest):
        """nodes whose canonical gene-mer holds the gene, in node order (:223-244)."""
        assert not (geneOfInterest[0] == "+" or geneOfInterest[0] == "-"), (
            "Strand information cannot be present for any specified genes")
        assert isinstance(geneOfInterest, str), "Gene of interest is the wrong type"
        v = self._v()
        ids = self._node_ids_containing([geneOfInterest])
        return [v.node_at(i) for i in ids]

    def _node_ids_containing(self, genes):
        v = self._v()
        toks = []
        for g in genes:
            r = self._vocab.rank.get(g)
            if r is not None:
                V = max(self._vocab.V, 1)
                toks += [V + r, V - 1 - r]
        nt = v.arrays["nodes"]["tokens"]
        if not toks or nt.size == 0:
            return []
        mask = np.isin(nt, np.asarray(toks, dtype=nt.dtype)).any(axis=1) & (v.alive != 0)
        return np.nonzero(mask)[0].tolist()

    # ------------------------------------------------------------------ incremental construction (:165-324)
    # The reference builds its graph through these calls; here the device builds it and they remain
    # for callers that edit a graph by hand (the reference's unit tests, its multi-process merge).
    # They work on the materialised host view and switch the graph to host-only (_device_pass).
    def add_node_to_read(self, node, readId, node_direction, node_position=None):
        v = self._v()
        self._host_edits = True
        if readId not in v.readNodes:
            v.readNodes[readId], v.readNodeDirections[readId], v.readNodePositions[readId] = [], [], []
        v.readNodes[readId].append(node.__hash__())
        v.readNodeDirections[readId].append(node_direction)
        v.readNodePositions[readId].append(node_position)
        return v.readNodes[readId]

    def add_node_to_nodes(self, node, nodeHash):
        self._host_edits = True
        self._v().nodes[nodeHash] = node

    def add_node(self, geneMer, reads):
        """insert-or-find by hash; the first GeneMer object seen stays on the node (:196-212)"""
        nodes = self._v().nodes
        h = geneMer.__hash__()
        node = nodes.get(h)
        if node is None:
            node = _GraphNode(geneMer)
            node._lazy(None)
            node.listOfReads = []
            self.add_node_to_nodes(node, h)
        for r in reads:
            node.add_read(r)
            self._host_edits = True
        return node

    def create_edges(self, sourceNode, targetNode, sourceGeneMerDirection, targetGeneMerDirection):
        """the adjacency as seen from either end: (A, B, dA, dB) and (B, A, -dB, -dA) (:246-262)"""
        return (Edge(sourceNode, targetNode, sourceGeneMerDirection, targetGeneMerDirection),
                Edge(targetNode, sourceNode, targetGeneMerDirection * -1, sourceGeneMerDirection * -1))

    def add_edge_to_edges(self, edge):
        edges = self._v().edges
        h = edge.__hash__()
        if h not in edges:   # the first object of a class is the one that stays (:268-277)
            self._host_edits = True
            edges[h] = edge
        return edges[h]

    def add_edges_to_graph(self, sourceToTargetEdge, reverseTargetToSourceEdge):
        return self.add_edge_to_edges(sourceToTargetEdge), self.add_edge_to_edges(reverseTargetToSourceEdge)

    def add_edge_to_node(self, node, edge):
        """forward or backward list by the STORED edge's source direction (:287-298)"""
        self._host_edits = True
        if edge.get_sourceNodeDirection() == 1:
            node.add_forward_edge_hash(edge.__hash__())
        if edge.get_sourceNodeDirection() == -1:
            node.add_backward_edge_hash(edge.__hash__())
        return node

    def add_edge(self, sourceGeneMer, targetGeneMer):
        sourceNode = self.add_node(sourceGeneMer, [])
        targetNode = self.add_node(targetGeneMer, [])
        forward, backward = self.add_edges_to_graph(*self.create_edges(
            sourceNode, targetNode, sourceGeneMer.get_geneMerDirection(), targetGeneMer.get_geneMerDirection()))
        self.add_edge_to_node(sourceNode, forward)
        self.add_edge_to_node(targetNode, backward)
        return forward, backward

    def remove_edge_from_edges(self, edgeHash):
        self._host_edits = True
        del self._v().edges[edgeHash]

    def remove_edge(self, edgeHash):
        """drop an edge from the graph and from its source node's list; unknown hashes are ignored (:409-428).
        On a graph that has not been edited by hand the edge is removed ON THE DEVICE (amg_remove_edges), so that
        device passes keep working afterwards, as filter_graph does after remove_edge in the reference."""
        edges = self._v().edges
        if edgeHash not in edges:
            return
        edge = edges[edgeHash]
        if not self._host_edits and getattr(edge, "_amg_id", None) is not None:
            self._settle_leases()
            self._engine.remove_edges([edge._amg_id])
            # a loop that removes many edges one at a time replays as ONE device call (pickling: rebuild + replay)
            if self._pass_log and self._pass_log[-1][0] == "remove_edges":
                self._pass_log[-1][1].append(int(edge._amg_id))
            else:
                self._pass_log.append(("remove_edges", [int(edge._amg_id)]))
            # (source id, target id, source direction, target direction): tip clipping wants to know whether the twin
            # of every edge removed one at a time is gone as well
            self._lone_edges.add((edge.get_sourceNode()._amg_id, edge.get_targetNode()._amg_id,
                                  edge.get_sourceNodeDirection(), edge.get_targetNodeDirection()))
            # the view follows in place instead of being thrown away (re-fetching every array of the graph per removed
            # edge made such a loop quadratic): the edge's alive flag, which the lazily made edge lists look at, the
            # source node's list if it exists already, and the edge's entry
            v = self._v()
            source = edge.get_sourceNode()   # (a list not made yet is made here, WITH the edge, and loses it below)
            if edge.get_sourceNodeDirection() == 1:
                source.remove_forward_edge_hash(edgeHash)
            if edge.get_sourceNodeDirection() == -1:
                source.remove_backward_edge_hash(edgeHash)
            v.arrays["edges"]["alive"][edge._amg_id] = 0
            del v.edges[edgeHash]
            return
        source = edge.get_sourceNode()
        if edge.get_sourceNodeDirection() == 1:
            source.remove_forward_edge_hash(edgeHash)
        if edge.get_sourceNodeDirection() == -1:
            source.remove_backward_edge_hash(edgeHash)
        self.remove_edge_from_edges(edgeHash)

    def remove_node_from_reads(self, node_to_remove):
        """every occurrence of the node becomes None on its reads, which join the reads to correct (:442-461)"""
        v = self._v()
        self._host_edits = True
        h = node_to_remove.__hash__()
        for read_id in node_to_remove.get_reads():
            keep = [x != h for x in v.readNodes[read_id]]
            for lists in (v.readNodes, v.readNodeDirections, v.readNodePositions):
                lists[read_id] = [x if k else None for x, k in zip(lists[read_id], keep)]
            self._extra_to_correct.add(read_id)

    def dfs_component(self, start, component_id, visited=None):
        visited = set() if visited is None else visited
        visited.add(start.__hash__())
        start.set_component(component_id)
        for neighbor in self.get_all_neighbors(start):
            if neighbor.__hash__() not in visited:
                self.dfs_component(neighbor, component_id, visited)

    def assign_component_ids(self):
        """component ids 1, 2, ... in order of each component's first node (:920-927); the device
        build has already done this — the method re-labels the host view (after host edits)"""
        visited, component_id = set(), 1
        for h, node in self.get_nodes().items():
            if h not in visited:
                self.dfs_component(node, component_id, visited)
                component_id += 1

    def calculate_mean_node_coverage(self):
        return statistics.mean(self.get_all_node_coverages())

    # ------------------------------------------------------------------ topology (:326-400)
    def get_degree(self, node):
        return len(node.get_forward_edge_hashes()) + len(node.get_backward_edge_hashes())

    def get_forward_edges(self, node):
        return [self.get_edge_by_hash(h) for h in node.get_forward_edge_hashes()]

    def get_backward_edges(self, node):
        return [self.get_edge_by_hash(h) for h in node.get_backward_edge_hashes()]

    def get_forward_neighbors(self, node):
        return [e.get_targetNode() for e in self.get_forward_edges(node)]

    def get_backward_neighbors(self, node):
        return [e.get_targetNode() for e in self.get_backward_edges(node)]

    def get_all_neighbors(self, node):
        return self.get_forward_neighbors(node) + self.get_backward_neighbors(node)

    def get_all_neighbor_hashes(self, node):
        return set(n.__hash__() for n in self.get_all_neighbors(node))

    def check_if_nodes_are_adjacent(self, sourceNode, targetNode):
        return (targetNode.__hash__() in self.get_all_neighbor_hashes(sourceNode)
                and sourceNode.__hash__() in self.get_all_neighbor_hashes(targetNode))

    def get_edge_hashes_between_nodes(self, sourceNode, targetNode):
        """(source->target hash, target->source hash); a pair of LISTS when either side has
        several (the reference's multi-edge quirk, :364-386)."""
        assert self.check_if_nodes_are_adjacent(sourceNode, targetNode)
        s2t = [e.__hash__() for e in self.get_forward_edges(sourceNode) + self.get_backward_edges(sourceNode)
               if e.get_targetNode() == targetNode]
        t2s = [e.__hash__() for e in self.get_forward_edges(targetNode) + self.get_backward_edges(targetNode)
               if e.get_targetNode() == sourceNode]
        if not (len(s2t) > 1 or len(t2s) > 1):
            return (s2t[0], t2s[0])
        return (s2t, t2s)

    def get_edges_between_nodes(self, sourceNode, targetNode):
        a, b = self.get_edge_hashes_between_nodes(sourceNode, targetNode)
        if not (isinstance(a, list) or isinstance(b, list)):
            return self.get_edge_by_hash(a), self.get_edge_by_hash(b)
        return [self.get_edge_by_hash(h) for h in a], [self.get_edge_by_hash(h) for h in b]

    # ------------------------------------------------------------------ removals + filter
    def remove_node(self, node):
        """remove a node, its edges (both directions) and mask it on its reads (:463-484)."""
        h = node.__hash__()
        assert h in self.get_nodes(), "This node is not in the graph"
        assert node == self.get_node_by_hash(h)
        if self._host_edits:   # host-edited graph: the reference's own steps on the view
            self.remove_node_from_reads(node)
            for edge_hash in set(node.get_forward_edge_hashes() + node.get_backward_edge_hashes()):
                target = self.get_edge_by_hash(edge_hash).get_targetNode()
                for e in self.get_edge_hashes_between_nodes(node, target):
                    self.remove_edge(e)
            del self.get_nodes()[h]
            return
        i = self._node_id(h)
        self._settle_leases()
        self._engine.remove_nodes([i])
        self._pass_log.append(("remove_nodes", [int(i)]))
        self._invalidate()

    def _remove_node_ids(self, ids):
        if len(ids):
            self._settle_leases()
            self._engine.remove_nodes(np.asarray(ids, dtype=np.int32))
            self._pass_log.append(("remove_nodes", np.asarray(ids, dtype=np.int32)))
            self._invalidate()

    def list_nodes_to_remove(self, minNodeCoverage):
        return {n for n in self.all_nodes() if not n.get_node_coverage() > minNodeCoverage - 1}

    def list_edges_to_remove(self, minEdgeCoverage, nodesToRemove):
        doomed = set()
        for h, e in self.get_edges().items():
            if not e.get_edge_coverage() > minEdgeCoverage - 1:
                doomed.add(h)
            if e.get_sourceNode() in nodesToRemove or e.get_targetNode() in nodesToRemove:
                doomed.add(h)
        return doomed

    def filter_graph(self, minNodeCoverage, minEdgeCoverage):
        """device coverage filter: nodes with coverage < minNodeCoverage, edges with coverage
        < minEdgeCoverage or a removed endpoint; reads through removed nodes are masked and
        queued for correction (:523-540)."""
        self._device_pass("filter_graph")
        self.set_minNodeCoverage(minNodeCoverage)
        self.set_minEdgeCoverage(minEdgeCoverage)
        self._engine.filter(max(int(minNodeCoverage), 0), max(int(minEdgeCoverage), 0))
        self._pass_log.append(("filter", (max(int(minNodeCoverage), 0), max(int(minEdgeCoverage), 0))))
        self._invalidate()
        return self

    def remove_low_coverage_components(self, min_component_coverage):
        self._device_pass("remove_low_coverage_components")
        self._engine.remove_low_coverage_components(max(int(min_component_coverage), 0))
        self._pass_log.append(("remove_low_coverage_components", max(int(min_component_coverage), 0)))
        self._invalidate()

    def remove_short_linear_paths(self, min_length, sample_genesOfInterest={}, _lazy_hashes=False):
        """tip clipping on the device (:679-720); returns the hashes of the removed nodes (a list, as the reference
        does; _lazy_hashes: the drivers of graph_utils, which ignore the result, ask for a Sequence whose hashes are
        only made when somebody looks at them)."""
        self._device_pass("remove_short_linear_paths")
        if any((t, s_, -td, -sd) not in self._lone_edges for (s_, t, sd, td) in self._lone_edges):
            # the walk of the device kernel takes an adjacency from either end; the reference's, on a graph where
            # remove_edge took ONE direction of an adjacency away, depends on which end it comes from
            raise RuntimeError("remove_short_linear_paths: remove_edge has removed one direction of an adjacency and "
                               "left its twin; remove the twin (the edge target -> source with both directions "
                               "negated) as well, as filter_graph and remove_node do")
        protect = None
        if sample_genesOfInterest:
            ids = self._node_ids_containing(list(sample_genesOfInterest))
            if ids:
                protect = np.zeros(self._engine.counts()["n_nodes"], np.uint8)
                protect[ids] = 1
        v = self._view   # the object view is NOT made for this: the hashes of the removed nodes come from their tokens
        tokens = None if v is not None else self._engine.nodes()["tokens"]
        removed = self._engine.remove_short_linear_paths(int(min_length), protect)
        self._pass_log.append(("remove_short_linear_paths", (int(min_length), protect)))
        if v is not None:
            hashes = [v.hash_at(i) for i in removed.tolist()]
        elif _lazy_hashes:
            # (the drivers, which ignore the result): the hashes — a sha256 per removed node, tens of thousands of
            # them in a first sweep — are made when somebody looks at them
            hashes = _LazyHashes(tokens[removed], self._vocab)
        else:
            hashes = [self._hash_of_tokens(tokens[i].tolist()) for i in removed.tolist()]
        if len(hashes):
            self._invalidate()
        return hashes

    def remove_non_AMR_associated_nodes(self, genesOfInterest):
        """drop every node none of whose reads touches a node holding a gene of interest (:2941-2959)."""
        v = self._v()
        amr_ids = self._node_ids_containing(list(genesOfInterest))
        tok_node = v.arrays["tok_node"]
        D = len(v.alive)
        is_amr = np.zeros(D + 1, bool)
        is_amr[amr_ids] = True
        live = tok_node >= 0
        read_of_tok = np.repeat(np.arange(len(self._read_ids)), np.diff(self._read_off))
        reads_hit = np.zeros(len(self._read_ids), bool)
        reads_hit[read_of_tok[live & is_amr[np.where(live, tok_node, D)]]] = True
        node_hit = np.zeros(D, bool)
        node_hit[tok_node[live & reads_hit[read_of_tok]]] = True
        self._remove_node_ids(np.nonzero((v.alive != 0) & ~node_hit)[0])

    def remove_junk_reads(self, error_rate):
        """split reads by their fraction of masked nodes; Python round() (:1398-1420)."""
        if not self._host_edits:
            return self._remove_junk_reads_arrays(error_rate)
        new_reads, new_positions, rejected_reads, rejected_read_positions = {}, {}, {}, {}
        for read_id, nodes in self.get_readNodes().items():
            allowed = round(len(nodes) * (1 - error_rate))
            masked = sum(1 for n in nodes if n is None)
            if masked <= allowed:
                new_reads[read_id] = self._reads[read_id]
                new_positions[read_id] = self._genePositions[read_id]
            else:
                rejected_reads[read_id] = self._reads[read_id]
                rejected_read_positions[read_id] = self._genePositions[read_id]
        return new_reads, new_positions, rejected_reads, rejected_read_positions

    def _remove_junk_reads_arrays(self, error_rate):
        """remove_junk_reads from the device's per-window node ids: masked windows per read by one segmented sum, the
        reference's round() (half to even, in double arithmetic) by numpy's rint of the same product"""
        tok_node, _ = self._engine.read_nodes()
        offs, k = self._read_off, self._kmerSize
        n_win = np.maximum(np.diff(offs) - k + 1, 0) if len(offs) > 1 else np.zeros(0, np.int64)
        rows = np.flatnonzero(n_win > 0)            # the reads get_readNodes() lists, in read order
        masked_tok = (tok_node == -2).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(masked_tok)])
        masked = csum[offs[rows] + n_win[rows]] - csum[offs[rows]]
        allowed = np.rint(n_win[rows].astype(np.float64) * (1 - error_rate))
        ok = masked <= allowed
        keep_rows, drop_rows = rows[ok], rows[~ok]
        if self._tokenized_io(self._genePositions is not None) and self._genePositions is not None:
            return (self._reads.subset(keep_rows), self._genePositions.subset(keep_rows),
                    self._reads.subset(drop_rows), self._genePositions.subset(drop_rows))
        ids, reads, pos = self._read_ids, self._reads, self._genePositions
        keep = [ids[i] for i in keep_rows.tolist()]
        drop = [ids[i] for i in drop_rows.tolist()]
        return ({r: reads[r] for r in keep}, {r: pos[r] for r in keep},
                {r: reads[r] for r in drop}, {r: pos[r] for r in drop})

    def get_valid_reads_only(self):
        fix = self.get_reads_to_correct()
        return {r: g for r, g in self._reads.items() if r not in fix}

    # ------------------------------------------------------------------ correction (:1123-1396)
    def correct_reads(self, fastq_data):
        """Re-thread every queued read through the filtered graph on the device and return
        (corrected gene calls, corrected gene positions) like the reference: untouched reads
        keep their original list objects, reads whose nodes were all removed disappear,
        self._genePositions is updated in place for changed reads."""
        self._device_pass("correct_reads")
        eng, vocab = self._engine, self._vocab
        have_pos = bool(self._genePositions)
        if self._positions_pending:   # a graph of build_many: its engine has not seen the positions yet
            self._upload_positions()
        if have_pos and hasattr(fastq_data, "lengths_array"):   # amira_amd.io.ReadLengths: no per-read loop
            eng.set_read_lengths(fastq_data.lengths_array(self._read_ids, getattr(self._reads, "source_rows", None),
                                                          getattr(self._reads, "source_ids", None)))
        elif have_pos:
            flags = eng.reads_to_correct()
            lengths = np.zeros(len(self._read_ids), np.int64)
            for r in np.nonzero(flags)[0].tolist():
                try:  # only consulted when a trailing gene needs an inferred end (:1685)
                    lengths[r] = len(fastq_data[self._read_ids[r]]["sequence"])
                except (KeyError, TypeError, IndexError):
                    lengths[r] = 0
            eng.set_read_lengths(lengths)
        n_reads, n_tokens = eng.correct_reads()   # (an earlier correction's output was fetched by _device_pass)
        if self._tokenized_io(have_pos):
            return self._corrected_as_arrays(eng.corrected_index(n_reads), n_reads, n_tokens, have_pos)
        out = eng.corrected(n_reads, n_tokens, have_pos)
        offs, toks = out["read_offsets"].tolist(), out["tokens"]
        corrected_genes, corrected_gene_positions = {}, {}
        for i, (orig, changed) in enumerate(zip(out["orig_read"].tolist(), out["changed"].tolist())):
            read_id = self._read_ids[orig]
            a, b = offs[i], offs[i + 1]
            if changed:
                corrected_genes[read_id] = vocab.decode(toks[a:b])
                if have_pos:
                    self._genePositions[read_id] = list(zip(out["gene_start"][a:b].tolist(),
                                                            out["gene_end"][a:b].tolist()))
            else:
                corrected_genes[read_id] = self._reads[read_id]
            if have_pos:
                corrected_gene_positions[read_id] = self._genePositions[read_id]
        return corrected_genes, corrected_gene_positions

    def _tokenized_io(self, have_pos):
        """the caller handed the reads (and positions) over as arrays (amira_amd.io.TokenizedReads /
        TokenizedPositions): corrections go back the same way, with no per-read Python work"""
        return isinstance(self._reads, TokenizedReads) and (not have_pos or isinstance(self._genePositions,
                                                                                         TokenizedPositions))

    def _corrected_as_arrays(self, out, n_reads, n_tokens, have_pos):
        """correct_reads' result for array inputs: the corrected calls as a TokenizedReads, their positions as a
        TokenizedPositions — both over a DeviceCorrected: the genes and positions stay on the device until somebody
        reads them on the host, and a GeneMerGraph built from the two takes them over device to device.  Reads the
        correction changed are redirected to their new positions in the caller's mapping (the reference updates
        gene_positions in place, :1282-1284, :1328)"""
        from .io import DeviceCorrected
        orig = out["orig_read"]
        if len(orig) == len(self._read_ids):   # nothing dropped (the order is kept): the very same reads
            ids = self._read_ids
        else:
            ids = self._read_ids_array()[orig].tolist()
        offs = out["read_offsets"]
        src = getattr(self._reads, "source_rows", None)
        on_device = DeviceCorrected(self._engine, n_reads, n_tokens, have_pos)
        reads = TokenizedReads(self._vocab, on_device, offs, ids,
                               source_rows=orig.astype(np.int64) if src is None else src[orig],
                               source_ids=self._reads.read_ids if src is None else self._reads.source_ids)
        if not have_pos:
            return reads, {}
        positions = TokenizedPositions(ids, offs, on_device, on_device)
        changed = np.flatnonzero(out["changed"])
        if len(changed):
            self._genePositions.replace_rows(orig[changed], positions, changed)
        return reads, positions

    def _read_ids_array(self):
        if getattr(self, "_read_ids_arr", None) is None:
            self._read_ids_arr = np.asarray(self._read_ids, dtype=object)
        return self._read_ids_arr

    def _names_of_rows(self, rows):
        """read names of a few rows (an int array), in that order: straight out of the list while the rows are few —
        an object array of a million names costs more to make than a few thousand lookups"""
        if getattr(self, "_read_ids_arr", None) is None and 8 * len(rows) < len(self._read_ids):
            ids = self._read_ids
            return [ids[i] for i in rows.tolist()]
        return self._read_ids_array()[rows].tolist()

    def correct_single_read(self, read_id, readNodes, fastq_data):
        """host twin of one read of the device correction (:1136-1151), for callers that correct a
        single read by hand; correct_reads runs all of them on the device"""
        if read_id not in self.get_reads_to_correct():
            return self.get_reads()[read_id]
        if all(n is None for n in readNodes[read_id]):
            return []
        start, end = self.find_read_boundaries(readNodes[read_id])
        new_genes_on_read = self.process_read_correction(read_id, readNodes, start, end, fastq_data)
        if self.get_gene_positions():
            assert len(new_genes_on_read) == len(self.get_gene_positions()[read_id])
        return new_genes_on_read

    def generate_replacement_dict(self, corrected, pair):
        first, last = pair
        return {pair: self.new_find_paths_between_nodes(corrected[first][0], corrected[last][0],
                                                        self.get_kmerSize() * 2, corrected[first][1])}

    def get_possible_paths(self, nodes_on_read, replacementDict, start, end):
        """every way of filling the read's None runs, as (node hashes, directions); windows without a
        node are skipped (the reference's upstream / downstream extension is disabled, :1205-1263)"""
        return [([n for n, _ in filled if n], [d for n, d in filled if n])
                for filled in self.insert_elements(nodes_on_read, replacementDict)]

    def process_read_correction(self, read_id, readNodes, start, end, fastq_data):
        k = self.get_kmerSize()
        directions = self.get_readNodeDirections()[read_id]
        nodes_on_read = [(readNodes[read_id][i], directions[i]) for i in range(len(readNodes[read_id]))]
        path_terminals = self.identify_path_terminals(readNodes[read_id], start, end)
        if len(path_terminals) == 0:   # only the ends were lost: slice genes and positions
            if self.get_gene_positions():
                self.get_gene_positions()[read_id] = self.get_gene_positions()[read_id][start:end + k]
            kept = nodes_on_read[start:end + 1]
            return self.get_annotation_for_read([n for n, _ in kept], [d for _, d in kept], read_id)
        replacementDict = {}
        for pair in path_terminals:
            replacementDict.update(self.generate_replacement_dict(nodes_on_read, pair))
        possible_paths = self.get_possible_paths(nodes_on_read, replacementDict, start, end)
        if possible_paths == []:
            return self.get_reads()[read_id]
        original = self.get_reads()[read_id]
        best_shared, best_coverage = 0, 0
        for nodes, dirs in possible_paths:   # most shared genes, then strictly higher mean coverage
            coverage = self.get_coverage_of_path(nodes)
            genes = self.get_annotation_for_read(nodes, dirs, read_id)
            shared = len(set(genes).intersection(original))
            if shared > best_shared or (shared == best_shared and coverage > best_coverage):
                closest, best_shared, best_coverage = genes, shared, coverage
        old_positions, taken, new_positions = self.get_gene_positions()[read_id], 0, []
        for new_gene, old_gene in self.needleman_wunsch(closest, original):
            if new_gene == "*":
                taken += 1
            elif old_gene != new_gene:
                new_positions.append((None, None))
            else:
                new_positions.append(old_positions[taken])
                taken += 1
        self.get_gene_positions()[read_id] = self.replace_invalid_gene_positions(new_positions, fastq_data, read_id)
        return closest

    def score(self, x, y):
        return int(x == y)

    def find_read_boundaries(self, readNode):
        start, end = 0, len(readNode) - 1
        for i, node in enumerate(readNode):
            if node:
                start = i
                break
        for i, node in enumerate(reversed(readNode)):
            if node:
                end = len(readNode) - 1 - i
                break
        return start, end

    def identify_path_terminals(self, corrected, start, end):
        terminals = []
        for i in range(len(corrected)):
            if start <= i <= end and not corrected[i]:
                if corrected[i - 1]:
                    path_start = i - 1
                if corrected[i + 1]:
                    terminals.append((path_start, i + 1))
        return terminals

    def insert_elements(self, base_list, insert_dict):
        if len(insert_dict) == 0:
            return [base_list]
        choices = [[(key, p) for p in paths] for key, paths in insert_dict.items()]
        results = []
        for combination in product(*choices):
            cur, shift = base_list[:], 0
            for (start, end), path in combination:
                cur[start + shift:end + shift + 1] = path
                shift += len(path) - (end - start + 1)
            results.append(cur)
        return results

    def new_find_paths_between_nodes(self, start_hash, end_hash, distance, current_direction,
                                     path=None, seen_nodes=None):
        """host twin of the device DFS in k_corr_gapped (:2292-2342)."""
        path = [] if path is None else path
        seen_nodes = set() if seen_nodes is None else seen_nodes
        path.append((start_hash, current_direction))
        seen_nodes.add(start_hash)
        if (end_hash and start_hash == end_hash and len(path) <= distance) or (
                end_hash is None and len(path) - 1 == distance):
            return [list(path)]
        if len(path) - 1 > distance:
            return []
        node = self.get_node_by_hash(start_hash)
        hops = (node.get_forward_edge_hashes() if current_direction == 1
                else node.get_backward_edge_hashes() if current_direction == -1 else [])
        found = []
        for edge_hash in hops:
            edge = self.get_edge_by_hash(edge_hash)
            nxt = edge.get_targetNode().__hash__()
            if nxt not in seen_nodes:
                found.extend(self.new_find_paths_between_nodes(
                    nxt, end_hash, distance, edge.get_targetNodeDirection(), list(path), seen_nodes | {nxt}))
        return found

    def get_coverage_of_path(self, path):
        return statistics.mean([self.get_node_by_hash(n).get_node_coverage() for n in path])

    def get_annotation_for_read(self, listOfNodes, listOfNodeDirections, read_id):
        assert len(listOfNodes) == len(listOfNodeDirections), (
            f"The number of nodes and node directions for read {read_id} are not the same")
        if not listOfNodeDirections:
            listOfNodeDirections = self.get_readNodeDirections()[read_id]
        oriented = lambda n, d: (self.get_gene_mer_genes(n) if d == 1 else self.get_reverse_gene_mer_genes(n))
        if len(listOfNodes) == 1:
            d = listOfNodeDirections[0]
            if d != 1 and d != -1:
                raise ValueError(f"Gene-mer direction for a node with 1 read cannot be {d}")
            return oriented(self.get_node_by_hash(listOfNodes[0]), d)
        genes = []
        for i, h in enumerate(listOfNodes):
            node, d = self.get_node_by_hash(h), listOfNodeDirections[i]
            if i == 0:
                genes += oriented(node, d)[:-1]
            if d:
                genes.append(oriented(node, d)[-1])
        assert None not in genes
        return genes

    def needleman_wunsch(self, x, y):
        """host twin of k_corr_nw: match 1 / mismatch 0 / gap -1, borders -index, ties broken
        by max over (score, pointer) tuples => UP > LEFT > DIAG (:1433-1480)."""
        N, M = len(x), len(y)
        if N * M > 16 and not os.environ.get("AMG_NW_PYTHON"):
            # beyond a handful of cells: the same recurrence and tie order on interned genes in libamg (amg_nw_align)
            code = {}
            try:
                xs = np.fromiter((code.setdefault(g, len(code)) for g in x), np.int32, N)
                ys = np.fromiter((code.setdefault(g, len(code)) for g in y), np.int32, M)
            except TypeError:
                xs = None   # (unhashable items: the table below compares them as they are)
            if xs is not None:
                ops, n_ops = np.empty(N + M, np.int8), C.c_int32(0)
                _ffi.check(_ffi.lib.amg_nw_align(_ffi.ptr(xs), N, _ffi.ptr(ys), M, _ffi.ptr(ops), C.byref(n_ops)))
                alignment, i, j = [], 0, 0
                for op in ops[:n_ops.value].tolist():
                    if op == 0:
                        alignment.append((x[i], y[j]))
                        i, j = i + 1, j + 1
                    elif op == 1:
                        alignment.append((x[i], "*"))
                        i += 1
                    else:
                        alignment.append(("*", y[j]))
                        j += 1
                return alignment
        DIAG, LEFT, UP = (-1, -1), (-1, 0), (0, -1)
        F, Ptr = {(-1, -1): 0}, {}
        for i in range(N):
            F[i, -1] = -i
        for j in range(M):
            F[-1, j] = -j
        for i in range(N):
            for j in range(M):
                F[i, j], Ptr[i, j] = max((F[i - 1, j - 1] + int(x[i] == y[j]), DIAG),
                                         (F[i - 1, j] - 1, LEFT), (F[i, j - 1] - 1, UP))
        alignment = []
        i, j = N - 1, M - 1
        while i >= 0 and j >= 0:
            step = Ptr[i, j]
            alignment.append((x[i], y[j]) if step == DIAG else (x[i], "*") if step == LEFT else ("*", y[j]))
            i, j = i + step[0], j + step[1]
        while i >= 0:
            alignment.append((x[i], "*"))
            i -= 1
        while j >= 0:
            alignment.append(("*", y[j]))
            j -= 1
        return alignment[::-1]

    def replace_invalid_gene_positions(self, new_positions, fastq_data, read_id):
        prev_end = 0
        for i, (start, end) in enumerate(new_positions):
            if end is not None:
                prev_end = end
            if start is None and end is None:
                next_start = next((p[0] for p in new_positions[i + 1:] if p[0] is not None), None)
                if next_start is not None:
                    new_positions[i] = (prev_end, next_start)
                else:
                    new_positions[i] = (prev_end, len(fastq_data[read_id]["sequence"]) - 1)
        return new_positions

    # ------------------------------------------------------------------ gene strings / unitigs
    def get_gene_mer_genes(self, sourceNode):
        return [convert_int_strand_to_string(g.get_strand()) + g.get_name()
                for g in sourceNode.get_canonical_geneMer()]

    def get_reverse_gene_mer_genes(self, sourceNode):
        return [convert_int_strand_to_string(g.get_strand()) + g.get_name()
                for g in sourceNode.get_reverse_geneMer()]

    def get_gene_mer_label(self, sourceNode):
        return "~~~".join(self.get_gene_mer_genes(sourceNode))

    def get_nodes_with_degree(self, degree):
        assert isinstance(degree, int), "The input degree must be an integer."
        return [n for n in self.all_nodes() if self.get_degree(n) == degree]

    def reverse_list_of_genes(self, list_of_genes):
        return [("-" if g[0] == "+" else "+") + g[1:] for g in reversed(list_of_genes)]

    def get_genes_in_unitig(self, listOfNodes):
        """gene strings spelled by a node path (:617-677): extend to the right, and if that
        fails at any step redo the whole path extending to the left."""
        if len(listOfNodes) == 1:
            return self.get_gene_mer_genes(self.get_node_by_hash(listOfNodes[0]))
        if not self._host_edits and not os.environ.get("AMG_BUBBLES_BY_OBJECTS"):
            genes = self._genes_in_unitig_from_arrays(listOfNodes)
            if genes is not None:
                return genes
        k1 = self._kmerSize - 1

        def first_orientation():
            a, b = self.get_node_by_hash(listOfNodes[0]), self.get_node_by_hash(listOfNodes[1])
            edge = self.get_edge_by_hash(self.get_edge_hashes_between_nodes(a, b)[0])
            return (self.get_gene_mer_genes(a) if edge.get_sourceNodeDirection() == 1
                    else self.get_reverse_gene_mer_genes(a))

        def walk(prepend):
            genes = []
            for n in range(len(listOfNodes) - 1):
                src = self.get_node_by_hash(listOfNodes[n])
                tgt = self.get_node_by_hash(listOfNodes[n + 1])
                self.get_edge_by_hash(self.get_edge_hashes_between_nodes(src, tgt)[0])
                if n == 0:
                    genes += first_orientation()
                fw, bw = self.get_gene_mer_genes(tgt), self.get_reverse_gene_mer_genes(tgt)
                if not prepend:
                    tail = genes[-k1:] if k1 else genes[0:]
                    if fw[:-1] == tail:
                        genes.append(fw[-1])
                    elif bw[:-1] == tail:
                        genes.append(bw[-1])
                    else:
                        return None
                else:
                    head = genes[:k1]
                    if fw[1:] == head:
                        genes.insert(0, fw[0])
                    elif bw[1:] == head:
                        genes.insert(0, bw[0])
                    else:
                        raise ValueError("Gene sequences do not match in alternative path.")
            return genes

        genes = walk(False)
        return genes if genes is not None else walk(True)

    def _genes_in_unitig_from_arrays(self, listOfNodes):
        """get_genes_in_unitig on the device's arrays (node tokens, live edge lists) instead of Node / Edge objects;
        None wherever the objects' way would not simply return (nodes that are not adjacent, several edges between
        two nodes, a path that spells nothing): it then runs and says what the reference says"""
        v = self._v()
        a = v.arrays
        n_tok, e_alive, e_tgt, e_sdir = a["nodes"]["tokens"], a["edges"]["alive"], a["edges"]["tgt"], a["edges"]["sdir"]
        adj_off, adj_edge = a["adj_off"], a["adj_edge"]
        id_of = v.node_of_hash
        ids = [id_of.get(h) for h in listOfNodes]
        if None in ids:
            return None
        flip = self._vocab.two_v - 1

        # per node, the first time a path runs over it: {target: (live edges to it, source direction of the first)} over
        # its forward list, then its backward list — what get_edge_hashes_between_nodes collects from — and its tokens
        # both ways
        node_info = a.setdefault("_unitig_nodes", {})

        def info(i):
            got = node_info.get(i)
            if got is None:
                lo, hi = int(adj_off[2 * i]), int(adj_off[2 * i + 2])
                es = adj_edge[lo:hi]
                es = es[e_alive[es] != 0]
                to = {}
                for t, sd in zip(e_tgt[es].tolist(), e_sdir[es].tolist()):
                    seen = to.get(t)
                    to[t] = (1, sd) if seen is None else (seen[0] + 1, seen[1])
                fw = n_tok[i].tolist()
                got = node_info[i] = (to, fw, [flip - t for t in reversed(fw)])
            return got

        infos = [info(i) for i in ids]
        k1 = self._kmerSize - 1
        genes = None
        for n in range(len(ids) - 1):
            there, back = infos[n][0].get(ids[n + 1]), infos[n + 1][0].get(ids[n])
            if there is None or back is None or there[0] != 1 or back[0] != 1:
                return None
            if n == 0:
                genes = list(infos[0][1] if there[1] == 1 else infos[0][2])
            _, fw, bw = infos[n + 1]
            tail = genes[-k1:] if k1 else genes[0:]
            if fw[:-1] == tail:
                genes.append(fw[-1])
            elif bw[:-1] == tail:
                genes.append(bw[-1])
            else:
                return None   # (the objects' way starts over, extending to the left)
        return self._vocab.decode(genes)

    # ------------------------------------------------------------------ GML output (:542-586, :873-909)
    def write_node_entry(self, node_id, node_string, node_coverage, reads, component_ID, nodeColor):
        rows = ["\tnode\t[", f"\t\tid\t{node_id}", f'\t\tlabel\t"{node_string}"',
                f"\t\tcoverage\t{node_coverage}"]
        if component_ID:
            rows.append(f"\t\tcomponent\t{component_ID}")
        rows.append('\t\treads\t"' + ",".join(reads) + '"')
        if nodeColor:
            rows.append(f'\t\tcolor\t"{nodeColor}"')
        rows.append("\t]")
        return "\n".join(rows)

    def write_edge_entry(self, source_node, target_node, source_edge_direction, target_edge_direction,
                         edge_coverage):
        return "\n".join(["\tedge\t[", f"\t\tsource\t{source_node}", f"\t\ttarget\t{target_node}",
                          f"\t\tsource_direction\t{source_edge_direction}",
                          f"\t\ttarget_direction\t{target_edge_direction}",
                          f"\t\tweight\t{edge_coverage}", "\t]"])

    def assign_Id_to_nodes(self):
        for i, node in enumerate(self.all_nodes()):
            assert node.assign_node_Id(i) == i, "This node was assigned an incorrect ID"

    def write_gml_to_file(self, output_file, gml_content):
        import os
        folder = os.path.dirname(output_file)
        if folder != "" and not os.path.exists(folder):
            os.mkdir(folder)
        with open(output_file + ".gml", "w") as fh:
            fh.write("\n".join(gml_content))

    def generate_gml(self, output_file, geneMerSize, min_node_coverage, min_edge_coverage):
        """nodes in dict order, each followed by its forward then backward edges (:873-909)"""
        graph_data = ["graph\t[", "multigraph 1"]
        self.assign_Id_to_nodes()
        for node in self.all_nodes():
            graph_data.append(self.write_node_entry(node.get_node_Id(), self.get_gene_mer_label(node),
                                                    node.get_node_coverage(), [r for r in node.get_reads()],
                                                    node.get_component(), node.get_color()))
            for edge in self.get_forward_edges(node) + self.get_backward_edges(node):
                if edge.get_edge_coverage() == 0:
                    continue
                graph_data.append(self.write_edge_entry(node.get_node_Id(), edge.get_targetNode().get_node_Id(),
                                                        edge.get_sourceNodeDirection(),
                                                        edge.get_targetNodeDirection(),
                                                        edge.get_edge_coverage()))
        graph_data.append("]")
        self.write_gml_to_file(".".join([output_file, str(geneMerSize), str(min_node_coverage),
                                         str(min_edge_coverage)]), graph_data)
        return graph_data

    # ------------------------------------------------------------------ linear paths (:722-861)
    def _linear_step(self, node, forward):
        hashes = node.get_forward_edge_hashes() if forward else node.get_backward_edge_hashes()
        if (forward and len(hashes) != 1) or (not forward and len(hashes) == 0):
            return False, None, None
        edge = self.get_edge_by_hash(hashes[0])
        target = edge.get_targetNode()
        extend = self.get_degree(target) in (1, 2) and target != node
        return extend, target, edge.get_targetNodeDirection()

    def get_forward_node_from_node(self, sourceNode):
        return self._linear_step(sourceNode, True)

    def get_backward_node_from_node(self, sourceNode):
        return self._linear_step(sourceNode, False)

    def get_forward_path_from_node(self, node, startDirection, wantBranchedNode=False):
        path = [node.__hash__()]
        extend, nxt, d = self._linear_step(node, startDirection == 1)
        while extend and path[0] != nxt.__hash__():
            path.append(nxt.__hash__())
            extend, nxt, d = self._linear_step(nxt, d == 1)
        if wantBranchedNode and nxt:
            path.append(nxt.__hash__())
        return path

    def get_backward_path_from_node(self, node, startDirection, wantBranchedNode=False):
        path = [node.__hash__()]
        extend, nxt, d = self._linear_step(node, startDirection != -1)
        while extend and path[-1] != nxt.__hash__():
            path.insert(0, nxt.__hash__())
            extend, nxt, d = self._linear_step(nxt, d != -1)
        if wantBranchedNode and nxt:
            path.insert(0, nxt.__hash__())
        return path

    def get_linear_path_for_node(self, node, wantBranchedNode=False):
        d0 = node.get_geneMer().get_geneMerDirection()
        backward = self.get_backward_path_from_node(node, -1 * d0, wantBranchedNode)
        assert backward[-1] == node.__hash__()
        forward = self.get_forward_path_from_node(node, d0, wantBranchedNode)
        assert forward[0] == node.__hash__()
        return backward[:-1] + [node.__hash__()] + forward[1:]

    def get_all_node_coverages(self):
        if self._host_edits:
            return [n.get_node_coverage() for n in self.all_nodes()]
        nodes = self._v().arrays["nodes"] if self._view is not None else self._engine.nodes()
        return nodes["coverage"][nodes["alive"] != 0].tolist()   # straight from the device arrays

    def get_mean_node_coverage(self):
        if self._host_edits:
            return statistics.mean(self.get_all_node_coverages())
        # statistics.mean of integers (:868-871) is the exact fraction, an int when it is one and the correctly rounded
        # float otherwise — which is what Python's own division of the two integers gives; the sum comes off the array
        nodes = self._v().arrays["nodes"] if self._view is not None else self._engine.nodes()
        cov = nodes["coverage"][nodes["alive"] != 0]
        if len(cov) == 0:
            return statistics.mean([])   # (StatisticsError, as the reference raises)
        total, n = int(cov.sum(dtype=np.int64)), int(len(cov))
        return total // n if total % n == 0 else total / n

    # ------------------------------------------------------------------ components (:911-958)
    def get_nodes_in_component(self, component):
        return [n for n in self.all_nodes() if n.get_component() == int(component)]

    def components(self):
        return sorted({n.get_component() for n in self.all_nodes()})

    def get_number_of_component(self):
        return len(self.components())

    def get_AMR_nodes(self, listOfGenes):
        out = {}
        for gene in listOfGenes:
            for node in self.get_nodes_containing(gene):
                out[node.__hash__()] = node
        return out

    def collect_reads_in_path(self, path):
        v = self._v()
        reads = set()
        for h in list(path):
            node = v.node_by_hash(h)
            if node is not None:
                reads.update(node.get_reads())   # (a list: added one by one, as the reference's loop does)
        return reads

    # ------------------------------------------------------------------ read-path clustering
    def find_sublist_indices(self, main_list, sublist):
        return _find_sublist_indices(main_list, sublist)

    def is_sublist(self, long_list, sub_list):
        return _is_sublist(long_list, sub_list)

    def _amr_occurrence_stats(self, AMRNodes):
        """What the read loop of get_AMR_anchors (:2644-2676) finds out about every AMR node, from the device's
        per-window node ids in one numpy pass: {hash: (stopped at an anchor occurrence, all(singletons), flags)}.
        A node's reads are listed in read order and its positions on a read ascend, so the loop meets the node's
        occurrences in ascending token order; every occurrence is, independently of the node, one of: the only window
        of its read (stop, flag True), a terminal window (flag True), an interior window with a non-AMR neighbour
        (anchor, stop), an interior window between AMR nodes (flag False)."""
        v = self._v()
        tok_node, offs, k = v.arrays["tok_node"], self._read_off, self._kmerSize
        nr_off, nr_idx = v.arrays["node_reads_off"], v.arrays["node_reads"]
        ids = [v.node_of_hash[h] for h in AMRNodes]
        D = len(v.alive)
        is_amr = np.zeros(D + 1, bool)            # [D]: windows without a node (None)
        is_amr[ids] = True
        rows = np.unique(np.concatenate([nr_idx[nr_off[i]:nr_off[i + 1]] for i in ids])) if ids else np.zeros(0, np.int64)
        rows = rows.astype(np.int64)
        a = offs[rows]
        n = offs[rows + 1] - a - k + 1
        starts = np.zeros(len(rows) + 1, np.int64)
        np.cumsum(n, out=starts[1:])
        total = int(starts[-1])
        within = np.arange(total, dtype=np.int64) - np.repeat(starts[:-1], n)
        flat = tok_node[np.repeat(a, n) + within].astype(np.int64)
        amr = is_amr[np.where(flat >= 0, flat, D)]
        n_of = np.repeat(n, n)
        first, last = within == 0, within == n_of - 1
        left = np.concatenate([[False], amr[:-1]])
        right = np.concatenate([amr[1:], [False]])
        # 0 flag False, 1 flag True (terminal), 2 anchor (stop), 3 single-window read (flag True, stop)
        kind = np.where(n_of == 1, 3, np.where(first | last, 1, np.where(~left | ~right, 2, 0)))
        out = {}
        for h, i in zip(AMRNodes, ids):
            occ = kind[np.flatnonzero(flat == i)]
            stops = np.flatnonzero(occ >= 2)
            cut = int(stops[0]) if len(stops) else len(occ)
            flags = (occ[:cut] == 1).tolist()
            is_anchor = cut < len(occ) and occ[cut] == 2
            if cut < len(occ) and occ[cut] == 3:
                flags.append(True)
            singleton_all = len(occ) == 0 or occ[0] == 3
            out[h] = (bool(is_anchor), bool(singleton_all), flags)
        return out

    def get_AMR_anchors(self, AMRNodes, _stats=None):
        """anchor selection (:2629-2691), including the reference's use of FORWARD neighbours
        for both sides of the first test.  _stats: {hash: (stopped at an anchor occurrence, all(singletons), number of
        terminal flags, number of True ones)} when the caller has walked the reads already (native clustering)."""
        if not self._host_edits:
            AMRNodes = list(AMRNodes)
            amr_set = set(AMRNodes)
            if _stats is None:
                _stats = {h: (a, s_, len(f), f.count(True))
                          for h, (a, s_, f) in self._amr_occurrence_stats(AMRNodes).items()}
            anchors = set()
            for h in AMRNodes:                      # the same add() sequence as the loop below
                node = self.get_node_by_hash(h)
                forward = self.get_forward_neighbors(node)
                if len([n for n in forward if n.__hash__() != h]) == 0:
                    anchors.add(h)
                is_anchor, singleton_all, n_flags, n_true = _stats[h]
                if is_anchor:
                    anchors.add(h)
                if singleton_all or n_true == n_flags:
                    fw_amr = [n for n in forward if n.__hash__() in amr_set]
                    bw_amr = [n for n in self.get_backward_neighbors(node) if n.__hash__() in amr_set]
                    if len(bw_amr) == 0 or len(fw_amr) == 0:
                        anchors.add(h)
            for h in AMRNodes:
                n_flags, n_true = _stats[h][2], _stats[h][3]
                if n_flags and n_true / n_flags > 0.3:
                    anchors.add(h)
            return anchors
        readNodes = self.get_readNodes()
        anchors, terminals = set(), {}
        per_read = {}  # read -> (AMR flag per node of the read, positions of every AMR node on it): a read is
                       # looked at once, not once per AMR node it carries

        def on_read_info(r, on_read):
            got = per_read.get(r)
            if got is None:
                amr, where = [0] * len(on_read), {}
                for i, n in enumerate(on_read):
                    if n in AMRNodes:
                        amr[i] = 1
                        where.setdefault(n, []).append(i)
                got = per_read[r] = (amr, where)
            return got

        for h in AMRNodes:
            flags = terminals[h] = []
            node = self.get_node_by_hash(h)
            fw_other = [n for n in self.get_forward_neighbors(node) if n.__hash__() != h]
            if len(fw_other) == 0:
                anchors.add(h)
            singletons, is_anchor = [], False
            for r in node.get_reads():
                on_read = readNodes[r]
                if len(on_read) == 1 and on_read[0] == h:
                    singletons.append(True)
                    flags.append(True)
                    break
                singletons.append(False)
                amr, where = on_read_info(r, on_read)
                for idx in where.get(h, ()):
                    if idx == 0 or idx == len(on_read) - 1:
                        flags.append(True)
                        continue
                    if amr[idx - 1] == 0 or amr[idx + 1] == 0:
                        is_anchor = True
                        break
                    flags.append(False)
                if is_anchor:
                    anchors.add(h)
                    break
            if all(singletons) or all(flags):
                fw_amr = [n for n in self.get_forward_neighbors(node) if n.__hash__() in AMRNodes]
                bw_amr = [n for n in self.get_backward_neighbors(node) if n.__hash__() in AMRNodes]
                if len(bw_amr) == 0 or len(fw_amr) == 0:
                    anchors.add(h)
        for h, flags in terminals.items():
            if flags and flags.count(True) / len(flags) > 0.3:
                anchors.add(h)
        return anchors

    def get_singleton_paths(self, all_seen_nodes, nodeAnchors, final_paths, final_path_coverages):
        for a in nodeAnchors:
            if a not in all_seen_nodes:
                key = tuple(self.get_genes_in_unitig([a]))
                node = self.get_node_by_hash(a)
                final_paths[key] = len(set(node.get_list_of_reads()))
                final_path_coverages[key] = [node.get_node_coverage()]

    def get_reads_supporting_path(self, path, suffix_tree):
        return {rid.replace("_reverse", "") for rid, _ in suffix_tree.find_all(list(path))}

    def _match_gene_lists(self, gene_lists):
        """Exact occurrences of every gene-string list in every read, on the device (kernel
        k_match, the batched form of is_sublist / find_sublist_indices / Tree.find_all).
        Returns one (read indices, start positions) pair of arrays per list, ordered by read, position."""
        table = self._vocab._tok
        pats, usable = [], []
        for genes in gene_lists:
            toks = [table.get(g) for g in genes]
            ok = len(toks) > 0 and None not in toks  # an unknown gene cannot occur in any read
            pats.append(toks if ok else None)
            if ok:
                usable.append(toks)
        found = []
        for lo in range(0, len(usable), 60000):
            off, hit_read, hit_pos = self._engine.match_patterns(0, usable[lo:lo + 60000])
            off = off.tolist()
            found.extend((hit_read[off[j]:off[j + 1]], hit_pos[off[j]:off[j + 1]]) for j in range(len(off) - 1))
        nothing = (np.zeros(0, np.int32), np.zeros(0, np.int32))
        it = iter(found)
        return [next(it) if p is not None else nothing for p in pats]

    def get_all_sublists(self, lst, gene_call_subset, threshold, geneOfInterest, cores):
        """windows of a block's gene path that hold every copy of the gene and are carried by
        >= threshold reads (:2711-2723, path_finding_utils.py:296-310).  The reference builds a
        suffix tree per window length in a process pool; here all windows (and their reverse
        complements) are matched against all reads in one device call.  A read supports a
        window if the window occurs in it in either orientation; reads without nodes (< k
        genes) are not in gene_call_subset and do not count."""
        combs, uniq = self._sublist_windows(lst, geneOfInterest)
        hits = self._match_gene_lists([list(c) for c in uniq] +
                                      [self.reverse_list_of_genes(list(c)) for c in uniq])
        return self._sublists_from_hits(combs, uniq, hits[:len(uniq)], hits[len(uniq):], gene_call_subset, threshold)

    def _sublist_windows(self, lst, geneOfInterest):
        """the windows of a block's gene path that hold every copy of the gene, in the reference's order (:296-310),
        and the distinct ones among them"""
        plus, minus = f"+{geneOfInterest}", f"-{geneOfInterest}"
        wanted = lst.count(plus) + lst.count(minus)
        combs = []
        for i in range(1, len(lst) + 1):
            for start in range(len(lst) - i + 1):
                comb = tuple(lst[start:start + i])
                if comb.count(plus) + comb.count(minus) == wanted:
                    combs.append(comb)
        return combs, list(dict.fromkeys(combs))

    def _sublists_from_hits(self, combs, uniq, fwd_hits, rev_hits, gene_call_subset, threshold):
        memo = getattr(self, "_subset_rows_memo", None)   # (the blocks of one gene come with the same subset)
        if memo is not None and memo[0] is gene_call_subset and memo[1] == len(gene_call_subset):
            in_subset = memo[2]
        else:
            in_subset = np.zeros(len(self._read_ids) + 1, bool)   # reads that count (by row)
            index = self._read_index
            rows = [index.get(rid) for rid in gene_call_subset]
            in_subset[[r for r in rows if r is not None]] = True
            self._subset_rows_memo = (gene_call_subset, len(gene_call_subset), in_subset)
        support = {}
        for j, comb in enumerate(uniq):
            reads = np.union1d(fwd_hits[j][0], rev_hits[j][0])
            support[comb] = int(in_subset[reads].sum())
        sublists = {}
        for comb in combs:
            if comb and support[comb] >= threshold:
                sublists[comb] = support[comb]
        return sublists

    def get_full_paths(self, node_tree, reads, nodeAnchors, threshold, gene_call_subset,
                       geneOfInterest, cores):
        full_blocks = {}
        for a1 in nodeAnchors:
            if hasattr(node_tree, "reversed_suffix_tree"):   # both steps at once, nothing re-interned
                sub_tree = node_tree.reversed_suffix_tree(a1)
            else:
                suffixes = get_suffixes_from_initial_tree(node_tree, a1)
                sub_tree = Tree({r: list(reversed(s)) for r, s in suffixes.items()})
            process_anchors(sub_tree, nodeAnchors, a1, full_blocks, reads, node_tree, threshold)
        return self._paths_from_full_blocks(full_blocks, gene_call_subset, threshold, geneOfInterest, cores)

    def _paths_from_full_blocks(self, full_blocks, gene_call_subset, threshold, geneOfInterest, cores, _batch=None):
        """the second half of get_full_paths (:2738-2782): gene paths of the full blocks, block filter, the
        differentiating path of every kept block"""
        gene_blocks = {}
        spelled = {}   # block -> its gene list (the graph does not change in here; the lists are only read)

        def genes_of(f):
            got = spelled.get(f)
            if got is None:
                got = spelled[f] = self.get_genes_in_unitig(list(f))
            return got

        # the windows of ALL blocks go to the device in one batch (the reference asks block by block, :2741-2746);
        # _batch: the caller searches for the windows of SEVERAL genes' blocks at once — the first call (no hits yet)
        # hands back what to look for, the second one comes with the answers
        if _batch is not None and "hits" not in _batch:
            plans = [(f,) + self._sublist_windows(genes_of(f), geneOfInterest) for f in full_blocks]
            fwd_lists = [list(c) for _, _, uniq in plans for c in uniq]
            _batch.update(plans=plans, spelled=spelled, fwd_lists=fwd_lists)
            return None
        if _batch is not None:
            plans, fwd_lists, hits = _batch["plans"], _batch["fwd_lists"], _batch["hits"]
            spelled.update(_batch["spelled"])
        else:
            plans = [(f,) + self._sublist_windows(genes_of(f), geneOfInterest) for f in full_blocks]
            fwd_lists = [list(c) for _, _, uniq in plans for c in uniq]
            hits = (self._match_gene_lists(fwd_lists + [self.reverse_list_of_genes(x) for x in fwd_lists])
                    if fwd_lists else [])
        at, n_all = 0, len(fwd_lists)
        for f, combs, uniq in plans:
            options = self._sublists_from_hits(combs, uniq, hits[at:at + len(uniq)],
                                               hits[n_all + at:n_all + at + len(uniq)], gene_call_subset, threshold)
            at += len(uniq)
            if len(options) > 0:
                gene_blocks[f] = options
        filtered_blocks = filter_blocks({f: full_blocks[f] for f in gene_blocks})
        final_paths, final_path_coverages, seen_nodes = {}, {}, set()
        plus, minus = f"+{geneOfInterest}", f"-{geneOfInterest}"
        for f1 in filtered_blocks:
            seen_nodes.update(f1)
            if f1 not in gene_blocks:
                continue
            differentiating = set()
            for o1 in gene_blocks[f1]:
                fwd, rev = list(o1), self.reverse_list_of_genes(list(o1))
                shared = False
                for f2 in filtered_blocks:
                    if f1 == f2:
                        continue
                    other = genes_of(f2)
                    if _is_sublist(other, fwd) or _is_sublist(other, rev):
                        shared = True
                        break
                if not shared:
                    differentiating.add(o1)
            if differentiating:
                chosen = sorted(list(differentiating),
                                key=lambda x: (x.count(plus) + x.count(minus), gene_blocks[f1][x], len(x)),
                                reverse=True)[0]
                final_paths[chosen] = gene_blocks[f1][chosen]
                final_path_coverages[chosen] = [self.get_node_by_hash(n).get_node_coverage() for n in list(f1)]
        return final_paths, seen_nodes, final_path_coverages

    def get_paths_for_gene(self, node_suffix_tree, gene_call_subset, nodeHashesOfInterest, threshold,
                           geneOfInterest, cores):
        anchors = self.get_AMR_anchors(nodeHashesOfInterest)
        final_paths, seen, coverages = self.get_full_paths(
            node_suffix_tree, self.get_readNodes(), anchors, threshold, gene_call_subset,
            geneOfInterest, cores)
        self.get_singleton_paths(seen, anchors, final_paths, coverages)
        return final_paths, coverages

    def split_into_subpaths(self, geneOfInterest, pathsOfinterest, path_coverages, path_reads,
                            mean_node_coverage=None, _hits=None):
        """allele clusters: reads holding a path exactly once, forward orientation first, the
        reverse complement only when there is no forward occurrence (:2360-2455).  The
        reference scans every read for every path on the host; here all paths and their
        reverse complements are matched against all reads in one device call."""
        allele_count = 1
        gene_clusters, read_tracking = {}, {}
        if mean_node_coverage is None:
            mean_node_coverage = self.get_mean_node_coverage()
        paths = list(pathsOfinterest)
        fwd_lists = [list(p) for p in paths]
        rev_lists = [self.reverse_list_of_genes(f) for f in fwd_lists]
        # (_hits: the occurrences of these lists, found by the caller together with other genes' paths)
        hits = _hits if _hits is not None else self._match_gene_lists(fwd_lists + rev_lists)
        for pi, path in enumerate(paths):
            fwd = fwd_lists[pi]
            named = list(path)
            fw_idx, rv_idx = {}, {}
            for g, gene in enumerate(fwd):
                if gene[1:] == geneOfInterest:
                    allele = f"{geneOfInterest}_{allele_count}"
                    fw_idx[g] = rv_idx[len(fwd) - g - 1] = allele
                    gene_clusters[allele], read_tracking[allele] = [], set()
                    named[g] = f"{gene[0]}{allele}"
                    allele_count += 1
            named = tuple(named)
            # reads that hold the path exactly once: forward occurrences decide; the reverse complement
            # only counts for reads without any forward occurrence (:2401-2439)
            (fw_reads, fw_pos), (rv_reads, rv_pos) = hits[pi], hits[pi + len(paths)]
            fw_u, fw_first, fw_cnt = np.unique(fw_reads, return_index=True, return_counts=True)
            rv_u, rv_first, rv_cnt = np.unique(rv_reads, return_index=True, return_counts=True)
            rv_only = ~np.isin(rv_u, fw_u)
            chosen = [(fw_u[fw_cnt == 1], fw_pos[fw_first[fw_cnt == 1]], fw_idx),
                      (rv_u[rv_only & (rv_cnt == 1)], rv_pos[rv_first[rv_only & (rv_cnt == 1)]], rv_idx)]
            rows = np.concatenate([c[0] for c in chosen])
            starts = np.concatenate([c[1] for c in chosen])
            which = np.concatenate([np.zeros(len(chosen[0][0]), np.int8), np.ones(len(chosen[1][0]), np.int8)])
            order = np.argsort(rows, kind="stable")   # read order == dict order of _reads
            ro, so, wo = rows[order], starts[order], which[order]
            rids = self._names_of_rows(ro)
            if rids:
                path_reads.setdefault(named, set()).update(rids)
            pos = self._genePositions
            if (fw_idx and rids and isinstance(self._reads, TokenizedReads) and isinstance(pos, TokenizedPositions)
                    and not pos._cache and pos._moved is None and self._gs is not None):
                # tokenised containers: the genes and positions of all chosen reads come out of the flat arrays
                # at once (one entry per read and allele, appended in read order as the loop below does)
                V = max(self._vocab.V, 1)
                want = self._vocab.rank[geneOfInterest]
                base = self._read_off[ro] + so
                for g, allele in fw_idx.items():
                    t = base + np.where(wo == 1, len(fwd) - g - 1, g)
                    tok = self._tokens[t].astype(np.int64)
                    assert bool(np.all(np.where(tok >= V, tok - V, V - 1 - tok) == want))
                    entries = [f"{rid}_{s_}_{e_}" for rid, s_, e_ in zip(rids, self._gs[t].tolist(), self._ge[t].tolist())]
                    gene_clusters[allele].extend(entries)
                    read_tracking[allele].update(entries)
                continue
            # (single genes / positions of a read: tokenised containers answer without decoding the read)
            gene_at = getattr(self._reads, "gene_at", None) or (lambda rid, i: self._reads[rid][i])
            pos_at = getattr(pos, "pos_at", None) or (lambda rid, i: pos[rid][i])
            for read_id, start, w in zip(rids, so.tolist(), wo.tolist()):
                idx = rv_idx if w else fw_idx
                for gene_index in idx:
                    assert gene_at(read_id, start + gene_index)[1:] == geneOfInterest
                    s_, e_ = pos_at(read_id, start + gene_index)
                    entry = f"{read_id}_{s_}_{e_}"
                    gene_clusters[idx[gene_index]].append(entry)
                    read_tracking[idx[gene_index]].add(entry)
        ranked = sorted([a for a in read_tracking], key=lambda x: len(read_tracking[x]), reverse=True)
        doomed = set()
        for i, a1 in enumerate(ranked):
            if a1 in doomed:
                continue
            for a2 in ranked[i + 1:]:
                if a1 != a2 and len(read_tracking[a1] & read_tracking[a2]) > 0:
                    doomed.add(a2)
        for d in doomed:
            del gene_clusters[d]
        return gene_clusters, path_reads

    def assign_final_alleles_to_components(self, finalAllelesOfInterest, clustered_reads, allele_counts,
                                           geneOfInterest):
        readNodes = self.get_readNodes()
        for allele in finalAllelesOfInterest:
            for entry in finalAllelesOfInterest[allele]:
                for node_hash in readNodes["_".join(entry.split("_")[:-2])]:
                    component = self.get_node_by_hash(node_hash).get_component()
                    break
                break
            gene_name = "_".join(allele.split("_")[:-1])
            if gene_name not in allele_counts:
                allele_counts[gene_name] = 1
            clustered_reads.setdefault(component, {}).setdefault(geneOfInterest, {})[
                f"{gene_name}_{allele_counts[gene_name]}"] = finalAllelesOfInterest[allele]
            allele_counts[gene_name] += 1

    def collect_component_missed_genes(self, component_nodeHashesOfInterest, clustered_reads,
                                       allele_counts, geneOfInterest, path_reads):
        for component, hashes in component_nodeHashesOfInterest.items():
            clustered_reads.setdefault(component, {}).setdefault(geneOfInterest, {})
            if len(clustered_reads[component][geneOfInterest]) != 0:
                continue
            if geneOfInterest not in allele_counts:
                allele_counts[geneOfInterest] = 1
            allele_name = f"{geneOfInterest}_{allele_counts[geneOfInterest]}"
            key = tuple([f"+{allele_name}"])
            bucket = clustered_reads[component][geneOfInterest][allele_name] = []
            for read_id in self.collect_reads_in_path(hashes):
                genes = self._reads[read_id]
                for i in [i for i, g in enumerate(genes) if g[1:] == geneOfInterest]:
                    s, e = self._genePositions[read_id][i]
                    bucket.append(f"{read_id}_{s}_{e}")
                path_reads.setdefault(key, set()).add(read_id)
            allele_counts[geneOfInterest] += 1

    def assign_reads_to_genes(self, listOfGenes, cores, allele_counts={}, mean_node_coverage=None,
                              path_threshold=5):
        """per gene of interest: anchors -> full blocks -> differentiating gene paths -> allele
        clusters (:2880-2939)."""
        import gc
        # No reference cycles are made below, but millions of live containers (a million-read graph)
        # make every pass of the cyclic collector expensive and the clustering allocates enough to
        # trigger thousands of them (measured: 8.6 of 21.7 s at BASELINE config 4): pause it.
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._assign_reads_to_genes(listOfGenes, cores, allele_counts, mean_node_coverage)
        finally:
            if gc_was_on:
                gc.enable()

    def _node_tree_from_device_ids(self, reads_with_gene):
        """construct_suffix_tree({r: readNodes[r]}) (path_finding_utils.py:79-85) without interning the
        256-bit node hashes item by item and without a Python loop over the reads: the search codes are the
        DEVICE node ids of the per-window array (None = -2) gathered for all reads at once, the sequences
        (lists of hashes, what suffixes are cut from) are the view's cached read lists.  Entry order
        as the reference builds it: the reads in the given order, then '<read>_reverse' for every read with
        more than one distinct node."""
        if not hasattr(Tree, "from_flat"):
            return None   # the external suffix_tree package is in use
        v = self._v()
        tok_node, offs, k = v.arrays["tok_node"], self._read_off, self._kmerSize
        index = self._read_index
        reads_with_gene = list(reads_with_gene)
        rows = np.fromiter((index[rid] for rid in reads_with_gene), dtype=np.int64, count=len(reads_with_gene))
        a = offs[rows]
        n = offs[rows + 1] - a - k + 1
        starts = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(n, out=starts[1:])
        total = int(starts[-1])
        within = np.arange(total, dtype=np.int64) - np.repeat(starts[:-1], n)
        fwd = tok_node[np.repeat(a, n) + within]
        if len(rows):
            lo = np.minimum.reduceat(fwd, starts[:-1])
            hi = np.maximum.reduceat(fwd, starts[:-1])
            late = np.flatnonzero(lo != hi)       # more than one distinct node (ids <-> hashes, -2 <-> None)
        else:
            late = np.zeros(0, np.int64)
        n_late = n[late]
        starts_late = np.zeros(len(late) + 1, dtype=np.int64)
        np.cumsum(n_late, out=starts_late[1:])
        within_l = np.arange(int(starts_late[-1]), dtype=np.int64) - np.repeat(starts_late[:-1], n_late)
        rev = fwd[np.repeat(starts[late] + n_late - 1, n_late) - within_l]
        flat = np.concatenate([fwd, rev])
        all_starts = np.concatenate([starts, starts_late[1:] + total])
        keys = reads_with_gene + [reads_with_gene[i] + "_reverse" for i in late.tolist()]
        # the sequences themselves (what suffixes are cut from): the reads' node-hash lists, cut out of ONE gather of
        # the hashes by device id (ids -2 / -1 land on the two None entries at the end of the table)
        hashes = v.node_hash_table(fwd)[fwd].tolist()
        cuts = starts.tolist()
        seqs = [hashes[cuts[i]:cuts[i + 1]] for i in range(len(rows))]
        cache = v.readNodes._cache   # the block search asks the view for the same reads' lists next
        for rid, lst in zip(reads_with_gene, seqs):
            cache.setdefault(rid, lst)
        seqs += [seqs[i][::-1] for i in late.tolist()]
        to_id = v.node_of_hash
        self._tree_rows = rows
        return Tree.from_flat(keys, seqs, flat, all_starts, lambda x: -2 if x is None else to_id.get(x))

    def _any_read_name_ends_with(self, suffix):
        if isinstance(self._reads, TokenizedReads):
            return self._reads.any_name_ends_with(suffix)
        memo = self.__dict__.setdefault("_suffix_memo", {})
        if suffix not in memo:
            memo[suffix] = any(r.endswith(suffix) for r in self._read_ids)
        return memo[suffix]

    def _reads_on_nodes(self, node_ids, _rows_of=None):
        """collect_reads_in_path (:1497-1504) for nodes given by device id, with the rows of the reads: the SET of read
        names is made by the same update() calls, node by node — the order in which it iterates later is the order of
        the reference's set — and the rows come out of the node -> reads lists themselves, not out of a name -> row
        table of the whole read set"""
        v = self._v()
        node_ids = [i for i in node_ids if v.alive[i]]
        # the reads of these few nodes by one batched device search over the per-window node ids (k_match with
        # one-node patterns: hits ordered by node, read, position) — not the node -> reads lists of every node
        # (_rows_of: {node id: rows} when the caller has searched for several genes' nodes at once)
        if _rows_of is None:
            off, hit_read, _ = self._engine.match_patterns(1, [[i] for i in node_ids])
            off = off.tolist()
            per_node = [hit_read[off[j]:off[j + 1]] for j in range(len(node_ids))]
        else:
            per_node = [_rows_of[i] for i in node_ids]
        # The reference's set is filled node by node with every read of the node (update() with a LIST: item by item, a
        # name already there changes nothing), so what the set becomes — and the order in which it iterates — follows
        # from the FIRST time each read shows up in that sequence alone: the rows' first occurrences, in order, are
        # found on the arrays and only those names are made and added (a read sits on ~k nodes of a gene: one name in
        # five of the sequence).
        seq = np.concatenate(per_node).astype(np.int64) if per_node else np.zeros(0, np.int64)
        reads = set()
        if len(seq) == 0:
            return reads, np.zeros(0, np.int64)
        at = np.arange(len(seq), dtype=np.int64)
        first_at = np.empty(int(seq.max()) + 1, np.int64)
        first_at[seq[::-1]] = at[::-1]                  # (written back to front: the first occurrence is what stays)
        order = seq[first_at[seq] == at]                # distinct rows in the order they first appear
        names = self._names_of_rows(order)
        reads.update(names)
        row_of = dict(zip(names, order.tolist()))
        self._known_rows.update(row_of)
        return reads, np.asarray(list(map(row_of.__getitem__, reads)), dtype=np.int64)

    def _cluster_genes_native(self, listOfGenes, mean_node_coverage, cores, allele_counts, clustered_reads, path_reads):
        """assign_reads_to_genes (:2896-2937) for ALL genes of interest with the block search in native code
        (amira_amd.clustering / amg_cluster_full_blocks): the reads, their node lists and the blocks stay integer
        arrays; node hashes are made for the nodes these reads run through, objects for the few nodes and edges the
        anchor selection and the unitig spelling look at.  The reference takes the genes one after another and scans
        the reads three times per gene (the reads on the gene's nodes, the windows of its blocks, its final paths);
        nothing a gene's scans look for depends on another gene's result, so each of the three searches is made ONCE,
        for all genes together (three k_match passes over the reads instead of three per gene), and only the last
        step — alleles into `clustered_reads` / `allele_counts` / `path_reads`, which the genes share — runs gene
        by gene in the caller's order."""
        v = self._v()
        threshold = mean_node_coverage / 20
        jobs = []
        for geneOfInterest in listOfGenes:
            node_ids = self._node_ids_containing([geneOfInterest])
            jobs.append({"gene": geneOfInterest, "node_ids": node_ids, "hashes": [v.hash_at(i) for i in node_ids]})
        # ---- search 1: the reads on every gene's nodes (one-node patterns over the per-window node ids)
        wanted = list(dict.fromkeys(i for j in jobs for i in j["node_ids"] if v.alive[i]))
        rows_of = {}
        if wanted:
            off, hit_read, _ = self._engine.match_patterns(1, [[i] for i in wanted])
            off = off.tolist()
            rows_of = {i: hit_read[off[n]:off[n + 1]] for n, i in enumerate(wanted)}
        offs, k = self._read_off, self._kmerSize
        for j in jobs:
            geneOfInterest, node_ids, hashes = j["gene"], j["node_ids"], j["hashes"]
            reads_with_gene, rows = self._reads_on_nodes(node_ids, _rows_of=rows_of)
            # the reads' node lists, laid end to end: gathered on the device (the per-window array of the whole read set
            # never crosses PCIe for this)
            a = offs[rows]
            seq, starts = self._engine.read_node_ids_of(a, offs[rows + 1] - a - k + 1)
            st = _clustering.anchor_stats(seq, starts, np.argsort(rows, kind="stable"), node_ids, len(v.alive))
            anchors = self.get_AMR_anchors(hashes, _stats={h: (bool(x[0]), bool(x[1]), int(x[2]), int(x[3]))
                                                           for h, x in zip(hashes, st.tolist())})
            uniq = np.unique(seq)
            uniq = uniq[uniq >= 0]
            nh = v.node_hash_table(uniq)
            memo = getattr(self, "_py_hash_memo", None)
            if memo is None or memo[0] is not v:
                memo = self._py_hash_memo = (v, np.zeros(len(v.alive), np.int64), np.zeros(len(v.alive), bool))
            py_hash, known = memo[1], memo[2]
            new = uniq[~known[uniq]]
            if len(new):
                py_hash[new] = [hash(h) for h in nh[new].tolist()]
                known[new] = True
            anchor_list = list(anchors)                   # the set's iteration order, as `for a1 in nodeAnchors` meets it
            by_hash = {h: i for i, h in enumerate(sorted(anchor_list))}
            anchor_ids = [v.node_of_hash[h] for h in anchor_list]
            j.update(anchors=anchors, rows=rows, nh=nh,
                     search=(seq, starts, anchor_ids, [by_hash[h] for h in anchor_list]))
        # the block searches of the genes (host C++, amg_cluster_full_blocks: no shared state, the interpreter lock is
        # released around the call) side by side; the node-hash table they read is complete by now
        py_hash = self._py_hash_memo[1] if jobs else None
        none_hash = hash(None)

        def search(j):
            return _clustering.full_block_ids(*j["search"], py_hash, none_hash)

        if len(jobs) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 16)) as pool:
                found = list(pool.map(search, jobs))
        else:
            found = [search(j) for j in jobs]
        for j, blocks in zip(jobs, found):
            geneOfInterest, rows, nh = j["gene"], j.pop("rows"), j.pop("nh")
            j.pop("search")
            full_blocks = {tuple(nh[b].tolist()): True for b in blocks}
            in_subset = np.zeros(len(self._read_ids) + 1, bool)
            in_subset[rows] = True
            subset = _SubsetRows(2 * len(rows))   # (reads and their "_reverse" twins: get_all_sublists only asks which rows count)
            j.update(full_blocks=full_blocks, subset=subset, in_subset=in_subset, batch={})
            self._paths_from_full_blocks(full_blocks, subset, threshold, geneOfInterest, cores, _batch=j["batch"])
        # ---- search 2: the windows of every gene's blocks, forward and reverse complemented
        fwd_all = [x for j in jobs for x in j["batch"]["fwd_lists"]]
        hits = self._match_gene_lists(fwd_all + [self.reverse_list_of_genes(x) for x in fwd_all]) if fwd_all else []
        at, n_all = 0, len(fwd_all)
        for j in jobs:
            m = len(j["batch"]["fwd_lists"])
            j["batch"]["hits"] = hits[at:at + m] + hits[n_all + at:n_all + at + m]
            at += m
            self._subset_rows_memo = (j["subset"], len(j["subset"]), j["in_subset"])
            paths, seen, coverages = self._paths_from_full_blocks(j["full_blocks"], j["subset"], threshold, j["gene"],
                                                                  cores, _batch=j["batch"])
            self.get_singleton_paths(seen, j["anchors"], paths, coverages)
            j.update(paths=paths, coverages=coverages, batch=None)
        # ---- search 3: every gene's final paths, forward and reverse complemented
        fwd_all = [list(p) for j in jobs for p in j["paths"]]
        rev_all = [self.reverse_list_of_genes(f) for f in fwd_all]
        hits = self._match_gene_lists(fwd_all + rev_all) if fwd_all else []
        at, n_all = 0, len(fwd_all)
        for j in jobs:   # the part the genes share, in the caller's order
            geneOfInterest, hashes = j["gene"], j["hashes"]
            m = len(j["paths"])
            own = hits[at:at + m] + hits[n_all + at:n_all + at + m]
            at += m
            alleles, _ = self.split_into_subpaths(geneOfInterest, j["paths"], j["coverages"], path_reads,
                                                  mean_node_coverage, _hits=own)
            self.assign_final_alleles_to_components(alleles, clustered_reads, allele_counts, geneOfInterest)
            by_component = {}
            for h in hashes:
                by_component.setdefault(self.get_node_by_hash(h).get_component(), set()).add(h)
            self.collect_component_missed_genes(by_component, clustered_reads, allele_counts, geneOfInterest, path_reads)

    def _assign_reads_to_genes(self, listOfGenes, cores, allele_counts, mean_node_coverage):
        clustered_reads, path_reads = {}, {}
        if mean_node_coverage is None:
            mean_node_coverage = self.get_mean_node_coverage()
        # (a read really called "<name>_reverse" would collide with the reference's name for a reversed read: the
        # Python block search, which carries the names around as the reference does, keeps such inputs)
        native = (not self._host_edits and hasattr(Tree, "from_flat") and not os.environ.get("AMG_CLUSTER_PYTHON")
                  and _clustering.emulation_ok() and not self._any_read_name_ends_with("_reverse"))
        if native:
            # a bounded number of genes at a time: what _cluster_genes_native holds per gene (its reads' node lists, the
            # block plans, a flag per read) is then bounded too; the genes' results are entered in the caller's order
            # either way (AMG_CLUSTER_GENES_PER_CALL: test switch)
            genes = list(listOfGenes)
            step = max(int(os.environ.get("AMG_CLUSTER_GENES_PER_CALL", "64")), 1)
            for at in range(0, len(genes), step):
                self._cluster_genes_native(genes[at:at + step], mean_node_coverage, cores, allele_counts,
                                           clustered_reads, path_reads)
            listOfGenes = ()
        for geneOfInterest in listOfGenes:
            hashes = [n.__hash__() for n in self.get_nodes_containing(geneOfInterest)]
            reads_with_gene = self.collect_reads_in_path(hashes)
            node_tree = None if self._host_edits else self._node_tree_from_device_ids(reads_with_gene)
            if node_tree is None:
                node_tree = construct_suffix_tree({r: self.get_readNodes()[r] for r in reads_with_gene})
            # (the reference hands the gene lists of these reads and of their reverse complements to a
            # suffix tree per window length; get_all_sublists here only asks WHICH reads count, so the
            # keys are enough and no read is decoded into strings)
            gene_call_subset = dict.fromkeys(reads_with_gene)
            gene_call_subset.update(dict.fromkeys([r + "_reverse" for r in reads_with_gene]))
            known = getattr(self, "_tree_rows", None)
            if (known is not None and len(known) == len(reads_with_gene) and isinstance(self._reads, TokenizedReads)
                    and not self._reads.any_name_ends_with("_reverse")):
                # which rows count for get_all_sublists: the rows of these reads, already known from the tree (the
                # "<read>_reverse" keys name no read of their own)
                in_subset = np.zeros(len(self._read_ids) + 1, bool)
                in_subset[known] = True
                self._subset_rows_memo = (gene_call_subset, len(gene_call_subset), in_subset)
            self._tree_rows = None
            paths, coverages = self.get_paths_for_gene(node_tree, gene_call_subset, hashes,
                                                       mean_node_coverage / 20, geneOfInterest, cores)
            alleles, path_reads = self.split_into_subpaths(geneOfInterest, paths, coverages, path_reads,
                                                           mean_node_coverage)
            self.assign_final_alleles_to_components(alleles, clustered_reads, allele_counts, geneOfInterest)
            by_component = {}
            for h in hashes:
                by_component.setdefault(self.get_node_by_hash(h).get_component(), set()).add(h)
            self.collect_component_missed_genes(by_component, clustered_reads, allele_counts,
                                                geneOfInterest, path_reads)
        self._subset_rows_memo = None
        return clustered_reads, path_reads
