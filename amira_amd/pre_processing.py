"""`process_pandora_json` under the reference's own name and signature (pre_processing.py:44-63): the gene-call JSON and
the gene-position JSON in, (annotatedReads, genesOfInterest, gene_position_dict) out — the two mappings array-backed
(amira_amd.io.TokenizedReads / TokenizedPositions: read-only mappings of the reference's shapes that GeneMerGraph and
the drivers take straight to the device), the list reduced to the genes of interest the reads contain.

The pandora SAM converter in front of it (pre_processing.py:190-284, pysam) is pre-processing and stays out of scope
(DESIGN.md section 7): the hot path starts at the gene calls.
"""
import ctypes as C
import json

import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .io import TokenizedPositions, load_gene_calls


def _present_in_first_use_order(reads, wanted):
    """the names of `wanted` that occur in the reads, in the order the reads first show them (read order, then gene
    order within the read): one native pass over the token array (amg_calls_first_use)"""
    rank = reads.vocab.rank
    names = [g for g in dict.fromkeys(wanted) if g in rank]
    if not names:
        return []
    ranks = np.fromiter((rank[g] for g in names), dtype=np.int32, count=len(names))
    first = np.empty(len(names), np.int64)
    tokens = np.ascontiguousarray(reads.tokens, np.int32)
    check(_ffi.lib.amg_calls_first_use(ptr(tokens), len(tokens), reads.vocab.two_v, ptr(ranks), len(names), ptr(first)))
    order = np.argsort(first, kind="stable")
    return [names[i] for i in order.tolist() if first[i] >= 0]


def process_pandora_json(pandoraJSON, genesOfInterest, gene_positions):
    """pre_processing.py:44-63.  Returns (annotatedReads, genesOfInterest, gene_position_dict):
      * annotatedReads — EVERY read of the file (the reference collects the reads without a gene of interest in
        `to_delete` and never deletes them), as a TokenizedReads;
      * genesOfInterest — `list(set)` of the wanted genes that occur in some read.  The reference's set is filled in
        first-appearance order; the same insertions are made here, so the list comes out in the order the reference's
        interpreter would give it (a set of str: it follows PYTHONHASHSEED, as there);
      * gene_position_dict — the positions of every read's genes, as a TokenizedPositions aligned with the reads.
    Gene names are compared raw (`annotatedReads[read][g][1:] in genesOfInterest`, :54).  The native loader stores a name
    with its blanks replaced (construct_gene.py:54-56); a file in which some gene name holds a blank — no pandora
    output does — is therefore answered from the file's own text by the reference's loop, as plain dicts."""
    reads, gs, ge, blanks = load_gene_calls(pandoraJSON, gene_positions, want_blanks=True)
    if blanks:
        # the mapping the reference returns holds the RAW strings ("+two words"), and they are what it writes back;
        # the token arrays cannot say them: such a file takes the reference's own route, plain dicts and all
        del reads, gs, ge
        with open(pandoraJSON) as i:
            raw = json.loads(i.read())
        with open(gene_positions) as i:
            raw_positions = json.loads(i.read())
        subsetted = set()
        for read in raw:
            for g in raw[read]:
                if g[1:] in genesOfInterest:
                    subsetted.add(g[1:])
        return raw, list(subsetted), raw_positions
    positions = TokenizedPositions(reads.read_ids, reads.read_offsets, gs, ge)
    subsetted = set()
    for g in _present_in_first_use_order(reads, genesOfInterest):
        subsetted.add(g)
    return reads, list(subsetted), positions


def trim_buffers():
    """give the loader's and the writers' cached work buffers back to the system (amg_calls_trim), and the arrays the
    Python-side loader keeps for its next call (amira_amd.io: up to six token / position arrays of the last loads), and
    the reads' nucleotide sequences the last cleaning run left on the device (amira_amd.bubble_popping)"""
    from . import bubble_popping, io
    io._pool.clear()
    bubble_popping.release_sequences()   # (the reads' bases a cleaning run left on the device)
    n = C.c_int64(0)
    check(_ffi.lib.amg_calls_trim(C.byref(n)))
    return n.value
