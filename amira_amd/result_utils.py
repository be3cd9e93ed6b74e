"""`write_pandora_gene_calls` under the reference's own name and signature (result_utils.py:1260-1264): the corrected
gene calls and their positions out as the two JSON files the reference dumps — byte for byte what `json.dumps` writes
for the same mappings (tests/test_calls_cpu.py).  Array-backed mappings (amira_amd.io.TokenizedReads /
TokenizedPositions, what GeneMerGraph.correct_reads and the drivers hand back) are written natively from their arrays,
the two files side by side; plain dicts take the reference's own two lines.

Everything else of the reference's result_utils.py (racon / minimap2 / samtools post-processing, TSV output) is
outside the hot path.
"""
import json
import threading

import numpy as np

from .io import TokenizedPositions, TokenizedReads, write_gene_calls, write_gene_positions


def write_pandora_gene_calls(output_dir, gene_position_dict, annotatedReads, outfile_1, outfile_2):
    """result_utils.py:1260-1264: json.dumps(annotatedReads) -> outfile_1, json.dumps(gene_position_dict) -> outfile_2
    (output_dir is not used there either)."""
    jobs = []
    # (reads bubble popping rewrote and positions a correction redirected are spelled into the arrays first: .settled())
    if isinstance(annotatedReads, TokenizedReads):
        r = annotatedReads.settled()
        jobs.append(lambda: write_gene_calls(outfile_1, r.vocab, r.tokens, r.read_offsets, r.read_ids))
    else:
        jobs.append(lambda: _dump(outfile_1, annotatedReads))
    if isinstance(gene_position_dict, TokenizedPositions):
        p = gene_position_dict.settled()
        jobs.append(lambda: write_gene_positions(outfile_2, p.gene_start, p.gene_end, p.read_offsets, p.read_ids))
    else:
        jobs.append(lambda: _dump(outfile_2, gene_position_dict))
    # the two files are independent: their writers run side by side (the native writers release the interpreter lock
    # and use the host's cores; each alone leaves most of a large box idle)
    errs = []

    def run(job):
        try:
            job()
        except BaseException as e:  # noqa: BLE001 - re-raised below, in the caller's thread
            errs.append(e)

    t = threading.Thread(target=run, args=(jobs[1],))
    t.start()
    run(jobs[0])
    t.join()
    if errs:
        raise errs[0]


def _dump(path, mapping):
    if isinstance(mapping, (TokenizedReads, TokenizedPositions)):
        mapping = {k: _listed(mapping[k]) for k in mapping}
    with open(path, "w") as o:
        o.write(json.dumps(mapping))


def _listed(v):
    return [list(x) if isinstance(x, tuple) else x for x in v]
