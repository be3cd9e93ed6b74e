"""Oracle of the JSON front end / write-back.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

Restates amira/pre_processing.py:44-63 (process_pandora_json) and amira/result_utils.py:1260-1264
(write_pandora_gene_calls) on plain dicts with the standard json module, quirks included: the reads without a gene of
interest are collected and never deleted, and the genes of interest come back as `list(set)` — an order that follows
the interpreter's string hashing (goldens are generated and compared with PYTHONHASHSEED=0).
"""
import json


def process_pandora_json(pandoraJSON, genesOfInterest, gene_positions):
    with open(pandoraJSON) as i:                      # :47-48
        annotatedReads = json.loads(i.read())
    with open(gene_positions) as i:                   # :49-50
        gene_position_dict = json.loads(i.read())
    to_delete = []                                    # :51 (filled, never used)
    subsettedGenesOfInterest = set()                  # :52
    for read in annotatedReads:                       # :53
        containsAMRgene = False
        for g in range(len(annotatedReads[read])):
            if annotatedReads[read][g][1:] in genesOfInterest:   # :56 raw name, strand character cut off
                containsAMRgene = True
                subsettedGenesOfInterest.add(annotatedReads[read][g][1:])
        if not containsAMRgene:
            to_delete.append(read)
    genesOfInterest = list(subsettedGenesOfInterest)  # :61
    return annotatedReads, genesOfInterest, gene_position_dict


def write_pandora_gene_calls(output_dir, gene_position_dict, annotatedReads, outfile_1, outfile_2):
    with open(outfile_1, "w") as o:                   # :1261-1262
        o.write(json.dumps(annotatedReads))
    with open(outfile_2, "w") as o:                   # :1263-1264
        o.write(json.dumps(gene_position_dict))
