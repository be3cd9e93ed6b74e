"""Writes tests/golden/api_surface.json: the NAMES of the reference's classes' methods and module functions on the
hot path (an index like SURVEY.md Appendix F, no source text), from /root/reference in the build container.
usage: python tests/golden/gen_api_surface.py"""
import ast
import json
import os

REF = "/root/reference/amira"
HERE = os.path.dirname(os.path.abspath(__file__))
out = {"classes": {}, "functions": {}}
for mod, cls in [("construct_graph", "GeneMerGraph"), ("construct_node", "Node"), ("construct_edge", "Edge"),
                 ("construct_gene", "Gene"), ("construct_gene_mer", "GeneMer"), ("construct_read", "Read")]:
    tree = ast.parse(open(os.path.join(REF, mod + ".py")).read())
    out["classes"][cls] = [n.name for c in tree.body if isinstance(c, ast.ClassDef) and c.name == cls
                           for n in c.body if isinstance(n, ast.FunctionDef)]
for mod in ("graph_utils", "path_finding_utils"):
    tree = ast.parse(open(os.path.join(REF, mod + ".py")).read())
    out["functions"][mod] = [n.name for n in tree.body if isinstance(n, ast.FunctionDef)]
json.dump(out, open(os.path.join(HERE, "api_surface.json"), "w"), indent=1)
print({k: len(v) for k, v in out["classes"].items()}, {k: len(v) for k, v in out["functions"].items()})
