#!/usr/bin/env python3
"""Record the public call surface of the reference's drop-in boundary as DATA (tests/golden/reference_signatures.json):
for every module a user imports flat (`from layers import MMA`, `from mma_conv import MMAConv`, ...), the top-level
functions and the classes' methods with their parameter names and literal defaults.

Build container only (reads /root/reference as TEXT with `ast`; nothing is imported or executed, so the graph-regression
files - which need the absent torch_geometric - are covered too).  The fixture holds names and defaults only.
Cited surfaces: layers.py:57-61,853  mma_conv.py:47-51,121-122,138,159-160  mask_aggr.py:13-23,53  scalers.py:10-64
models.py:10-16,62  utils.py (load_data, accuracy, ...)."""
import ast
import json
import os

REF = "/root/reference"
FILES = {"layers": "node_classification/layers.py", "scalers": "node_classification/scalers.py",
         "models": "node_classification/models.py", "utils": "node_classification/utils.py",
         "mma_conv": "graph_regression/mma_conv.py", "mask_aggr": "graph_regression/mask_aggr.py"}


def params(fn):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    out = [{"name": n, "default": d} for n, d in zip(names, defaults)]
    if a.vararg:
        out.append({"name": "*" + a.vararg.arg, "default": None})
    for x, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": x.arg, "default": ast.unparse(d) if d is not None else None, "kwonly": True})
    if a.kwarg:
        out.append({"name": "**" + a.kwarg.arg, "default": None})
    return out


def main():
    res = {}
    for mod, rel in FILES.items():
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        m = {"functions": {}, "classes": {}}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                m["functions"][node.name] = params(node)
            elif isinstance(node, ast.ClassDef):
                m["classes"][node.name] = {"bases": [ast.unparse(b) for b in node.bases],
                                           "methods": {f.name: params(f) for f in node.body if isinstance(f, ast.FunctionDef)}}
        res[mod] = m
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
