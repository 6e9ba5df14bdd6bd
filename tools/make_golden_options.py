#!/usr/bin/env python3
"""Fixture generator (build container only): the reference's option DATA -- options/default.json and options/constant_args.json,
two json files of keys and values -- into tests/golden/ref_options.json, so the CPU tests can push the reference's own option
sets through this build's parser without reading /root/reference at test time."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/options"
out = {name: json.load(open(os.path.join(REF, name + ".json"))) for name in ("default", "constant_args")}
with open(os.path.join(ROOT, "tests", "golden", "ref_options.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote", len(out["default"]), "+", len(out["constant_args"]), "option values")
