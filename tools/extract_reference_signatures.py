#!/usr/bin/env python3
"""Pins the reference's public routine signatures the Rust adaptor (bindings/rend3-routine-amd) must repeat:
reads them out of /root/reference (where it exists) and writes tests/golden/rust_signatures.json; tests/test_rust_bindings.py
compares the adaptor's signatures with the fixture, and the fixture with the reference tree when that is present.
A signature = the text from `pub fn` to the body's `{`, whitespace collapsed, a leading `_` of parameter names dropped."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "rust_signatures.json")
# (reference file, adaptor file, function name, occurrence index among `pub fn <name>` in the reference file)
PINNED = [
    ("rend3-routine/src/base.rs", "base.rs", "new", 1),            # BaseRenderGraph::new (DepthTargets::new is occurrence 0)
    ("rend3-routine/src/base.rs", "base.rs", "add_to_graph", 0),
    ("rend3-routine/src/culling/culler.rs", "culler.rs", "new", 0),
    ("rend3-routine/src/culling/culler.rs", "culler.rs", "add_object_uniform_upload_to_graph", 0),
    ("rend3-routine/src/culling/culler.rs", "culler.rs", "add_culling_to_graph", 0),
    ("rend3-routine/src/pbr/routine.rs", "base.rs", "new", 0),     # PbrRoutine::new (the adaptor keeps it in base.rs, first `new` there)
    ("rend3-routine/src/forward.rs", "forward.rs", "new", 0),
    ("rend3-routine/src/forward.rs", "forward.rs", "add_forward_to_graph", 0),
    ("rend3-routine/src/hi_z.rs", "hi_z.rs", "new", 0),
    ("rend3-routine/src/hi_z.rs", "hi_z.rs", "add_hi_z_to_graph", 0),
    ("rend3-routine/src/tonemapping.rs", "tonemapping.rs", "new", 0),
    ("rend3-routine/src/tonemapping.rs", "tonemapping.rs", "add_to_graph", 0),
    ("rend3-routine/src/skinning.rs", "skinning.rs", "add_skinning_to_graph", 0),
]


def signatures(text, name):
    """Every `pub fn <name>` of a Rust source, normalised, in order of appearance, with its line number."""
    out = []
    for m in re.finditer(r"pub fn %s\b" % re.escape(name), text):
        depth, i = 0, m.end()
        while i < len(text):  # the body's brace is the first `{` outside the parameter list's brackets
            ch = text[i]
            if ch in "(<[":
                depth += 1
            elif ch in ")>]":
                depth -= 0 if (ch == ">" and text[i - 1] == "-") else 1
            elif ch == "{" and depth <= 0:
                break
            i += 1
        sig = " ".join(text[m.start():i].split())
        sig = re.sub(r"\b_(\w+):", r"\1:", sig)        # unused-parameter underscores
        sig = re.sub(r",\s*\)", ")", sig).replace("( ", "(").replace(" )", ")")
        sig = re.sub(r"\s*,\s*where", " where", sig).rstrip(", ")
        out.append((sig, text.count("\n", 0, m.start()) + 1))
    return out


def from_reference(ref_root):
    out = {}
    for ref_file, _adaptor_file, name, idx in PINNED:
        sig, line = signatures(open(os.path.join(ref_root, ref_file)).read(), name)[idx]
        out[f"{ref_file}::{name}#{idx}"] = {"signature": sig, "line": line}
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    data = from_reference(ref)
    json.dump(data, open(FIXTURE, "w"), indent=1, sort_keys=True)
    for k, v in sorted(data.items()):
        print(f"{k}:{v['line']}\n    {v['signature']}")
