#!/usr/bin/env python3
"""Which reference (rend3 / rend3-routine / rend3-types) items the Rust adaptor crate (bindings/rend3-routine-amd) names, and
whether each is PUBLIC API there -- the adaptor is uncompiled in this image (no Rust toolchain), so "it only uses what a
downstream crate may use" is checked on the sources:

  * every `rend3::..` / `rend3_routine::..` path of a `use` declaration or a qualified expression is resolved through the
    reference's module tree: each module segment must be `pub mod` (or reachable through a `pub use`), the item itself
    `pub struct|enum|trait|fn|type|const` in that module or re-exported into it by a `pub use`;
  * every `data_core.<field>`, `eval_output.<field>`, `renderer.<field>` the node bodies read must be a `pub` field, and every
    method called on a manager reached that way (`data_core.object_manager.buffer::<M>()`) a `pub fn` of that manager's type;
  * methods called on reference types listed in METHODS (camera state, camera specifier) must be `pub fn`.

usage: reference_visibility.py [/root/reference]   -> writes tests/golden/rust_visibility.json (the fixture the CPU test compares
with when the reference tree is absent)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTOR = os.path.join(ROOT, "bindings", "rend3-routine-amd", "src")
HOOKS = os.path.join(ROOT, "bindings", "rend3-hooks", "amd.rs")  # installed as rend3/src/util/amd.rs by bindings/rend3-hooks.patch
FIXTURE = os.path.join(ROOT, "tests", "golden", "rust_visibility.json")
CRATES = {"rend3": "rend3/src", "rend3_routine": "rend3-routine/src", "rend3_types": "rend3-types/src"}
ITEM = r"pub\s+(?:unsafe\s+)?(?:struct|enum|trait|fn|type|const|static|union)\s+{name}\b"
# (type, reference file, methods the adaptor calls on values of that type)
METHODS = [("CameraState", "rend3/src/managers/camera.rs", ["view", "view_proj", "world_frustum"]),
           ("CameraSpecifier", "rend3-routine/src/common/camera.rs", ["to_shader_index"])]
# field owner -> (struct name, file)
OWNERS = {"data_core": ("RendererDataCore", "rend3/src/renderer/mod.rs"), "eval_output": ("InstructionEvaluationOutput", "rend3/src/graph/graph.rs"),
          "renderer": ("Renderer", "rend3/src/renderer/mod.rs")}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def block_after(text, start):
    """text of the `{ ... }` block that opens at or after `start`"""
    i = text.index("{", start)
    depth, j = 0, i
    while True:
        c = text[j]
        depth += c == "{"
        depth -= c == "}"
        if depth == 0:
            return text[i + 1:j]
        j += 1


def expand_use(tree, prefix=""):
    """`a::{b::*, C, d::{E as F}}` -> [("a::b", "*", "*"), ("a", "C", "C"), ("a::d", "F", "E")]: (module path, exported name, original name)"""
    tree = tree.strip()
    depth, start, parts = 0, 0, []
    # split the top level at commas
    for i, c in enumerate(tree):
        depth += c == "{"
        depth -= c == "}"
        if c == "," and depth == 0:
            parts.append(tree[start:i])
            start = i + 1
    parts.append(tree[start:])
    out = []
    for part in parts:
        part = part.strip()
        if not part:
            continue
        if "{" in part:
            head = part[:part.index("{")].rstrip(":").strip()
            inner = part[part.index("{") + 1:part.rindex("}")]
            out += expand_use(inner, (prefix + "::" + head).strip(":") if head else prefix)
        else:
            path, _, alias = part.partition(" as ")
            path = path.strip()
            mod, _, name = path.rpartition("::")
            full_mod = (prefix + "::" + mod).strip(":") if mod else prefix
            out.append((full_mod, (alias.strip() or name), name))
    return out


class Module:
    def __init__(self, crate, text, directory, stem):
        self.crate, self.text, self.dir, self.stem = crate, text, directory, stem

    def child(self, name, need_pub):
        """the module `name` declared in this module (file or inline block); None if absent or (need_pub and) private"""
        m = re.search(r"(?m)^\s*(pub(?:\([^)]*\))?\s+)?mod\s+%s\s*(;|\{)" % re.escape(name), self.text)
        if not m:
            return None
        if need_pub and (m.group(1) is None or "(" in m.group(1)):
            return None
        if m.group(2) == "{":  # an inline module's file children live in <dir of the parent's children>/<name>/
            return Module(self.crate, block_after(self.text, m.start()), os.path.join(self.subdir(), name), "mod")
        for cand in (os.path.join(self.subdir(), name + ".rs"), os.path.join(self.subdir(), name, "mod.rs")):
            if os.path.exists(cand):
                return Module(self.crate, strip_comments(open(cand).read()), os.path.dirname(cand), "mod" if cand.endswith("mod.rs") else name)
        return None

    def subdir(self):
        if self.stem in ("lib", "mod"):
            return self.dir
        return os.path.join(self.dir, self.stem)

    def has_pub_item(self, name, ref, depth=0):
        if re.search(ITEM.format(name=re.escape(name)), self.text):
            return True
        if depth > 4:
            return False
        # re-exports: pub use a::b::{X, c::*, Y as Z}; pub use a::b::X; pub use a::b::*;
        for m in re.finditer(r"(?<![A-Za-z_0-9(])pub\s+use\s+([^;]+);", self.text):
            for mod_path, exported, original in expand_use(m.group(1)):
                if exported != name and exported != "*":
                    continue
                first = mod_path.split("::")[0] if mod_path else ""
                external = bool(first) and first not in CRATES and first not in ("crate", "self", "super") and self.child(first, need_pub=False) is None
                if external:
                    # a re-export from a crate outside the reference (wgpu-types, glam): public by the `pub use` itself; globs of
                    # external crates cannot be enumerated here and do not count
                    if exported == name:
                        return True
                    continue
                target = self.resolve_module(mod_path, ref)
                if target is None:
                    continue
                want = name if exported == "*" else original
                if target.has_pub_item(want, ref, depth + 1) or (exported != "*" and target.child(want, need_pub=False) is not None):
                    return True
        return False

    def resolve_module(self, path, ref):
        segs = [s for s in path.split("::") if s]
        if not segs:
            return self
        if segs[0] in CRATES:  # another crate of the reference
            mod = crate_root(segs[0], ref)
            segs = segs[1:]
        elif segs[0] == "crate":
            mod = crate_root(self.crate, ref)
            segs = segs[1:]
        elif segs[0] == "self":
            mod, segs = self, segs[1:]
        elif segs[0] == "super":
            return None  # not needed for the paths the adaptor uses
        else:
            mod = self
        for s in segs:
            nxt = mod.child(s, need_pub=False)  # a re-export may name a private module of its own crate
            if nxt is None:
                return None
            mod = nxt
        return mod


def crate_root(crate, ref):
    d = os.path.join(ref, CRATES[crate])
    return Module(crate, strip_comments(open(os.path.join(d, "lib.rs")).read()), d, "lib")


def resolve_public(path, ref):
    """(ok, why) for a `crate::mod::..::Item` path as a downstream crate sees it"""
    segs = path.split("::")
    if segs[0] not in CRATES:
        return True, "not a reference crate"
    mod = crate_root(segs[0], ref)
    for s in segs[1:-1]:
        nxt = mod.child(s, need_pub=True)
        if nxt is None:
            return False, f"module `{s}` is not public in {path}"
        mod = nxt
    if len(segs) == 1:
        return True, "crate"
    name = segs[-1]
    if mod.child(name, need_pub=True) is not None:
        return True, "pub mod"
    return (True, "pub item") if mod.has_pub_item(name, ref) else (False, f"`{name}` is not a public item of {'::'.join(segs[:-1])}")


def adaptor_paths():
    """every rend3 / rend3_routine path the adaptor names: use declarations (expanded) and qualified paths in expressions / types"""
    out = {}
    for f in sorted(os.listdir(ADAPTOR)):
        text = strip_comments(open(os.path.join(ADAPTOR, f)).read())
        for m in re.finditer(r"\buse\s+((?:rend3|rend3_routine|rend3_types)(?:::[A-Za-z_0-9]+)*)(?:::\{([^}]*)\})?\s*;", text):
            head, group = m.group(1), m.group(2)
            if group is None:
                out.setdefault(head, set()).add(f)
            else:
                for n in group.split(","):
                    n = n.strip().split(" as ")[0].strip()
                    if n == "self":
                        out.setdefault(head, set()).add(f)
                    elif n:
                        out.setdefault(head + "::" + n, set()).add(f)
        for m in re.finditer(r"(?<![A-Za-z_0-9:])((?:rend3|rend3_routine)(?:::[A-Za-z_][A-Za-z_0-9]*)+)", text):
            path = m.group(1)
            if text[max(0, m.start() - 4):m.start()].strip().endswith("use"):
                continue
            segs = path.split("::")
            # an associated item / variant after a type (Type::Variant, Type::CONST): cut at the first CamelCase segment
            for i, sgm in enumerate(segs[1:], 1):
                if sgm[0].isupper():
                    segs = segs[:i + 1]
                    break
            out.setdefault("::".join(segs), set()).add(f)
    return out


def field_checks(ref):
    """pub-ness of the fields / manager methods the node bodies reach through data_core / eval_output / renderer"""
    res = {}
    for f in sorted(os.listdir(ADAPTOR)):
        text = strip_comments(open(os.path.join(ADAPTOR, f)).read())
        for owner, (struct, file) in OWNERS.items():
            src = strip_comments(open(os.path.join(ref, file)).read())
            body = block_after(src, re.search(r"pub\s+struct\s+%s\b" % struct, src).start())
            for m in re.finditer(r"\b%s\s*\.\s*([a-z_][a-z_0-9]*)\s*(\.\s*([a-z_][a-z_0-9]*)\s*(?:::<[^>]*>)?\s*\()?" % owner, text):
                field = m.group(1)
                if re.match(r"^(clone|as_ref|lock)$", field):
                    continue
                if m.group(3) in ("into", "from", "clone", "as_ref", "iter", "len"):  # std traits / containers, whatever the field's type
                    m = re.match(r"\b%s\s*\.\s*([a-z_][a-z_0-9]*)" % owner, m.group(0))
                fm = re.search(r"(?m)^\s*(pub\s+)?%s\s*:\s*([^,\n]+)" % re.escape(field), body)
                key = f"{struct}.{field}"
                if fm is None:
                    # a method of the owner itself (renderer.add_graph_data(..))
                    ok = re.search(r"pub\s+fn\s+%s\b" % re.escape(field), src) is not None
                    res[key + "()"] = ok
                    continue
                res[key] = fm.group(1) is not None
                if m.lastindex and m.lastindex >= 3 and m.group(3):
                    ty = re.sub(r"[^A-Za-z_0-9]", " ", fm.group(2)).split()
                    ty = [t for t in ty if t[0].isupper()][-1]
                    method, found = m.group(3), None
                    for d, _dirs, files in os.walk(os.path.join(ref, "rend3", "src")):
                        for g in files:
                            if g.endswith(".rs"):
                                t = strip_comments(open(os.path.join(d, g)).read())
                                if re.search(r"pub\s+struct\s+%s\b" % ty, t):
                                    if re.search(r"\bfn\s+%s\b" % re.escape(method), t) is None:
                                        found = "foreign"  # not a method of the reference type (Vec::iter, Into::into ...)
                                    else:
                                        found = re.search(r"pub\s+fn\s+%s\b" % re.escape(method), t) is not None
                    if found != "foreign":
                        res[f"{ty}::{method}()"] = bool(found)
    for ty, file, methods in METHODS:
        src = strip_comments(open(os.path.join(ref, file)).read())
        for mth in methods:
            res[f"{ty}::{mth}()"] = re.search(r"pub\s+fn\s+%s\b" % mth, src) is not None
    return res


def patched_reference(ref):
    """A copy of the reference's three crates with bindings/rend3-hooks.patch applied (tools/make_hooks_patch.py's edits, written
    out directly): the tree the adaptor is written against -- `rend3::util::amd::AmdContext`, `Renderer::amd` exist only there."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_hooks_patch
    dst = tempfile.mkdtemp(prefix="rend3_patched_")
    for crate_dir in ("rend3", "rend3-routine", "rend3-types"):
        shutil.copytree(os.path.join(ref, crate_dir), os.path.join(dst, crate_dir))
    shutil.copy(os.path.join(ref, "Cargo.toml"), os.path.join(dst, "Cargo.toml"))
    for rel, (_old, new) in make_hooks_patch.patched_files(ref).items():
        os.makedirs(os.path.dirname(os.path.join(dst, rel)), exist_ok=True)
        open(os.path.join(dst, rel), "w").write(new)
    return dst


def survey(ref):
    """`ref`: the reference checkout; the survey runs on its PATCHED copy (patched_reference)."""
    import shutil
    patched = patched_reference(ref)
    try:
        return survey_tree(patched)
    finally:
        shutil.rmtree(patched, ignore_errors=True)


def survey_tree(ref):
    paths = {}
    for path, files in sorted(adaptor_paths().items()):
        ok, why = resolve_public(path, ref)
        paths[path] = {"public": ok, "how": why, "used_in": sorted(files)}
    return {"paths": paths, "members": dict(sorted(field_checks(ref).items()))}


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    doc = survey(ref)
    json.dump(doc, open(FIXTURE, "w"), indent=1, sort_keys=True)
    bad = [p for p, v in doc["paths"].items() if not v["public"]] + [m for m, ok in doc["members"].items() if not ok]
    print(len(doc["paths"]), "paths,", len(doc["members"]), "members; not public:", bad)
