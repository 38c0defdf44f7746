"""CPU: bindings/rend3-hooks.patch -- the reference-side hooks (Renderer::new creating the AmdContext, every manager's upload
mirrored into it) -- is an APPLICABLE patch of the reference tree, and it is the one tools/make_hooks_patch.py derives.

The reference checkout exists in the build container only (/root/reference); elsewhere the test checks what can be checked
without it: the patch touches exactly the sites INTEGRATION.md section 2 names and every r3n_* call of the installed file is a
declared symbol (tests/test_rust_bindings.py covers the latter)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "bindings", "rend3-hooks.patch")
SITES = ["Cargo.toml", "rend3/Cargo.toml", "rend3/src/lib.rs", "rend3/src/util/amd.rs", "rend3/src/util/freelist/buffer.rs",
         "rend3/src/renderer/mod.rs", "rend3/src/renderer/setup.rs", "rend3/src/renderer/eval.rs", "rend3/src/managers/mesh.rs",
         "rend3/src/managers/object.rs", "rend3/src/managers/material.rs", "rend3/src/managers/directional.rs", "rend3/src/managers/point.rs"]


def test_patch_names_the_documented_sites():
    text = open(PATCH).read()
    files = re.findall(r"^diff --git a/(\S+) b/\S+$", text, flags=re.M)
    assert sorted(files) == sorted(SITES)
    # the installed file is bindings/rend3-hooks/amd.rs, line for line
    new = text[text.index("+++ b/rend3/src/util/amd.rs"):]
    new = new[new.index("@@"):].split("\n", 1)[1]
    end = new.find("\ndiff --git")
    body = "\n".join(l[1:] for l in (new if end < 0 else new[:end]).split("\n") if l.startswith("+"))
    assert body.strip() == open(os.path.join(ROOT, "bindings", "rend3-hooks", "amd.rs")).read().strip()
    # every manager upload of INTEGRATION.md section 2 has its mirror call in a hunk
    for call in ("amd.mesh_buffer_write(range.start", "amd.mesh_buffer_write(index_range.start", "amd.objects_write(", "amd.materials_write(",
                 "self.amd.texture_fill(", "renderer.amd.texture_remove(", "renderer.amd.textures_flush()", "renderer.amd.lights_write_directional(",
                 "renderer.amd.lights_write_point(", "AmdContext::from_env()"):
        assert any(l.startswith("+") and call in l for l in text.splitlines()), call
    docs = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "rend3-hooks.patch" in docs


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rend3", "src")), reason="the reference checkout exists in the build container only")
def test_patch_applies_to_the_reference_and_is_current(tmp_path):
    import make_hooks_patch
    assert make_hooks_patch.make(REF) == open(PATCH).read(), "bindings/rend3-hooks.patch is stale: run python tools/make_hooks_patch.py"
    work = tmp_path / "rend3"
    work.mkdir()
    shutil.copy(os.path.join(REF, "Cargo.toml"), work / "Cargo.toml")
    shutil.copytree(os.path.join(REF, "rend3"), work / "rend3")
    res = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=work, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    res = subprocess.run(["git", "apply", PATCH], cwd=work, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    # applied == what the generator writes (git apply and the string edits agree)
    for rel, (_old, new) in make_hooks_patch.patched_files(REF).items():
        assert open(work / rel).read() == new, rel
    # the patched tree still balances its braces / parentheses in every touched file (a cut-off hunk would not)
    for rel in SITES:
        if rel.endswith(".rs"):
            t = re.sub(r"//[^\n]*", "", open(work / rel).read())
            t = re.sub(r'"(?:\\.|[^"\\])*"', '""', t)
            t = re.sub(r"'(?:\\.|[^'\\])'", "' '", t)
            assert t.count("{") == t.count("}") and t.count("(") == t.count(")"), rel
    # the adaptor crate resolves against the PATCHED tree (rend3::util::amd::AmdContext, Renderer::amd are public there)
    import reference_visibility as V
    doc = V.survey(REF)
    assert doc["paths"]["rend3::util::amd::AmdContext"]["public"] and doc["paths"]["rend3::util::amd::texture_format_id"]["public"]
    amd = open(work / "rend3" / "src" / "renderer" / "mod.rs").read()
    assert re.search(r"pub amd: Arc<crate::util::amd::AmdContext>", amd)
