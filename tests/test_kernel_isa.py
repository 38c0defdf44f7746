"""CPU: a guard on the generated gfx950 code of the work-item rasteriser.

k_raster_big keeps the NEXT work item's record in flight while it scans the current one: an `s_load_dwordx16` issued through inline
asm (the compiler would sink a plain load to its first use) whose destination registers the compiler believes to be written at
once.  If register pressure makes it copy or spill those registers before the matching `s_waitcnt lgkmcnt(0)` statement, the copy
reads whatever the registers held -- a corrupt record, then wild addresses (round 4 lost a variant to exactly this, round 5 a build
that raised a GPU memory fault).  Nothing at run time catches that deterministically; the ISA does: between the asm load and the asm
wait no instruction may name a register of the destination range.  The test compiles csrc/r3n.hip for the device (about a minute)
and checks every instantiation."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def prefetch_hazards(asm_text, kernel_prefix):
    """[(kernel, destination range, [instructions that name it on some path to the wait], wait reached)] for every asm-issued
    s_load_dwordx16: a walk over the function's control-flow graph (labels + s_branch / s_cbranch_*) from the instruction behind the
    load, every path followed until it meets the asm `s_waitcnt lgkmcnt(0)` (block layout in the text is not execution order)."""
    out = []
    for func in re.split(r"\n(?=_Z\w+:)", asm_text):
        name = func.split(":", 1)[0]
        if not name.startswith(kernel_prefix):
            continue
        lines = func.split("\n")
        label_at = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\w+):", l)] if m}

        def is_wait(j):
            return "s_waitcnt lgkmcnt(0)" in lines[j] and j > 0 and "ASMSTART" in lines[j - 1]

        for i, line in enumerate(lines):
            if "s_load_dwordx16" not in line or i == 0 or "ASMSTART" not in lines[i - 1]:
                continue
            m = re.search(r"s_load_dwordx16 s\[(\d+):(\d+)\]", line)
            lo, hi = int(m.group(1)), int(m.group(2))
            touched, waited, seen, todo = [], False, set(), [i + 1]
            while todo:
                j = todo.pop()
                while j < len(lines) and j not in seen:
                    seen.add(j)
                    lj = lines[j]
                    if is_wait(j):
                        waited = True
                        break
                    code = lj.split(";")[0].strip()
                    if code and not code.startswith(".") and not code.endswith(":"):
                        regs = [int(r) for r in re.findall(r"\bs(\d+)\b", code)]
                        spans = [(int(a), int(b)) for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", code)]
                        if j != i and (any(lo <= r <= hi for r in regs) or any(not (b < lo or a > hi) for a, b in spans)):
                            touched.append(code)
                        br = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)", code)
                        if br:
                            if br.group(2) in label_at:
                                todo.append(label_at[br.group(2)])
                            if br.group(1) == "s_branch":
                                break
                        if code.startswith("s_endpgm") or code.startswith("s_setpc"):
                            break
                    j += 1
            out.append((name, (lo, hi), touched, waited))
    return out


def test_work_item_prefetch_registers_are_left_alone_until_the_wait():
    from rend3_amd import build
    src = os.path.join(build.CSRC, "r3n.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "r3n.s")
        res = subprocess.run([build.hipcc()] + build.FLAGS + ["--cuda-device-only", "-S", "-o", asm, src], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-3000:]
        text = open(asm).read()
    sites = prefetch_hazards(text, "_Z12k_raster_big")
    assert len(sites) >= 9, f"expected one prefetch per k_raster_big instantiation, found {len(sites)}"
    for name, (lo, hi), touched, waited in sites:
        assert waited, f"{name}: no asm `s_waitcnt lgkmcnt(0)` behind the prefetch into s[{lo}:{hi}]"
        assert not touched, f"{name}: s[{lo}:{hi}] is named before the wait while the load may still be in flight: {touched[:4]}"


def test_the_scanner_sees_a_planted_hazard():
    planted = """_Z12k_raster_bigPlanted:
	s_mov_b32 s1, 0
	;;#ASMSTART
	s_load_dwordx16 s[40:55], s[2:3], 0x0
	;;#ASMEND
	v_writelane_b32 v9, s41, 3
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
"""
    (name, span, touched, waited), = prefetch_hazards(planted, "_Z12k_raster_big")
    assert span == (40, 55) and waited and touched == ["v_writelane_b32 v9, s41, 3"]
