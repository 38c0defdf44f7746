#!/usr/bin/env python3
"""Folds the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles-style JSON: per kernel and launch HBM bytes
(FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction for wide coalesced loads -- calibrated in the same run on
k_mark_visible / k_hiz_head, whose byte counts are known -- plus WRITE_SIZE; both counters are in KiB), VALU wave-instructions,
VALU busy fraction and instruction-cache figures.  Stamped with the sha of the kernel sources so that bench.py quotes the
figures only for a build of the same sources.   usage: make_traffic.py <gpurun_out/tag> [bench args...]"""
import csv
import datetime
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}


def _targ(name, i):
    """i-th template argument of a kernel name ('false', '1', 'true', ...), '' when it has fewer"""
    inner = name[name.find("<") + 1: name.rfind(">")] if "<" in name else ""
    parts = [x.strip() for x in inner.split(",")]
    return parts[i] if i < len(parts) else ""


def short(name):
    base = name.split("(")[0].replace("void ", "").strip()
    return base


def main():
    out = sys.argv[1]
    args = sys.argv[2:]
    passes = {}
    for name in ("fetch", "write", "sq", "ic", "mix", "mix2"):
        files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
        passes[name] = per_kernel(files[0]) if files else {}
    kernels = sorted(set().union(*[set(p) for p in passes.values()]))
    table = {}
    for k in kernels:
        f = passes["fetch"].get(k, {}).get("FETCH_SIZE")
        w = passes["write"].get(k, {}).get("WRITE_SIZE")
        sq = passes["sq"].get(k, {})
        ic = passes["ic"].get(k, {})
        row = {}
        if f is not None and w is not None:
            row["fetch_bytes_x2"] = int(2 * f * 1024)
            row["write_bytes"] = int(w * 1024)
            row["hbm_bytes"] = row["fetch_bytes_x2"] + row["write_bytes"]
        if sq:
            row.update({c.lower(): v for c, v in sq.items()})
            if sq.get("GRBM_GUI_ACTIVE") and sq.get("SQ_ACTIVE_INST_VALU") is not None:
                # SQ_ACTIVE_INST_VALU is in quad-cycles summed over the SIMDs; capacity = 256 CU x 4 SIMD x (GUI_ACTIVE / 8 XCDs) / 4
                row["valu_busy"] = round(sq["SQ_ACTIVE_INST_VALU"] / (256 * 4 * (sq["GRBM_GUI_ACTIVE"] / 8.0) / 4.0), 4)
        if ic:
            row.update({c.lower(): v for c, v in ic.items()})
        for extra_pass in ("mix", "mix2"):
            row.update({c.lower(): v for c, v in passes[extra_pass].get(k, {}).items()})
        table[short(k)] = row
    variant = ("instanced" if "--instanced" in args else "unique") + ("-untextured" if "--untextured" in args else "-textured") + \
              ("-s4" if "--samples" in args and args[args.index("--samples") + 1] == "4" else "-s1") + \
              ("-fast" if "--shade-mode" in args and args[args.index("--shade-mode") + 1] == "fast" else "") + \
              ("-cfg4" if "--config" in args and args[args.index("--config") + 1] == "4" else "") + ("-v2" if "--bistro-v2" in args else "")
    import bench
    pick = {"k_resolve_opaque": [k for k in table if k.startswith("k_resolve_opaque")],
            "k_triangle_cull": [k for k in table if k.startswith("k_triangle_cull")],
            "k_raster_small<depth>": [k for k in table if k.startswith("k_raster_small<true")],
            "k_raster_big<depth>": [k for k in table if k.startswith("k_raster_big<true")],
            "k_shadow_tiles": [k for k in table if k.startswith("k_shadow_tiles")],
            # viewport rasterisers: <DEPTH_ONLY=false, S, TEX, ...>; the cutout key's launches are the instantiations with TEX (textured
            # alpha) or without NOCUT -- k_raster_small<false, S, TEX, NOCUT, SHORTA>, k_raster_big<false, S, TEX, BLEND, NOCUT, SHORTA>
            "k_raster_small<vis>": [k for k in table if k.startswith("k_raster_small<false") and _targ(k, 3) == "true"],
            "k_raster_big<vis>": [k for k in table if k.startswith("k_raster_big<false") and _targ(k, 4) == "true"],
            "k_raster_small<vis,cutout>": [k for k in table if k.startswith("k_raster_small<false") and _targ(k, 3) != "true"],
            "k_raster_big<vis,cutout>": [k for k in table if k.startswith("k_raster_big<false") and _targ(k, 3) != "true" and _targ(k, 4) != "true"],
            "k_object_pass_chained": [k for k in table if k.startswith("k_object_pass_chained<true")],
            "k_hiz_head": [k for k in table if k.startswith("k_hiz_head")]}
    doc = {"source": "tools/gpu_pmc.sh: rocprofv3 --kernel-trace --pmc, one counter set per run; FETCH_SIZE x2 (gfx950 wide-load correction) + WRITE_SIZE",
           "taken": datetime.date.today().isoformat(), "variant": variant, "kernel_sources_sha": bench.kernel_sources_sha(),
           "bytes_per_launch": {}, "valu_busy": {}, "valu_insts_per_launch": {}, "useful_flops_per_launch": {}, "useful_flops_unpacked_count": {}, "kernels": table}
    for key, names in pick.items():
        if names and "hbm_bytes" in table[names[0]]:
            doc["bytes_per_launch"][key] = table[names[0]]["hbm_bytes"]
        if names and "valu_busy" in table[names[0]]:
            doc["valu_busy"][key] = table[names[0]]["valu_busy"]
        if names and "sq_insts_valu" in table[names[0]]:
            doc["valu_insts_per_launch"][key] = int(table[names[0]]["sq_insts_valu"])
        if names and "sq_insts_valu_add_f32" in table[names[0]]:
            # f32 arithmetic only -- moves, selects, compares, min / max, floor, division fix-ups, integer / address work and
            # conversions are NOT flops.  SQ_INSTS_VALU_FLOPS_FP32 weighs an instruction by its flops per lane: add / sub / mul /
            # transcendental 1, fma 2, the PACKED forms twice that (calibrated per instruction on the box: tools/valu_counter_probe.hip,
            # profiles/r05_valu_counter_probe.txt).  The per-class counters count a packed instruction ONCE, so the sum
            # add + mul + 2 x fma + transcendental (rounds 4-5) missed half of every v_pk_mul_f32 / v_pk_add_f32 -- half of the
            # resolve's f32 arithmetic instructions are packed; it is kept beside the counter as `..._unpacked_count`.
            t = table[names[0]]
            by_class = int(64 * (t["sq_insts_valu_add_f32"] + t["sq_insts_valu_mul_f32"] + 2 * t["sq_insts_valu_fma_f32"] + t.get("sq_insts_valu_trans_f32", 0.0)))
            doc["useful_flops_unpacked_count"][key] = by_class
            doc["useful_flops_per_launch"][key] = int(64 * t["sq_insts_valu_flops_fp32"]) if t.get("sq_insts_valu_flops_fp32") else by_class
    # one entry per workload variant (bench.py quotes the entry of the variant it runs): merged into the file the previous
    # variants of THIS build left in the same directory tree (gpurun_out/<tag>/traffic.json), or into profiles/traffic.json
    merged = {"source": doc["source"], "kernel_sources_sha": doc["kernel_sources_sha"], "variants": {}}
    for prior in (os.path.join(os.path.dirname(out.rstrip("/")), "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json")):
        try:
            old = json.load(open(prior))
            if old.get("kernel_sources_sha") == doc["kernel_sources_sha"] and "variants" in old:
                merged["variants"].update(old["variants"])
                break
        except (OSError, ValueError):
            pass
    merged["variants"][variant] = {k: doc[k] for k in ("taken", "bytes_per_launch", "valu_busy", "valu_insts_per_launch", "useful_flops_per_launch", "useful_flops_unpacked_count", "kernels")}
    json.dump(merged, open(os.path.join(os.path.dirname(out.rstrip("/")), "traffic.json"), "w"), indent=1)
    json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    for k, row in table.items():
        print(k[:60], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in row.items()})


if __name__ == "__main__":
    main()
