#!/usr/bin/env python3
"""Copies what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ under the round's names (gpurun_out/ is scratch;
profiles/ is what is committed and cited).   usage: python tools/collect_profiles.py r02"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for name in ("bench", "bench_fast", "bench_instanced", "bench_untextured", "bench_msaa4", "bench_cfg4", "bench_cfg4_traffic", "bench_bistro_v2", "bench_scene",
             "bench_exchange_rows", "bench_exchange_objects", "bench_exchange_rows_python", "bench_exchange_spatial", "bench_exchange_slots", "bench_under_rocprof", "bench_under_rocprof_serial", "bench_under_rocprof_single"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
        json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)
for sub, out in (("kt", "kernel_stats"), ("kts", "kernel_stats_serial"), ("kt1", "kernel_stats_single_stream"),
                 ("kt1_cfg4", "cfg4_kernel_stats_single_stream"), ("kt1_v2", "bistro_v2_kernel_stats_single_stream")):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, f"{tag}_bench_{out}.csv"))
for sub, out in (("pmc", "pmc_kernels"), ("pmc_cfg4", "pmc_kernels_cfg4"), ("pmc_untextured", "pmc_kernels_untextured")):
    p = os.path.join(src, sub, "traffic.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{out}.json"))
p = os.path.join(src, "traffic.json")  # one entry per workload variant (tools/make_traffic.py): what bench.py quotes, stale-checked against the kernel sources
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, "traffic.json"))
for name in ("host_rate.txt", "host_rate_nodes.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        keep = [l for l in open(p).read().splitlines() if "ms/frame" in l or "ms per frame" in l or "host time" in l]
        open(os.path.join(dst, f"{tag}_{name}"), "w").write("\n".join(keep) + "\n")
for name, out in (("exact_math.txt", "exact_math.txt"), ("stall/pmc_raster_table.md", "pmc_raster_table.md")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{out}"))
p = os.path.join(src, "configs.jsonl")
if os.path.exists(p):
    rows = [json.loads(l) for l in open(p).read().splitlines() if l.startswith("{")]
    groups = {"cfg1_scifi": [r for r in rows if "configs[1]" in r["config"]], "cfg3_emerald": [r for r in rows if "configs[3]" in r["config"]],
              "cfg4_skinning": [r for r in rows if "configs[4]" in r["config"]]}
    for name, rs in groups.items():
        if rs:
            json.dump(rs if len(rs) > 1 else rs[0], open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)
print(sorted(f for f in os.listdir(dst) if f.startswith(tag) or f == "traffic.json"))
