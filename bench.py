#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric ("culled objects/s + shaded Mpixels/s @4K, Bistro scene") on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched as python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one whole frame of the hot path over the synthetic Bistro-like scene of BASELINE.json configs[2]
(rend3_amd/scenes.py::bistro_like: ~3 000 objects each owning its geometry -- ~2.8 M unique triangles, a ~216 MB mesh
buffer --, 130 PBR materials, 4 directional lights with 2048^2 shadow views, 3840x2160, camera dollying down the street
so the residual pass has real work; --instanced is the round-1 variant whose 11 shared meshes fit in L2):
  per shadow view: uniform bake -> object frustum cull + slot scan -> triangle cull + compaction -> depth raster;
  viewport: bake -> raster predicted -> Hi-Z -> cull -> raster residual -> PBR resolve -> tonemap.
All inputs are resident in HBM before the timed region; per-frame camera blocks (a few KB) are precomputed on the
host and uploaded through the C ABI exactly as the Rust side would.  value = W*H*K / time (whole job, all ranks).
At N=1 the first two frames (camera steps 0 and 1) are rendered with a read-back of step 1, and the CPU oracle -- which
renders the same steps for the `cpu_baseline` figure anyway -- checks that frame: visible-object sets, per-triangle
pass / residual sets, visibility keys, shadow atlas and the Rgba16Float target bit for bit, the tonemapped framebuffer
within 1e-3 (`"parity"` in the JSON line).  Prints ONE JSON line on rank 0.
"""
import hashlib
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 3840, 2160
AMBIENT = (0.1, 0.1, 0.1, 1.0)
CLEAR = (0.25, 0.45, 0.8, 1.0)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3  # same guide: peak FP32 vector rate (packed FMA: 2 lanes x 2 flop per lane-slot and issue cycle)


def kernel_sources_sha():
    """sha256 over the library's kernel sources: profiles/traffic.json is stamped with it when the PMC passes are taken
    (tools/make_traffic.py), and the counters are only quoted by a build of the same sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rend3_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def compare_frames(fo, fp, fast=False):
    """Oracle frame `fo` against the HIP frame `fp` (read-back dicts): what must be bit-exact is counted, the tonemapped
    framebuffer is measured.  Same checks as tests/test_gpu_parity.py::compare_frames."""
    import numpy as np
    n = len(fo["pass"])
    baked_slots = (fo["objects"][:, 29] != 0) & fo["visible"].astype(bool)  # the slots whose matrices are read (frustum-visible)
    out = {
        "baked_matrices_equal": bool(np.array_equal(fo["baked"].view(np.uint32)[baked_slots], fp["baked"].view(np.uint32)[baked_slots])),
        "visible_objects_differ": int((fo["visible"] != fp["visible"]).sum()),
        "pass_triangles_differ": int((fo["pass"] != fp["pass"][:n]).sum()),
        "residual_triangles_differ": int((fo["residual"] != fp["residual"][:n]).sum()),
        "shadow_views_sets_differ": int(sum(int((so["visible"] != sp["visible"]).sum()) + int((so["pass"] != sp["pass"][: len(so["pass"])]).sum())
                                            for so, sp in zip(fo["shadows"], fp["shadows"]))),
        "visibility_keys_differ_px": int((fo["vis"] != fp["vis"]).sum()),
        "shadow_atlas_differ_texels": int((fo["atlas"].view(np.uint32) != fp["atlas"].view(np.uint32)).sum()),
        "hdr_f16_differ_px": int((fo["hdr16"] != fp["hdr16"]).any(axis=2).sum()),
        "framebuffer_max_abs": float(np.abs(fo["rgba_f32"] - fp["rgba_f32"]).max()),
        "rgba8_max_lsb": int(np.abs(fo["rgba8"].astype(np.int16) - fp["rgba8"].astype(np.int16)).max()),
        "visible_objects": int(fo["visible"].sum()), "pass_triangles": int(fo["pass"].sum()),
        "residual_triangles": int(fo["residual"].sum()), "covered_px": int((fo["vis"] != 0).sum()),
    }
    exact_keys = [k for k in out if k.endswith(("_differ", "_differ_px", "_differ_texels")) and not (fast and k == "hdr_f16_differ_px")]
    out["ok"] = bool(out["baked_matrices_equal"] and all(out[k] == 0 for k in exact_keys)
                     and out["framebuffer_max_abs"] <= 1e-3 and out["rgba8_max_lsb"] <= 1)
    if fast:
        out["note"] = "shade_mode fast: sets, keys and atlas bit-exact; the HDR target differs by rounding, the framebuffer is held to 1e-3"
    return out


def camera_path(hm, view0, step):
    """Dolly 0.15 m per frame down the street with a slow yaw sweep: view_k = Ry(yaw_k) * T(0,0,+d_k) * view0 (RH: forward = -Z)."""
    d = 0.15 * step
    yaw = 0.02 * math.sin(0.25 * step)
    return hm.mat4_mul(hm.mat4_mul(hm.rotation_y(yaw), hm.translation((0.0, 0.0, d))), view0)


def build_workload(args, r, hm, mk):
    """The benchmarked world on renderer `r` (the HIP renderer, or the oracle in the cpu_baseline leg): the synthetic stand-in of
    the named config, or -- `--scene` -- a real asset through the scene-viewer harness.  Returns the generator's info plus
    ambient / clear / workload description / data kind."""
    import rend3_amd.scenes as S
    if args.scene:
        from rend3_amd import scene_viewer as sv
        info = dict(sv.build(r, hm, mk, sv.settings_from(args)))
        info["data"] = "asset"
        info["workload"] = (f"scene viewer (examples/src/scene_viewer/mod.rs) on {info['file']}: {info['objects']} objects, {info['triangles']} triangles, "
                            f"{WIDTH}x{HEIGHT}, " + ("MSAA x4, " if args.msaa == 4 else "") + f"{len([l for l in r.dir_lights if l is not None])} directional shadow view(s), "
                            "camera dolly from the given camera")
        return info
    if args.config == 4:
        info = dict(S.emerald_like(r, hm, mk, n_lights=4, n_objects=getattr(args, "emerald_objects", 1 << 20)))
        info.update(ambient=AMBIENT, clear=CLEAR, data="synthetic",
                    workload="BASELINE.json configs[3] stand-in: emerald_like (seed 0xE5A0), 1 048 576 objects / 55 M triangles, 3840x2160, full PBR "
                             "opaque + 4 directional shadow views (2048^2), factor-only materials, camera dolly" + (", MSAA x4" if args.samples == 4 else ""))
        return info
    info = dict(S.bistro_like(r, hm, mk, n_objects=args.objects, target_tris=args.tris, textured=not args.untextured, unique=not args.instanced,
                              v2=getattr(args, "bistro_v2", False)))
    if getattr(args, "bistro_v2", False):
        info.update(ambient=AMBIENT, clear=CLEAR, data="synthetic",
                    workload="BASELINE.json configs[2] stand-in, Bistro-faithful variant: bistro_like v2 (seed 0xB157), 3840x2160, full PBR opaque + cutout + 4 "
                             "directional shadow views (2048^2), camera dolly; 24 texture sets of 2048^2 base colour + normal and 1024^2 AO/roughness/metallic maps "
                             "delivered as BC7 with stored mips (decoded at upload: ~1.2 GB texel pool), a fifth of the triangles alpha-tested foliage cards on the "
                             "cutout key, a third of the props instanced")
        return info
    info.update(ambient=AMBIENT, clear=CLEAR, data="synthetic",
                workload="BASELINE.json configs[2] stand-in: bistro_like (seed 0xB157, " + ("11 instanced meshes" if args.instanced else "every object owns its geometry")
                         + "), 3840x2160, full PBR opaque + 4 directional shadow views (2048^2), camera dolly, " + ("MSAA x4, " if args.samples == 4 else "")
                         + ("factor-only materials" if args.untextured else "130 materials with base colour + normal + AO/roughness/metallic maps (RGBA8, mips, trilinear)"))
    return info


XGMI_LINK_GBS = 153.0   # per direction and link, 7 links per GPU, fully connected inside a node
XGMI_EFFICIENCY = 0.7   # assumed fraction of the link rate a large RCCL message reaches; no multi-GPU box to measure it on
COLLECTIVE_LATENCY_MS = 0.02


def split_model(stage_ms, launches, world, width, height, samples, n_views, setup_frac=0.6):
    """Predicted per-rank GPU time of one frame (ms) under the two splits the library issues natively, from THIS GPU's stand-alone
    stage times of the unsharded frame (HIP events, single stream) and a direct-exchange model of the node's xGMI mesh
    (every peer pair has its own link: a rank receives (N - 1) shares concurrently).
      rows    (sort-first): the viewport's cull and per-triangle setup are REPLICATED, pixel work / N; depth bands all-gathered.
      objects (north_star): the viewport's cull, setup and pixel work / N; depth plane MAX all-reduce + key MAX reduce-scatter.
    Both: shadow views by view (ceil(V / N) per rank; with N > V a view's rows are split into N // V bands, one rank each: the
    view's culls and per-triangle setup replicated, its pixel work / bands -- r3n.hip shadow_parts) + broadcast, resolve / N, row
    gather.  setup_frac: share of the per-triangle pass that is per-triangle work (fetch + setup + work-item emission) rather than
    pixels -- an assumption, stated in the line."""
    per = lambda st: stage_ms[st] / max(launches.get(st, 0) or 1, 1)  # noqa: E731
    cams = 1 + n_views
    # the stage table sums every camera's launches: a camera's share = the per-launch average (2 viewport draws: predicted + residual)
    cull_vp = per("bake") * (1 if launches.get("bake") else 0) + per("object_cull") + per("triangle_cull")
    small_vp, big_vp = stage_ms["raster"] + stage_ms.get("raster_cut", 0.0), stage_ms["raster_big"] + stage_ms.get("raster_big_cut", 0.0)
    shadow_cull = (cams - 1) * (per("object_cull") + per("triangle_cull"))
    shadow = stage_ms["shadow_raster"] + stage_ms["shadow_raster_big"] + shadow_cull
    views_here = -(-n_views // world) if n_views else 0
    bands = world // n_views if n_views and world > n_views else 1  # rows of a view split over the ranks that would own none
    shadow_banded = shadow_cull + stage_ms["shadow_raster"] * (setup_frac + (1.0 - setup_frac) / bands) + stage_ms["shadow_raster_big"] / bands
    px = width * height
    link = XGMI_LINK_GBS * 1e9 * XGMI_EFFICIENCY
    share = lambda bytes_total: 1e3 * (bytes_total / world) / link + COLLECTIVE_LATENCY_MS  # noqa: E731  one band / shard per link, all links at once
    depth_bytes = px * (4 if samples == 1 else 8 * samples)
    common = (shadow_banded * views_here / n_views if n_views else 0.0) + (stage_ms["shade"] + stage_ms["vertex"]) / world + stage_ms["clear"] + stage_ms["hiz"] \
        + ((1e3 * views_here * 4.0 * 2048 * 2048 / bands / link + COLLECTIVE_LATENCY_MS) if n_views and world > 1 else 0.0) + (share(4.0 * px) if world > 1 else 0.0)
    rows = common + cull_vp + small_vp * (setup_frac + (1.0 - setup_frac) / world) + big_vp / world + (share(depth_bytes) if world > 1 else 0.0)
    objects = common + (cull_vp + small_vp + big_vp) / world + ((2.0 * share(depth_bytes) + share(8.0 * samples * px)) if world > 1 else 0.0)
    single = stage_ms["shade"] + stage_ms["vertex"] + stage_ms["clear"] + stage_ms["hiz"] + shadow + cull_vp + small_vp + big_vp
    return {"rows_ms": round(rows, 4), "objects_ms": round(objects, 4), "single_gpu_ms": round(single, 4),
            "predicted_speedup": {"rows": round(single / rows, 2), "objects": round(single / objects, 2)},
            "choice": "rows" if rows <= objects else "objects",
            "inputs_ms": {"viewport_cull": round(cull_vp, 4), "viewport_per_triangle_pass": round(small_vp, 4), "viewport_work_items": round(big_vp, 4),
                          "shadow_views_total": round(shadow, 4), "shadow_bands_per_view": bands, "resolve": round(stage_ms["shade"] + stage_ms["vertex"], 4)},
            "assumptions": f"stand-alone stage sums (no overlap between streams or frames), xGMI {XGMI_LINK_GBS:.0f} GB/s per link x {XGMI_EFFICIENCY} efficiency, "
                           f"direct exchange (one share per peer link), {COLLECTIVE_LATENCY_MS} ms per collective, setup_frac {setup_frac}; "
                           "nothing here was measured on more than one GPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--objects", type=int, default=3000)
    ap.add_argument("--tris", type=int, default=2_800_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-op-tally", action="store_true", help="skip the oracle's op tally of the resolve (oracle/tally.h; ~15 s of CPU)")
    ap.add_argument("--untextured", action="store_true",
                    help="factor-only materials (the round-1 workload before textures were built); default: every material "
                         "has base colour + normal + AO/roughness/metallic maps")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU exchange path (RCCL all-reduce / all-gather) even with one rank: plumbing check")
    ap.add_argument("--samples", type=int, default=1, choices=(1, 4),
                    help="SampleCount of the viewport targets (the reference's scene_viewer defaults to SampleCount::One, "
                         "examples/src/scene_viewer/mod.rs:317; --msaa 4 is its option); the headline metric is quoted at 1")
    ap.add_argument("--cpu-sample-frames", type=int, default=5, help="steady-state oracle frames timed for cpu_baseline (median)")
    ap.add_argument("--shade-mode", choices=("exact", "fast"), default="exact",
                    help="fragment-stage arithmetic: exact (default; bit-identical to the oracle) or fast (r3n_set_shade_mode(R3N_SHADE_FAST): "
                         "fused multiply-add + hardware rcp / rsqrt, framebuffer within 1e-3 after tonemap)")
    ap.add_argument("--bistro-v2", action="store_true",
                    help="the Bistro-faithful variant of the configs[2] stand-in (rend3_amd/scenes.py bistro_like v2): 2048^2 maps delivered as BC7 in 24 "
                         "texture sets, a fifth of the triangles alpha-tested foliage on the cutout key, a third of the props instanced")
    ap.add_argument("--instanced", action="store_true",
                    help="the round-1 stand-in: 3 000 objects instancing 11 shared meshes (1.5 MB of geometry, L2-resident) instead "
                         "of one mesh per object (~216 MB)")
    ap.add_argument("--config", type=int, default=3, choices=(3, 4),
                    help="3 (default): BASELINE.json configs[2] stand-in, the config the metric is quoted on; 4: configs[3] stand-in "
                         "(emerald_like, 1 048 576 objects / 55 M triangles, 4 shadow views), the workload whose per-rank work is milliseconds")
    ap.add_argument("--partition", choices=("auto", "rows", "objects", "spatial", "slots"), default="auto",
                    help="N > 1: how the viewport is sharded -- auto (default): rows or objects, whichever the printed cost model (split_model in "
                         "the line: replicated cull / setup against exchanged bytes over xGMI, from this rank's own stage times) predicts faster; "
                         "objects: BASELINE.json north_star's object-range split issued by the library itself (r3n_comm_set_split: contiguous slot "
                         "ranges balanced by triangles, MAX all-reduce of the pass-1 depth plane, MAX reduce-scatter of the pass-2 keys); "
                         "rows (sort-first): every rank culls and draws every object but rasterises "
                         "only its band of rows; the depth bands are all-gathered in front of Hi-Z and no keys are exchanged; slots: contiguous "
                         "object-slot ranges balanced by triangles, whole-target MAX collectives of depth (pass 1) and keys (pass 2); spatial: owner bytes from the Morton order of the bounding-sphere centres, with the pass-1 / pass-2 "
                         "exchanges limited to the rows inside each rank's conservative screen extent (pays only when partitions are compact on "
                         "screen: measured extents in DESIGN.md section 6)")
    ap.add_argument("--python-exchange", action="store_true",
                    help="--partition rows through rend3_amd/parallel.py's Exchange (torch.distributed calls from the frame's callbacks) instead "
                         "of the library's own RCCL calls (r3n_comm_init): A/B of the host cost")
    ap.add_argument("--scene", default=None, metavar="FILE.glb|FILE.gltf",
                    help="run the benchmark on a real glTF asset through the scene-viewer harness (rend3_amd/scene_viewer.py; its flags "
                         "below; `data` becomes \"asset\"): hand it Bistro.glb with tools/scene_viewer.py's --bistro flags and the line is "
                         "the reference's own configs[2]")
    ap.add_argument("--resolution", default=None, help="WxH of the target (default 3840x2160, the metric's)")
    from rend3_amd import scene_viewer as sv
    sv.add_arguments(ap)
    args = ap.parse_args(sv.normalize_argv(sys.argv[1:]))
    if args.resolution:
        global WIDTH, HEIGHT
        WIDTH, HEIGHT = (int(v) for v in args.resolution.lower().split("x"))
    if args.scene:
        args.samples = args.msaa

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    # R3N_BENCH_SHARE_GPU=1 (tests/test_zzz_bench_ranks_gpu.py on the one-GPU boxes): every rank on cuda:0, torch's collectives over gloo
    # and the library's over the library R3N_RCCL_LIB names (RCCL refuses two ranks on one device) -- exercises this file's N > 1
    # branches (cost-model probe, split choice, native exchanges, max-over-ranks timing); its numbers mean nothing
    share_gpu = os.environ.get("R3N_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or args.force_exchange
    import rend3_amd as r3
    import rend3_amd.scenes  # noqa: F401
    from rend3_amd import parallel

    # ---------------------------------------------------------------- scene: replicated on every rank
    # the context comes FIRST: its streams must take their hardware queues before the communication library creates its own
    # (rend3_amd/csrc/r3n.hip r3n_create; 1.19 vs 1.65 ms per frame at N = 1 with the exchange forced, profiles/r02_summary.md)
    r = r3.Renderer(r3.host.RIGHT, np.float32(WIDTH) / np.float32(HEIGHT), device=local_rank)
    if distributed:
        if "MASTER_ADDR" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    info = build_workload(args, r, r3.host, r3.material_record)
    view0 = info["camera"][0]
    ambient, clear = info["ambient"], info["clear"]
    if args.shade_mode == "fast":
        r.set_shade_mode(1)
    hbm_measured = r.hbm_copy_rate(1 << 30, 5)
    exchange, native_comm, comm_note, split = None, False, None, None
    if distributed:
        r.evaluate_instructions()  # flush the world: the object buffer's capacity is final
        counts = np.zeros(r.capacity, dtype=np.int64)
        spheres = np.zeros((r.capacity, 4), dtype=np.float64)
        for h, m in r.object_meta.items():
            counts[h] = r.meshes[m["mesh"]].index_count // 3
            spheres[h] = m["sphere"]
        rows = parallel.row_ranges(HEIGHT, world)
        if args.partition == "auto":
            # a few unsharded frames on this rank with stage timing: the inputs of the cost model; rank 0's choice is everybody's
            r.set_multi_stream(False)
            r.timing_enable(True)
            for k in range(4):
                r.set_camera_data(camera_path(r3.host, view0, k), info["camera"][1])
                r.render(WIDTH, HEIGHT, samples=args.samples, ambient=ambient, clear_color=clear, readback=False)
                if k == 0:
                    r.sync()
                    r.stage_times(reset=True)  # the first frame allocates
            r.sync()
            probe = r.stage_times(reset=True)
            r.timing_enable(False)
            r.set_multi_stream(True)
            n_views_probe = len([l for l in r.dir_lights if l is not None])
            split = split_model({st: ms / 3 for st, (ms, _n) in probe.items()}, {st: n / 3 for st, (_ms, n) in probe.items()}, world, WIDTH, HEIGHT,
                                args.samples, n_views_probe)
            box = [split]
            dist.broadcast_object_list(box, src=0)
            split = box[0]
            args.partition = split["choice"]
        native_comm = args.partition in ("rows", "objects") and not args.python_exchange
        if native_comm:
            # the library issues the exchanges itself (r3n_comm_init): no Exchange object, no torch collective on the frame path
            try:
                r.comm_init_torch()
            except Exception as exc:  # noqa: BLE001  (the same on every rank: RCCL not loadable / an id that cannot be shared) -> the Python exchange
                native_comm, comm_note = False, f"r3n_comm_init failed ({exc}); exchanges through torch.distributed instead"
                args.python_exchange = True
        if not native_comm:
            exchange = parallel.Exchange(r, device)
            exchange.rows_equal = HEIGHT % world == 0
        if native_comm:
            if args.partition == "objects":  # the library's own object-range exchanges (r3n_comm_set_split)
                r.comm_set_split(True)
                begin, end = parallel.partition_objects(counts, world)[rank]
                r.set_object_range(begin, end)
        elif args.partition == "rows":
            exchange.set_row_sharding(rows[rank][0], rows[rank][1])
        elif args.partition == "spatial" and exchange.rows_equal and args.samples == 1:
            owners = parallel.partition_objects_spatial(spheres[:, :3], counts, world)
            exchange.set_spatial_partition(owners, parallel.partition_bounds(owners, spheres[:, :3], spheres[:, 3], counts, world))
        else:
            begin, end = parallel.partition_objects(counts, world)[rank]
            r.set_object_range(begin, end)
        if not native_comm:
            r._check(r.lib.r3n_set_row_range(r.ctx, rows[rank][0], rows[rank][1]), "r3n_set_row_range")

    base = r3.BaseRenderGraph(r)

    # the camera path is an input of the workload: its view matrices are computed up front, a frame hands one to the renderer
    n_views = 2 + args.warmup + args.steps + min(args.steps, 20) + 1
    views = [camera_path(r3.host, view0, k) for k in range(n_views)]

    def frame(k, readback=False):
        r.set_camera_data(views[k], info["camera"][1])
        out = r.render(WIDTH, HEIGHT, samples=args.samples, ambient=ambient, clear_color=clear, readback=readback, base=base, exchange=exchange)
        if exchange is not None:
            exchange.gather_rows(WIDTH, HEIGHT, world)
        return out

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- parity frames (N=1): camera steps 0 and 1, read-back of step 1
    check = world == 1 and not args.no_cpu_baseline and not args.force_exchange
    hip_frame1 = None
    step0 = 0
    if check:
        frame(0)
        hip_frame1 = frame(1, readback=True)
        step0 = 2
    # ---------------------------------------------------------------- warmup + timed region
    for k in range(args.warmup):
        frame(step0 + k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        frame(step0 + args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------------------------------------------------------- host cost of a frame (outside the timed region)
    # In the timed loop the host runs ahead of the GPU until it waits for a pinned frame image still in flight, so its enqueue time
    # converges to the GPU's frame time.  Its OWN time per frame (Python glue + r3n_host_evaluate_frame + r3n_render_frame + the HIP
    # runtime's launches): bursts of 3 frames right after a full synchronisation, median.
    host_ms = None
    if True:  # with the exchange too: the collectives' enqueue calls are part of the host's frame
        bursts, k0 = [], step0 + args.warmup
        for b in range(12):
            r.sync()
            tb = time.perf_counter()
            for j in range(3):
                frame(k0 + (3 * b + j) % max(args.steps, 1))
            bursts.append((time.perf_counter() - tb) / 3)
        r.sync()
        host_ms = round(1e3 * sorted(bursts)[len(bursts) // 2], 4)

    # ---------------------------------------------------------------- instrumented pass (outside the timed region)
    # HIP events on the context's stream around every kernel launch of each stage (include/r3n.h R3N_STAGE_*)
    # (single-stream here: a kernel's duration is inflated while kernels of other streams are co-resident)
    r.sync()
    r.set_multi_stream(False)
    r.timing_enable(True)
    r.stage_times(reset=True)
    n_inst = min(args.steps, 20)
    if exchange is not None:
        exchange.timed = True
    for k in range(n_inst):
        frame(step0 + args.warmup + args.steps + k)
    r.sync()
    stages = r.stage_times(reset=True)
    r.timing_enable(False)
    import ctypes
    span_overhead = ctypes.c_double(0.0)
    r._check(r.lib.r3n_timing_overhead(r.ctx, ctypes.byref(span_overhead)), "r3n_timing_overhead")
    exchange_ms, exchange_bytes = None, None
    if exchange is not None:  # HIP events on the context's stream around every collective of the instrumented frames
        exchange.timed = False
        exchange_ms = {k: round(v / n_inst, 4) for k, v in exchange.drain_timings().items()}
        exchange_bytes = dict(exchange.bytes)
    elif native_comm:  # the library's own stage events (R3N_STAGE_EXCHANGE_*)
        by_objects = args.partition == "objects"
        exchange_ms = {"shadow": round(stages["exchange_shadow"][0] / n_inst, 4), "pass1": round(stages["exchange_depth"][0] / n_inst, 4),
                       "pass2": round(stages["exchange_keys"][0] / n_inst, 4), "rows": round(stages["exchange_rows"][0] / n_inst, 4)}
        exchange_bytes = {"shadow": 4 * 4 * 2048 * 2048 if not args.scene else None,
                          "pass1": (4 if args.samples == 1 else 8 * args.samples) * WIDTH * HEIGHT // (1 if by_objects else world),
                          "pass2": 8 * args.samples * WIDTH * HEIGHT if by_objects else 0, "rows": 4 * WIDTH * HEIGHT // world}
    r.set_multi_stream(True)
    last = frame(step0 + args.warmup + args.steps + n_inst, readback=(world == 1))

    result = None
    # HBM bytes / VALU instructions per launch from the PMC passes (collected in their own rocprofv3 runs, tools/profile_round.sh
    # + tools/make_traffic.py).  Quoted only when they were taken on these kernel sources and this workload variant.
    traffic, valu_busy, valu_insts, useful_flops, traffic_note = {}, {}, {}, {}, "no PMC pass on record"
    variant = ("instanced" if args.instanced else "unique") + ("-untextured" if args.untextured else "-textured") + f"-s{args.samples}" + \
              ("-fast" if args.shade_mode == "fast" else "") + ("-cfg4" if args.config == 4 else "") + ("-v2" if getattr(args, "bistro_v2", False) else "") + \
              ("-scene:" + os.path.basename(args.scene) if args.scene else "") + ("" if (WIDTH, HEIGHT) == (3840, 2160) else f"-{WIDTH}x{HEIGHT}")
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            committed = json.load(fh)
        entry = committed.get("variants", {}).get(variant)  # one entry per workload variant (tools/make_traffic.py)
        if committed.get("kernel_sources_sha") != kernel_sources_sha():
            traffic_note = "stale: PMC passes were taken on other kernel sources (" + str(committed.get("kernel_sources_sha")) + ")"
        elif entry is None:
            traffic_note = "PMC passes on record are for workload variants " + ", ".join(sorted(committed.get("variants", {}))) + "; this is " + variant
        else:
            traffic = entry.get("bytes_per_launch", {})
            valu_busy = entry.get("valu_busy", {})
            valu_insts = entry.get("valu_insts_per_launch", {})
            useful_flops = entry.get("useful_flops_per_launch", {})
            traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* passes of " + str(entry.get("taken", "?")) + ", same kernel sources, workload variant " + variant
    except (OSError, ValueError):
        pass
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        mpix = WIDTH * HEIGHT * args.steps / elapsed / 1e6
        cameras = 1 + len([l for l in r.dir_lights if l is not None])
        stage_ms = {s: (ms / n_inst) for s, (ms, _n) in stages.items()}
        launches = {s: n / n_inst for s, (_ms, n) in stages.items()}
        cull_ms = stage_ms["bake"] + stage_ms["object_cull"] + stage_ms["triangle_cull"]
        tri_cull_ms_per_launch = stage_ms["triangle_cull"] / max(launches["triangle_cull"], 1)
        # roofline objects: algorithmic bytes per launch (SURVEY.md section 8d / BASELINE.md section 5: the minimum traffic the
        # inputs and outputs imply, not this implementation's) / average launch duration from HIP events recorded on the
        # context's stream around that kernel's launches (instrumented pass above, single stream).
        rooflines = {}

        def roofline_of(kernel, stage, bytes_per_launch, traffic_key, note, extra=None):
            ms = stage_ms[stage] / max(launches[stage], 1)
            # Stage figures are HIP-event spans minus the span of an EMPTY launch (stage_timing.span_overhead_ms): right for
            # kernels of hundreds of microseconds, but that span over-corrects a kernel of ~20 us (round 4: the triangle cull at
            # 17.8 us here against 20.7 in the rocprofv3 kernel trace -> a fraction of 0.53 instead of 0.46).  A launch under 50 us
            # is therefore priced at its UNCORRECTED span: the fraction can only be understated.
            if 0.0 < ms < 0.05:
                ms += span_overhead.value
                note += "; launch under 50 us: duration = the uncorrected HIP-event span (no empty-launch correction), so the fraction is a lower bound"
            ach = bytes_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            d = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(traffic_key) if world == 1 else None,
                 "peak_measured": round(hbm_measured, 1), "frac_of_measured_peak": round(ach / hbm_measured, 5) if hbm_measured else None,
                 "bytes_per_launch": int(bytes_per_launch), "ms_per_launch": round(ms, 5), "ms_per_frame": round(stage_ms[stage], 5),
                 "traffic_source": traffic_note, "note": note}
            vi = valu_insts.get(traffic_key) if world == 1 else None
            if vi:  # SQ_INSTS_VALU wave-instructions per launch x 64 lanes / launch time against the vector peak: the peak counts a
                # packed FMA as 4 flop per lane-slot, a plain (non-packed, non-fused) f32 op is 1 -- so 0.25 is the ceiling of such code
                lane_ops = vi * 64.0 / (ms * 1e-3) / 1e12
                d["valu"] = {"insts_per_launch": int(vi), "lane_ops_T_per_s": round(lane_ops, 2), "peak_tflops": VALU_PEAK_TFLOPS,
                             "issue_frac_of_vector_peak": round(lane_ops / VALU_PEAK_TFLOPS, 4), "busy": valu_busy.get(traffic_key),
                             "note": "issue_frac counts EVERY vector instruction (moves, selects, address math) as one lane-op; useful_* = 64 x "
                                     "SQ_INSTS_VALU_FLOPS_FP32: f32 add / sub / mul / transcendental 1, fma 2, packed forms twice that, everything else "
                                     "(min / max, floor, compares, division fix-ups, moves, integer, conversions) 0 -- weights calibrated per "
                                     "instruction on the box (profiles/r05_valu_counter_probe.txt)"}
                uf = useful_flops.get(traffic_key)
                if uf:
                    d["valu"]["useful_tflops"] = round(uf / (ms * 1e-3) / 1e12, 2)
                    d["valu"]["useful_frac_of_vector_peak"] = round(uf / (ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)
            # counted traffic BELOW the algorithmic bytes: the kernel's inputs are served from the Infinity Cache (the frame's
            # working set fits its 256 MB), so `achieved` is an on-die rate, not HBM bandwidth -- say so instead of calling it HBM
            tr = d["traffic"]
            if tr is not None and tr < 0.8 * bytes_per_launch:
                d["cache_served"] = True
                d["note"] += (f"; counted memory-side traffic ({tr / 1e6:.1f} MB) is below the algorithmic bytes ({bytes_per_launch / 1e6:.1f} MB): the inputs come "
                              "from the Infinity Cache, `achieved` is algorithmic bytes / time, not HBM bandwidth")
            if extra:
                d.update(extra)
            rooflines[stage] = d
            return d

        if last is not None:
            vis_tris = 0
            per_launch = []
            counts = np.zeros(r.capacity, dtype=np.int64)
            for h, m in r.object_meta.items():
                counts[h] = r.meshes[m["mesh"]].index_count // 3
            cams = [last] + last["shadows"]
            for c in cams:
                t_in = int(counts[c["visible"].astype(bool)].sum())
                n_pass = int(c["pass"].sum())
                n_new = int(c["residual"].sum())
                # triangle cull: 48 B in + 12 B * pass + 12 B * new + 0.25 B out per triangle slot the launch processes
                per_launch.append(48.0 * t_in + 12.0 * n_pass + 12.0 * n_new + 0.25 * t_in)
                vis_tris += t_in
            roofline_of("k_triangle_cull", "triangle_cull", float(np.mean(per_launch)), "k_triangle_cull",
                        "48 B in + 12 B per passing + 12 B per newly visible + 0.25 B out per triangle of the frustum-visible objects, mean over the 5 cameras",
                        {"triangles_per_launch": vis_tris // len(cams)})
            # shadow views: <= 72 B per drawn triangle (8 B list entry + 12 B indices + 36 B positions, + setup hand-over) and 4 B per
            # covered atlas texel (the final depth: overdraw is this implementation's, not the algorithm's)
            drawn = float(np.mean([int(c["pass"].sum()) for c in last["shadows"]])) if last["shadows"] else 0.0
            covered = float((last["atlas"] != 0).sum()) / max(len(last["shadows"]), 1)
            for stage, kname in (("shadow_raster", "k_raster_small<depth>"), ("shadow_raster_big", "k_raster_big<depth>")):
                if launches.get(stage):
                    roofline_of(kname, stage, 56.0 * drawn + 4.0 * covered, kname,
                                "per shadow view: 56 B per drawn triangle + 4 B per covered texel, charged to each of the view's raster launches "
                                "(both walk the same triangles / texels between them)", {"drawn_triangles": int(drawn), "covered_texels": int(covered)})
            # viewport rasterisers (opaque.wgsl:91-135 + the depth test; VERDICT r5 missing #3): the two passes of a frame between them
            # read every drawn triangle once per pass it is drawn in (56 B: list entry + indices + positions + setup hand-over) and
            # leave one 8-B key per covered sample.  Predicted pass: last frame's passing triangles (this frame's count stands in for
            # it: the camera moves by a fraction of a pixel per frame), residual pass: the newly visible ones.  Charged per key
            # (opaque / cutout launches are separate stages) by the key's share of the drawn triangles.
            tri_base = np.concatenate([[0], np.cumsum(counts)])[:-1]  # canonical slot base per object (exclusive scan of the triangle counts)
            tri_obj = np.searchsorted(tri_base, np.arange(len(last["pass"])), side="right") - 1
            mkeys = np.asarray([k for _rec, k in r.materials], dtype=np.uint8)
            okeys = np.zeros(r.capacity, dtype=np.uint8)
            for h, m in r.object_meta.items():
                okeys[h] = mkeys[m["material"]]
            tkeys = okeys[np.clip(tri_obj, 0, r.capacity - 1)]
            covered_samples = float((last["vis"] != 0).sum())
            drawn_by_key = [float(((last["pass"] != 0) & (tkeys == k)).sum() + ((last["residual"] != 0) & (tkeys == k)).sum()) for k in (0, 1)]
            drawn_all = max(sum(drawn_by_key), 1.0)
            for k, (st_small, st_big, tag) in enumerate((("raster", "raster_big", ""), ("raster_cut", "raster_big_cut", ",cutout"))):
                for stage, kname in ((st_small, f"k_raster_small<vis{tag}>"), (st_big, f"k_raster_big<vis{tag}>")):
                    if launches.get(stage) and drawn_by_key[k]:
                        per_frame = 56.0 * drawn_by_key[k] + 8.0 * covered_samples * drawn_by_key[k] / drawn_all
                        roofline_of(kname, stage, per_frame / launches[stage], kname,
                                    "viewport, " + ("cutout" if k else "opaque") + " key: 56 B per drawn triangle (predicted + residual passes) + 8 B per covered sample "
                                    "(this key's share of the drawn triangles), per frame / launches of the stage; both kernels of a pass walk the same "
                                    "triangles / pixels between them" + ("; the alpha test samples the albedo map per fragment (opaque.wgsl:207-235): texels excluded" if k else ""),
                                    {"drawn_triangles_per_frame": int(drawn_by_key[k]), "covered_samples": int(covered_samples)})
            # object pass (uniform_prep.wgsl + batching.rs:144-148 on the GPU): per slot 20 B of the SoA view + 1 B flag + 4 B slot base,
            # per visible object 8 B list entry, per baked slot 64 B transform in + 128 B matrices out; mean over the cameras
            if launches.get("object_cull"):
                vis_mean = float(np.mean([int(c["visible"].astype(bool).sum()) for c in cams]))
                roofline_of("k_object_pass_chained", "object_cull", 25.0 * r.capacity + (8.0 + 192.0) * vis_mean, "k_object_pass_chained",
                            "per camera: 25 B per object slot (sphere + meta read, flag + slot base written) + 200 B per frustum-visible object (list entry, "
                            "transform read, two matrices written); includes the fused uniform bake", {"slots": int(r.capacity), "visible_objects": int(vis_mean)})
            # Hi-Z (hi_z.rs:161-234): the keys' depth words read (8 B per sample), mip 0 written (4 B per pixel), the chain above it 4/3 B per pixel
            if launches.get("hiz"):
                hz = rooflines_hiz = (8.0 * args.samples + 4.0 + 4.0 / 3.0) * WIDTH * HEIGHT
                d = roofline_of("k_hiz_head + k_hiz_tail", "hiz", hz / launches["hiz"], "k_hiz_head",
                                "per frame: 8 B per key sample read + 4 B per pixel of mip 0 + 4/3 B per pixel of the levels above, over the stage's launches "
                                "(head: the first levels in one pass; tail: the last levels in one workgroup, latency-bound by construction)")
                d["bytes_per_frame"] = int(hz)
        # the deferred PBR resolve (one launch per frame).  HBM floor: 8 B key read + 8 B Rgba16Float write per pixel (BASELINE.md
        # section 5: ">= 16 B / shaded pixel") + 4 B for the Rgba8UnormSrgb blit fused into it.  VALU-bound, not HBM-bound, so its
        # fraction of the HBM roofline is small by construction; `valu` says how close it is to its own bound.
        roofline_of("k_resolve_opaque", "shade", 20.0 * WIDTH * HEIGHT / world, "k_resolve_opaque",
                    f"8 B key + 8 B HDR + 4 B sRGB per pixel, texels / triangle records / shadow texels excluded; per pixel: {cameras - 1} lights x "
                    "(5-tap PCF + GGX)" + ("" if args.untextured else ", 3 trilinear maps, tangent frame"))
        # the resolve is bound by vector issue, not by HBM: its roofline is the f32 vector peak, `frac` = useful flops (64 x SQ_INSTS_VALU_FLOPS_FP32:
        # add / mul / transcendental 1, fma 2, packed forms twice that; calibrated by tools/valu_counter_probe.hip) / launch time / 157.3 TFLOP/s; the HBM figure stays beside it
        rs = rooflines["shade"]
        if rs.get("valu", {}).get("useful_tflops"):
            rs.update({"bound": "valu", "frac_hbm": rs["frac"], "achieved_hbm_GBps": rs["achieved"], "peak_hbm_GBps": rs["peak"],
                       "achieved": rs["valu"]["useful_tflops"], "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": rs["valu"]["useful_frac_of_vector_peak"],
                       "frac_note": "useful f32 flops / vector peak; every vector instruction counted as one lane-op gives valu.issue_frac_of_vector_peak "
                                    "(0.25 is the ceiling of unpacked, unfused f32 code), the ALUs are busy valu.busy of the launch",
                       "limiter": "measured (profiles/r04_summary.md section 6b, profiles/r05_paired_texel_loads.txt): neither roofline binds.  "
                                  "10 % fewer vector instructions, one round trip less in the chain key -> record -> material -> descriptors -> "
                                  "texels -> shadow lookups, or an occupancy cap move the launch by <= 1 %; a third fewer vector-MEMORY "
                                  "instructions (footprint rows as one 8-byte load, round 5) moved it by 7 %, as the PCF's twelve loads -> four "
                                  "had in round 3: the kernel is short of the rate at which the vector L1 takes memory instructions, at four "
                                  "waves per SIMD.  `bound` keeps the label of the nearer roofline (the HBM figure is `frac_hbm`)"})
        else:
            rs["bound_note"] = "VALU-bound kernel (no PMC pass of these sources on record: only the HBM figure can be quoted)"
        # Every other stage against the bound it has: a stage whose counted traffic is below its algorithmic
        # bytes is served by the Infinity Cache -- `bound` says so, `achieved` stays an on-die rate; a stage whose vector ALUs are
        # busy >= 75 % of the launch is vector-issue bound: `frac` = useful f32 flops / vector peak, the HBM figure beside it.
        for st_name, rr in rooflines.items():
            if st_name == "shade":
                continue
            vv = rr.get("valu") or {}
            if rr.get("cache_served"):
                rr["bound"] = "cache"
                rr["bound_note"] = "inputs served by the 256-MB Infinity Cache (counted memory-side traffic below the algorithmic bytes): `achieved` is algorithmic bytes / time, not HBM bandwidth"
            elif (vv.get("busy") or 0.0) >= 0.75 and vv.get("useful_tflops"):
                rr.update({"bound": "valu", "frac_hbm": rr["frac"], "achieved_hbm_GBps": rr["achieved"], "peak_hbm_GBps": rr["peak"],
                           "achieved": vv["useful_tflops"], "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": vv["useful_frac_of_vector_peak"],
                           "frac_note": f"vector ALUs busy {vv['busy']:.2f} of the launch (SQ_ACTIVE_INST_VALU): bound by vector issue; frac = executed f32 flops "
                                        "(64 x SQ_INSTS_VALU_FLOPS_FP32) / launch time / 157.3 TFLOP/s -- low because most of the instructions are not "
                                        "flops (edge tests, address and mask arithmetic, conversions, exact-division sequences)"})
            elif vv.get("busy") is not None:
                rr["bound_note"] = (f"neither roofline binds: vector ALUs busy {vv['busy']:.2f}, HBM fraction {rr['frac']:.3f}; the rasterisers are bound by "
                                    "instruction issue (scalar + vector) and the rate of memory-side atomics (profiles/r06_summary.md section 2: with every "
                                    "atomic ablated the shadow work-item launch keeps two thirds of its time)")
        dominant = max(rooflines, key=lambda st: stage_ms[st])
        roof = dict(rooflines[dominant], dominant_by="largest kernel time per frame in the HIP-event stage table")
        # the whole frame against the HBM roofline: the sum of the stages' algorithmic bytes per frame over the frame time
        frame_bytes = sum(v["bytes_per_launch"] * max(launches.get(k, 0), 0) for k, v in rooflines.items())
        obj_bytes = (196.0 + 20.0 + 20.0) * info["objects"] * cameras          # bake + object cull + IndirectCall
        px_bytes = (5.33 + 8.0) * WIDTH * HEIGHT                               # Hi-Z + the resolve pre-pass's key read
        frame_bytes += obj_bytes + px_bytes
        frame_roofline = {"bound": "hbm", "bytes_per_frame": int(frame_bytes), "achieved": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "sum over the frame of every stage's algorithmic bytes (triangle culls, shadow rasterisers, resolve, object passes, Hi-Z) / "
                                  "ms_per_step; the frame is not one streaming kernel -- its largest stage is VALU-bound, the geometry stages issue- / latency-bound"}
        result = {
            "metric": "shaded Mpixels/s @4K (whole frame: cull+compact all cameras, 4 shadow views, PBR opaque, tonemap)",
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": info["data"],
            "config": {"workload": info["workload"],
                       "shade_mode": args.shade_mode,
                       "objects": info["objects"], "triangles": info["triangles"], "cameras": cameras,
                       "split_model": split,
                       "parallelism": "single GPU" if world == 1 else (
                           f"object-range split issued by the library (r3n_comm_init + r3n_comm_set_split): viewport objects by contiguous slot range balanced by "
                           f"triangles x{world}, shadow views by view (broadcast on a shadow lane's stream), RCCL MAX all-reduce of the pass-1 depth plane in front "
                           f"of Hi-Z, MAX reduce-scatter of the pass-2 keys onto the bands of {HEIGHT // world} rows, image rows all-gathered"
                           + ("" if split is None else f"; chosen by the cost model: predicted {split['predicted_speedup']['objects']}x against {split['predicted_speedup']['rows']}x for rows")
                           if native_comm and args.partition == "objects" else
                           f"sort-first: every rank culls + draws every object into its band of {HEIGHT // world} rows (x{world}), shadow views by view "
                           "(broadcast on a shadow lane's stream), pass-1 depth bands all-gathered over RCCL in front of Hi-Z, no key exchange, image rows all-gathered"
                           + (" -- exchanges issued by the library itself (r3n_comm_init)" if native_comm else " -- exchanges through torch.distributed")
                           + ("" if split is None else f"; chosen by the cost model: predicted {split['predicted_speedup']['rows']}x against {split['predicted_speedup']['objects']}x for objects")
                           if native_comm or (exchange is not None and exchange.by_rows) else
                           f"viewport objects by spatial partition (Morton order, owner bytes) x{world}, shadow views by view (broadcast), pass-1 depth and pass-2 keys "
                           "MAX-reduced onto the row-band owners over RCCL all-to-all limited to each rank's screen-row extent, depth bands + image rows all-gathered"
                           if exchange is not None and exchange.sparse is not None else
                           f"viewport objects by slot range x{world}, shadow views by view (broadcast), RCCL MAX all-reduce of the pass-1 depth plane, MAX reduce-scatter of "
                           "the pass-2 keys, row all-gather")},
            "fps": round(args.steps / elapsed, 2),
            "host_ms_per_frame": host_ms,
            "culled_objects_per_s": round(info["objects"] * cameras / (cull_ms * 1e-3), 1) if cull_ms > 0 else None,
            "culled_mtris_per_s": round(info["triangles"] * cameras / (cull_ms * 1e-3) / 1e6, 1) if cull_ms > 0 else None,
            "stage_ms_per_frame": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_launches_per_frame": launches,
            "stage_timing": {"span_overhead_ms": round(span_overhead.value, 5),
                             "note": "HIP events on the stream around every stage's launches, single stream, outside the timed region; the span of an "
                                     "EMPTY launch between two events (measured on this box, above) is taken off every span, so the figures are kernel "
                                     "time as rocprofv3's kernel trace reports it (profiles/r04_bench_kernel_stats_single_stream.csv)"},
            "roofline": roof,
            "rooflines": {k: v for k, v in rooflines.items() if k != dominant},
            "frame_roofline": frame_roofline,
            "hbm_copy_rate_measured_GBps": round(hbm_measured, 1),
            "exchange_ms_per_frame": exchange_ms,
            "exchange_bytes_per_frame": exchange_bytes,
            "exchange_note": comm_note,
            "mesh_buffer_bytes": info.get("mesh_bytes"), "unique_triangles": info.get("unique_triangles"),
        }

    # ---------------------------------------------------------------- CPU baseline: the oracle ("port"), rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, parity = cpu_baseline(args, hip_frame1)
        result["cpu_baseline"] = cb
        result["parity"] = parity
        result["vs_cpu_baseline"] = round(result["value"] / cb["value"], 1) if cb["value"] else None
    # ---------------------------------------------------------------- the oracle's op tally beside the resolve's counter figure (N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_op_tally and not args.scene:
        try:
            ot = op_tally(args)
        except Exception as e:  # noqa: BLE001  (checker-side extra: never costs the line)
            ot = {"error": repr(e)}
        rs = result["roofline"] if result["roofline"].get("kernel") == "k_resolve_opaque" else result["rooflines"].get("shade")
        if rs is not None and ot and "algorithmic_flops_per_shaded_pixel" in ot:
            ms = rs["ms_per_launch"]
            px = WIDTH * HEIGHT * (args.samples if args.samples > 1 else 1)
            tf = ot["algorithmic_flops_per_shaded_pixel"] * px / (ms * 1e-3) / 1e12
            ot["tflops_at_this_launch"] = round(tf, 2)
            ot["frac_of_vector_peak"] = round(tf / VALU_PEAK_TFLOPS, 4)
            ot["note"] = ("fs_main's f32 flops as the ORACLE executes them x every pixel of the target / the resolve's launch time / the vector peak.  An upper "
                          "estimate of the algorithm's minimum (the oracle recomputes the mip footprint for each of a material's maps and filters every "
                          "texel in f32; background pixels are charged like shaded ones); the HIP kernel's executed flops are roofline.valu.useful_tflops")
        if rs is not None:
            rs["algorithmic"] = ot
    if rank == 0:
        print(json.dumps(result))
    r.close()
    if distributed:
        dist.destroy_process_group()


def op_tally(args):
    """"exact count from the oracle's op tally" -- the ALGORITHMIC f32 flops of the resolve per shaded pixel,
    counted by the oracle itself (oracle/tally.h: r3o.c compiled with a counting f32, bit-identical frames), on the workload's own
    materials / lights / texture classes with a cut-down geometry (the count per pixel depends on the material class and the
    light count, not on how many triangles there are) at 480x270.  Single-threaded: a few seconds.  CHECKER-SIDE code: the figure
    stands beside the HIP kernel's counter figure in `roofline`, nothing of the product runs through it."""
    import copy
    import numpy as np
    from oracle import host as oh
    from oracle import lib as olib
    from oracle.world import OracleRenderer, material_record as omk
    t0 = time.perf_counter()
    a2 = copy.copy(args)
    a2.objects, a2.tris, a2.emerald_objects = min(args.objects, 250), min(args.tris, 80000), 30000
    w, h = 480, 270
    o = OracleRenderer(oh.RIGHT, np.float32(w) / np.float32(h), lib=olib.OracleLib(tally=True))
    info_o = build_workload(a2, o, oh, omk)
    o.set_camera_data(camera_path(oh, info_o["camera"][0], 1), info_o["camera"][1])
    fo = o.render(w, h, samples=args.samples, ambient=info_o["ambient"], clear_color=info_o["clear"])
    shaded = int((fo["vis"] != 0).sum())
    t = o.stage_tally.get("shade", {})
    if not shaded or not t:
        return None
    fs, vs, ff = t["fs_main"], t["vs_main"], t["fixed_function"]
    return {"algorithmic_flops_per_shaded_pixel": round(fs["flops"] / shaded, 1),
            "fs_main_per_shaded_pixel": {k: round(v / shaded, 2) for k, v in fs.items()},
            "vs_main_flops_per_vertex": round(vs["flops"] / (3.0 * shaded), 1),
            "fixed_function_flops_per_shaded_pixel": round(ff["flops"] / shaded, 1),
            "shaded_samples": shaded, "seconds": round(time.perf_counter() - t0, 1),
            "source": "oracle/tally.h: the oracle's own restatement of opaque.wgsl:203-551 + math/brdf.wgsl + shadow/pcf.wgsl with every f32 operation "
                      "counted (add / sub / mul / div / sqrt / pow 1, fmaf 2; min / max / floor / compares / conversions 0), on the workload's materials and "
                      f"lights, geometry cut down to {info_o['objects']} objects, {w}x{h}.  algorithmic_flops_per_shaded_pixel = fs_main only (texture "
                      "filtering arithmetic included); the oracle redoes vs_main (three vertices) and the triangle setup + attribute interpolation for "
                      "every pixel: those are counted apart (vs_main per vertex; fixed-function work per pixel) and are NOT in the figure"}


def cpu_baseline(args, hip_frame1):
    """The C oracle (oracle/r3o.c; OpenMP over objects / triangles / rows in every stage) on the host cores, on a bounded sample
    of the same workload: the same scene and camera path, camera step 0 as the history frame, then `--cpu-sample-frames` timed
    steady-state frames at 4K -- median frame time and the stage split of BASELINE.md section 3.  kind "port": the reference
    itself (Rust + wgpu on lavapipe) cannot be built here.  The frame of camera step 1 is also what the
    HIP path's read-back of the same step is checked against (`parity`)."""
    import numpy as np
    from oracle import host as oh
    from oracle.world import OracleRenderer, material_record as omk
    import rend3_amd as r3
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    o = OracleRenderer(oh.RIGHT, np.float32(WIDTH) / np.float32(HEIGHT))
    info_o = build_workload(args, o, oh, omk)
    view0 = info_o["camera"][0]
    AMBIENT, CLEAR = info_o["ambient"], info_o["clear"]
    o.set_camera_data(camera_path(oh, view0, 0), info_o["camera"][1])
    o.render(WIDTH, HEIGHT, samples=args.samples, ambient=AMBIENT, clear_color=CLEAR)  # history frame (untimed)
    times, splits, parity = [], [], None
    for k in range(max(args.cpu_sample_frames, 1)):
        o.set_camera_data(camera_path(oh, view0, 1 + k), info_o["camera"][1])
        t0 = time.perf_counter()
        fo = o.render(WIDTH, HEIGHT, samples=args.samples, ambient=AMBIENT, clear_color=CLEAR)
        times.append(time.perf_counter() - t0)
        splits.append(dict(o.stage_s))
        if k == 0 and hip_frame1 is not None:
            parity = dict(compare_frames(fo, hip_frame1, fast=args.shade_mode == "fast"), frame="camera step 1 of the bench's camera path (first frame with temporal history), "
                                                                f"{WIDTH}x{HEIGHT}, oracle vs HIP read-back")
        del fo
    med = float(np.median(times))
    split = {k: round(float(np.median([sp[k] for sp in splits])), 3) for k in splits[0]}
    cb = {"value": round(WIDTH * HEIGHT / med / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
          "s_per_frame_median": round(med, 3), "stage_s_median": split,
          "sample": f"median of {len(times)} steady-state frame(s) (camera steps 1..{len(times)}) of the same scene/camera path at "
                    f"{WIDTH}x{HEIGHT}, after one history frame; oracle C with OpenMP in every stage on {cores} threads; "
                    f"{sum(times):.1f} s of CPU wall time"}
    return cb, parity


if __name__ == "__main__":
    main()
