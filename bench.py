#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric ("culled objects/s + shaded Mpixels/s @4K, Bistro scene") on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched as python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one whole frame of the hot path over the synthetic Bistro-like scene of BASELINE.json configs[2]
(rend3_amd/scenes.py::bistro_like: ~3 000 objects, ~2.8 M triangles, 130 PBR materials, 4 directional lights with
2048^2 shadow views, 3840x2160, camera dollying down the street so the residual pass has real work):
  per shadow view: uniform bake -> object frustum cull + slot scan -> triangle cull + compaction -> depth raster;
  viewport: bake -> raster predicted -> Hi-Z -> cull -> raster residual -> PBR resolve -> tonemap.
All inputs are resident in HBM before the timed region; per-frame camera blocks (a few KB) are precomputed on the
host and uploaded through the C ABI exactly as the Rust side would.  value = W*H*K / time (whole job, all ranks).
Prints ONE JSON line on rank 0.
"""
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


def camera_path(hm, view0, step):
    """Dolly 0.15 m per frame down the street with a slow yaw sweep: view_k = Ry(yaw_k) * T(0,0,+d_k) * view0 (RH: forward = -Z)."""
    d = 0.15 * step
    yaw = 0.02 * math.sin(0.25 * step)
    return hm.mat4_mul(hm.mat4_mul(hm.rotation_y(yaw), hm.translation((0.0, 0.0, d))), view0)


def build_scene(r3, objects, tris, width, height):
    r = r3.Renderer(r3.host.RIGHT, float(width) / float(height))
    info = r3.scenes.bistro_like(r, r3.host, r3.material_record, n_objects=objects, target_tris=tris)
    return r, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--objects", type=int, default=3000)
    ap.add_argument("--tris", type=int, default=2_800_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--untextured", action="store_true",
                    help="factor-only materials (the round-1 workload before textures were built); default: every material "
                         "has base colour + normal + AO/roughness/metallic maps")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU exchange path (RCCL all-reduce / all-gather) even with one rank: plumbing check")
    ap.add_argument("--samples", type=int, default=1, choices=(1, 4),
                    help="SampleCount of the viewport targets (the reference's scene_viewer default is 4); the headline metric is quoted at 1")
    ap.add_argument("--cpu-sample-frames", type=int, default=1)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or args.force_exchange
    if distributed:
        if "MASTER_ADDR" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import rend3_amd as r3
    import rend3_amd.scenes  # noqa: F401
    from rend3_amd import parallel

    # ---------------------------------------------------------------- scene: replicated on every rank
    r = r3.Renderer(r3.host.RIGHT, np.float32(WIDTH) / np.float32(HEIGHT), device=local_rank)
    info = r3.scenes.bistro_like(r, r3.host, r3.material_record, n_objects=args.objects, target_tris=args.tris,
                                 textured=not args.untextured)
    view0 = info["camera"][0]
    exchange = None
    if distributed:
        counts = np.zeros(r.capacity, dtype=np.int64)
        for h, m in r.object_meta.items():
            counts[h] = r.meshes[m["mesh"]].index_count // 3
        begin, end = parallel.partition_objects(counts, world)[rank]
        r.set_object_range(begin, end)
        exchange = parallel.Exchange(r, device)
        rows = parallel.row_ranges(HEIGHT, world)
        exchange.rows_equal = HEIGHT % world == 0
        r._check(r.lib.r3n_set_row_range(r.ctx, rows[rank][0], rows[rank][1]), "r3n_set_row_range")

    base = r3.BaseRenderGraph(r)

    def frame(k, readback=False):
        r.set_camera_data(camera_path(r3.host, view0, k), info["camera"][1])
        out = r.render(WIDTH, HEIGHT, samples=args.samples, ambient=AMBIENT, clear_color=CLEAR, readback=readback, base=base, exchange=exchange)
        if exchange is not None:
            exchange.gather_rows(WIDTH, HEIGHT, world)
        return out

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- warmup + timed region
    for k in range(args.warmup):
        frame(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        frame(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------------------------------------------------------- instrumented pass (outside the timed region)
    # HIP events on the context's stream around every kernel launch of each stage (include/r3n.h R3N_STAGE_*)
    # (single-stream here: a kernel's duration is inflated while kernels of other streams are co-resident)
    r.sync()
    r.set_multi_stream(False)
    r.timing_enable(True)
    r.stage_times(reset=True)
    n_inst = min(args.steps, 20)
    for k in range(n_inst):
        frame(args.warmup + args.steps + k)
    r.sync()
    stages = r.stage_times(reset=True)
    r.timing_enable(False)
    r.set_multi_stream(True)
    last = frame(args.warmup + args.steps + n_inst, readback=(world == 1))

    result = None
    traffic, valu_busy = {}, {}
    try:  # HBM bytes per launch from the committed PMC passes (collected separately, see profiles/r01_summary.md)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            committed = json.load(fh)
            traffic = committed.get("bytes_per_launch", {})
            valu_busy = committed.get("valu_busy", {})
    except OSError:
        pass
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        mpix = WIDTH * HEIGHT * args.steps / elapsed / 1e6
        cameras = 1 + 4
        stage_ms = {s: (ms / n_inst) for s, (ms, _n) in stages.items()}
        launches = {s: n / n_inst for s, (_ms, n) in stages.items()}
        cull_ms = stage_ms["bake"] + stage_ms["object_cull"] + stage_ms["triangle_cull"]
        tri_cull_ms_per_launch = stage_ms["triangle_cull"] / max(launches["triangle_cull"], 1)
        # roofline objects: algorithmic bytes per launch (SURVEY.md section 8d / BASELINE.md section 5) / average launch duration
        # from HIP events recorded on the context's stream around that kernel's launches (instrumented pass above)
        roof = roof_cull = None
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
            bytes_per_launch = float(np.mean(per_launch))
            achieved = bytes_per_launch / (tri_cull_ms_per_launch * 1e-3) / 1e9
            roof_cull = {"kernel": "k_triangle_cull", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic.get("k_triangle_cull") if world == 1 else None,
                         "bytes_per_launch": int(bytes_per_launch), "ms_per_launch": round(tri_cull_ms_per_launch, 5),
                         "triangles_per_launch": vis_tris // len(cams)}
        # dominant kernel by GPU time: the deferred PBR resolve (one launch per frame).  Its HBM floor is 8 B key read +
        # 8 B Rgba16Float write per pixel (">= 16 B / shaded pixel") + 4 B for the Rgba8UnormSrgb
        # blit fused into it.  The kernel is VALU-bound, not HBM-bound (SQ_ACTIVE_INST_VALU = the whole SIMD issue
        # capacity of the launch, profiles/r01_summary.md), so the fraction of the HBM roofline is small by construction.
        shade_ms = stage_ms["shade"] / max(launches["shade"], 1)
        shade_bytes = 20.0 * WIDTH * HEIGHT / world
        ach = shade_bytes / (shade_ms * 1e-3) / 1e9
        roof = {"kernel": "k_resolve_opaque", "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get("k_resolve_opaque") if world == 1 else None,
                "bytes_per_launch": int(shade_bytes),
                "ms_per_launch": round(shade_ms, 5),
                "valu_busy": valu_busy.get("k_resolve_opaque" if not args.untextured else "k_resolve_opaque_untextured") if world == 1 else None,
                "note": "dominant kernel by time; VALU-bound, not HBM-bound: valu_busy = SQ_ACTIVE_INST_VALU / SIMD issue cycles of the "
                        "launch (SQ counter pass, profiles/r01_summary.md section 2); about "
                        + ("1500" if args.untextured else "2400") + " vector instructions per pixel (4 lights x (5-tap PCF + GGX)"
                        + ("" if args.untextured else ", 3 trilinear maps, tangent frame") + ", all IEEE div/sqrt); bytes = key + HDR + sRGB per pixel, texels excluded"}
        result = {
            "metric": "shaded Mpixels/s @4K (whole frame: cull+compact all cameras, 4 shadow views, PBR opaque, tonemap)",
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2] stand-in: bistro_like (seed 0xB157), 3840x2160, "
                                   "full PBR opaque + 4 directional shadow views (2048^2), camera dolly, " + ("MSAA x4, " if args.samples == 4 else "")
                                   + ("factor-only materials" if args.untextured else
                                      "130 materials with base colour + normal + AO/roughness/metallic maps (RGBA8, mips, trilinear)"),
                       "objects": info["objects"], "triangles": info["triangles"], "cameras": cameras,
                       "parallelism": "single GPU" if world == 1 else f"object-range sharding x{world}, RCCL max all-reduce of depth keys"},
            "fps": round(args.steps / elapsed, 2),
            "culled_objects_per_s": round(info["objects"] * cameras / (cull_ms * 1e-3), 1) if cull_ms > 0 else None,
            "culled_mtris_per_s": round(info["triangles"] * cameras / (cull_ms * 1e-3) / 1e6, 1) if cull_ms > 0 else None,
            "stage_ms_per_frame": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_launches_per_frame": launches,
            "roofline": roof,
            "roofline_triangle_cull": roof_cull,
        }

    # ---------------------------------------------------------------- CPU baseline: the oracle ("port"), rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, info)
    if rank == 0:
        print(json.dumps(result))
    r.close()
    if distributed:
        dist.destroy_process_group()


def cpu_baseline(args, info):
    """The C oracle (oracle/r3o.c, OpenMP over objects/rows) timed on the host cores on a bounded sample of the same
    workload: the same scene and camera path, `--cpu-sample-frames` steady-state frames at 4K after one history frame.
    kind "port": the reference itself (Rust + wgpu on lavapipe) cannot be built here."""
    import numpy as np
    from oracle import host as oh
    from oracle.world import OracleRenderer, material_record as omk
    import rend3_amd as r3
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    o = OracleRenderer(oh.RIGHT, np.float32(WIDTH) / np.float32(HEIGHT))
    info_o = r3.scenes.bistro_like(o, oh, omk, n_objects=args.objects, target_tris=args.tris, textured=not args.untextured)
    view0 = info_o["camera"][0]
    o.set_camera_data(camera_path(oh, view0, 0), info_o["camera"][1])
    o.render(WIDTH, HEIGHT, samples=args.samples, ambient=AMBIENT, clear_color=CLEAR)  # history frame (untimed)
    t0 = time.perf_counter()
    for k in range(args.cpu_sample_frames):
        o.set_camera_data(camera_path(oh, view0, 1 + k), info_o["camera"][1])
        o.render(WIDTH, HEIGHT, samples=args.samples, ambient=AMBIENT, clear_color=CLEAR)
    dt = time.perf_counter() - t0
    return {"value": round(WIDTH * HEIGHT * args.cpu_sample_frames / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores,
            "kind": "port", "sample": f"{args.cpu_sample_frames} steady-state frame(s) of the same scene/camera path at "
                                      f"{WIDTH}x{HEIGHT} (oracle C, OpenMP cull+shade, single-thread raster), {dt:.1f} s"}


if __name__ == "__main__":
    main()
