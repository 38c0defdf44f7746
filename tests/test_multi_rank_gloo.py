"""CPU, world_size 2, gloo: the multi-GPU algorithm of DESIGN.md section 6 (object-range sharding of the viewport, shadow
views sharded by view and broadcast from their owners, MAX all-reduce of the pass-1 depth plane and of the pass-2 visibility
keys, row-split resolve with all-gather) is exact: two ranks, each running the
ORACLE over its own object range and exchanging through rend3_amd.parallel's collectives, end up with the same
per-triangle sets (union), visibility keys and image as one unsharded oracle, bit for bit, over several frames."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, FRAMES = 160, 96, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    import math
    import scenes
    from oracle import host as oh
    from oracle.world import OracleRenderer, material_record as omk
    r = OracleRenderer(oh.LEFT, np.float32(W) / np.float32(H))
    scenes.build_random_scene(r, oh, omk, 150, 0xE5A0, lights=2, shadow_res=128, with_cutout=True)
    return r, oh, math


def _camera(r, oh, math, f):
    ang = 0.3 * f
    r.set_camera_data(oh.look_at_lh((3.0 * math.sin(ang), 1.0, -3.0 * math.cos(ang)), (0, 0, 6), (0, 1, 0)), ("perspective", 60.0, 0.1))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rend3_amd import parallel
    try:
        full, oh, math = _scene()
        shard, _, _ = _scene()
        counts = (shard.objects[:, 21] // 3) * (shard.objects[:, 29] != 0)
        ranges = parallel.partition_objects(counts, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(counts) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        shard.object_range = ranges[rank]
        # shadow views by view: view v is rendered whole by rank v mod N
        shard.shadow_views_owned = {v for v in range(len(shard.dir_lights)) if parallel.shadow_view_owner(v, world) == rank}
        # (two ranks, two views: one each; more ranks than views: view v on rank v mod N, the others own none and only receive)
        assert len(shard.shadow_views_owned) == sum(1 for v in range(len(shard.dir_lights)) if v % world == rank)
        assert world > 2 or (shard.shadow_views_owned and len(shard.shadow_views_owned) < len(shard.dir_lights))
        rows = parallel.row_ranges(H, world)

        def exchange(what, arr, shadows=None):
            if what == "shadow":
                parallel.exchange_shadow_views_(torch.from_numpy(arr), shadows, rank, world)
            elif what == "pass1_depth":
                parallel.allreduce_max_(torch.from_numpy(arr))  # f32 depth plane, depth >= 0
            else:
                parallel.allreduce_max_(torch.from_numpy(arr.view(np.int64)))

        for f in range(FRAMES):
            _camera(full, oh, math, f)
            _camera(shard, oh, math, f)
            ref = full.render(W, H, ambient=(0.1, 0.1, 0.1, 1), clear_color=(0.1, 0.2, 0.3, 1))
            got = shard.render(W, H, ambient=(0.1, 0.1, 0.1, 1), clear_color=(0.1, 0.2, 0.3, 1), exchange=exchange)
            # after the pass-2 exchange every rank holds the full keys -> identical image on every rank
            assert np.array_equal(ref["vis"], got["vis"]), f"vis frame {f}"
            assert np.array_equal(ref["atlas"].view(np.uint32), got["atlas"].view(np.uint32)), f"atlas frame {f}"
            assert np.array_equal(ref["rgba8"], got["rgba8"]), f"image frame {f}"
            # L1/L2: each rank's sets are the reference's restricted to its object range; the union is exact
            b, e = ranges[rank]
            mask = np.zeros(len(ref["visible"]), dtype=bool)
            mask[b:e] = True
            assert np.array_equal(got["visible"].astype(bool), ref["visible"].astype(bool) & mask), f"L1 frame {f}"
            tri_obj = np.searchsorted(ref["tri_base"], np.arange(len(ref["pass"])), side="right") - 1
            tmask = mask[tri_obj]
            assert np.array_equal(got["pass"].astype(bool), ref["pass"].astype(bool) & tmask), f"L2 pass frame {f}"
            assert np.array_equal(got["residual"].astype(bool), ref["residual"].astype(bool) & tmask), f"L2 residual frame {f}"
            union = torch.from_numpy(got["pass"].astype(np.int32))
            dist.all_reduce(union, op=dist.ReduceOp.SUM)
            assert np.array_equal(union.numpy().astype(bool), ref["pass"].astype(bool))
            # row-split output gather: each rank contributes only its rows
            img = torch.from_numpy(got["rgba8"].copy().reshape(-1))
            r0, r1 = rows[rank]
            keep = img.clone()
            img.zero_()
            img[r0 * W * 4:r1 * W * 4] = keep[r0 * W * 4:r1 * W * 4]
            parallel.allgather_rows_(img, rank, world)
            # pass-2 exchange as a reduce-scatter to the row owners (falls back to the all-reduce on gloo): own rows complete
            keys = torch.from_numpy(got["vis"].copy().reshape(-1).view(np.int64))
            parallel.reduce_scatter_max_rows_(keys, rank, world)
            n = keys.numel() // world
            assert np.array_equal(keys.numpy()[rank * n:(rank + 1) * n].view(np.uint64), ref["vis"].reshape(-1)[rank * n:(rank + 1) * n])
            assert np.array_equal(img.numpy().reshape(H, W, 4), ref["rgba8"]), f"gather frame {f}"
            # the direct (all-to-all + local MAX) forms of the two reductions give the collectives' results
            part = torch.from_numpy(got["vis"].copy().reshape(-1).view(np.int64))
            for o in range(world):  # this rank's partial values of the OTHER ranks' rows: something smaller
                if o != rank:
                    part[o * n:(o + 1) * n] >>= 1
            parallel.direct_reduce_scatter_max_(part, rank, world)
            assert np.array_equal(part.numpy()[rank * n:(rank + 1) * n].view(np.uint64), ref["vis"].reshape(-1)[rank * n:(rank + 1) * n])
            plane = torch.from_numpy(np.where((np.arange(W * H) % world) == rank, got["atlas"].reshape(-1)[:W * H], np.float32(0.0)).astype(np.float32))
            want = torch.from_numpy(got["atlas"].reshape(-1)[:W * H].copy())
            parallel.direct_allreduce_max_(plane, rank, world)
            assert np.array_equal(plane.numpy().view(np.uint32), want.numpy().view(np.uint32)), f"direct all-reduce frame {f}"
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(exc) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _worker_sparse(rank, world, port, q):
    """The spatial partition + row-limited exchanges: objects by Morton order of their bounding-sphere centres (owner bytes),
    pass-1 depth and pass-2 keys reduced onto the row-band owners moving only the rows inside each rank's conservative screen
    extent (parallel.rows_alltoall_max_ over extents every rank derives from the replicated world), merged depth bands
    all-gathered, image rows all-gathered: equal to the unsharded oracle bit for bit."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rend3_amd import parallel
    try:
        full, oh, math = _scene()
        shard, _, _ = _scene()
        counts = ((shard.objects[:, 21] // 3) * (shard.objects[:, 29] != 0)).astype(np.int64)
        spheres = shard.objects[:, 16:20].view(np.float32)
        owners = parallel.partition_objects_spatial(spheres[:, :3], counts, world)
        assert set(np.unique(owners[counts > 0])) == set(range(world))
        loads = [int(counts[(owners == r) & (counts > 0)].sum()) for r in range(world)]
        assert max(loads) <= 0.7 * sum(loads)
        bounds = parallel.partition_bounds(owners, spheres[:, :3], spheres[:, 3], counts, world)
        shard.object_owners = (owners, rank)
        shard.shadow_views_owned = {v for v in range(len(shard.dir_lights)) if parallel.shadow_view_owner(v, world) == rank}
        bands = parallel.row_ranges(H, world)
        sent = {"pass1_depth": 0, "pass2": 0}
        narrow = []

        def exchange(what, arr, shadows=None):
            if what == "shadow":
                parallel.exchange_shadow_views_(torch.from_numpy(arr), shadows, rank, world)
                return
            extents = parallel.partition_row_extents(bounds, shard.camera.view_proj, H)
            narrow.append(extents[rank][1] - extents[rank][0] < H)
            if what == "pass1_depth":
                plane = torch.from_numpy(arr)  # f32 depth plane, depth >= 0
                sent[what] += parallel.rows_alltoall_max_(plane.view(H, W), extents, bands, rank, world)
                parallel.allgather_rows_(plane, rank, world)
            else:
                keys = torch.from_numpy(arr.view(np.int64))
                sent[what] += parallel.rows_alltoall_max_(keys.view(H, W), extents, bands, rank, world)

        r0, r1 = bands[rank]
        for f in range(FRAMES):
            _camera(full, oh, math, f)
            _camera(shard, oh, math, f)
            ref = full.render(W, H, ambient=(0.1, 0.1, 0.1, 1), clear_color=(0.1, 0.2, 0.3, 1))
            got = shard.render(W, H, ambient=(0.1, 0.1, 0.1, 1), clear_color=(0.1, 0.2, 0.3, 1), exchange=exchange)
            # the Hi-Z every rank culled against is the global one; after the pass-2 exchange the rank's OWN rows are complete
            assert np.array_equal(ref["hiz"].view(np.uint32), got["hiz"].view(np.uint32)), f"hi-z frame {f}"
            assert np.array_equal(ref["vis"][r0:r1], got["vis"][r0:r1]), f"vis rows frame {f}"
            assert np.array_equal(ref["atlas"].view(np.uint32), got["atlas"].view(np.uint32)), f"atlas frame {f}"
            assert np.array_equal(ref["rgba8"][r0:r1], got["rgba8"][r0:r1]), f"image rows frame {f}"
            mask = owners == rank
            assert np.array_equal(got["visible"].astype(bool), ref["visible"].astype(bool) & mask), f"L1 frame {f}"
            tri_obj = np.searchsorted(ref["tri_base"], np.arange(len(ref["pass"])), side="right") - 1
            tmask = mask[tri_obj]
            assert np.array_equal(got["pass"].astype(bool), ref["pass"].astype(bool) & tmask), f"L2 pass frame {f}"
            assert np.array_equal(got["residual"].astype(bool), ref["residual"].astype(bool) & tmask), f"L2 residual frame {f}"
            img = torch.from_numpy(got["rgba8"].copy().reshape(-1))
            parallel.allgather_rows_(img, rank, world)
            assert np.array_equal(img.numpy().reshape(H, W, 4), ref["rgba8"]), f"gather frame {f}"
        # the extents did cut rows for somebody on some frame, and less than whole targets moved
        flags = torch.tensor([1 if any(narrow) else 0, sent["pass2"]], dtype=torch.int64)
        dist.all_reduce(flags, op=dist.ReduceOp.SUM)
        assert flags[0].item() >= 1, "no rank ever had a partial row extent: the sparse path was not exercised"
        assert flags[1].item() < FRAMES * world * (world - 1) / world * W * H * 8, "the row-limited exchange moved as much as the dense one"
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(exc) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _worker_rows(rank, world, port, q, height=None):
    """Sort-first (R3N_SHARD_ROWS, Exchange.set_row_sharding): every rank culls and draws every object but keeps only its rows;
    the depth bands are all-gathered in front of Hi-Z, nothing is exchanged after pass 2.  The visible sets are then the
    UNSHARDED ones on every rank, the own rows of keys / image the unsharded ones, the gathered image the unsharded image."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rend3_amd import parallel
    try:
        full, oh, math = _scene()
        shard, _, _ = _scene()
        H_ = height or H
        rows = parallel.row_ranges(H_, world)
        even = H_ % world == 0
        shard.row_band = rows[rank]
        shard.shadow_views_owned = {v for v in range(len(shard.dir_lights)) if parallel.shadow_view_owner(v, world) == rank}
        calls = []

        def exchange(what, arr, shadows=None):
            calls.append(what)
            if what == "shadow":
                parallel.exchange_shadow_views_(torch.from_numpy(arr), shadows, rank, world)
            elif what in ("pass1_depth", "pass1"):  # the bands are final: gather, no reduction
                flat = torch.from_numpy(arr.reshape(-1).view(np.int64 if arr.dtype == np.uint64 else arr.dtype))
                if even:
                    parallel.allgather_rows_(flat, rank, world)
                else:  # ragged bands (Exchange.__call__ without rows_equal): rows a rank does not own hold the clear value, MAX merges them
                    parallel.allreduce_max_(flat)
            # pass2: nothing

        for samples in (1, 4):
            for f in range(FRAMES):
                _camera(full, oh, math, f)
                _camera(shard, oh, math, f)
                kw = dict(samples=samples, ambient=(0.1, 0.1, 0.1, 1), clear_color=(0.1, 0.2, 0.3, 1))
                ref = full.render(W, H_, **kw)
                got = shard.render(W, H_, exchange=exchange, **kw)
                r0, r1 = rows[rank]
                assert np.array_equal(ref["vis"][r0:r1], got["vis"][r0:r1]), f"own rows of the keys, frame {f}"
                assert not got["vis"][:r0].any() and not got["vis"][r1:].any()
                assert np.array_equal(ref["atlas"].view(np.uint32), got["atlas"].view(np.uint32)), f"atlas frame {f}"
                assert np.array_equal(ref["hiz"].view(np.uint32), got["hiz"].view(np.uint32)), f"Hi-Z pyramid frame {f}"
                for k in ("visible", "pass", "residual"):
                    assert np.array_equal(got[k], ref[k]), f"{k} frame {f}: the sets are the unsharded ones on every rank"
                assert np.array_equal(ref["rgba8"][r0:r1], got["rgba8"][r0:r1]), f"own rows of the image, frame {f}"
                if even:
                    img = torch.from_numpy(got["rgba8"].copy().reshape(-1))
                    parallel.allgather_rows_(img, rank, world)
                    assert np.array_equal(img.numpy().reshape(H_, W, 4), ref["rgba8"]), f"gathered image frame {f}"
            assert ref["residual"].sum() > 0
        assert "pass1_depth" in calls and "pass1" in calls and "shadow" in calls
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + repr(exc) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, H), (2, H - 1), (4, H), (4, H - 1)])
def test_two_rank_gloo_rows_exact(world, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rows, args=(r, world, port, q, height)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_spatial_partition_helpers():
    from rend3_amd import parallel
    rng = np.random.default_rng(7)
    centres = rng.uniform(-100, 100, (4000, 3))
    tris = rng.integers(0, 300, 4000)
    for world in (2, 4, 8):
        owners = parallel.partition_objects_spatial(centres, tris, world)
        live = tris > 0
        loads = [tris[(owners == r) & live].sum() for r in range(world)]
        assert max(loads) <= 1.1 * sum(loads) / world
        # compact: a partition's bounding box is much smaller than the world's
        vol = [np.prod(np.ptp(centres[(owners == r) & live], axis=0)) for r in range(world)]
        assert np.median(vol) <= 1.2 * np.prod(np.ptp(centres, axis=0)) / world * 2
        assert (owners[~live] == 0).all()
    bounds = parallel.partition_bounds(owners, centres, np.full(4000, 0.5), tris, 8)
    # an orthographic-like camera looking down -z from far away: extents are conservative and inside the target
    vp = np.array([0.01, 0, 0, 0, 0, 0.01, 0, 0, 0, 0, 0.001, 0, 0, 0, 0.5, 1], dtype=np.float32)
    ext = parallel.partition_row_extents(bounds, vp, 1000)
    for boxes, (y0, y1) in zip(bounds, ext):
        lo, hi = np.min([b[0] for b in boxes], axis=0), np.max([b[1] for b in boxes], axis=0)
        rows = (1.0 - np.array([lo[1], hi[1]]) * 0.01) * 500.0
        assert 0 <= y0 <= max(rows.min(), 0) and min(rows.max(), 1000) <= y1 <= 1000 and y0 <= y1
    # a corner behind the eye plane: the whole target
    vp_persp = np.array([0.01, 0, 0, 0, 0, 0.01, 0, 0, 0, 0, 0, 1, 0, 0, 0.1, 0], dtype=np.float32)  # w = z, a very wide view
    ext = parallel.partition_row_extents(bounds, vp_persp, 1000)
    assert (0, 1000) in ext and all(0 <= a <= b <= 1000 for a, b in ext)  # boxes straddling the eye plane inside the view: whole target


def test_two_rank_gloo_sparse_exact():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_partition_and_rows():
    from rend3_amd import parallel
    counts = np.array([10, 0, 0, 5000, 20, 20, 3000, 1, 1, 1, 4000, 7], dtype=np.int64)
    for world in (1, 2, 3, 4, 8):
        rs = parallel.partition_objects(counts, world)
        assert len(rs) == world and rs[0][0] == 0 and rs[-1][1] == len(counts)
        assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(rs, rs[1:]))
        if world == 2:
            loads = [counts[a:b].sum() for a, b in rs]
            assert max(loads) <= 0.75 * counts.sum()
    assert parallel.row_ranges(2160, 8) == [(270 * i, 270 * (i + 1)) for i in range(8)]
    assert parallel.row_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_gloo_exact(world):
    """(world 4, VERDICT r4 item 4: two shadow views on four ranks -- ranks 2 and 3 own no view and only receive; three peers in
    every reduction)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_stale_partition_bounds_fall_back_to_whole_targets():
    """parallel.Exchange.set_spatial_partition captures host-side bounds of the world as it is then; after ANY world edit
    (Renderer.world_version: add / move / remove an object, new joint matrices or poses) the row-limited exchanges must use
    whole targets again, or a rank that now draws outside its old extent would silently not send those rows."""
    from rend3_amd import parallel

    class FakeRenderer:
        world_version = 7

        def current_view_proj(self):
            raise AssertionError("stale bounds must not be projected")

    ex = parallel.Exchange.__new__(parallel.Exchange)
    ex.world, ex.full_extent_frames = 2, 0
    ex.sparse = dict(bounds=None, version=6)  # the partition was set one edit ago
    assert ex._row_extents(FakeRenderer(), 96) == [(0, 96), (0, 96)]
    ex.sparse = dict(bounds=[], version=7)
    projected = []
    FakeRenderer.current_view_proj = lambda self: projected.append(1) or np.eye(4, dtype=np.float32)
    ex._row_extents(FakeRenderer(), 96)  # up to date: the bounds are used
    assert projected == [1]
