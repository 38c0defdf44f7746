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
        assert shard.shadow_views_owned and len(shard.shadow_views_owned) < len(shard.dir_lights)
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
            part[(1 - rank) * n:(2 - rank) * n] >>= 1  # this rank's partial values of the OTHER rank's rows: something smaller
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


def test_two_rank_gloo_exact():
    world = 2
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
