"""GPU, TWO processes: the multi-GPU path of the product itself (rend3_amd/parallel.py::Exchange driven from r3n_render_frame's
exchange callbacks) with world size 2 -- one MI355X each over RCCL when the box has two, else both ranks on the one GPU with
the collectives staged through the host (gloo): what is exercised either way is the product's own code: two HIP contexts in
two processes, object sharding (slot ranges, and owner bytes from the spatial partition), shadow views by view + broadcast,
the pass-1 depth exchange in front of Hi-Z, the pass-2 key reduction onto the row owners (dense and row-limited), the split
resolve and the row gather; and the sort-first scheme (rows: every object on every rank, each rank rasterises its rows, the depth
bands are all-gathered, no key exchange) -- and the result is compared with the SAME process's unsharded HIP render, bit for bit, over
frames with camera motion (predicted / residual passes, frames in flight).

Every run goes through tests/mp_harness.py: file rendezvous, breadcrumbs of the worker and of the library, faulthandler, a
limit of R3N_MP_LIMIT seconds, children always reaped -- and the file sorts LAST (behind the single-process parity tests and the
fuzz slice), so the worst a multi-process defect can cost is this file (a silent 900-s wait here cost twelve tests
and the smoke).  tools/soak_native.py runs the same workers in a loop."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
W, FRAMES = 320, 4


def worker(rank, world, run_dir, crumb, mode, H=192, samples=1, by_objects=False, n_objects=200, shim="sync", serial=1):
    """One rank (tests/mp_harness.py::run_ranks calls this in a spawned process; an exception is the rank's failure).
    shim: "sync" | "async" -- tests/rccl_shim.cpp's mode (async: a collective runs when its stream reaches it, one at a time per
    process); serial: the library's comm_serial tunable (1: one total order of the collectives per device, the default)."""
    import math
    if shim == "async":
        os.environ["R3N_SHIM_ASYNC"] = "1"
        os.environ["R3N_SHIM_TIMEOUT"] = "30"
    os.environ["R3N_TUNE"] = (os.environ.get("R3N_TUNE", "") + f" comm_serial={serial}").strip()
    import torch
    import mp_harness
    n_dev = torch.cuda.device_count()
    dev = rank if n_dev >= world else 0
    torch.cuda.set_device(dev)
    device = torch.device("cuda", dev)
    import rend3_amd as r3
    import scenes
    from rend3_amd import parallel
    hm = r3.host
    f32 = np.float32

    def make():
        r = r3.Renderer(hm.LEFT, f32(W) / f32(H), device=dev)
        scenes.build_random_scene(r, hm, r3.material_record, n_objects, 0xE5A1, lights=2, shadow_res=256, with_cutout=True)
        return r

    if mode == "native" and n_dev < world:
        # RCCL refuses two ranks on one device: the library binds tests/rccl_shim.cpp instead (the same entry points between
        # processes that share a GPU, staged through shared memory), so its N = 2 branches run on a one-GPU box too
        import rccl_shim
        os.environ["R3N_RCCL_LIB"] = rccl_shim.build()
    crumb(f"torch up, {n_dev} device(s)")
    shard, full = make(), make()  # contexts first, the communication library second (r3n_create binds the hardware queues)
    crumb("contexts up")
    mp_harness.init_group(rank, world, run_dir, device if n_dev >= world else None)
    crumb("process group up")
    shard.evaluate_instructions()
    counts = np.zeros(shard.capacity, dtype=np.int64)
    centres = np.zeros((shard.capacity, 3), dtype=np.float64)
    radii = np.zeros(shard.capacity, dtype=np.float64)
    for h, m in shard.object_meta.items():
        counts[h] = shard.meshes[m["mesh"]].index_count // 3
        centres[h], radii[h] = m["sphere"][:3], m["sphere"][3]
    ex = None
    if mode == "native":  # r3n_comm_init: the exchanges issued inside r3n_render_frame over RCCL, no exchange object
        shard.comm_init_torch()
        crumb("communicators up")
        mask = np.ones(shard.capacity, dtype=bool)
        if by_objects:  # north_star's object-range split, issued by the library: depth all-reduce + key reduce-scatter
            shard.comm_set_split(True)
            b, e = parallel.partition_objects(counts, world)[rank]
            shard.set_object_range(b, e)
            mask[:] = False
            mask[b:e] = True
    else:
        ex = parallel.Exchange(shard, device)
        ex.rows_equal = H % world == 0
    if mode == "native":
        pass
    elif mode == "slots":
        b, e = parallel.partition_objects(counts, world)[rank]
        shard.set_object_range(b, e)
        mask = np.zeros(shard.capacity, dtype=bool)
        mask[b:e] = True
    elif mode == "rows":  # sort-first: every object on every rank, each rank rasterises its rows only
        rb, re_ = parallel.row_ranges(H, world)[rank]
        ex.set_row_sharding(rb, re_)
        mask = np.ones(shard.capacity, dtype=bool)
    else:
        owners = parallel.partition_objects_spatial(centres, counts, world)
        ex.set_spatial_partition(owners, parallel.partition_bounds(owners, centres, radii, counts, world))
        mask = owners == rank
    rows = parallel.row_ranges(H, world)
    if ex is not None:
        shard._check(shard.lib.r3n_set_row_range(shard.ctx, rows[rank][0], rows[rank][1]), "r3n_set_row_range")
    kw = dict(ambient=(0.1, 0.1, 0.1, 1.0), clear_color=(0.1, 0.2, 0.3, 1.0))
    for f in range(FRAMES):
        ang = 0.25 * f
        view = hm.look_at_lh((3.0 * math.sin(ang), 1.5, -3.0 * math.cos(ang) - 6.0), (0, 0, 6), (0, 1, 0))
        for r in (shard, full):
            r.set_camera_data(view, ("perspective", 60.0, 0.1))
        crumb(f"frame {f}: unsharded render")
        ref = full.render(W, H, samples=samples, **kw)
        crumb(f"frame {f}: sharded render + exchange")
        got = shard.render(W, H, samples=samples, exchange=ex, **kw)
        if ex is not None:
            ex.gather_rows(W, H, world)
        crumb(f"frame {f}: sync")
        shard.sync()
        crumb(f"frame {f}: compare")
        r0, r1 = rows[rank]
        # this rank's rows of the keys are the unsharded ones after the pass-2 exchange; the atlas is whole everywhere
        assert np.array_equal(ref["vis"][r0:r1], got["vis"][r0:r1]), f"{mode} rank {rank} frame {f}: keys of the own rows"
        assert np.array_equal(ref["atlas"].view(np.uint32), got["atlas"].view(np.uint32)), f"{mode} rank {rank} frame {f}: atlas"
        bad = (ref["hdr16"][r0:r1] != got["hdr16"][r0:r1]).any(axis=2)
        if bad.any():
            ys, xs = np.nonzero(bad)
            y, x = int(ys[0]) + r0, int(xs[0])
            slot = int(ref["vis"][y, x] & np.uint64(0xFFFFFFFF)) - 1
            tri_base = np.concatenate([[0], np.cumsum(counts)])[:-1]
            obj = int(np.searchsorted(tri_base, slot, side="right") - 1) if slot >= 0 else -1
            raise AssertionError(f"{mode} rank {rank} frame {f}: HDR of the own rows differs in {int(bad.sum())} px, rows {ys.min() + r0}..{ys.max() + r0}, "
                                 f"cols {xs.min()}..{xs.max()}; first ({x}, {y}): ref {ref['hdr16'][y, x]} got {got['hdr16'][y, x]}, object {obj} "
                                 f"owned here: {bool(mask[obj]) if obj >= 0 else None}; differing px on own objects: "
                                 f"{int(sum(1 for yy, xx in zip(ys[:2000], xs[:2000]) if (lambda sl: sl >= 0 and mask[int(np.searchsorted(tri_base, sl, side='right') - 1)])(int(ref['vis'][yy + r0, xx] & np.uint64(0xFFFFFFFF)) - 1)))} of {min(len(ys), 2000)}")
        # the gathered image: every row from its owner
        out = np.zeros((H, W, 4), dtype=np.uint8)
        shard._check(shard.lib.r3n_readback_output(shard.ctx, out.ctypes.data, None), "r3n_readback_output")
        assert np.array_equal(out, ref["rgba8"]), f"{mode} rank {rank} frame {f}: gathered image"
        # L1 / L2 sets: the unsharded sets restricted to this rank's objects (the Hi-Z it culled against was the global one)
        assert np.array_equal(got["visible"].astype(bool), ref["visible"].astype(bool) & mask), f"{mode} L1 frame {f}"
        tri_base = np.concatenate([[0], np.cumsum(counts)])[:-1]
        tri_obj = np.searchsorted(tri_base, np.arange(len(ref["pass"])), side="right") - 1
        tmask = mask[np.clip(tri_obj, 0, len(mask) - 1)]
        n = int(counts.sum())
        assert np.array_equal(got["pass"][:n].astype(bool), ref["pass"][:n].astype(bool) & tmask[:n]), f"{mode} L2 pass frame {f}"
        assert np.array_equal(got["residual"][:n].astype(bool), ref["residual"][:n].astype(bool) & tmask[:n]), f"{mode} L2 residual frame {f}"
    assert n_objects < 100 or (ref["residual"].sum() > 0 and ref["pass"].sum() > 100)  # (the scene exercised both lists)
    crumb("closing")
    shard.close(); full.close()


MODES = ["slots", "spatial", "rows", "native", "native-ragged", "native-msaa", "native-objects", "native-objects-ragged", "native-objects-msaa"]
MANY = [
    (4, "native-ragged", 200),            # sort-first rows, 190 rows over 4 ranks: bands of 48 / 48 / 47 / 47, one broadcast per band
    (4, "native-objects-ragged", 200),    # object ranges + MAX all-reduces; two shadow views on four ranks: each view's rows split in two bands
    (4, "native-objects-msaa", 200),      # equal bands (192 rows), four samples: reduce-scatter of the keys' bands
    (8, "native-ragged", 200),            # 190 rows over 8 ranks: seven peers in the ragged band broadcasts
    (8, "native-objects-ragged", 6),      # six objects on eight ranks: EMPTY object ranges; two shadow views in four bands of 64 rows each
    (8, "native-objects", 200),           # equal bands at eight ranks: in-place reduce-scatter onto the row owners
]


def worker_args(mode, world=2, n_objects=200, shim="sync", serial=1):
    """(mode, H, samples, by_objects, n_objects, shim, serial) of a parametrisation; heights: 192 = equal bands, 191 / 190 ragged ones."""
    ragged = (191 if world == 2 else 190) if "ragged" in mode else 192
    return (mode.split("-")[0], ragged, 4 if mode.endswith("msaa") else 1, "objects" in mode, n_objects, shim, serial)


@pytest.mark.gpu
@pytest.mark.timeout(240)
@pytest.mark.parametrize("mode", MODES)
def test_two_processes_exchange_matches_unsharded(mode):
    """native*: r3n_comm_init + the collectives r3n_render_frame issues itself -- sort-first rows (in-place all-gather of equal
    bands; ragged: one broadcast per band, height 191; msaa: the keys' bands, four samples) and native-objects*: the
    object-range split (r3n_comm_set_split: depth MAX all-reduce in front of Hi-Z, key MAX reduce-scatter onto the row bands, an
    all-reduce when they are ragged) -- over RCCL with one GPU per rank, else over tests/rccl_shim.cpp with both ranks on the
    one GPU."""
    import torch
    import mp_harness
    assert torch.cuda.is_available()
    results, problem, rep, _ = mp_harness.run_ranks(worker, 2, worker_args(mode))
    mp_harness.check(results, problem, rep, 2)


@pytest.mark.gpu
@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,mode,n_objects", MANY)
def test_many_processes_exchange_matches_unsharded(world, mode, n_objects):
    """the library's own exchange with MORE than two ranks -- states two ranks cannot reach (more ranks than
    shadow views: a view's rows are split into world / views bands, band p of view v drawn by rank v + p * views and broadcast from
    there -- the atlas every rank ends up with is compared bit for bit; empty object ranges; seven peers in comm_gather_bands'
    ragged broadcasts).
    One GPU per rank over RCCL where the box has them, else every rank on the one GPU through tests/rccl_shim.cpp (it takes any
    rank count).  Each rank compares its rows / its objects' sets with its own unsharded render, bit for bit, over four frames."""
    import torch
    import mp_harness
    assert torch.cuda.is_available()
    results, problem, rep, _ = mp_harness.run_ranks(worker, world, worker_args(mode, world, n_objects))
    mp_harness.check(results, problem, rep, world)


ASYNC = [(2, "native", 200), (2, "native-objects-msaa", 200), (4, "native-objects-ragged", 200), (8, "native-ragged", 200), (8, "native-objects", 200)]


@pytest.mark.gpu
@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,mode,n_objects", ASYNC)
def test_collectives_in_one_total_order_under_the_asynchronous_shim(world, mode, n_objects):
    """VERDICT r5 item 6.  The three communicators (main / shadow / rows) are driven from three streams; the synchronous shim
    completes every collective at enqueue and can never show a dependence on the order in which a DEVICE executes them.  In its
    asynchronous mode a collective runs when its stream reaches it, one at a time per process (a device with room for one collective
    kernel): ranks whose streams reach two communicators' collectives in different orders time out in the barrier.  With the
    library's comm_serial order (default) every device executes the collectives in program order: the frames must come out bit for
    bit as in the synchronous runs.  (On a box with one GPU per rank real RCCL is used and this is a plain repeat.)"""
    import torch
    import mp_harness
    assert torch.cuda.is_available()
    results, problem, rep, _ = mp_harness.run_ranks(worker, world, worker_args(mode, world, n_objects, shim="async", serial=1))
    mp_harness.check(results, problem, rep, world)
