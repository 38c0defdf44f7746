"""CPU: properties of the multi-GPU partition functions (rend3_amd/parallel.py) over drawn inputs.

The gloo / shim tests run a few fixed scenes through whole frames with 2, 4 and 8 ranks; these check the pieces for EVERY size the
driver could launch: object ranges and row bands tile their domain whatever the world size (more ranks than objects or rows
included), the spatial partition assigns every live slot exactly once and balances the load, a shadow view has one owner, and the
conservative row extents of the spatial split really contain every pixel row a partition's geometry can reach -- the exchanges
skip the rows outside them, so an extent that is too small loses fragments silently."""
import math

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import host as oh
from rend3_amd import parallel

WORLDS = st.integers(min_value=1, max_value=16)


@settings(max_examples=300, deadline=None, derandomize=True, database=None)
@given(st.lists(st.integers(min_value=0, max_value=5000), min_size=0, max_size=300), WORLDS)
def test_object_ranges_tile_the_slots_and_balance_the_load(counts, world):
    ranges = parallel.partition_objects(counts, world)
    assert len(ranges) == world
    assert ranges[0][0] == 0 and ranges[-1][1] == len(counts)
    assert all(b <= e for b, e in ranges) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    if counts:
        cost = np.asarray(counts, dtype=np.int64) + 1  # the function's load measure: an empty slot still costs one
        loads = [int(cost[b:e].sum()) for b, e in ranges]
        # a contiguous split by prefix sums: no range exceeds its share by more than the largest single object
        assert max(loads) <= math.ceil(cost.sum() / world) + int(cost.max())


@settings(max_examples=300, deadline=None, derandomize=True, database=None)
@given(st.integers(min_value=1, max_value=5000), WORLDS)
def test_row_bands_tile_the_target_like_the_library_does(height, world):
    rows = parallel.row_ranges(height, world)
    assert rows[0][0] == 0 and rows[-1][1] == height and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    sizes = [e - b for b, e in rows]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # rend3_amd/csrc/r3n.hip band_rows (the native exchanges' split): base = h / world, the first h mod world bands one row taller
    base, rem = height // world, height % world
    for r, (b, e) in enumerate(rows):
        assert b == r * base + min(r, rem) and e == b + base + (1 if r < rem else 0)


@settings(max_examples=120, deadline=None, derandomize=True, database=None)
@given(st.integers(min_value=0, max_value=2 ** 31 - 1), st.integers(min_value=0, max_value=400), WORLDS)
def test_spatial_partition_assigns_every_live_slot_once(seed, n, world):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-50, 50, (n, 3))
    counts = rng.integers(0, 2000, n) * (rng.uniform(size=n) < 0.8)
    owners = parallel.partition_objects_spatial(centres, counts, world)
    assert owners.dtype == np.uint8 and len(owners) == n
    assert (owners < world).all() and (owners[counts == 0] == 0).all()
    live = np.flatnonzero(counts > 0)
    if world > 1 and len(live):
        cost = counts[live] + 1
        loads = np.bincount(owners[live], weights=cost, minlength=world)
        assert loads.max() <= math.ceil(cost.sum() / world) + cost.max()
        # runs of the Morton order: walking the live slots in that order the owner never decreases
        order = live[np.argsort(parallel.morton_codes(centres[live]), kind="stable")]
        assert (np.diff(owners[order].astype(int)) >= 0).all()


def test_every_shadow_view_has_one_owner():
    for world in range(1, 17):
        for views in range(0, 9):
            owners = [parallel.shadow_view_owner(v, world) for v in range(views)]
            assert all(0 <= o < world for o in owners) and owners == [v % world for v in range(views)]


def _rows_of_points(points, view_proj, height):
    """Pixel rows of world-space points that are in front of the eye plane and inside the view volume's sides (what a rasterised
    fragment of geometry through that point could touch), computed in f64 from the column-major f32[16] view_proj."""
    m = np.asarray(view_proj, dtype=np.float64).reshape(4, 4).T
    clip = np.c_[points, np.ones(len(points))] @ m.T
    x, y, w = clip[:, 0], clip[:, 1], clip[:, 3]
    ok = (w > 1e-6) & (np.abs(x) <= w) & (np.abs(y) <= w)
    return ((1.0 - y[ok] / w[ok]) * 0.5 * height)


@settings(max_examples=80, deadline=None, derandomize=True, database=None)
@given(st.integers(min_value=0, max_value=2 ** 31 - 1), st.integers(min_value=2, max_value=8), st.booleans(), st.booleans())
def test_row_extents_of_the_spatial_split_are_conservative(seed, world, left_handed, orthographic):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 200))
    centres = rng.uniform(-30, 30, (n, 3))
    radii = rng.uniform(0.1, 4.0, n)
    counts = rng.integers(1, 500, n)
    owners = parallel.partition_objects_spatial(centres, counts, world)
    bounds = parallel.partition_bounds(owners, centres, radii, counts, world)
    height = int(rng.integers(16, 2200))
    hand = oh.LEFT if left_handed else oh.RIGHT
    eye = tuple(rng.uniform(-40, 40, 3))
    target = tuple(rng.uniform(-10, 10, 3))
    look = oh.look_at_lh if left_handed else oh.look_at_rh
    proj = ("orthographic", (25.0, 14.0, 200.0)) if orthographic else ("perspective", float(rng.uniform(20, 120)), 0.1)
    cam = oh.CameraState(look(eye, target, (0, 1, 0)), proj, hand, np.float32(16.0 / 9.0))
    vp = np.asarray(cam.view_proj, dtype=np.float32).reshape(16)
    extents = parallel.partition_row_extents(bounds, vp, height)
    assert len(extents) == world
    for r in range(world):
        y0, y1 = extents[r]
        assert 0 <= y0 <= y1 <= height
        sel = np.flatnonzero(owners == r)
        if not len(sel):
            continue
        # points on and inside the bounding spheres of the partition's objects (its geometry lies inside them, object.rs:268-269)
        k = 40
        d = rng.normal(size=(len(sel), k, 3))
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        pts = (centres[sel, None, :] + d * (radii[sel, None, None] * rng.uniform(0, 1, (len(sel), k, 1)))).reshape(-1, 3)
        rows = _rows_of_points(pts, vp, height)
        rows = rows[(rows >= 0) & (rows < height)]
        if len(rows):
            assert y0 <= math.floor(rows.min()) and math.ceil(rows.max()) <= y1, (extents[r], rows.min(), rows.max())
