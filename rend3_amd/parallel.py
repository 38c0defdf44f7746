"""Multi-GPU object-range sharding (the reference is single-device, so this layer is new).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests).  Every rank holds
the full mesh/object/material buffers.  The VIEWPORT's opaque / cutout objects are split into contiguous object-slot
ranges (balanced by triangle count); the SHADOW VIEWS are split by view -- view v is rendered whole by rank v mod N and by
nobody else.  Exchange steps per frame (reverse-Z: nearest == max; the 64-bit visibility key orders by depth first, so
MAX of keys is exactly the depth-tested composite):

  "shadow": every view's atlas rectangle is broadcast from its owner        (16 MiB per 2048^2 view, received once)
  "pass1" : element-wise MAX all-reduce of the pass-1 DEPTH plane (f32)     (33 MB at 4K: every rank then culls against the
                                                   GLOBAL Hi-Z, which keeps the per-triangle visible sets bit-exact; only the
                                                   depth has to be global here -- the keys stay local until pass 2.  Multisampled
                                                   targets exchange the keys instead: min-over-samples and max-over-ranks do not
                                                   commute)
  "pass2" : element-wise MAX of the 64-bit visibility keys, as a reduce-scatter to the row owners when the rows split
            evenly (nobody needs the other ranks' rows any more)

Screen rows are split across ranks for resolve + tonemap and the Rgba8 rows are all-gathered.
"""
import numpy as np


def partition_objects(tri_counts, world_size):
    """Contiguous object-slot ranges balanced by triangle count (prefix sum of index_count/3).
    Returns [(begin, end)] * world_size covering [0, len(tri_counts))."""
    tri_counts = np.asarray(tri_counts, dtype=np.int64)
    n = len(tri_counts)
    if world_size <= 1:
        return [(0, n)]
    prefix = np.concatenate([[0], np.cumsum(tri_counts + 1)])  # +1: empty objects still cost a slot
    total = prefix[-1]
    cuts = [0]
    for r in range(1, world_size):
        cuts.append(int(np.searchsorted(prefix, total * r / world_size, side="left")))
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]


def morton_codes(points, bits=10):
    """Interleaved-bit (Z-order) codes of 3-D points quantised to `bits` bits per axis over their bounding box."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros(0, dtype=np.uint64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    q = np.minimum(((p - lo) / span * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    code = np.zeros(len(p), dtype=np.uint64)
    for b in range(bits):
        for axis in range(3):
            code |= ((q[:, axis] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + axis)
    return code


def partition_objects_spatial(centres, tri_counts, world_size):
    """Owner rank per object slot (uint8): the slots in Morton order of their bounding-sphere centres, cut into `world_size`
    runs of equal triangle load -- every rank gets a compact region of the world, hence (for most cameras) of the screen, which is
    what lets the exchanges skip the rows a rank cannot have touched (partition_row_extents).  Slots without triangles
    (free / disabled) go to rank 0."""
    tri_counts = np.asarray(tri_counts, dtype=np.int64)
    owners = np.zeros(len(tri_counts), dtype=np.uint8)
    live = np.flatnonzero(tri_counts > 0)
    if world_size <= 1 or len(live) == 0:
        return owners
    order = live[np.argsort(morton_codes(np.asarray(centres, dtype=np.float64)[live]), kind="stable")]
    load = np.cumsum(tri_counts[order] + 1)
    cuts = np.searchsorted(load, load[-1] * np.arange(1, world_size) / world_size, side="left")
    owners[order] = np.searchsorted(cuts, np.arange(len(order)), side="right").astype(np.uint8)
    return owners


def partition_bounds(owners, centres, radii, tri_counts, world_size, pieces=64):
    """World-space boxes of every rank's partition: its objects (in Morton order) cut into up to `pieces` runs, each run's
    bounding-sphere AABB (lo, hi).  Small boxes instead of one per rank: a partition that reaches behind the camera only loses
    the boxes that really straddle the eye plane inside the view to "whole target" (partition_row_extents)."""
    centres = np.asarray(centres, dtype=np.float64).reshape(-1, 3)
    radii = np.asarray(radii, dtype=np.float64).reshape(-1)
    out = []
    for r in range(world_size):
        sel = np.flatnonzero((np.asarray(owners) == r) & (np.asarray(tri_counts) > 0))
        boxes = []
        if len(sel):
            sel = sel[np.argsort(morton_codes(centres[sel]), kind="stable")]
            for part in np.array_split(sel, min(pieces, len(sel))):
                boxes.append(((centres[part] - radii[part, None]).min(axis=0), (centres[part] + radii[part, None]).max(axis=0)))
        out.append(boxes)
    return out


def partition_row_extents(bounds, view_proj, height, pad=2):
    """Conservative pixel-row extent [y0, y1) of every partition for a camera, from its boxes (partition_bounds): a box wholly
    outside one of the view volume's side / near planes (clip-space tests on its eight corners through view_proj, column-major
    f32[16]) contributes nothing; a box with a corner at or behind the eye plane makes the extent the whole target; any other
    box contributes the rows of its projected corners.  Every rank computes every rank's extent from the same replicated data,
    so the exchanges' split sizes are known on the host without a message and without reading anything back from the GPU.
    Outside its extent a rank's keys are the clear value, which never wins a MAX."""
    m = np.asarray(view_proj, dtype=np.float64).reshape(4, 4).T  # row-major
    out = []
    for boxes in bounds:
        y0, y1 = height, 0
        for lo, hi in boxes:
            corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
            clip = corners @ m.T
            if not np.isfinite(clip).all():
                y0, y1 = 0, height
                break
            x, y, z, w = clip[:, 0], clip[:, 1], clip[:, 2], clip[:, 3]
            # outside: all corners beyond the same plane of -w <= x <= w, -w <= y <= w, z <= w (reverse-Z near: z > w is in front
            # of the near plane for the infinite projection; orthographic cameras have w == 1).  The margin keeps it conservative.
            e = 1e-4 * (np.abs(w) + 1.0)
            if (x < -w - e).all() or (x > w + e).all() or (y < -w - e).all() or (y > w + e).all() or (z > w + e).all():
                continue
            if (w <= 1e-6).any():
                y0, y1 = 0, height
                break
            rows = (1.0 - y / w) * 0.5 * height  # the rasteriser's (w - y) * H / 2
            y0 = min(y0, int(np.clip(np.floor(rows.min()) - pad, 0, height)))
            y1 = max(y1, int(np.clip(np.ceil(rows.max()) + pad, 0, height)))
        out.append((y0, max(y0, y1)) if y1 > y0 else (0, 0))
    return out


def rows_alltoall_max_(buf2d, extents, bands, rank, world_size, group=None):
    """Element-wise MAX reduction of `buf2d` (rows x row elements; visibility keys as int64, or the f32 depth plane) onto the band
    owners, moving only the rows a rank can have touched: rank r sends band j the rows band_j & extent_r -- its extent's rows
    are contiguous in the buffer and the bands are in rank order, so the send buffer is a VIEW of the buffer and the split sizes
    follow from the extents (no packing, no host read-back); band owner j MAX-merges what it receives.  Afterwards rank j's band
    rows hold the reduced values.  `extents`, `bands`: [(y0, y1)] per rank.  Returns the bytes this rank sent."""
    import torch
    import torch.distributed as dist
    if not _device_collectives(buf2d, group):
        return _staged(rows_alltoall_max_, buf2d, extents, bands, rank, world_size, group)
    width = buf2d.shape[1]
    y0, y1 = extents[rank]
    def inter(a, b):
        lo, hi = max(a[0], b[0]), min(a[1], b[1])
        return (lo, max(lo, hi))
    send_rows = [inter(bands[j], (y0, y1)) for j in range(world_size)]
    recv_rows = [inter(bands[rank], extents[s]) for s in range(world_size)]
    in_split = [(b - a) * width for a, b in send_rows]
    out_split = [(b - a) * width for a, b in recv_rows]
    lo = min((a for a, b in send_rows if b > a), default=0)
    send = buf2d.reshape(-1)[lo * width: lo * width + sum(in_split)]
    recv = torch.empty(sum(out_split), dtype=buf2d.dtype, device=buf2d.device)
    dist.all_to_all_single(recv, send.contiguous(), out_split, in_split, group=group)
    at = 0
    for s in range(world_size):
        a, b = recv_rows[s]
        n = (b - a) * width
        if n and s != rank:
            dst = buf2d[a:b].reshape(-1)
            torch.maximum(dst, recv[at:at + n], out=dst)
        at += n
    return (sum(in_split) - in_split[rank]) * buf2d.element_size()


def row_ranges(height, world_size):
    base, rem = divmod(height, world_size)
    out, y = [], 0
    for r in range(world_size):
        h = base + (1 if r < rem else 0)
        out.append((y, y + h))
        y += h
    return out


def _device_collectives(tensor, group=None):
    """True when collectives can run on `tensor` where it is: CPU tensors always; device tensors on the RCCL ("nccl") backend.
    Device tensors on another backend (gloo: the one-GPU two-process test) are staged through the host by the callers."""
    import torch.distributed as dist
    return (not tensor.is_cuda) or dist.get_backend(group) == "nccl"


def _staged(fn, tensor, *args, **kw):
    """Run the in-place collective `fn(tensor, ...)` on a host copy of a device tensor and copy the result back."""
    host = tensor.cpu()
    out = fn(host, *args, **kw)
    tensor.copy_(host)
    return out


def allreduce_max_(tensor, group=None):
    """In-place element-wise MAX all-reduce (the only data-path collective of the pipeline).  Visibility keys are
    passed as int64: depth lies in [0,1] so bit 63 is never set and signed MAX equals unsigned MAX."""
    import torch.distributed as dist
    if not _device_collectives(tensor, group):
        _staged(allreduce_max_, tensor, group)
        return tensor
    dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group)
    return tensor


def reduce_scatter_max_rows_(tensor, rank, world_size, group=None):
    """Element-wise MAX over ranks, delivered only where it is needed: afterwards rank r's r-th equal chunk of `tensor`
    holds the reduced values (the rows it resolves); the other chunks keep this rank's own partial values.  Half the
    traffic of an all-reduce.  Backends without reduce-scatter (gloo) fall back to the all-reduce."""
    import torch
    import torch.distributed as dist
    assert tensor.numel() % world_size == 0
    chunk = tensor.numel() // world_size
    if not _device_collectives(tensor, group):
        _staged(reduce_scatter_max_rows_, tensor, rank, world_size, group)
        return tensor
    if tensor.is_cuda and hasattr(dist, "reduce_scatter_tensor"):
        out = torch.empty(chunk, dtype=tensor.dtype, device=tensor.device)
        dist.reduce_scatter_tensor(out, tensor, op=dist.ReduceOp.MAX, group=group)
        tensor[rank * chunk:(rank + 1) * chunk].copy_(out)
    else:
        dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group)
    return tensor


def _scratch(cache, key, like):
    """A receive buffer shaped like `like`, kept in `cache` (a dict the caller owns, e.g. one per Exchange) across frames."""
    import torch
    if cache is None:
        return torch.empty_like(like)
    t = cache.get(key)
    if t is None or t.shape != like.shape or t.dtype != like.dtype or t.device != like.device:
        t = cache[key] = torch.empty_like(like)
    return t


def direct_reduce_scatter_max_(tensor, rank, world_size, group=None, cache=None):
    """The reduce-scatter of reduce_scatter_max_rows_ as ONE all-to-all + a local MAX: rank r sends its partial chunk j straight to
    rank j -- over the full xGMI mesh that is one point-to-point transfer per link, all seven links of a GPU busy at once, each
    carrying 1/N of the buffer -- and reduces the N chunks it received on its own.  Same result as the collective (MAX is exact and
    order-free).  Opt-in (R3N_EXCHANGE_DIRECT=1 / Exchange(direct=True)): not measured on a multi-GPU node yet."""
    import torch
    import torch.distributed as dist
    assert tensor.numel() % world_size == 0
    chunk = tensor.numel() // world_size
    if not _device_collectives(tensor, group):
        _staged(direct_reduce_scatter_max_, tensor, rank, world_size, group)
        return tensor
    recv = _scratch(cache, ("a2a", tensor.dtype), tensor)  # the full-size receive buffer is not reallocated every frame
    dist.all_to_all_single(recv, tensor, group=group)
    tensor[rank * chunk:(rank + 1) * chunk].copy_(recv.view(world_size, chunk).amax(dim=0))
    return tensor


def direct_allreduce_max_(tensor, rank, world_size, group=None, cache=None):
    """MAX all-reduce as direct_reduce_scatter_max_ + an all-gather of the reduced chunks (buffers whose size divides by N)."""
    import torch.distributed as dist
    if not _device_collectives(tensor, group):
        _staged(direct_allreduce_max_, tensor, rank, world_size, group)
        return tensor
    direct_reduce_scatter_max_(tensor, rank, world_size, group, cache)
    chunk = tensor.numel() // world_size
    mine = tensor[rank * chunk:(rank + 1) * chunk].clone()
    if hasattr(dist, "all_gather_into_tensor") and tensor.is_cuda:
        dist.all_gather_into_tensor(tensor, mine, group=group)
    else:
        dist.all_gather([tensor[r * chunk:(r + 1) * chunk] for r in range(world_size)], mine, group=group)
    return tensor


def allgather_rows_(full, rank, world_size, group=None):
    """`full` is the whole flat image; rank r has valid data in its r-th equal chunk.  Gathers every chunk in place."""
    import torch.distributed as dist
    assert full.numel() % world_size == 0
    if not _device_collectives(full, group):
        _staged(allgather_rows_, full, rank, world_size, group)
        return full
    chunk = full.numel() // world_size
    mine = full[rank * chunk:(rank + 1) * chunk].clone()
    if hasattr(dist, "all_gather_into_tensor") and full.is_cuda:
        dist.all_gather_into_tensor(full, mine, group=group)
    else:
        parts = [full[r * chunk:(r + 1) * chunk] for r in range(world_size)]
        dist.all_gather(parts, mine, group=group)
    return full


def shadow_view_owner(view, world_size):
    """Shadow views are sharded by view: view v belongs to rank v mod N (it draws every object there, nothing elsewhere)."""
    return int(view) % int(world_size)


def exchange_shadow_views_(atlas2d, shadows, rank, world_size, group=None, clock=None):
    """atlas2d: (atlas_h, atlas_w) f32 tensor (a view of the atlas buffer); shadows: [{"offset": (x, y), "size": s}].
    Every view's rectangle travels once: contiguous staging copy on the owner, broadcast, copy into the rectangle elsewhere."""
    import torch.distributed as dist
    for v, sh in enumerate(shadows):
        x, y, size = int(sh["offset"][0]), int(sh["offset"][1]), int(sh["size"])
        owner = shadow_view_owner(v, world_size)
        rect = atlas2d[y:y + size, x:x + size]
        stage = rect.contiguous() if rank == owner else rect.new_empty((size, size))
        src = owner if group is None else dist.get_global_rank(group, owner)
        if _device_collectives(stage, group):
            dist.broadcast(stage, src=src, group=group)
        else:
            host = stage.cpu()
            dist.broadcast(host, src=src, group=group)
            stage.copy_(host)
        if rank != owner:
            rect.copy_(stage)
    return atlas2d


class _DevArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, count, dtype_str, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, (int(count),), dtype_str), device=device)


class Exchange:
    """The callable BaseRenderGraph.add_to_graph(exchange=...) expects, over torch.distributed.  `timings` (ms per call
    site, HIP events on the context's stream) is filled when `timed` is set: bench.py --gpus N reports it."""

    _rows_groups = {}  # one extra communicator per (process, parent group), however many Exchange objects are made

    def __init__(self, renderer, device, group=None, timed=False, direct=None, rows_group=None):
        import ctypes
        import os
        self.direct = (os.environ.get("R3N_EXCHANGE_DIRECT", "0") == "1") if direct is None else bool(direct)
        import torch
        import torch.distributed as dist
        self.dist, self.torch, self.group = dist, torch, group
        self.r = renderer
        self.device = device
        self.stream = torch.cuda.ExternalStream(renderer.lib.r3n_stream(renderer.ctx), device=device)
        self._ct = ctypes
        self.rows_equal = False  # set by the caller when every rank resolves an equal, contiguous block of rows
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # the row all-gather follows the resolve on the resolve's stream and has its own communicator: on the frame's main
        # communicator it would queue in front of the NEXT frame's pass-1 exchange and hold that frame's culling back until this
        # frame's resolve is done (collectives of one communicator run in issue order)
        # (dist.new_group is collective over the default group: made once per parent group and reused; pass `rows_group` to supply one)
        if rows_group is None:
            key = id(group) if group is not None else None
            if key not in Exchange._rows_groups:
                Exchange._rows_groups[key] = dist.new_group(ranks=list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group))
            rows_group = Exchange._rows_groups[key]
        self.group_rows = rows_group
        # the shadow-view exchange runs on a shadow lane's stream beside the viewport's passes (r3n_exchange_shadow_stream); for
        # the same reason as the rows it has a communicator of its own: on the main one the pass-1 exchange would queue behind it
        skey = ("shadow", id(group) if group is not None else None)
        if skey not in Exchange._rows_groups:
            Exchange._rows_groups[skey] = dist.new_group(ranks=list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group))
        self.group_shadow = Exchange._rows_groups[skey]
        self.by_rows = False  # set_row_sharding: sort-first
        self.sparse = None    # spatial partition: dict(bounds=[...]) -- the exchanges then move only the rows a rank can have touched
        self.full_extent_frames = 0
        self._streams = {}
        self._scratch = {}    # receive buffers of the direct reductions, reused across frames
        self.timed = timed
        self.events = []      # (what, start event, end event)
        self.bytes = {}       # what -> bytes this rank hands to the collective per frame

    def set_spatial_partition(self, owners, bounds):
        """Shard the viewport's objects by owner byte (partition_objects_spatial) instead of by slot range, and let the pass-1 /
        pass-2 exchanges skip the rows outside a rank's screen extent (partition_row_extents of `bounds` = partition_bounds).
        The first frame afterwards still exchanges whole targets: its pass 1 draws the PREVIOUS partition's predicted lists."""
        owners = np.ascontiguousarray(owners, dtype=np.uint8)
        r = self.r
        r._check(r.lib.r3n_set_object_owners(r.ctx, owners.ctypes.data, len(owners), self.rank), "r3n_set_object_owners")
        # the bounds describe the world as it is NOW: any later edit (add / move / remove an object, new joint matrices) makes
        # them stale -- a rank that rasterises outside its old extent would not send those rows -- so the exchanges fall back
        # to whole targets until a new partition is set (Renderer.world_version)
        self.sparse = dict(bounds=bounds, version=r.world_version)
        self.full_extent_frames = 1

    def set_row_sharding(self, row_begin, row_end):
        """Sort-first: this rank culls and draws EVERY object but rasterises only its rows (r3n_set_shard_mode(R3N_SHARD_ROWS)).
        Its rows of the pass-1 depth plane are then final: the pass-1 exchange becomes an all-gather of the row bands (every rank
        culls against the whole Hi-Z pyramid, so the visible sets are the unsharded ones on every rank), and pass 2 needs no
        exchange at all -- the 8 B per pixel key reduction of the object-sharded scheme disappears."""
        r = self.r
        r._check(r.lib.r3n_set_shard_mode(r.ctx, 1), "r3n_set_shard_mode")
        r._check(r.lib.r3n_set_row_range(r.ctx, int(row_begin), int(row_end)), "r3n_set_row_range")
        self.by_rows = True

    def _row_extents(self, renderer, height):
        if self.full_extent_frames > 0 or renderer.world_version != self.sparse.get("version"):
            return [(0, height)] * self.world
        return partition_row_extents(self.sparse["bounds"], renderer.current_view_proj(), height)

    def owns_shadow_view(self, view):
        return shadow_view_owner(view, self.world) == self.rank

    def assign_shadow_views(self, n_views):
        """Shadow views by view: the ones this rank owns draw EVERY object slot, the others are not rendered here at all."""
        for v in range(n_views):
            if self.owns_shadow_view(v):
                self.r.set_camera_object_range(v, 0, 0xFFFFFFFE)

    def _buffers(self):
        ct = self._ct
        vis, vis_n, atlas, atlas_n = ct.c_void_p(), ct.c_uint64(), ct.c_void_p(), ct.c_uint64()
        r = self.r
        r._check(r.lib.r3n_exchange_buffers(r.ctx, ct.byref(vis), ct.byref(vis_n), ct.byref(atlas), ct.byref(atlas_n)),
                 "r3n_exchange_buffers")
        return vis.value, vis_n.value, atlas.value, atlas_n.value

    def __call__(self, what, renderer, ev=None, samples=1):
        dist, torch = self.dist, self.torch
        self._height = renderer.current_resolution()[1]
        if what == "shadow":
            # on the shadow lane's stream, not the main one: the viewport's passes (which never read the atlas) do not wait for it
            if ev is None or not ev.shadows:
                return
            ct = self._ct
            atlas, atlas_n, sp = ct.c_void_p(), ct.c_uint64(), ct.c_void_p()
            r = self.r
            r._check(r.lib.r3n_exchange_shadow_stream(r.ctx, ct.byref(atlas), ct.byref(atlas_n), ct.byref(sp)), "r3n_exchange_shadow_stream")
            stream = self._streams.get(sp.value)
            if stream is None:
                stream = self._streams[sp.value] = torch.cuda.ExternalStream(sp.value, device=self.device)
            with torch.cuda.stream(stream):
                t0 = t1 = None
                if self.timed:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(stream)
                aw, ah = ev.shadow_target_size
                a2 = device_tensor(atlas.value, atlas_n.value, "<f4", self.device).view(ah, aw)
                exchange_shadow_views_(a2, ev.shadows, dist.get_rank(self.group_shadow), self.world, self.group_shadow)
                self.bytes[what] = sum(4 * int(sh["size"]) ** 2 for sh in ev.shadows)
                if self.timed:
                    t1.record(stream)
                    self.events.append((what, t0, t1))
            return
        # the context's stream is made torch's current stream, so the collective is ordered after the kernels
        # already enqueued on it and the kernels enqueued next wait for the collective
        with torch.cuda.stream(self.stream):
            t0 = t1 = None
            if self.timed:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.stream)
            if self.by_rows:
                if what == "pass1":
                    ct = self._ct
                    r = self.r
                    if samples == 1:
                        plane, n = ct.c_void_p(), ct.c_uint64()
                        r._check(r.lib.r3n_exchange_depth(r.ctx, ct.byref(plane), ct.byref(n)), "r3n_exchange_depth")
                        t = device_tensor(plane.value, n.value, "<f4", self.device)
                    else:  # multisampled targets: Hi-Z reads the keys
                        vis, vis_n, _atlas, _atlas_n = self._buffers()
                        t = device_tensor(vis, vis_n, "<i8", self.device)
                    if self.rows_equal:
                        allgather_rows_(t, self.rank, self.world, self.group)
                        self.bytes[what] = t.numel() * t.element_size() // self.world
                    else:  # ragged bands: rows a rank does not own hold the clear value, which never wins a MAX
                        allreduce_max_(t, self.group)
                        self.bytes[what] = t.numel() * t.element_size()
                else:
                    self.bytes[what] = 0  # pass 2: every rank's rows are complete
                if self.timed:
                    t1.record(self.stream)
                    self.events.append((what, t0, t1))
                return
            if what == "pass1" and samples == 1:
                ct = self._ct
                plane, n = ct.c_void_p(), ct.c_uint64()
                r = self.r
                r._check(r.lib.r3n_exchange_depth(r.ctx, ct.byref(plane), ct.byref(n)), "r3n_exchange_depth")
                t = device_tensor(plane.value, n.value, "<f4", self.device)  # depth >= 0: float MAX
                if self.sparse is not None and self.rows_equal:
                    # reduce onto the row-band owners moving only the rows each rank can have touched, then every rank gets
                    # every merged band (the Hi-Z cull needs the whole plane)
                    h = self._height
                    sent = rows_alltoall_max_(t.view(h, -1), self._row_extents(renderer, h), row_ranges(h, self.world), self.rank, self.world, self.group)
                    allgather_rows_(t, self.rank, self.world, self.group)
                    self.bytes[what] = sent + 4 * n.value // self.world
                    if self.timed:
                        t1.record(self.stream)
                        self.events.append((what, t0, t1))
                    return
                if self.direct and n.value % self.world == 0:
                    direct_allreduce_max_(t, self.rank, self.world, self.group, self._scratch)
                else:
                    allreduce_max_(t, self.group)
                self.bytes[what] = 4 * n.value
            elif what == "pass2" and self.rows_equal:
                # only the rows this rank resolves have to be complete from here on
                vis, vis_n, atlas, atlas_n = self._buffers()
                t = device_tensor(vis, vis_n, "<i8", self.device)
                if self.sparse is not None:
                    h = self._height
                    self.bytes[what] = rows_alltoall_max_(t.view(h, -1), self._row_extents(renderer, h), row_ranges(h, self.world), self.rank, self.world, self.group)
                    self.full_extent_frames = max(0, self.full_extent_frames - 1)
                    if self.timed:
                        t1.record(self.stream)
                        self.events.append((what, t0, t1))
                    return
                if self.direct:
                    direct_reduce_scatter_max_(t, self.rank, self.world, self.group, self._scratch)
                else:
                    reduce_scatter_max_rows_(t, self.rank, self.world, self.group)
                self.bytes[what] = 8 * vis_n
            else:
                vis, vis_n, atlas, atlas_n = self._buffers()
                allreduce_max_(device_tensor(vis, vis_n, "<i8", self.device), self.group)
                self.bytes[what] = 8 * vis_n
            if self.timed:
                t1.record(self.stream)
                self.events.append((what, t0, t1))

    def gather_rows(self, width, height, world_size):
        """All-gather the Rgba8 rows each rank tonemapped (equal row counts required), on the stream the resolve ran on."""
        import ctypes
        dist, torch = self.dist, self.torch
        assert height % world_size == 0
        out, nbytes, sp = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p()
        r = self.r
        r._check(r.lib.r3n_output_buffer_async(r.ctx, ctypes.byref(out), ctypes.byref(nbytes), ctypes.byref(sp)), "r3n_output_buffer_async")
        stream = self._streams.get(sp.value)
        if stream is None:
            stream = self._streams[sp.value] = torch.cuda.ExternalStream(sp.value, device=self.device)
        with torch.cuda.stream(stream):
            t0 = t1 = None
            if self.timed:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(stream)
            full = device_tensor(out.value, nbytes.value, "|u1", self.device)
            allgather_rows_(full, dist.get_rank(self.group_rows), world_size, self.group_rows)
            self.bytes["rows"] = nbytes.value
            if self.timed:
                t1.record(stream)
                self.events.append(("rows", t0, t1))
        r._check(r.lib.r3n_output_work_enqueued(r.ctx), "r3n_output_work_enqueued")

    def drain_timings(self):
        """ms per exchange site summed over the recorded calls (synchronises the device), and the number of frames' worth."""
        self.torch.cuda.synchronize(self.device)
        out = {}
        for what, t0, t1 in self.events:
            out[what] = out.get(what, 0.0) + t0.elapsed_time(t1)
        self.events = []
        return out
