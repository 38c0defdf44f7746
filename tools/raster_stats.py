import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
import rend3_amd as r3, rend3_amd.scenes
from rend3_amd import _ffi
import bench
r = r3.Renderer(r3.host.RIGHT, np.float32(3840/2160))
info = r3.scenes.bistro_like(r, r3.host, r3.material_record)
base = r3.BaseRenderGraph(r)
for k in range(4):
    r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
    r.render(3840,2160, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base)
st = np.zeros(64, dtype=np.uint32); r.lib.r3n_readback_raster_stats(r.ctx, _ffi.ptr(st)); print("big items per forward call:", st[:14])
calls = np.zeros((6,5), dtype=np.uint32); r.lib.r3n_readback_draw_calls(r.ctx, 0xFFFFFFFF, _ffi.ptr(calls)); print("viewport calls (pred/resid tris):", calls[:,0]//3)
for si in range(4):
    r.lib.r3n_readback_draw_calls(r.ctx, si, _ffi.ptr(calls)); print("shadow", si, calls[:3,0]//3)
