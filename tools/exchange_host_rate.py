import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
import numpy as np, torch, torch.distributed as dist
import rend3_amd as r3, rend3_amd.scenes as S, rend3_amd.parallel as P
import bench
dev=torch.device("cuda:0"); torch.cuda.set_device(dev)
ORDER = os.environ.get("X_ORDER", "nccl_first")
if ORDER == "nccl_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
r = r3.Renderer(r3.host.RIGHT, np.float32(bench.WIDTH)/np.float32(bench.HEIGHT))
info = S.bistro_like(r, r3.host, r3.material_record, textured=True)
base = r3.BaseRenderGraph(r)
if ORDER != "nccl_first":
    if ORDER == "renderer_first":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
ex = P.Exchange(r, dev); ex.assign_shadow_views(len(r.dir_lights)); ex.rows_equal=True
r._check(r.lib.r3n_set_row_range(r.ctx, 0, bench.HEIGHT), "rows")
def frame(k, e):
    r.set_camera_data(bench.camera_path(r3.host, info["camera"][0], k), info["camera"][1])
    r.render(bench.WIDTH, bench.HEIGHT, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base, exchange=e)
    if e is not None: e.gather_rows(bench.WIDTH, bench.HEIGHT, 1)
for e in (None, ex):
    for k in range(10): frame(k, e)
    torch.cuda.synchronize()
    K=100; t0=time.perf_counter()
    for k in range(K): frame(10+k, e)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("exchange" if e else "plain", f"enqueue {1e3*(t1-t0)/K:.3f} ms/frame, until done {1e3*(t2-t0)/K:.3f}")
dist.destroy_process_group()
