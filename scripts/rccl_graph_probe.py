import os, sys, time, torch, torch.distributed as dist
def P(*a):
    print(*a, flush=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
P("init"); dist.init_process_group("nccl", device_id=torch.device("cuda", 0)); P("init done")
n = 4 << 20
a = torch.arange(n, device="cuda", dtype=torch.float32); b = torch.zeros_like(a); g = torch.ones(1 << 16, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        dist.all_to_all_single(b, a); dist.all_reduce(g)
torch.cuda.synchronize(); P("eager collectives ok"); time.sleep(0.3)
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        for i in range(4):
            a.mul_(1.0001)
            dist.all_to_all_single(b, a)
            b.add_(1.0)
            dist.all_reduce(g)
            w = dist.all_to_all_single(a, b, async_op=True)
            g.mul_(0.5)
            w.wait()
    P("captured")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    P("RCCL-in-graph OK: %.1f us per replay (4 x [a2a 16MB, allreduce, a2a 16MB]); a[5]=%g g[0]=%g" % (dt / 200 * 1e6, float(a[5]), float(g[0])))
    time.sleep(1.0)
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    P("second burst ok")
except Exception as e:
    P("RCCL-in-graph FAILED:", type(e).__name__, str(e)[:400])
dist.destroy_process_group()
