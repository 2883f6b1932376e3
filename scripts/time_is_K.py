import sys, time, torch
sys.path.insert(0, "/root/repo")
from nmf_toolbox_amd.engine import Engine
def run(m, n, K, div, path, iters=6):
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    V = torch.rand((n, m), generator=g, device="cuda:0").clamp_(min=1e-6); W = torch.rand((K, m), generator=g, device="cuda:0").clamp_(min=1e-6); H = torch.rand((n, K), generator=g, device="cuda:0").clamp_(min=1e-6)
    e = Engine(V, W, H, divergence=div, path=path, use_dist=False); e.init()
    c = torch.zeros(iters + 2, dtype=torch.float64, device="cuda:0")
    e.iterate(2, c); torch.cuda.synchronize(); t0 = time.perf_counter(); e.iterate(iters, c); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print("%s %dx%d K=%d path %d: %.3f ms / iteration, path_kind %s" % (div, m, n, K, path, dt * 1e3, e.path_kind)); e.close()
for K in (128, 160, 192):
    for path in (0, 1):
        run(8192, 32768, K, "is", path)
