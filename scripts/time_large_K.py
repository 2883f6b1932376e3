"""Dev aid: per-iteration time of nmf at K = 256 / 320 / 512 (8192 x 32768), i.e. the fused kernels against the paths K > 256 takes."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from nmf_toolbox_amd.engine import Engine
def run(m, n, K, div, path, iters=6):
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    V = torch.rand((n, m), generator=g, device="cuda:0").clamp_(min=1e-6); W = torch.rand((K, m), generator=g, device="cuda:0").clamp_(min=1e-6); H = torch.rand((n, K), generator=g, device="cuda:0").clamp_(min=1e-6)
    e = Engine(V, W, H, divergence=div, path=path, use_dist=False); e.init()
    c = torch.zeros(iters + 2, dtype=torch.float64, device="cuda:0")
    e.iterate(2, c); torch.cuda.synchronize(); t0 = time.perf_counter(); e.iterate(iters, c); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    fl = 8.0 * m * n * K if div == "kl" else 6.0 * m * n * K
    print("%s %dx%d K=%d path %d: %.3f ms / iteration, %.1f TFLOP/s by the fused paths' flop count (%.2f of peak), path_kind %s" % (div, m, n, K, path, dt * 1e3, fl / dt / 1e12, fl / dt / 157.3e12, e.path_kind))
    e.close()
Ks = [int(k) for k in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 320, 512]
divs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["kl", "euclidean"]
for K in Ks:
    for div in divs:
        run(8192, 32768, K, div, 0)
