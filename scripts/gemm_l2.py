import sys, torch
sys.path.insert(0, '/root/repo')
from nmf_toolbox_amd import _lib
lib = _lib.load()
dev = "cuda:0"
def run(M, N, Kc, lda, ldb, note, reps=20):
    A = torch.rand((M * Kc,), device=dev); B = torch.rand((Kc * N,), device=dev); Cm = torch.zeros((M * N,), device=dev)
    ws = torch.empty((1 << 30,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        _lib.check(lib.nmfx_gemm_f32(st, 1, 0, M, N, Kc, A.data_ptr(), None, lda, 0, B.data_ptr(), None, ldb, 0, Cm.data_ptr(), M, 0, ws.data_ptr(), ws.numel()))
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-34s %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)" % (note, ms, 2.0 * M * N * Kc / ms / 1e9, 2.0 * M * N * Kc / ms / 1e9 / 157.3), flush=True)
M, N, Kc = 512, 16384, 4096
run(M, N, Kc, Kc, Kc, "Q-GEMM as is")
run(M, N, Kc, Kc, 0, "B: every column the same 16 KB")
run(M, N, Kc, 0, Kc, "A: every row the same 16 KB")
run(M, N, Kc, 0, 0, "both operands one line each")
M, N, Kc = 8192, 8192, 8192
run(M, N, Kc, Kc, Kc, "8192^3 TN as is", 5)
run(M, N, Kc, 0, 0, "8192^3 both operands one line", 5)
