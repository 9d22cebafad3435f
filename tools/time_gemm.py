import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sidekit_amd as sk
eng = sk.default_engine()
def run(M, N, K, kind, dt, ta=False, tb=False):
    A = torch.randn((K, M) if ta else (M, K), dtype=dt, device=eng.device)
    B = torch.randn((N, K) if tb else (K, N), dtype=dt, device=eng.device)
    C = eng.gemm(A, B, transA=ta, transB=tb, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.gemm(A, B, C=C, transA=ta, transB=tb, kind=kind)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / 5
    print("M=%d N=%d K=%d kind=%d ta=%d tb=%d: %.3f ms  %.1f TFLOP/s" % (M, N, K, kind, ta, tb, dtm * 1e3, 2.0 * M * N * K / dtm / 1e12))
f32, f64 = torch.float32, torch.float64
run(4096, 4096, 4096, 0, f32)
run(4096, 4096, 4096, 1, f32)
run(4096, 4096, 4096, 0, f32, ta=True)
run(4096, 4096, 4096, 0, f32, tb=True)
run(4096, 4096, 4096, 2, f64)
run(1024, 86400, 2048, 1, f32)
run(2049, 86400, 1024, 4, f32, ta=True)
run(4096, 400, 122880, 1, f32)
