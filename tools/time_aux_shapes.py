"""Cost of the ragged rank dimension (R = 400 = 3 x 128 + 16) in the two rank-sized TV products:
aux = f~ T (M = sessions, N = R, K = 2048 per chunk) and C += e_h' f~ (M = R, N = C D, K = sessions of a batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sidekit_amd as sk
eng = sk.default_engine()
f32 = torch.float32
def run(M, N, K, ta=False, reps=5, kind=None):
    A = torch.randn((K, M) if ta else (M, K), dtype=f32, device=eng.device)
    B = torch.randn((K, N), dtype=f32, device=eng.device)
    kind = eng.GEMM_F32_F64OUT if kind is None else kind
    C = eng.gemm(A, B, transA=ta, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.gemm(A, B, C=C, transA=ta, beta=1.0, kind=kind)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("M=%6d N=%6d K=%5d ta=%d kind=%d: %7.3f ms  %6.1f TFLOP/s" % (M, N, K, ta, kind, dt * 1e3, 2.0 * M * N * K / dt / 1e12), flush=True)
for N in (384, 400, 512):
    run(50000, N, 2048)
for M in (384, 400, 512):
    run(M, 122880, 4096, ta=True)
run(50000, 384, 2048, kind=eng.GEMM_F32_F32)
run(50000, 400, 2048, kind=eng.GEMM_F32S_F32)
run(50000, 400, 2048, kind=eng.GEMM_F32S_F64OUT)
run(50000, 400, 4096, kind=eng.GEMM_F32S_F64OUT)
run(50000, 400, 4096)
run(400, 122880, 4096, ta=True, kind=eng.GEMM_F32S_F64OUT)
run(4096, 80800, 2048, kind=eng.GEMM_F32_F32)
run(4096, 80800, 2048, kind=eng.GEMM_F32S_F32)
