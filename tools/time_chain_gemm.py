"""Is the batched float64 chain of the TV E-step bound by bytes?  The trailing update of the blocked Cholesky
(64 x (R - r0) output per matrix, K = r0, operands inside the R x R matrix) over batches whose matrices fit L2,
the Infinity Cache, or only HBM.   python tools/time_chain_gemm.py"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sidekit_amd as sk
eng = sk.default_engine()
R = 400
def run(nb, r0, dt, kind, reps=20):
    esz = 8 if dt == torch.float64 else 4
    Mb = torch.randn((nb, R, R), dtype=dt, device=eng.device)
    base = Mb.data_ptr()
    bs, N, K = 64, R - r0, r0
    A = base + r0 * esz
    B = base + r0 * esz
    C = base + (r0 * R + r0) * esz
    def go():
        eng._check(eng.lib.skm_gemm(eng.ctx, kind, ctypes.c_void_p(A), ctypes.c_void_p(B), ctypes.c_void_p(C), bs, N, K,
                                    R, R, R, 1, 0, -1.0, 1.0, nb, R * R, R * R, R * R, eng._stream()))
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): go()
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / reps
    fl = 2.0 * bs * N * K * nb
    by = (bs * K + N * K + 2 * bs * N) * esz * nb
    print("%s nb=%5d (%7.1f MB) r0=%3d: %.3f ms  %.1f TFLOP/s  %.2f TB/s minimal traffic" % (
        "f64" if esz == 8 else "f32", nb, nb * R * R * esz / 1e6, r0, dtm * 1e3, fl / dtm / 1e12, by / dtm / 1e12), flush=True)
for dt, kind in ((torch.float64, eng.GEMM_F64_F64), (torch.float32, eng.GEMM_F32_F32)):
    for r0 in (64, 192, 320):
        for nb in (16, 64, 128, 512, 4096):
            run(nb, r0, dt, kind, reps=50 if nb <= 128 else 10)
