"""Kernel time of the bf16 x 3 split product (skm_gemm_split) at the shapes of the TV E-step: the L build
(sessions x C) (C x P) and the A accumulation (C x sessions) (sessions x P), P = R (R + 1) / 2 = 80 200.
   python tools/time_gemm_split.py [sessions]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sidekit_amd as sk
eng = sk.default_engine()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def run(M, N, K, what):
    A = torch.randn((M, K), dtype=torch.float32, device=eng.device)
    B = torch.randn((K, N), dtype=torch.float32, device=eng.device)
    C = eng.gemm_split(A, B)
    ref = (A[:64].double() @ B.double())
    err = ((C[:64].double() - ref).abs().max() / ref.abs().max()).item()
    eng.profile(True)
    for _ in range(3):
        eng.gemm_split(A, B)
    torch.cuda.synchronize()
    prof0 = eng.profile_read()
    for _ in range(3):
        eng.gemm_split(A, B, C=C)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    ms = prof["k_gemm_b3"][1] / prof["k_gemm_b3"][0]
    ms0 = prof0["k_gemm_b3"][1] / prof0["k_gemm_b3"][0]
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print("%-8s M=%d N=%d K=%d: C = A B %.3f ms; C += A B %.3f ms  %.1f TF algorithmic (%.2f of the f32 MFMA peak), issued %.2f of the bf16 peak;"
          " split %.3f ms; max err %.1e" % (what, M, N, K, ms0, ms, tf, tf / 157.3, 6 * tf / 2500.0,
                                          prof["k_b3_split"][1] / 3, err), flush=True)


run(S, 80200, 2048, "lbuild")
if os.environ.get("ONLY") != "lbuild":
    run(2048, 80200, S, "aacc")
    run(4096, 4096, 4096, "square")
    run(S, 80200, 1024, "K=1024")
    run(S, 80200, 4096, "K=4096")
