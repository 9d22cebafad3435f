"""Cycle stamps of one tile of k_estep_b3 (library built with -DB3E_TRACE, SKM_LIB=tools/ab/libskm_b3trace.so):
0 tile start, 1 end of logits MFMAs, 2 before barrier, 3 after barrier, 4 before the statistics groups, 5 end of tile."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SKM_SPLIT"] = "bf16x3"
import torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
eng = sk.default_engine()
C, D, T = 512, 60, 10_000_000
ubm = random_ubm(sk, C, D)
packed = eng.pack_stats(ubm.w, ubm.mu, ubm.invcov)
X = synth_frames_device(torch, T, D, 1, eng.device)
buf = torch.zeros(16, dtype=torch.int64, device=eng.device)
eng.lib.skm_b3_trace_buffer.argtypes = [ctypes.c_void_p]
eng.lib.skm_b3_trace_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3):
    eng.estep(X, packed)
torch.cuda.synchronize()
t = buf.cpu().numpy()
names = ["logits", "lse+exchange write", "barrier", "combine + split ct0 + loads", "statistics"]
print("tile total", t[5] - t[0])
for i, n in enumerate(names):
    print("%-28s %6d" % (n, t[i + 1] - t[i]))
