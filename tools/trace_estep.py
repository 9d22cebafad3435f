"""Cycle stamps of one tile of the fused E-step kernel (needs the -DSKM_TRACE build: SKM_LIB=.../libskm_trace.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
C = int(os.environ.get("C", 512)); T = int(os.environ.get("T", 10_000_000)); D = 60
eng = sk.default_engine()
X = synth_frames_device(torch, T, D, 1234, eng.device)
ubm = random_ubm(sk, C, D)
packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
buf = torch.zeros(8 * 16, dtype=torch.int64, device=eng.device)
eng.lib.skm_trace_buffer.argtypes = [ctypes.c_void_p]
eng.lib.skm_trace_buffer.restype = None
eng.estep(X, packed)
torch.cuda.synchronize()
eng.lib.skm_trace_buffer(ctypes.c_void_p(buf.data_ptr()))
eng.estep(X, packed)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(8, 16)
tt = t.copy(); tt[:, 12] = 0
base = tt[tt > 0].min()
names = ["start", "logits", "lse_part", "staged", "bar/stash", "exp", "stats", "fence", "bar(late)", "lg_q2", "lg_q4", "lg_loopend", "spins"]
print("stagger", os.environ.get("SKM_STAGGER", "1"), " stamps relative to the earliest, cycles")
print("wave " + " ".join("%10s" % n for n in names))
for w in range(8):
    print("%4d " % w + " ".join("%10d" % ((t[w, k] - base if k != 12 else t[w, k] - 1000000) if t[w, k] else -1) for k in range(len(names))))
