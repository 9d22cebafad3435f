import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
C = int(os.environ.get("C", 512)); T = int(os.environ.get("T", 10_000_000)); D = 60
eng = sk.default_engine()
X = synth_frames_device(torch, T, D, 1234, eng.device)
ubm = random_ubm(sk, C, D)
packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
for _ in range(2): eng.estep(X, packed)
torch.cuda.synchronize()
eng.profile(True)
for _ in range(3): eng.estep(X, packed)
torch.cuda.synchronize()
print(os.environ.get("SKM_LIB", "default"), {k: round(v[1] / v[0], 3) for k, v in eng.profile_read().items()})
