import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
C = int(os.environ.get("C", 2048)); S = int(os.environ.get("S", 2000)); L = int(os.environ.get("L", 2000)); D = 60
eng = sk.default_engine()
X = synth_frames_device(torch, S * L, D, 1234, eng.device)
ubm = random_ubm(sk, C, D)
packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
seg = torch.arange(0, S + 1, dtype=torch.int64, device=eng.device) * L
if os.environ.get("SPLIT"):
    seg = np.arange(0, S + 1, dtype=np.int64) * L           # host offsets: long sessions are cut over the teams
s0 = torch.empty((S, C), dtype=torch.float64, device=eng.device)
s1 = torch.empty((S, C * D), dtype=torch.float64, device=eng.device)
eng.bw_stats(X, seg, packed, s0, s1)
torch.cuda.synchronize()
eng.profile(True)
for _ in range(2): eng.bw_stats(X, seg, packed, s0, s1)
torch.cuda.synchronize()
pr = {k: round(v[1] / v[0], 3) for k, v in eng.profile_read().items()}
t = sum(pr.values())
print("C=%d %d x %d frames" % (C, S, L), pr, "%.3g frames/s" % (S * L / (t * 1e-3)), "%.1f TF (6DC)" % (6.0 * D * C * S * L / (t * 1e-3) / 1e12))
