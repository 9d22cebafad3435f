"""How much the f32 accumulation interval costs: E-step statistics with SKM_FLUSH (frame tiles between two folds into
the f64 slab) = 64 as the reference point against longer intervals, same frames, same model.
   python tools/flush_precision.py [C] [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
D = 60
eng = sk.default_engine()
X = synth_frames_device(torch, T, D, 1234, eng.device)
ubm = random_ubm(sk, C, D)
packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
res = {}
for fl in (64, 256, 512, 1024, 2048, 1 << 20):
    os.environ["SKM_FLUSH"] = str(fl)
    res[fl] = eng.to_host(eng.estep(X, packed)).copy()
ref = res[64]
parts = {"N": slice(0, C), "F": slice(C, C + C * D), "S": slice(C + C * D, C + 2 * C * D), "llk": slice(-1, None)}
for fl, v in res.items():
    out = []
    for k, sl in parts.items():
        a, b = v[sl], ref[sl]
        big = np.abs(b) >= 0.05 * np.abs(b).max()
        out.append("%s %.1e/%.1e" % (k, np.abs(a - b).max() / np.abs(b).max(), (np.abs(a - b)[big] / np.abs(b)[big]).max()))
    print("flush %8d tiles: max|d|/max|ref| / element-wise on large entries: %s" % (fl, "  ".join(out)), flush=True)
