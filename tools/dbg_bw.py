import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import sidekit_amd as sk
from oracle import sidekit_ref as orc
eng = sk.default_engine()
C, D = int(sys.argv[1]), int(sys.argv[2])
gmm = orc.random_gmm(C, D, seed=C + D)
lens = [0, 0, 1, 31, 32, 33, 0, 257, 64, 1, 0, 700, 95, 0, 0]
frames = orc.synth_frames(sum(lens), dim=D, seed=10)
off = np.concatenate([[0], np.cumsum(lens)])
X = eng.upload_frames(frames)
packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
s0, s1 = eng.bw_stats(X, off, packed)
s0, s1 = eng.to_host(s0), eng.to_host(s1)
r0, r1 = orc.accumulate_stat(gmm, [frames[off[i]:off[i + 1]].astype(np.float64) for i in range(len(lens))])
err = np.abs(s1 - r1).reshape(len(lens), C, D)
sc = np.abs(r1).max()
print("teams", eng.lib.skm_bw_team_count(eng.ctx, C, D))
for i in range(len(lens)):
    e = err[i] / sc
    if e.max() > 1e-4:
        bad = np.argwhere(e > 1e-4)
        comps = sorted(set(bad[:, 0] // 32)); cols = sorted(set(bad[:, 1] // 32))
        print("session", i, "len", lens[i], "max", e.max(), "comp tiles", comps[:20], "col tiles", cols, "n bad", len(bad))
e0 = np.abs(s0 - r0) / np.abs(r0).max()
print("stat0 max err", e0.max())
