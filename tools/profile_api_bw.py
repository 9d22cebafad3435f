"""cProfile of StatServer.accumulate_stat at the API level (host frames in, host statistics out)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sidekit_amd as sk
from bench import random_ubm
S, L, C, D = 2000, 2000, 2048, 60
rs = np.random.RandomState(0)
feats = {"s%05d" % i: rs.randn(L, D).astype(np.float32) for i in range(S)}
fs = sk.FeaturesServer(features=feats)
ubm = random_ubm(sk, C, D, seed=11)
idmap = sk.IdMap()
idmap.leftids = np.array(sorted(feats)); idmap.rightids = idmap.leftids.copy()
idmap.start = np.empty(S, "|O"); idmap.stop = np.empty(S, "|O")
def run(keep):
    ss = sk.StatServer(idmap, distrib_nb=C, feature_size=D)
    t0 = time.perf_counter()
    ss.accumulate_stat(ubm, fs, keep_on_device=keep)
    return time.perf_counter() - t0
print("warm", run(False))
pr = cProfile.Profile(); pr.enable(); dt = run(False); pr.disable()
print("host out %.3f s = %.2e frames/s" % (dt, S * L / dt))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
