"""cProfile of Mixture.EM_split on frames resident in HBM (the bench em_split leg): where the host time goes."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sidekit_amd as sk
from bench import synth_frames_device
T, D, C = 10_000_000, 60, 512
eng = sk.default_engine()
X = synth_frames_device(torch, T, D, 1234, eng.device)
fs = sk.FeaturesServer(features={"all": X}, is_sharded=True)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sk.Mixture().EM_split(fs, ["all"], distrib_nb=C)
    torch.cuda.synchronize(); print("run %d: %.4f s" % (k, time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
sk.Mixture().EM_split(fs, ["all"], distrib_nb=C)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
