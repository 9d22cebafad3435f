"""Build-container check (needs /root/reference, never runs on the GPU box): the oracle's E-step is timed beside
the real reference's Mixture._expectation on the same frames, to show that bench.py's cpu_baseline ("port") runs at
the reference's own speed (SURVEY.md 8d asks for agreement within ~10 %)."""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SIDEKIT", "libsvm=false")
sys.dont_write_bytecode = True
import gen_golden  # noqa: E402
from oracle import sidekit_ref as orc  # noqa: E402


def main():
    sidekit = gen_golden.import_reference()
    for C, T in ((512, 100000), (2048, 50000)):
        g = orc.random_gmm(C, 60, seed=7)
        cep = orc.synth_frames(T, dim=60, seed=1234).astype(np.float64)
        ubm = gen_golden.make_ubm(sidekit, g.w, g.mu, g.invcov)
        acc = copy.deepcopy(ubm)
        acc._reset()
        ubm._expectation(acc, cep[:1000])
        t0 = time.perf_counter()
        ubm._expectation(acc, cep)
        t_ref = time.perf_counter() - t0
        orc.expectation(g, cep[:1000])
        t0 = time.perf_counter()
        orc.expectation(g, cep)
        t_or = time.perf_counter() - t0
        print("C=%d T=%d  reference %.0f frames/s   oracle %.0f frames/s   ratio %.2f"
              % (C, T, T / t_ref, T / t_or, t_ref / t_or))


if __name__ == "__main__":
    main()
