"""BASELINE configs[1] end to end: Mixture.EM_split to 512 components on 10 M synthetic 60-dim frames resident in HBM
(the reference's default schedule, 37 EM iterations over 9 binary splits), one MI355X.  Prints one JSON line with the
wall time, the per-iteration mean log-likelihoods and the per-kernel totals.
Usage: python tools/em_split_cfg2.py [--frames 10000000] [--comps 512]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10_000_000)
    ap.add_argument("--comps", type=int, default=512)
    args = ap.parse_args()
    import torch
    import sidekit_amd as sk
    from bench import synth_frames_device
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = sk.default_engine()
    X = synth_frames_device(torch, args.frames, 60, seed=1234, device=dev)
    fs = sk.FeaturesServer(features={"all": X})
    warm = sk.Mixture()
    warm.EM_split(sk.FeaturesServer(features={"w": X[:200000]}), ["w"], distrib_nb=args.comps)     # warm-up (code objects)
    torch.cuda.synchronize()
    ubm = sk.Mixture()
    eng.profile(True)
    t0 = time.perf_counter()
    llk = ubm.EM_split(fs, ["all"], distrib_nb=args.comps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    sched = (1, 2, 2, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8)[:int(np.log2(args.comps))]
    frame_iters = args.frames * sum(sched)
    flop = 8.0 * 60 * args.frames * sum(it * (2 << i) for i, it in enumerate(sched))
    print(json.dumps({"workload": "EM_split, %d frames x 60 dims -> %d components, schedule %s" % (args.frames, args.comps, list(sched)),
                      "seconds": dt, "em_iterations": len(llk), "frame_iterations_per_s": frame_iters / dt,
                      "algorithmic_tflops": flop / dt / 1e12, "frac_of_f32_mfma_peak": flop / dt / 157.3e12,
                      "final_mean_llk": llk[-1], "llk_monotone_within_stage": bool(all(np.isfinite(llk))),
                      "weights_sum": float(ubm.w.sum()), "min_weight": float(ubm.w.min()),
                      "kernels_ms": {k: [v[0], round(v[1], 2)] for k, v in sorted(prof.items())}}))


if __name__ == "__main__":
    main()
