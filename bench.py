#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X GMM-UBM / i-vector statistics hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "UBM 512 diag comps, 60-dim synthetic MFCC, 10M frames,
EM_split on 1 MI355X"): one STEP = one EM iteration of Mixture at C=512 -- E-step over the
rank's 10 M float32 frames resident in HBM, one all-reduce of the packed [N|F|S|llk] float64
buffer when N > 1, the float64 M-step and the re-pack of the model -- i.e. the body of the
EM_split loop (reference mixture.py:686-710).  Weak scaling: every rank holds its own 10 M frames.
`value` = frames processed by all ranks / max-over-ranks time.

`--gpus N` with N > 1 and no torchrun environment: this process touches no GPU, starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py` as a CHILD (fresh processes, one per GPU)
and relays rank 0's JSON line.  Under torchrun (WORLD_SIZE set) it is one of the ranks.

Also reported on the same JSON line (rank 0):
  roofline      dominant kernel against the f32 MFMA peak, timed with HIP events on the launch stream
                inside the library (skm_prof_*), algorithmic flops only
  roofline_path the whole E-step (8*D*C flop/frame) over the sum of its kernels
  cpu_baseline  the NumPy float64 oracle (a port of the reference's call sequence) on a bounded
                sample of the same workload, host cores stated (N=1 only)
  em_split      BASELINE configs[1] whole: Mixture.EM_split to 512 components, 37 EM iterations over 9 splits
  em_c2048      BASELINE configs[2] per-GPU shard: EM iteration at C=2048 on 12.5 M frames per GPU (weak scaling)
  cfg1          BASELINE configs[0] shape (the reference's conf.yaml UBM): EM iteration + Baum-Welch at C=64, D=132
  bw_stats      BASELINE configs[3]: Baum-Welch N/F, C=2048, 50 000 sessions x 2 000 frames in total, sessions
                split over the ranks (strong scaling); `api` = host arrays in -> host arrays out on a bounded sample
  tv            BASELINE configs[4]: TV training iteration + extract_ivectors, R=400, C=2048, 50 000 sessions in
                total, split over the ranks (strong scaling); `api` as above
  scoring, feat the "next" rows either side of the path
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F64_MFMA_TFLOPS = 78.6    # MI355X_MICROARCH.md, "Peak FP64 (matrix)"
PEAK_HBM_GBS = 8000.0


def synth_frames_device(torch, n, dim, seed, device, clusters=64, chunk=1 << 20):
    """CMVN-like MFCC frames from a fixed random diagonal GMM, generated on the GPU."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gp = torch.Generator(device=device)
    gp.manual_seed(1234)                       # the mixture itself is the same on every rank
    means = torch.randn(clusters, dim, generator=gp, device=device)
    std = torch.sqrt(0.3 + 0.7 * torch.rand(clusters, dim, generator=gp, device=device))
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        lab = torch.randint(0, clusters, (b - a,), generator=g, device=device)
        out[a:b] = means[lab] + std[lab] * torch.randn(b - a, dim, generator=g, device=device)
    return out


def random_ubm(sk, C, D, seed=7):
    rs = np.random.RandomState(seed)
    m = sk.Mixture()
    m.w = rs.dirichlet(np.full(C, 5.0))
    m.mu = rs.randn(C, D)
    m.invcov = 1.0 / rs.uniform(0.5, 1.5, size=(C, D))
    m.cov_var_ctl = 1.0 / m.invcov
    m._compute_all()
    return m


def em_iterations(ubm, X_dev, n):
    """n passes of the body of the EM_split loop (mixture.py:686-710) on frames resident in HBM, exactly as
    Mixture.EM_split runs a stage: per iteration the E-step, the all-reduce of the packed statistics and the
    M-step + variance control + re-pack of the model on the device; at the end of the stage the parameters and the
    per-iteration mean log-likelihoods come back to the host (a full sync).  Returns the last log-likelihood."""
    return ubm._em_iterations_device(X_dev, n)[-1]


def cpu_baseline(C, D, seconds_target=15.0):
    """The oracle's E-step + M-step (reference call sequence: 4 dot + 2 exp passes per 50k-frame
    chunk) on a bounded sample, all host BLAS threads."""
    from oracle import sidekit_ref as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    gmm = orc.random_gmm(C, D, seed=7)
    chunk = orc.synth_frames(50000, dim=D, seed=1234).astype(np.float64)
    t0 = time.perf_counter()
    orc.expectation(gmm, chunk[:10000])
    probe = time.perf_counter() - t0
    n_chunks = int(max(1, min(80, seconds_target / max(probe * 5.0, 1e-3))))
    t0 = time.perf_counter()
    N, F, S, llk = orc.expectation_chunked(gmm, [chunk] * n_chunks)
    orc.maximization(gmm, N, F, S)
    dt = time.perf_counter() - t0
    frames = 50000 * n_chunks
    return {"value": frames / dt, "unit": "frames/s", "cores": int(cores), "kind": "port",
            "sample": "oracle (NumPy f64) E-step+M-step, C=%d D=%d, %d frames in %d chunks of 50k, %.1f s"
                      % (C, D, frames, n_chunks, dt)}


def _host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] or [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_baseline_bw(C, D, seconds_target=8.0):
    """The oracle's accumulate_stat body (reference call sequence per session: 2 dot + exp passes + 1 dot) on a
    bounded number of 2000-frame sessions."""
    from oracle import sidekit_ref as orc
    gmm = orc.random_gmm(C, D, seed=11)
    sess = orc.synth_frames(2000, dim=D, seed=1234).astype(np.float64)
    t0 = time.perf_counter()
    orc.accumulate_stat(gmm, [sess])
    probe = time.perf_counter() - t0
    n = int(max(8, min(64, seconds_target / max(probe, 1e-3))))
    t0 = time.perf_counter()
    orc.accumulate_stat(gmm, [sess] * n)
    dt = time.perf_counter() - t0
    return {"value": 2000 * n / dt, "unit": "frames/s", "cores": _host_threads(), "kind": "port",
            "sample": "oracle (NumPy f64) accumulate_stat, C=%d D=%d, %d sessions x 2000 frames, %.1f s" % (C, D, n, dt)}


def cpu_baseline_tv(C, D, R, seconds_target=10.0):
    """The oracle's e_on_batch (the reference's per-session (F.T * n).dot(F) + inv, factor_analyser.py:86-90) on a
    bounded number of sessions; it is the body of both TV training and extract_ivectors."""
    from oracle import sidekit_ref as orc
    gmm = orc.random_gmm(C, D, seed=11)
    rs = np.random.RandomState(3)
    Fm = 0.1 * rs.randn(C * D, R)

    def stats(n):
        s0 = (2000.0 * rs.dirichlet(np.full(C, 0.3), size=n)).astype(np.float32)
        s1 = (np.repeat(s0, D, axis=1) * gmm.mu.flatten() + rs.randn(n, C * D)).astype(np.float32)
        return s0, s1
    s0, s1 = stats(1)
    t0 = time.perf_counter()
    orc.e_on_batch(s0, s1, gmm, Fm)
    probe = time.perf_counter() - t0
    n = int(max(1, min(32, seconds_target / max(probe, 1e-3))))
    s0, s1 = stats(n)
    t0 = time.perf_counter()
    orc.e_on_batch(s0, s1, gmm, Fm)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "sessions/s (= i-vectors/s)", "cores": _host_threads(), "kind": "port",
            "sample": "oracle (NumPy/SciPy f64) e_on_batch, C=%d D=%d R=%d, %d sessions, %.1f s" % (C, D, R, n, dt)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n fresh rank processes (one per GPU) through torch.distributed.run and relay rank 0's JSON line.
    The calling process has not touched the GPU (nothing is re-executed after a HIP call); it only waits."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line)
    return rc if rc != 0 or line is not None else 1


def launch_check(world, rank, local_rank):
    """--launch-check: what the launcher handed to this rank, collected over gloo (no GPU is touched)."""
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    got = [None] * world
    td.all_gather_object(got, {"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid()})
    if rank == 0:
        print(json.dumps({"metric": "launch-check", "n_gpus": world, "ranks": got}))
    td.barrier()
    td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=10_000_000, help="frames per GPU of the headline workload")
    ap.add_argument("--comps", type=int, default=512)
    ap.add_argument("--dim", type=int, default=60)
    ap.add_argument("--em2048-frames", type=int, default=100_000_000,
                    help="frames of the em_c2048 leg IN TOTAL (BASELINE configs[2]: 100 M), split over the ranks")
    ap.add_argument("--bw-sessions", type=int, default=50_000, help="sessions of the bw_stats leg, all ranks together")
    ap.add_argument("--tv-sessions", type=int, default=50_000, help="sessions of the tv leg, all ranks together")
    ap.add_argument("--tv-batch", type=int, default=4096, help="sessions per TV E-step batch (training)")
    ap.add_argument("--tv-extract-batch", type=int, default=16384, help="sessions per batch of extract_ivectors")
    ap.add_argument("--api-sessions", type=int, default=4000, help="sessions of the host-in / host-out samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs")
    ap.add_argument("--no-split", action="store_true", help="skip the split-precision (SKM_SPLIT=bf16x3) repeat of the headline")
    ap.add_argument("--legs", default="em_split,em2048,bw,cfg1,scoring,feat,tv",
                    help="secondary legs (comma list of em_split, em2048, bw, cfg1, scoring, feat, tv)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only report the rank layout the launcher produced (gloo, no GPU); used by the CPU tests")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        launch_check(world, rank, local_rank)
        return

    if os.environ.get("SKM_DUMP_MAPS"):
        # where a crash inside exit() handlers lives: the address map as it is when Python's atexit hooks run
        import atexit
        import shutil
        atexit.register(lambda: shutil.copyfile("/proc/self/maps", os.environ["SKM_DUMP_MAPS"]))

    t_start = time.perf_counter()
    import torch
    import torch.distributed as td
    rehearse = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SKM_BENCH_REHEARSE=1: N ranks on ONE GPU with gloo collectives -- a rehearsal of the world > 1 branches of
        # this file and of the drop-in classes on a one-GPU box (RCCL refuses two ranks on one device).  Its numbers
        # are not a measurement: the ranks share the card and the collectives go through host memory.
        rehearse = os.environ.get("SKM_BENCH_REHEARSE") == "1"
        torch.cuda.set_device(0 if rehearse else local_rank)
        td.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import sidekit_amd as sk
    eng = sk.default_engine()
    C, D, T = args.comps, args.dim, args.frames
    legs = set() if args.no_extra else set(x for x in args.legs.split(",") if x)
    api_stats = None

    # cfg3 is ONE job of 100 M frames whatever the GPU count (strong scaling): a rank holds its array_split share,
    # all of it (24 GB) at N = 1
    lo_e, hi_e = sk.dist.shard_bounds(args.em2048_frames, world, rank)
    Te_rank = hi_e - lo_e
    T_all = max(T, Te_rank if "em2048" in legs else 0)
    X_all = synth_frames_device(torch, T_all, D, seed=1234 + rank, device=dev)
    X = X_all[:T]
    ubm = random_ubm(sk, C, D)

    verbose = os.environ.get("SKM_BENCH_VERBOSE") == "1"

    def note(what):
        if verbose:
            print("[bench rank %d] %s  t=%.1fs" % (rank, what, time.perf_counter() - t_start), file=sys.stderr, flush=True)

    def sync():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def timed(fn, reps):
        """barrier + synchronize on both sides of `reps` calls, MAX over ranks of the wall time (seconds)."""
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, r

    note("headline warm-up")
    if args.warmup > 0:
        em_iterations(ubm, X, args.warmup)
    note("headline timed")
    dt, llk = timed(lambda: em_iterations(ubm, X, args.steps), 1)
    ms_per_step = 1e3 * dt / args.steps
    value = world * T * args.steps / dt

    result = {
        "metric": "frames/sec for UBM EM + Baum-Welch stats; i-vectors/sec (TV rank 400)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "UBM EM iteration (E-step + all-reduce + M-step), C=%d diag comps, D=%d, "
                               "%d float32 frames per GPU resident in HBM (BASELINE configs[1])" % (C, D, T),
                   "components": C, "dim": D, "frames_per_gpu": T, "parallelism": "dp%d" % world,
                   "final_mean_llk": float(llk)},
    }

    if rehearse:
        result["config"]["rehearsal"] = "%d ranks share ONE GPU, gloo collectives: not a measurement" % world

    # ---- per-kernel timing with HIP events on the launch stream (separate, un-timed pass).  Every rank runs
    # it (the iteration contains the all-reduce); rank 0 reports its own kernels.
    eng.profile(True)
    nprof = 3
    em_iterations(ubm, X, nprof)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    if rank == 0:
        kt = {k: v[1] / v[0] for k, v in prof.items()}            # avg ms per launch
        flop_path = 8.0 * D * C * T         # log-lik GEMM + statistics GEMM
        single = [k for k in ("k_estep_fat", "k_estep_fused") if k in kt]
        if single:
            # single-sweep kernel (C <= 512): the whole E-step is one kernel, all 8*D*C*T flops are its own
            dom = single[0]
            t_dom, flop_dom, issued = kt[dom], flop_path, (4.0 * D + 4.0 * 64) * C * T
            t_path = sum(kt.get(k, 0.0) for k in (dom, "k_reduce_slabs", "k_mstep_pack"))
        else:
            # two sweeps: the statistics GEMM gamma^T [x, x^2] is the algorithmic work of k_stats_em; the
            # logits it recomputes are NOT counted
            dom, t_dom, flop_dom = "k_stats_em", kt.get("k_stats_em", float("nan")), 4.0 * D * C * T
            issued = (8.0 * D + 4.0 * 64) * C * T
            t_path = sum(kt.get(k, 0.0) for k in ("k_lse_partial", "k_lse_combine", "k_stats_em", "k_reduce_slabs", "k_mstep_pack"))
        ach = flop_dom / (t_dom * 1e-3) / 1e12
        result["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach,
                              "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                              "traffic": None, "avg_launch_ms": t_dom,
                              "algorithmic_flop_per_launch": flop_dom}
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same workload (a process
        # cannot read its own PMC counters); the committed summary is used only when the workload matches
        pmc_path = os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.json")
        for name in ("r3_pmc_hbm_traffic.json", "r2_pmc_hbm_traffic.json"):
            if os.path.exists(os.path.join(ROOT, "profiles", name)):
                pmc_path = os.path.join(ROOT, "profiles", name)
                break
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("workload") == {"C": C, "D": D, "frames": T}:
                for name, rec in pmc["kernels"].items():
                    if name.startswith(dom):
                        result["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                        result["roofline"]["traffic_source"] = "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE)" % os.path.basename(pmc_path)
                        result["roofline"]["algorithmic_bytes_per_launch"] = 4.0 * D * T
                        if "mfma_util" in rec:
                            result["roofline"]["mfma_util_pmc"] = rec["mfma_util"]
        achp = flop_path / (t_path * 1e-3) / 1e12
        result["roofline_path"] = {"bound": "mfma", "kernels": {k: round(v, 4) for k, v in kt.items()},
                                   "achieved": achp, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": achp / PEAK_F32_MFMA_TFLOPS, "algorithmic_flop_per_step": flop_path,
                                   "issued_mfma_flop_per_step": issued}

    # ---- the same EM iteration with the opt-in split-precision E-step (SKM_SPLIT=bf16x3, csrc/skm_gmm_b3.hip): the two
    # GEMMs as bf16 x 3 products on the bf16 matrix cores, float32 accumulation, float32-grade results (parity bars of
    # tests/test_gpu_split.py).  Reported beside the headline, never as it: `value` above is the float32-MFMA path.
    if not args.no_extra and not args.no_split:
        note("split-precision E-step")
        os.environ["SKM_SPLIT"] = "bf16x3"
        try:
            ubm_b = random_ubm(sk, C, D)
            em_iterations(ubm_b, X, max(1, args.warmup))
            dtb, llk_b = timed(lambda: em_iterations(ubm_b, X, args.steps), 1)
            eng.profile(True)
            em_iterations(ubm_b, X, nprof)
            torch.cuda.synchronize()
            prof_b = eng.profile_read()
            eng.profile(False)
        finally:
            os.environ.pop("SKM_SPLIT", None)
        if rank == 0:
            kb = {k: v[1] / v[0] for k, v in prof_b.items()}
            if "k_estep_b3" in kb:
                achb = 8.0 * D * C * T / (kb["k_estep_b3"] * 1e-3) / 1e12
                result["split_precision"] = {
                    "mode": "SKM_SPLIT=bf16x3 (opt-in): x, x^2, W and the posteriors as three bf16 terms each, 6 of the 9 "
                            "partial products on v_mfma_f32_32x32x16_bf16, float32 accumulation",
                    "value": world * T * args.steps / dtb, "unit": "frames/s", "ms_per_step": 1e3 * dtb / args.steps,
                    "speedup_over_headline": dt / dtb, "final_mean_llk": float(llk_b),
                    "kernels_ms": {k: round(v, 4) for k, v in kb.items()},
                    "kernel_algorithmic_tflops": achb,
                    "frac_of_f32_mfma_peak": achb / PEAK_F32_MFMA_TFLOPS,
                    "issued_bf16_mfma_tflops": 3072.0 * C * T / (kb["k_estep_b3"] * 1e-3) / 1e12,
                    "issued_frac_of_bf16_mfma_peak": 3072.0 * C * T / (kb["k_estep_b3"] * 1e-3) / 1e12 / 2500.0,
                    "note": "frac_of_f32_mfma_peak = algorithmic float32 flops (8 D C per frame) over the F32 MFMA peak of "
                            "157.3 TF: above 1 is what the bf16 matrix cores buy, not a float32 roofline fraction.  The "
                            "kernel issues 6 bf16 MFMA products per float32 product on K padded to 128 (logits) and "
                            "128 columns (statistics): 3072 C flop per frame against the dense bf16 peak of 2.5 PF"}
        del ubm_b

    # The secondary legs run on EVERY rank (weak scaling: each rank owns its sessions / frames); values are
    # whole-job aggregates over the MAX-over-ranks wall time, kernel tables are rank 0's.
    def shard(n):
        """this rank's share of n units that the whole job splits (strong-scaling legs)"""
        lo, hi = sk.dist.shard_bounds(n, world, rank)
        return hi - lo

    if "em_split" in legs:
        note("leg em_split")
        # ---- BASELINE configs[1] whole: Mixture.EM_split to C components on the rank's frames, the reference's
        # default schedule (37 EM iterations over 9 binary splits at C = 512); frames sharded, statistics all-reduced
        fs_all = sk.FeaturesServer(features={"all": X}, is_sharded=True)
        # warm-up: two full passes.  150 short launches with host syncs between the stages: in the first second of GPU
        # work of a fresh process the same kernels (identical event times) leave 0.3 ms gaps between launches -- power
        # state, not code (tools/profile_em_split.py: 0.320, 0.339, 0.298, 0.298 s for four passes in a row)
        for _ in range(2):
            sk.Mixture().EM_split(fs_all, ["all"], distrib_nb=C)
        ubm_s = sk.Mixture()
        eng.profile(True)
        dts, llk_s = timed(lambda: ubm_s.EM_split(fs_all, ["all"], distrib_nb=C), 1)
        prof = eng.profile_read()
        eng.profile(False)
        sched = (1, 2, 2, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8)[:int(np.log2(C))]
        flop = 8.0 * D * T * sum(it * (2 << i) for i, it in enumerate(sched))
        result["em_split"] = {"value": world * T * sum(sched) / dts, "unit": "frame-iterations/s", "n_gpus": world,
                              "seconds": dts, "em_iterations": len(llk_s),
                              "workload": "Mixture.EM_split, %d frames per GPU x %d dims -> %d components, schedule %s "
                                          "(BASELINE configs[1])" % (T, D, C, list(sched)),
                              "final_mean_llk": float(llk_s[-1]),
                              "kernels_ms": {k: round(v[1], 2) for k, v in sorted(prof.items())},
                              "path_frac_of_mfma_peak": flop / dts / 1e12 / PEAK_F32_MFMA_TFLOPS}
        if not args.no_split:
            # the same call with the opt-in split-precision E-step: the last stage (512 components, 8 of the 37 iterations)
            # runs on k_estep_b3, the stages below it on the same kernels as above
            os.environ["SKM_SPLIT"] = "bf16x3"
            try:
                sk.Mixture().EM_split(fs_all, ["all"], distrib_nb=C)
                ubm_sb = sk.Mixture()
                dtsb, llk_sb = timed(lambda: ubm_sb.EM_split(fs_all, ["all"], distrib_nb=C), 1)
            finally:
                os.environ.pop("SKM_SPLIT", None)
            result["em_split"]["split_precision"] = {"seconds": dtsb, "speedup": dts / dtsb,
                                                     "final_mean_llk": float(llk_sb[-1]),
                                                     "mode": "SKM_SPLIT=bf16x3 (opt-in): the 512-component stage on k_estep_b3"}
            del ubm_sb
        del fs_all, ubm_s

    if "bw" in legs:
        note("leg bw")
        # ---- Baum-Welch N/F accumulation, BASELINE configs[3]: C=2048, 50 000 sessions x 2 000 frames in the whole
        # job, the sessions split over the ranks.  No data-path collective: every session row has one owner.
        Cb, L = 2048, 2000
        S = shard(args.bw_sessions)
        ubm_b = random_ubm(sk, Cb, D, seed=11)
        packed = eng.pack(ubm_b.w, ubm_b.mu, ubm_b.invcov)
        seg = torch.arange(0, S + 1, dtype=torch.int64, device=dev) * L
        Xb = X_all[:S * L] if S * L <= T_all else synth_frames_device(torch, S * L, D, seed=4321 + rank, device=dev)
        s0 = torch.empty((S, Cb), dtype=torch.float64, device=dev)
        s1 = torch.empty((S, Cb * D), dtype=torch.float64, device=dev)
        nw = min(S, 500)
        eng.bw_stats(Xb[:nw * L], seg[:nw + 1], packed, s0[:nw], s1[:nw])          # warm-up
        eng.profile(True)
        dtb, _ = timed(lambda: eng.bw_stats(Xb, seg, packed, s0, s1), 1)
        prof = eng.profile_read()
        eng.profile(False)
        k_bw = "k_bw_team" if "k_bw_team" in prof else "k_stats_bw"
        t_bw = prof[k_bw][1] / prof[k_bw][0]
        fl_bw = (6.0 if k_bw == "k_bw_team" else 2.0) * D * Cb * S * L
        result["bw_stats"] = {"value": args.bw_sessions * L / dtb, "unit": "frames/s", "n_gpus": world, "scaling": "strong",
                              "seconds": dtb,
                              "workload": "accumulate_stat body, C=%d, %d sessions x %d frames in total (BASELINE configs[3]), "
                                          "%d sessions on rank 0" % (Cb, args.bw_sessions, L, S),
                              "roofline": {"bound": "mfma", "kernel": k_bw,
                                           "achieved": fl_bw / (t_bw * 1e-3) / 1e12,
                                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": fl_bw / (t_bw * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                           "avg_launch_ms": t_bw},
                              "path_frac_of_mfma_peak": 6.0 * D * Cb * S * L / dtb / 1e12 / PEAK_F32_MFMA_TFLOPS}
        del s0, s1
        if Xb.data_ptr() != X_all.data_ptr():
            del Xb
        torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            # API level (host arrays in -> host arrays out) on a bounded sample: StatServer.accumulate_stat fed by an
            # in-memory features server; upload of the frames and download of stat0 / stat1 included
            Sa = min(args.api_sessions, S)
            host = X_all[:Sa * L].cpu().numpy()
            names = ["s%05d" % i for i in range(Sa)]
            fsa = sk.FeaturesServer(features={n: host[i * L:(i + 1) * L] for i, n in enumerate(names)})
            ssw = sk.StatServer(sk.IdMap().set(names, names), ubm=ubm_b, engine=eng)       # warm-up: staging buffers exist
            ssw.accumulate_stat(ubm=ubm_b, feature_server=fsa, seg_indices=range(min(Sa, 1100)), gather=False)
            del ssw
            ssa = sk.StatServer(sk.IdMap().set(names, names), ubm=ubm_b, engine=eng)
            t0 = time.perf_counter()
            ssa.accumulate_stat(ubm=ubm_b, feature_server=fsa, seg_indices=range(Sa), gather=False)
            dta = time.perf_counter() - t0
            ssb = sk.StatServer(sk.IdMap().set(names, names), ubm=ubm_b, engine=eng)
            t0 = time.perf_counter()
            ssb.accumulate_stat(ubm=ubm_b, feature_server=fsa, seg_indices=range(Sa), gather=False, keep_on_device=True)
            dtk = time.perf_counter() - t0
            result["bw_stats"]["api"] = {"value": Sa * L / dta, "unit": "frames/s", "seconds": dta,
                                         "sample": "StatServer.accumulate_stat, %d sessions x %d frames on one GPU: host float32 "
                                                   "frames in (%.1f GB), host float64 stat0/stat1 out (%.1f GB)"
                                                   % (Sa, L, host.nbytes / 1e9, (ssa.stat0.nbytes + ssa.stat1.nbytes) / 1e9),
                                         "seconds_keep_on_device": dtk}
            api_stats = ssb                    # device-resident rows feed the tv api sample below
            del host, fsa, ssa
        if world > 1:
            td.barrier()

    if "em2048" in legs:
        note("leg em2048")
        # ---- EM iteration at the target size, BASELINE configs[2] WHOLE: C=2048, 100 M frames in total split over the
        # ranks like numpy.array_split (sidekit_mpi.py:408-410), one all-reduce of the packed statistics per iteration
        Ce, Te = 2048, Te_rank
        ubm_e = random_ubm(sk, Ce, D, seed=11)
        Xe = X_all[:Te]
        em_iterations(ubm_e, Xe, 1)
        eng.profile(True)
        dte, _ = timed(lambda: em_iterations(ubm_e, Xe, 2), 1)
        dte /= 2
        prof = eng.profile_read()
        eng.profile(False)
        kte = {k: v[1] / v[0] for k, v in prof.items()}
        t_k = sum(kte.values())
        result["em_c2048"] = {"value": args.em2048_frames / dte, "unit": "frames/s", "n_gpus": world,
                              "scaling": "strong", "ms_per_iteration": 1e3 * dte,
                              "workload": "UBM EM iteration (E-step + all-reduce + M-step), C=%d, D=%d, %d frames in total "
                                          "(BASELINE configs[2]), %d on rank 0" % (Ce, D, args.em2048_frames, Te),
                              "kernels_ms": {k: round(v, 3) for k, v in kte.items()},
                              "roofline_path": {"bound": "mfma", "achieved": 8.0 * D * Ce * Te / (t_k * 1e-3) / 1e12,
                                                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                "frac": 8.0 * D * Ce * Te / (t_k * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}
        if not args.no_split:
            # the same iteration on the opt-in split-precision route: two sweeps of k_estep_b3 over blocks of 512 components
            os.environ["SKM_SPLIT"] = "bf16x3"
            try:
                ubm_b = random_ubm(sk, Ce, D, seed=11)
                em_iterations(ubm_b, Xe, 1)
                eng.profile(True)
                dtb2, _ = timed(lambda: em_iterations(ubm_b, Xe, 2), 1)
                dtb2 /= 2
                prof = eng.profile_read()
                eng.profile(False)
            finally:
                os.environ.pop("SKM_SPLIT", None)
            kb2 = {k: v[1] / v[0] for k, v in prof.items()}
            if "k_b3_sweep2" in kb2:
                result["em_c2048"]["split_precision"] = {
                    "mode": "SKM_SPLIT=bf16x3 (opt-in): two sweeps of the split-precision kernel over blocks of 512 components",
                    "value": args.em2048_frames / dtb2, "unit": "frames/s", "ms_per_iteration": 1e3 * dtb2,
                    "speedup": dte / dtb2, "kernels_ms": {k: round(v, 3) for k, v in kb2.items()}}
            del ubm_b
        del ubm_e, Xe

    if "cfg1" in legs:
        note("leg cfg1")
        # ---- the reference's own configuration (BASELINE configs[0]: conf.yaml UBM, 64 diagonal components on 132-dim
        # rows = 44 coefficients + deltas + double deltas): one EM iteration and the Baum-Welch pass, both on the
        # wide-row single-sweep kernels (one component tile per 512-register wave)
        C1, D1, T1, L1 = 64, 132, 4_000_000, 2000
        X1 = synth_frames_device(torch, T1, D1, seed=77 + rank, device=dev)
        ubm_1 = random_ubm(sk, C1, D1, seed=13)
        em_iterations(ubm_1, X1, 1)
        eng.profile(True)
        dt1, _ = timed(lambda: em_iterations(ubm_1, X1, 2), 1)
        dt1 /= 2
        prof = eng.profile_read()
        eng.profile(False)
        kt1 = {k: v[1] / v[0] for k, v in prof.items()}
        packed1 = eng.pack(ubm_1.w, ubm_1.mu, ubm_1.invcov)
        seg1 = torch.arange(0, T1 // L1 + 1, dtype=torch.int64, device=dev) * L1
        eng.bw_stats(X1, seg1, packed1)
        eng.profile(True)
        dtb1, _ = timed(lambda: eng.bw_stats(X1, seg1, packed1), 1)
        profb = eng.profile_read()
        eng.profile(False)
        result["cfg1"] = {"value": world * T1 / dt1, "unit": "frames/s", "n_gpus": world, "ms_per_iteration": 1e3 * dt1,
                          "workload": "UBM EM iteration and Baum-Welch statistics, C=%d, D=%d, %d frames per GPU in sessions of "
                                      "%d (BASELINE configs[0] shape, synthetic frames)" % (C1, D1, T1, L1),
                          "kernels_ms": {k: round(v, 3) for k, v in kt1.items()},
                          "em_frac_of_mfma_peak": 8.0 * D1 * C1 * T1 / dt1 / 1e12 / PEAK_F32_MFMA_TFLOPS,
                          "bw_frames_per_s": world * T1 / dtb1,
                          "bw_kernels_ms": {k: round(v[1] / v[0], 3) for k, v in profb.items()},
                          "bw_frac_of_mfma_peak": 6.0 * D1 * C1 * T1 / dtb1 / 1e12 / PEAK_F32_MFMA_TFLOPS}
        del X1, ubm_1, packed1, seg1
        torch.cuda.empty_cache()

    if "scoring" in legs:
        note("leg scoring")
        # ---- GMM-UBM LLR scoring (next row after the statistics path, gmm_scoring.py:56-119): every enrolled model
        # against every test segment = one log-likelihood pass per model over all test frames + segment means.
        # Test segments are sharded over ranks, no collective.
        Cs, n_models, n_seg, Ls = 2048, 16, min(500, T // 2000), 2000
        ubm_s = random_ubm(sk, Cs, D, seed=11)
        Xs = X[:n_seg * Ls]
        seg = torch.arange(0, n_seg + 1, dtype=torch.int64, device=dev) * Ls
        rs = np.random.RandomState(5)
        packs = [eng.pack(ubm_s.w, ubm_s.mu + 0.05 * rs.randn(Cs, D), ubm_s.invcov) for _ in range(n_models)]
        p_ubm = eng.pack(ubm_s.w, ubm_s.mu, ubm_s.invcov)

        def score_all():
            base = eng.segment_mean(eng.loglik(Xs, p_ubm), seg)
            return torch.stack([eng.segment_mean(eng.loglik(Xs, p), seg) - base for p in packs])
        score_all()
        eng.profile(True)
        dts, _ = timed(score_all, 1)
        prof = eng.profile_read()
        eng.profile(False)
        fl = 4.0 * D * Cs * n_seg * Ls * (n_models + 1)       # the two GEMMs of compute_log_posterior_probabilities
        result["scoring"] = {"value": world * n_models * n_seg / dts, "unit": "trials/s", "n_gpus": world,
                             "model_frames_per_s": world * (n_models + 1) * n_seg * Ls / dts,
                             "workload": "gmm_scoring body, C=%d, %d models x %d segments x %d frames per GPU"
                                         % (Cs, n_models, n_seg, Ls),
                             "kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof.items())},
                             "path_frac_of_mfma_peak": fl / dts / 1e12 / PEAK_F32_MFMA_TFLOPS}
        del Xs, packs
        # cosine scoring of i-vectors (iv_scoring.py:90-99): length normalisation + enroll . test' in float64
        ne = 8192
        iv_e = torch.randn(ne, 400, dtype=torch.float64, device=dev)
        iv_t = torch.randn(ne, 400, dtype=torch.float64, device=dev)

        def cosine_all():
            return eng.gemm(eng.row_normalize_(iv_e.clone()), eng.row_normalize_(iv_t.clone()), transB=True)
        cosine_all()
        dtc, _ = timed(cosine_all, 3)
        dtc /= 3
        result["scoring"]["cosine"] = {"value": world * ne * ne / dtc, "unit": "trials/s",
                                       "workload": "cosine_scoring body, %d x %d i-vectors of rank 400 per GPU, float64" % (ne, ne),
                                       "f64_tflops": 2.0 * ne * ne * 400 / dtc / 1e12}
        # Mahalanobis scoring (iv_scoring.py:138-142): 4 rank-sized products, 2 cross-term GEMMs, 2 quadratic forms
        Mm = torch.eye(400, dtype=torch.float64, device=dev) + 0.01 * torch.randn(400, 400, dtype=torch.float64, device=dev)
        from importlib import import_module
        ivs = import_module(sk.cosine_scoring.__module__)

        def maha_all():
            return ivs._pair_scores(eng, iv_e, iv_t, Mm, None, 0.5, -0.5, -0.5)
        maha_all()
        dtm, _ = timed(maha_all, 3)
        dtm /= 3
        result["scoring"]["mahalanobis"] = {"value": world * ne * ne / dtm, "unit": "trials/s",
                                            "workload": "mahalanobis_scoring body, %d x %d i-vectors of rank 400 per GPU, float64" % (ne, ne),
                                            "f64_tflops": (4.0 * ne * ne * 400 + 4.0 * ne * 400 * 400) / dtm / 1e12}
        # WCCN / LDA within-class scatter (statserver.py:913-920): class means, centring, ONE float64 GEMM Xc'Xc
        ns, nk = 50_000, 5_000
        ss = sk.StatServer(engine=eng)
        ss.modelset = np.array(["s%05d" % (i % nk) for i in range(ns)])
        ss.segset = ss.modelset
        ss.stat0 = np.ones((ns, 1))
        ss.stat1 = np.random.RandomState(3).randn(ns, 400)

        def wccn_scatter():
            Xc = ss._class_centred(True)[2]
            return eng.gram(Xc, alpha=1.0 / nk)
        wccn_scatter()
        eng.profile(True)
        dtw, _ = timed(wccn_scatter, 2)
        dtw /= 2
        prof = eng.profile_read()
        eng.profile(False)
        result["scoring"]["wccn_scatter"] = {"value": world * ns / dtw, "unit": "i-vectors/s",
                                             "workload": "within-class scatter of %d i-vectors (rank 400) of %d speakers, host upload included" % (ns, nk),
                                             "kernels_ms_per_call": {k: round(v[1] / 2, 3) for k, v in sorted(prof.items())}}
        del iv_e, iv_t, ss

    if "feat" in legs:
        note("leg feat")
        # ---- Feature post-processing in front of the path (FeaturesServer.post_processing, features_server.py:202-246;
        # the experiment's configuration model_interface.py:45-52: RASTA + delta + double delta + cmvn), raw 20-dim
        # cepstra + VAD labels in HBM -> normalised 60-dim float32 frames of the labelled rows.  HBM-bound streaming;
        # sessions are sharded over ranks, no collective.
        D0, Sf, Lf = 20, min(5000, T // 2000), 2000
        gen = torch.Generator(device=dev)
        gen.manual_seed(99 + rank)
        raw = 3.0 * torch.randn(Sf * Lf, D0, generator=gen, device=dev) + 5.0
        vad = (torch.rand(Sf * Lf, generator=gen, device=dev) > 0.3).to(torch.uint8)
        segf = np.arange(0, Sf + 1, dtype=np.int64) * Lf

        def feat_all():
            return eng.feat_postproc(raw, segf, label=vad, rasta=True, ndelta=2, norm=2, keep_all=False)
        fo, _, _ = feat_all()
        rows_out = int(fo.shape[0])
        eng.profile(True)
        dtf, _ = timed(feat_all, 3)
        dtf /= 3
        prof = eng.profile_read()
        eng.profile(False)
        ktf = {k: v[1] / v[0] for k, v in prof.items()}
        # algorithmic bytes: raw frames and labels read once, labelled output rows and their labels written once
        bytes_alg = Sf * Lf * (D0 * 4 + 1) + rows_out * (3 * D0 * 4 + 1)
        t_w = ktf.get("k_feat_write", float("nan"))
        result["feat"] = {"value": world * Sf * Lf / dtf, "unit": "frames/s", "n_gpus": world,
                          "workload": "post_processing body (RASTA + delta + double delta + cmvn + VAD selection), "
                                      "%d sessions x %d frames x %d raw dims per GPU -> %d-dim float32" % (Sf, Lf, D0, 3 * D0),
                          "kernels_ms": {k: round(v, 3) for k, v in sorted(ktf.items())},
                          "roofline": {"bound": "hbm", "kernel": "k_feat_write",
                                       "achieved": bytes_alg / (t_w * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": bytes_alg / (t_w * 1e-3) / 1e9 / PEAK_HBM_GBS, "avg_launch_ms": t_w,
                                       "algorithmic_bytes_per_launch": bytes_alg, "traffic": None},
                          "path_GBps": bytes_alg / dtf / 1e9}
        # HBM bytes of the write walk from the committed rocprofv3 --pmc passes of this same leg
        pmc_feat = os.path.join(ROOT, "profiles", "r1_pmc_feat.json")
        if os.path.exists(pmc_feat) and (Sf, Lf, D0) == (5000, 2000, 20):
            for name, rec in json.load(open(pmc_feat))["kernels"].items():
                if name.startswith("k_feat_walk<2, 1, 1"):
                    result["feat"]["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                    result["feat"]["roofline"]["traffic_source"] = "profiles/r1_pmc_feat.json (FETCH_SIZE x2 + WRITE_SIZE)"
        # 'cmvn_sliding' on the same batch (features_server.py:282-287): un-normalised pass in float64, rolling
        # statistics over 301 labelled frames, compaction; the label vector crosses PCIe once to build the work list
        fsl = sk.FeaturesServer(feat_norm="cmvn_sliding", delta=True, double_delta=True, rasta=True,
                                keep_all_features=False, engine=eng)

        def feat_sliding():
            return fsl._post_process_device(raw, segf, vad, False)
        feat_sliding()
        eng.profile(True)
        dts, _ = timed(feat_sliding, 2)
        dts /= 2
        prof = eng.profile_read()
        eng.profile(False)
        result["feat"]["cmvn_sliding"] = {"value": world * Sf * Lf / dts, "unit": "frames/s",
                                          "kernels_ms_per_call": {k: round(v[1] / 2, 3) for k, v in sorted(prof.items())}}
        del raw, vad, fo

    if "tv" in legs:
        note("leg tv")
        # ---- TV training iteration + i-vector extraction, BASELINE configs[4]: R=400, C=2048, 50 000 sessions in the
        # whole job, split over the ranks; training reduce-scatters the accumulators over components, solves its C/N
        # components and all-gathers the new rows (factor_analyser._tv_iteration).
        del X, X_all
        torch.cuda.empty_cache()
        Ct, R = 2048, 400
        St = shard(args.tv_sessions)
        ubm_t = random_ubm(sk, Ct, D, seed=11)
        gen = torch.Generator(device=dev)
        gen.manual_seed(4321 + rank)
        mu_sv = torch.as_tensor(ubm_t.get_mean_super_vector(), device=dev)
        sd_sv = torch.as_tensor(1.0 / np.sqrt(ubm_t.get_invcov_super_vector()), device=dev)
        stat0 = torch.empty((St, Ct), dtype=torch.float32, device=dev)
        stat1 = torch.empty((St, Ct * D), dtype=torch.float32, device=dev)
        for a0 in range(0, St, 2048):                  # synthetic statistics, built in slices (float64 temporaries)
            b0 = min(St, a0 + 2048)
            post = torch.softmax(2.0 * torch.randn(b0 - a0, Ct, generator=gen, device=dev), dim=1)
            stat0[a0:b0] = (2000.0 * post).to(torch.float32)
            n_rep = stat0[a0:b0].repeat_interleave(D, dim=1).to(torch.float64)
            stat1[a0:b0] = (n_rep * mu_sv + torch.sqrt(n_rep) * sd_sv *
                            torch.randn(b0 - a0, Ct * D, generator=gen, device=dev, dtype=torch.float64)).to(torch.float32)
            del n_rep, post
        mu_t, ic_t = ubm_t.get_mean_super_vector(), ubm_t.get_invcov_super_vector()
        eng.tv_whiten(stat0[:64], stat1[:64], mu_t, ic_t)                       # first-call costs out of the timings
        dt_wh, (ft, naug) = timed(lambda: eng.tv_whiten(stat0, stat1, mu_t, ic_t), 1)
        # extract_ivectors' whitening: f~ straight into the rows image of its bf16 x 3 planes (no float32 f~ kept)
        eng.tv_whiten_planes(stat0[:64], stat1[:64], mu_t, ic_t)
        dt_whp, (ft_rows, _naug2) = timed(lambda: eng.tv_whiten_planes(stat0, stat1, mu_t, ic_t), 1)
        del stat1, _naug2
        torch.cuda.empty_cache()
        fa = sk.FactorAnalyser()
        tv_state = {"T": torch.as_tensor(0.1 * np.random.RandomState(7).randn(Ct * D, R), device=dev)}

        # f~ is a constant of a training run: FactorAnalyser splits it once into the bf16 x 3 planes that the two
        # products reading it (aux, C accumulation) take (timed here, outside the iteration, like the whitening)
        torch.cuda.synchronize()
        t_pl = time.perf_counter()
        ft_planes = eng.tv_ftilde_planes(ft, Ct, D)
        torch.cuda.synchronize()
        t_pl = time.perf_counter() - t_pl

        planes_used = ft_planes is not None

        def tv_iter():
            tv_state["T"] = fa._tv_iteration(ft, naug, tv_state["T"], Ct, D, R, args.tv_sessions, True, args.tv_batch,
                                             ft_planes=ft_planes)
        tv_iter()                                                               # warm-up iteration
        eng.profile(True)
        dt_tv, _ = timed(tv_iter, 1)
        prof_tv = eng.profile_read()
        del ft_planes                        # extraction is a single pass over f~: no planes (nothing to amortise)
        torch.cuda.empty_cache()
        T32, G = eng.tv_prepare(tv_state["T"], Ct, D)
        ex = {"ft_planes": ft_rows} if ft_rows is not None else {}
        ft_ex = None if ft_rows is not None else ft             # as FactorAnalyser.extract_ivectors runs it
        dt_iv, _ = timed(lambda: eng.tv_estep(naug, ft_ex, T32, G, Ct, D, batch=args.tv_extract_batch, uncertainty=False, **ex), 1)
        prof_iv = eng.profile_read()
        dt_ivu, _ = timed(lambda: eng.tv_estep(naug, ft_ex, T32, G, Ct, D, batch=args.tv_extract_batch, uncertainty=True, **ex), 1)
        eng.profile_read()
        eng.profile(False)
        flop_train, flop_iv = 9.18e8, 4.5e8          # algorithmic flop per session
        chain_dtype = "f64" if os.environ.get("SKM_TV_CHAIN", "").lower() == "f64" else "f32"
        chain_peak = PEAK_F64_MFMA_TFLOPS if chain_dtype == "f64" else PEAK_F32_MFMA_TFLOPS
        Sall = args.tv_sessions
        # float64 chain of the E-step (Cholesky, triangular inverse, X X'): algorithmic R^3 flop per session
        t_f64 = sum(v[1] for k, v in prof_tv.items()
                    if k in ("k_chol_diag", "k_gemm_chol_panel", "k_gemm_chol_update", "k_gemm_trinv_a", "k_gemm_trinv_b",
                             "k_gemm_xxt", "k_chol_row", "k_trinv_row"))
        result["tv"] = {
            "workload": "TV EM iteration (E-step + M-step + min-div) and extract_ivectors, R=%d, C=%d, D=%d, %d synthetic "
                        "sessions in total (BASELINE configs[4]), %d on rank 0" % (R, Ct, D, Sall, St),
            "n_gpus": world, "scaling": "strong",
            "train_sessions_per_s": Sall / dt_tv, "train_s_per_iteration": dt_tv,
            "train_algorithmic_tflops": flop_train * Sall / dt_tv / 1e12,
            "train_frac_of_f32_mfma_peak": flop_train * Sall / dt_tv / 1e12 / PEAK_F32_MFMA_TFLOPS / world,
            "ivectors_per_s": Sall / dt_iv,
            "ivectors_with_uncertainty_per_s": Sall / dt_ivu,
            "extract_algorithmic_tflops": flop_iv * Sall / dt_iv / 1e12,
            "extract_frac_of_f32_mfma_peak": flop_iv * Sall / dt_iv / 1e12 / PEAK_F32_MFMA_TFLOPS / world,
            # per-session chain (Cholesky, triangular inverse, X X'): float32 storage and MFMAs by default, float64 with
            # SKM_TV_CHAIN=f64; R^3 algorithmic flop per session, priced against the peak of the type it runs in
            "chain": {"dtype": chain_dtype, "ms": round(t_f64, 3),
                      "algorithmic_tflops": float(R) ** 3 * St / (t_f64 * 1e-3) / 1e12 if t_f64 else None,
                      "frac_of_mfma_peak": float(R) ** 3 * St / (t_f64 * 1e-3) / 1e12 / chain_peak if t_f64 else None,
                      "bound": "hbm (64-wide panel products: 8 flop/B in float64, 16 in float32; tools/time_chain_gemm.py)"},
            # L build and A accumulation (2 C P flop per session each): bf16 x 3 split products on the bf16 matrix cores
            # (csrc/skm_gemm_b3.hip; float32-grade products, float32 accumulation) unless SKM_TV_SPLIT=0; their rate is
            # still quoted against the FLOAT32 MFMA peak, so it may exceed 1
            "split_gemms": {
                "mode": "f32 mfma" if os.environ.get("SKM_TV_SPLIT", "1") == "0" else "bf16x3 split on the bf16 MFMA",
                "lbuild_ms": round(prof_tv.get("k_gemm_lbuild", (0, 0.0))[1], 3),
                "aacc_ms": round(prof_tv.get("k_gemm_aacc", (0, 0.0))[1], 3),
                "lbuild_frac_of_f32_mfma_peak": (2.0 * Ct * (R * (R + 1) // 2) * St / (prof_tv["k_gemm_lbuild"][1] * 1e-3) / 1e12
                                                 / PEAK_F32_MFMA_TFLOPS) if prof_tv.get("k_gemm_lbuild", (0, 0))[1] else None,
                "aacc_frac_of_f32_mfma_peak": (2.0 * Ct * (R * (R + 1) // 2) * St / (prof_tv["k_gemm_aacc"][1] * 1e-3) / 1e12
                                               / PEAK_F32_MFMA_TFLOPS) if prof_tv.get("k_gemm_aacc", (0, 0))[1] else None},
            "whitening": {"seconds_float32_ftilde": dt_wh, "seconds_rows_planes": dt_whp,
                          "extraction_from_planes": ft_rows is not None,
                          "note": "the whitening of the statistics (factor_analyser.py:76-79) is outside train_s_per_iteration "
                                  "and outside the extraction times, as in rounds 1-2; extract_ivectors uses the variant that "
                                  "writes f~ as bf16 x 3 planes (its aux product then runs on the bf16 matrix cores)"},
            "ftilde_planes": {"used": planes_used, "seconds_once_per_training_run": t_pl,
                              "note": "f~ split into three bf16 terms in both orientations (12 B per element) once per "
                                      "training run; every iteration's aux = f~ T and C += e_h' f~ then run on the bf16 "
                                      "matrix cores; not part of train_s_per_iteration (nor is the whitening)"},
            "train_kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof_tv.items())},
            "extract_kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof_iv.items())},
            "note": "M-step cost (R x R inverses of the rank's C/N components) is independent of the session count; it is "
                    "included in train_s_per_iteration"}
        del ft, naug, T32, G, stat0, ft_rows, ex
        torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            # API level on a bounded sample: FactorAnalyser.total_variability (one iteration) and extract_ivectors on
            # an in-memory StatServer -- host statistics in, host F / i-vectors out -- and the same calls on a
            # StatServer whose rows accumulate_stat left in HBM (keep_on_device=True): no upload of the statistics
            Sa = min(args.api_sessions, St)
            rs = np.random.RandomState(5)
            ssh = sk.StatServer(engine=eng)
            ssh.modelset = np.array(["m%05d" % i for i in range(Sa)])
            ssh.segset = ssh.modelset.copy()
            ssh.start = np.empty(Sa, "|O")
            ssh.stop = np.empty(Sa, "|O")
            if api_stats is not None and api_stats.stat0.shape[0] >= Sa:
                ssh.stat0, ssh.stat1 = api_stats.stat0[:Sa], api_stats.stat1[:Sa]
            else:
                ssh.stat0 = 2000.0 * rs.dirichlet(np.full(Ct, 0.3), size=Sa)
                ssh.stat1 = np.repeat(ssh.stat0, D, axis=1) * ubm_t.get_mean_super_vector() + rs.randn(Sa, Ct * D)
            tv0 = 0.1 * np.random.RandomState(7).randn(Ct * D, R)
            fa_api = sk.FactorAnalyser(engine=eng)
            fa_api.total_variability_single(ssh, ubm_t, R, nb_iter=1, min_div=True, tv_init=tv0)     # warm-up: workspaces and
            fa_api.extract_ivectors(ubm_t, ssh)                                                     # staging buffers exist
            t0 = time.perf_counter()
            fa_api.total_variability_single(ssh, ubm_t, R, nb_iter=1, min_div=True, tv_init=tv0)
            dt_a = time.perf_counter() - t0
            t0 = time.perf_counter()
            ivs = fa_api.extract_ivectors(ubm_t, ssh)
            dt_e = time.perf_counter() - t0
            api = {"sample": "%d sessions on one GPU, host float64 statistics in (%.1f GB), host F / i-vectors out; the "
                             "M-step of all %d components is included in the training figure" % (Sa, (ssh.stat0.nbytes + ssh.stat1.nbytes) / 1e9, Ct),
                   "train_sessions_per_s": Sa / dt_a, "train_seconds": dt_a,
                   "ivectors_per_s": Sa / dt_e, "extract_seconds": dt_e,
                   "iv_finite": bool(np.isfinite(ivs.stat1).all())}
            if api_stats is not None and api_stats.resident_rows() is not None and api_stats.stat0.shape[0] == Sa:
                t0 = time.perf_counter()
                fa_api.total_variability_single(api_stats, ubm_t, R, nb_iter=1, min_div=True, tv_init=tv0)
                api["train_seconds_resident"] = time.perf_counter() - t0
                t0 = time.perf_counter()
                fa_api.extract_ivectors(ubm_t, api_stats)
                api["extract_seconds_resident"] = time.perf_counter() - t0
            result["tv"]["api"] = api
        if world > 1:
            td.barrier()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(C, D)
        # the same host cores on the other two units of the metric (bounded samples, ~10 s each)
        if "bw_stats" in result:
            result["bw_stats"]["cpu_baseline"] = cpu_baseline_bw(2048, D)
        if "tv" in result:
            result["tv"]["cpu_baseline"] = cpu_baseline_tv(2048, D, 400)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()
    # release the library context (events, workspace) while the HIP runtime is still up
    torch.cuda.synchronize()
    eng.close()


if __name__ == "__main__":
    main()
