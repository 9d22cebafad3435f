"""Do two TV E-steps on two streams of ONE GPU finish sooner than back to back?  An upper bound on what overlapping
the MFMA-bound products of one batch with the memory-bound per-session chain of another can buy.
   python tools/probe_overlap.py [sessions per stream] [train]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sidekit_amd as sk

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
train = len(sys.argv) > 2 and sys.argv[2] == "train"
C, D, R = 2048, 60, 400
batch = 2048
dev = torch.device("cuda:0")
engs = [sk.HipEngine(0), sk.HipEngine(0)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
gen = torch.Generator(device=dev)
gen.manual_seed(1)
T = 0.1 * torch.randn((C * D, R), generator=gen, device=dev, dtype=torch.float64)
data = []
for e in engs:
    post = torch.softmax(2.0 * torch.randn(S, C, generator=gen, device=dev), dim=1)
    naug = torch.zeros((S, e.lib.skm_tv_naug_stride(C)), dtype=torch.float32, device=dev)
    naug[:, :C] = 2000.0 * post
    ft = torch.randn((S, C * D), generator=gen, device=dev, dtype=torch.float32)
    T32, G = e.tv_prepare(T, C, D)
    acc = None
    if train:
        acc = (torch.zeros((C + 1, R * R), dtype=torch.float64, device=dev), torch.zeros((R, C * D), dtype=torch.float64, device=dev))
    data.append((naug, ft, T32, G, acc))
torch.cuda.synchronize()


def job(i):
    naug, ft, T32, G, acc = data[i]
    with torch.cuda.stream(streams[i]):
        if acc is None:
            engs[i].tv_estep(naug, ft, T32, G, C, D, batch=batch)
        else:
            engs[i].tv_estep(naug, ft, T32, G, C, D, batch=batch, A_aug=acc[0], Cacc=acc[1])


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def both():
    th = [threading.Thread(target=job, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for i in (0, 1):
    job(i)
torch.cuda.synchronize()
for rep in range(3):
    t_seq = timed(lambda: (job(0), job(1)))
    t_par = timed(both)
    print("%s, 2 x %d sessions: back to back %.1f ms, two streams %.1f ms  (x%.3f)" % (
        "train" if train else "extract", S, t_seq * 1e3, t_par * 1e3, t_seq / t_par), flush=True)
