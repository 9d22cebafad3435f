import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import sidekit_amd as sk
eng = sk.default_engine()
rs = np.random.RandomState(0)
sess = [rs.randn(n, 60).astype(np.float32 if i % 2 else np.float64) for i, n in enumerate([0, 1, 33, 2000, 700000, 5, 1200000, 17])]
ref = np.concatenate([s.astype(np.float32) for s in sess])
X = eng.upload_sessions(sess)
torch.cuda.synchronize()
assert X.shape == ref.shape and np.array_equal(X.cpu().numpy(), ref)
big = rs.randn(3_000_000, 60)
Y = eng.upload_frames(big)
assert np.array_equal(Y.cpu().numpy(), big.astype(np.float32))
sess2 = [rs.randn(2000, 60).astype(np.float32) for _ in range(4000)]
for f in (lambda: eng.upload_frames(np.concatenate(sess2)), lambda: eng.upload_sessions(sess2)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); print("%.3f s" % (time.perf_counter() - t0))
print("ok")
