import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('exp-trmf-nips16_amd', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, trmf
from helpers import make_model
from trmf import session
dtype = np.float32
T0, Tn, n, k, lags = 260, 20, 31, 5, [1, 2, 7]
d = trmf.Model.syn_gen(T0 + Tn, n, k, lags, seed=4, dtype=np.float64)
Y = np.ascontiguousarray(3.0 * d['Y'] + 5.0 + 0.1 * np.random.RandomState(4).randn(T0 + Tn, n), dtype=dtype)
hyper = dict(lambdaI=0.5, lambdaAR=125.0, lambdaLag=2.0)
m0 = trmf.Model.initialize(Y[:T0], lags, k, seed=1, transform=True)
dev = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
s = session.Session(Y[:T0], dev, missing=False, **hyper)
s.set_transform(m0.transform); s.run(3).download()
first = make_model(dev.W, dev.H, dev.lag_val, dev.lag_set); first.transform = m0.transform
m1 = trmf.Model.initialize(Y, lags, k, seed=1, warm_start_model=first)
s.append_rows(Y[T0:]); s.set_transform(m1.transform)
dev1 = make_model(m1.W, m1.H, m1.lag_val, m1.lag_set); s.model = dev1
s.download(); print('W after append == host warm start:', np.array_equal(dev1.W, m1.W), np.array_equal(dev1.H, m1.H))
for it in range(1, 4):
    s.run(1).download()
    host1 = make_model(m1.W, m1.H, m1.lag_val, m1.lag_set)
    with session.Session(np.ascontiguousarray(m1.transform.preprocess(Y).astype(dtype)), host1, missing=False, **hyper) as s3:
        s3.run(it).download()
    print(it, 'W', np.abs(dev1.W - host1.W).max(), 'H', np.abs(dev1.H - host1.H).max(), 'Th', np.abs(dev1.lag_val - host1.lag_val).max())
