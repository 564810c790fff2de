#!/usr/bin/env python3
"""Rolling-window forecast evaluation with the settings of the paper's experiment scripts
(python/exp-scripts/run_electricity.py / run_traffic.py of the reference), through this package's front end.

    python examples/rolling_forecast.py [--data electricity.npy] [--preset electricity|traffic] [--windows 7] [--iters 40]

Without --data a synthetic low-rank + autoregressive matrix of the data set's shape is generated (the data sets are
not redistributable).  Every window is trained on ONE resident GPU session: the series are uploaded once, each window
appends its 24 new timestamps and re-applies the per-series normalisation on the device."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'exp-trmf-nips16_amd'))
import trmf  # noqa: E402

LAGS = list(range(1, 25)) + list(range(7 * 24, 8 * 24))              # one day back, and the same day a week before
PRESETS = {                                                         # shape, rank and regularisation of the two scripts
    'electricity': dict(shape=(26304, 370), k=60, lambdaI=0.5, lambdaAR=125, lambdaLag=2),
    'traffic': dict(shape=(10560, 963), k=40, lambdaI=2, lambdaAR=625, lambdaLag=0.5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', help='.npy file, timestamps x series')
    ap.add_argument('--preset', default='electricity', choices=sorted(PRESETS))
    ap.add_argument('--windows', type=int, default=7)
    ap.add_argument('--iters', type=int, default=40)
    args = ap.parse_args()
    cfg = PRESETS[args.preset]
    if args.data:
        Y = np.ascontiguousarray(np.load(args.data), dtype=np.float64)
    else:
        T, n = cfg['shape']
        d = trmf.Model.syn_gen(T, n, cfg['k'], LAGS, seed=0, dtype=np.float64)
        level = np.random.RandomState(0).lognormal(3.0, 1.5, n)
        Y = np.ascontiguousarray((d['Y'] + 0.05 * np.random.RandomState(1).randn(T, n)) * level + 2.0 * level)
    t0 = time.time()
    metrics = trmf.rolling_validate(Y, LAGS, cfg['k'], 24, args.windows, cfg['lambdaI'], cfg['lambdaAR'], cfg['lambdaLag'],
                                    max_iter=args.iters, threshold=None, transform=True, seed=0, missing=False)
    print('{} x {} series, {} windows x {} iterations: {:.2f} s'.format(Y.shape[0], Y.shape[1], args.windows, args.iters, time.time() - t0))
    print(metrics)


if __name__ == '__main__':
    main()
