"""Caller-side Bayesian-optimisation loop for BASELINE config 5 (end-to-end BO over the window start).

NOT part of the accelerated path: the reference's loop (BayesianOptimization.py:99-192) is 13 points of
sklearn GP regression on the CPU and SURVEY.md 8 keeps it out of scope.  The reference module cannot be
imported here (it imports cv2 and, circularly, the BO script), so this file offers a loop with the SAME call
signature and return value for users who want the whole pipeline from one package:

    xp, yp = bayesian_optimisation(n_iters, sample_loss, val_loader, nn_model, criterion, bounds,
                                   x0=None, n_pre_samples=5, gp_params=None, random_search=False,
                                   alpha=1e-5, epsilon=1e-7)

Same ingredients (RBF-kernel GP with normalize_y, expected improvement with greater_is_better=True, random
re-draw on duplicates).  One deliberate difference: the domain is the integer range [bounds[0][0],
bounds[0][1]] (the reference starts one L-BFGS-B run per integer, BayesianOptimization.py:85-90, and
sample_loss truncates its argument to int anyway), so the acquisition is maximised by evaluating EI on every
integer -- exact, and cheap because every sample_loss call is a table look-up after the engine's one
batched pass.
"""
import random

import numpy as np


def expected_improvement(x, gaussian_process, evaluated_loss, greater_is_better=False, n_params=1):
    """EI(x) (BayesianOptimization.py:16-54), returned positive; 0 where the GP is certain."""
    from scipy.stats import norm
    x = np.asarray(x, dtype=np.float64).reshape(-1, n_params)
    mu, sigma = gaussian_process.predict(x, return_std=True)
    best = np.max(evaluated_loss) if greater_is_better else np.min(evaluated_loss)
    sign = 1.0 if greater_is_better else -1.0
    imp = sign * (mu - best)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = imp / sigma
        ei = imp * norm.cdf(z) + sigma * norm.pdf(z)
    ei[sigma == 0.0] = 0.0
    return ei


def bayesian_optimisation(n_iters, sample_loss, val_loader, nn_model, criterion, bounds, x0=None, n_pre_samples=5,
                          gp_params=None, random_search=False, alpha=1e-5, epsilon=1e-7, rng=None):
    """-> (xp [n_pre+n_iters, 1], yp [n_pre+n_iters]) like BayesianOptimization.py:99-192."""
    import sklearn.gaussian_process as gp
    rng = rng or random
    bounds = np.asarray(bounds)
    lo, hi = int(bounds[0][0]), int(bounds[0][1])
    x_list, y_list = [], []
    starts = x0 if x0 is not None else [[rng.randint(lo, hi)] for _ in range(n_pre_samples)]
    for params in starts:
        x_list.append(list(params))
        y_list.append(sample_loss(params, val_loader, nn_model, criterion))
    if gp_params is not None:
        model = gp.GaussianProcessRegressor(**gp_params)
    else:
        model = gp.GaussianProcessRegressor(kernel=gp.kernels.RBF(), alpha=alpha, n_restarts_optimizer=10,
                                            normalize_y=True)
    grid = np.arange(lo, hi + 1, dtype=np.float64).reshape(-1, 1)
    for _ in range(n_iters):
        xp, yp = np.array(x_list, dtype=np.float64), np.array(y_list, dtype=np.float64)
        model.fit(xp, yp)
        if random_search:
            next_sample = [rng.randint(lo, hi)]
        else:
            ei = expected_improvement(grid, model, yp, greater_is_better=True)
            next_sample = [float(grid[int(np.argmax(ei)), 0])]
        if np.any(np.abs(np.asarray(next_sample) - xp) <= epsilon):      # duplicates break the GP
            next_sample = [rng.randint(lo, hi)]
        x_list.append(list(next_sample))
        y_list.append(sample_loss(next_sample, val_loader, nn_model, criterion))
    return np.array(x_list), np.array(y_list)
