"""CPU restatement of the reference optimiser step -- TEST INFRASTRUCTURE ONLY (see the header of oracle/raster_oracle.py).

Follows scene/OurAdam.py:249-337 (_single_tensor_adam, the row-sparse update used by train_single.py:171-176 and
train_coarse.py:133-134) and :339-420 (_single_tensor_adam2, dense, taken when relevant.size(0) == 0), non-capturable,
no amsgrad, in float64 so that it can arbitrate between float32 implementations.

Pinned: scene/OurAdam.py is pure Python and imports on CPU, so tests/golden/make_golden.py runs the REFERENCE class
itself on seeded inputs and stores its outputs (tests/golden/ref_adam_golden.npz); tests/test_adam_cpu.py checks this
restatement against them.
"""
import math

import numpy as np


def adam_rows(param, grad, exp_avg, exp_avg_sq, step, relevant, *, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0):
    """One step for one tensor; arrays are [P, ...]; `step` is the step count AFTER the increment
    (scene/OurAdam.py:268).  relevant: int row indices, or None / empty for the dense update.
    Returns (param, exp_avg, exp_avg_sq) as new float64 arrays."""
    p = np.array(param, dtype=np.float64)
    m = np.array(exp_avg, dtype=np.float64)
    v = np.array(exp_avg_sq, dtype=np.float64)
    g = np.asarray(grad, dtype=np.float64)
    rows = slice(None) if relevant is None or len(relevant) == 0 else np.asarray(relevant).reshape(-1)
    gr, pr = g[rows], p[rows]
    if weight_decay != 0:
        gr = gr + weight_decay * pr                                    # :270-271
    mr = m[rows] * beta1 + (1 - beta1) * gr                            # :275
    vr = v[rows] * beta2 + (1 - beta2) * gr * gr                       # :276
    bc1 = 1 - beta1 ** step                                            # :305-306
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(vr) / math.sqrt(bc2) + eps                         # :319
    p[rows] = pr - (lr / bc1) * (mr / denom)                           # :322
    m[rows] = mr                                                       # :325-328
    v[rows] = vr
    return p, m, v
