"""Golden vectors recorded from the reference (see gen_golden.py).  Data only."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    """Return {key: np.ndarray}; fp16-packed exact inputs are widened back to fp32."""
    out = {}
    with np.load(os.path.join(_HERE, name + ".npz")) as z:
        for k in z.files:
            a = z[k]
            out[k] = a.astype(np.float32) if a.dtype == np.float16 else a
    return out
