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


def rope_scaling(g, i):
    """the rope_scaling dict of case i of rotary.npz (None: plain RoPE): llama3 parameters are stored as an array of
    four numbers, every other variant as the JSON text of the dict handed to the reference's get_rope"""
    import json
    if f"c{i}_scaling" in g:
        f = g[f"c{i}_scaling"]
        return {"rope_type": "llama3", "factor": float(f[0]), "low_freq_factor": float(f[1]),
                "high_freq_factor": float(f[2]), "original_max_position_embeddings": int(f[3])}
    if f"c{i}_scaling_json" in g:
        return json.loads(str(g[f"c{i}_scaling_json"]))
    return None
