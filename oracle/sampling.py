"""CPU oracle for the sampler (TEST INFRASTRUCTURE ONLY - see oracle/ops.py header).

Two layers, both restating nn/layers/sampler.py of the reference (paths relative to
``/root/reference/scratchpad``):

1. ``*_reference`` functions: the reference's torch formulation (sort, fp32 cumsum, masks) -
   sampler.py:195-232.  Pinned by ``tests/golden/sampling.npz`` (the reference's own functions run
   on CPU with ``torch.multinomial`` intercepted, tests/golden/gen_golden.py::gen_sampling).

2. ``select`` / ``sample`` / ``renorm``: the same filter stated WITHOUT a sort, on exact integers,
   which is the definition the HIP kernel implements bit for bit:
     * ``fx(p) = floor(p * 2^48)`` (exact in float64),
     * tokens are ranked by (p descending, token id ascending); a token is kept iff
       rank < top_k  and  sum of fx over higher-ranked tokens <= fx(top_p)  and  p >= p_max*min_p
       (fp32 product) - i.e. sampler.py:205-213 with the cumulative sum carried in integers,
     * the sample for a uniform u is the token whose half-open interval of the kept cumulative
       mass, taken in token-id order, contains ``r = min(floor(u * total), total - 1)``.
   The integer formulation removes the dependence on summation order (the reference's fp32 cumsum
   over a 128k vocabulary wobbles by ~1e-6), so every TP rank draws the same token from the same
   logits - the property sampler.py:146-157 asks of the sampling kernels.
   Layer 2 agrees with layer 1 except for tokens whose exclusive cumulative mass is within fp32
   rounding of top_p, and in the order of exactly tied probabilities (torch.sort is not stable);
   tests/test_oracle_golden.py checks both statements on the fixtures.
"""
import numpy as np
import torch

FX_BITS = 48
FX_ONE = float(2 ** FX_BITS)


# --------------------------------------------------------------------------- layer 1
def softmax_temperature(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    """sampler.py:71-73: logits.div_(temperatures); softmax(dim=-1) in fp32."""
    return torch.softmax(logits.float() / temperatures.float().view(-1, 1), dim=-1)


def filter_sorted_reference(probs, top_ks, top_ps, min_ps=None):
    """sampler.py:202-213: returns (probs_sort with dropped entries zeroed, probs_idx)."""
    probs_sort, probs_idx = probs.sort(dim=-1, descending=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    rank = torch.arange(probs.shape[-1]).view(1, -1)
    probs_sort[rank >= top_ks.view(-1, 1)] = 0.0
    probs_sort[(probs_sum - probs_sort) > top_ps.view(-1, 1)] = 0.0
    if min_ps is not None:
        thr = probs_sort[:, 0] * min_ps
        probs_sort[probs_sort < thr.view(-1, 1)] = 0.0
    return probs_sort, probs_idx


def top_p_normalize_reference(probs, top_ps):
    """sampler.py:224-232 (top_p_normalize_probs_torch)."""
    probs_sort, probs_idx = probs.sort(dim=-1, descending=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    probs_sort[(probs_sum - probs_sort) > top_ps.view(-1, 1)] = 0.0
    probs_sort.div_(probs_sort.sum(dim=-1, keepdim=True))
    return torch.zeros_like(probs_sort).scatter_(-1, probs_idx, probs_sort)


# --------------------------------------------------------------------------- layer 2
def fx(p: np.ndarray) -> np.ndarray:
    return np.floor(p.astype(np.float64) * FX_ONE).astype(np.uint64)


def select(probs_row: np.ndarray, top_k: int, top_p: float, min_p: float = 0.0):
    """Keep mask (bool[vocab]) and total kept mass (python int, fixed point) of one row."""
    p = np.asarray(probs_row, dtype=np.float32)
    n = p.shape[0]
    order = np.lexsort((np.arange(n), -p.astype(np.float64)))      # p desc, id asc
    w = [int(x) for x in fx(p[order])]
    top_p_fx = int(np.floor(np.float64(np.float32(top_p)) * FX_ONE))
    thr = np.float32(p.max()) * np.float32(min_p)                  # fp32 product
    keep = np.zeros(n, dtype=bool)
    acc = 0
    for rank, (tok, wt) in enumerate(zip(order, w)):
        if rank >= top_k or acc > top_p_fx or p[tok] < thr:
            break
        keep[tok] = True
        acc += wt
    return keep, acc


def sample(probs_row: np.ndarray, top_k: int, top_p: float, min_p: float, u: float) -> int:
    keep, total = select(probs_row, top_k, top_p, min_p)
    if total == 0:
        return 0
    w = fx(np.asarray(probs_row, dtype=np.float32))
    r = int(np.floor(np.float64(np.float32(u)) * np.float64(total)))
    r = min(r, total - 1)
    acc = 0
    for tok in np.flatnonzero(keep):
        acc += int(w[tok])
        if acc > r:
            return int(tok)
    raise AssertionError("unreachable: r < total")


def renorm(probs_row: np.ndarray, top_k: int, top_p: float, min_p: float = 0.0) -> np.ndarray:
    """Kept probabilities divided by the kept mass, dropped ones zero (fp32)."""
    keep, total = select(probs_row, top_k, top_p, min_p)
    p = np.asarray(probs_row, dtype=np.float32)
    out = np.zeros_like(p)
    if total:
        out[keep] = p[keep] / np.float32(np.float64(total) / FX_ONE)
    return out
