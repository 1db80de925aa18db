"""Shared test helpers: tolerances, golden loading, random paged-KV problems."""
import numpy as np
import torch

from tests import golden

DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def T(a, device="cpu", dtype=None):
    t = torch.from_numpy(np.asarray(a))
    if dtype is not None and t.is_floating_point():
        t = t.to(dtype)
    return t.to(device)


def tol(dtype, kind="attn"):
    """(rtol, atol_scale): |hip - ref| <= rtol*|ref| + atol_scale*max|ref|.

    North-star bar: <= 1e-3 relative at fp16.  bf16 outputs carry 8 significant bits, so one
    output rounding alone is up to 2^-8 relative: bf16 bar = 2^-8 (output quantisation) + 1e-3.
    fp32 bar: 2e-5 (summation order / v_exp_f32 differ from the CPU libm path)."""
    if dtype == torch.float32:
        return (2e-5, 2e-5) if kind == "attn" else (2e-6, 2e-6)
    if dtype == torch.float16:
        return 1e-3, 1e-3
    return 2.0 ** -8 + 1e-3, 1e-3


def assert_close(hip, ref, dtype, kind="attn", what="", both_rounded=False, vmax=None):
    """both_rounded: `ref` is itself a HIP output in `dtype` (e.g. another split size): allow one
    more output quantisation step.
    vmax: max |V| of the attended rows.  Like the reference's kernels (decode_attention.py:427,
    extend_attention.py:155 `p = p.to(v.dtype)`) the matrix-core kernels feed P to the P.V product
    in the KV dtype; that rounding perturbs an output by at most eps_dtype/2 * max|V| (sum p = l),
    independent of the output's own size.  Half of that bound is allowed on top of the 1e-3 bar."""
    hip = hip.detach().float().cpu().double()
    ref = ref.detach().float().cpu().double()
    assert hip.shape == ref.shape, (hip.shape, ref.shape)
    assert torch.isfinite(hip).all(), f"{what}: non-finite output"
    rtol, ascale = tol(dtype, kind)
    if both_rounded and dtype != torch.float32:
        rtol += {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]
    scale = float(ref.abs().max()) if ref.numel() else 0.0
    err = (hip - ref).abs()
    bound = rtol * ref.abs() + ascale * scale
    if vmax is not None and dtype != torch.float32:
        bound = bound + 0.5 * {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype] * 0.5 * float(vmax)
    bad = err > bound
    rel_scaled = float(err.max()) / scale if scale > 0 else 0.0
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {float(err.max()):.3e} "
        f"= {rel_scaled:.3e} of max|ref| (scale {scale:.3e}, rtol {rtol:.2e}), "
        f"max err/bound {float((err / bound.clamp_min(1e-30)).max()):.3f}")


UNIT_ROUNDOFF = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}


def attn_error_units(hip, ref, abs_ref, dtype):
    """max over elements of |hip - ref| / (u * (|ref| + abs_ref)), u = the dtype's unit roundoff.

    The error model of a 16-bit attention kernel with fp32 accumulation: the output is rounded once
    (<= u |ref|), and the probabilities enter P.V rounded to the KV dtype like the reference's own
    `p.to(v.dtype)` (decode_attention.py:427, extend_attention.py:155), which moves an output by at
    most u * sum_i p_i |v_i| / l = u * A, where A is the attention output computed with |V|
    (`abs_ref`).  Everything else (fp32 scores, exp2, sums) is orders of magnitude below u.  So a
    correct kernel stays below ~1 unit; where |ref| ~ A (no cancellation) one unit is a relative
    error of 2u = 9.8e-4 at fp16 - the north-star's "1e-3 relative fp16" - without the blanket
    `1e-3 * max|ref|` allowance of assert_close."""
    hip = hip.detach().float().cpu().double()
    ref = ref.detach().float().cpu().double()
    abs_ref = abs_ref.detach().float().cpu().double()
    assert hip.shape == ref.shape == abs_ref.shape, (hip.shape, ref.shape, abs_ref.shape)
    assert torch.isfinite(hip).all(), "non-finite output"
    u = UNIT_ROUNDOFF[dtype]
    denom = u * (ref.abs() + abs_ref) + 1e-30
    return float(((hip - ref).abs() / denom).max())


def assert_attn_close(hip, ref, abs_ref, dtype, what="", units=1.0):
    """16-bit attention output against the fp32 oracle under the error model of attn_error_units:
    at most one unit (measured on MI355X: 0.28 - 0.81 units over every kernel, dtype, shape and
    full-size configuration of the suite)."""
    got = attn_error_units(hip, ref, abs_ref, dtype)
    print(f"[parity] {what}: max error {got:.3f} units of u(|ref| + A), u = 2^{int(torch.log2(torch.tensor(UNIT_ROUNDOFF[dtype])))}"
          f" (bound {units})")
    assert got <= units, f"{what}: max error {got:.3f} units of u*(|ref| + A) exceeds {units}"


def paged_problem(seed, bs, Hq, Hkv, D, seq_lens, dtype, device, extra_slots=7, extra_rows=3,
                  ctx_pad=4, scale=1.0):
    """Random KV pool with a random slot permutation (fragmented free-list) + q."""
    g = torch.Generator().manual_seed(seed)
    seq = torch.as_tensor(seq_lens, dtype=torch.int64)
    total = int(seq.sum())
    P = total + extra_slots
    k_buf = (torch.randn(P + 1, Hkv, D, generator=g) * scale).to(dtype)
    v_buf = (torch.randn(P + 1, Hkv, D, generator=g) * scale).to(dtype)
    r2t = torch.zeros(bs + extra_rows, int(seq.max()) + ctx_pad, dtype=torch.int32)
    req = torch.randperm(bs + extra_rows, generator=g)[:bs].to(torch.int64)
    perm = torch.randperm(P, generator=g) + 1
    off = 0
    for b in range(bs):
        L = int(seq[b])
        r2t[req[b], :L] = perm[off:off + L].to(torch.int32)
        off += L
    q = (torch.randn(bs, Hq, D, generator=g) * scale).to(dtype)
    d = lambda t: t.to(device)
    return dict(q=d(q), k_buffer=d(k_buf), v_buffer=d(v_buf), req_to_token=d(r2t),
                req_pool_indices=d(req), seq_lens=d(seq))


def cpu(d):
    return {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
