"""Model-level parity at REAL width (Llama-3-8B-shaped decoder layers: hidden 4096, 32 / 8 heads of 128,
intermediate 14336; vocabulary cut to 32000 so the CPU oracle stays in test time): a ragged prefill and decode
steps through ScheduleBatch -> ModelRunner -> the HIP kernels, against the CPU oracle evaluated (a) in fp32 and
(b) in the same 16-bit dtype (every op rounding its output, as torch does).  The reference's own parity bar is a
32-character text match against HF (tests/e2e/test_engine.py:8-57); BASELINE.json asks for <= 1e-3 logit deviation.

Two cases:
  * "n002": 2 layers, N(0, 0.02) weights (the bench's initialisation; max |logit| ~ 6): every projection
    shrinks its input (std 0.02 * sqrt(K) ~ 1.3 .. 2.4 per unit input, but the residual stream stays at the
    embedding's 0.02 scale), so the few roundings of the hidden state weigh ~2e-3 of the logit scale in fp16.
  * "fanin": 8 layers, fan-in-scaled weights (std 1/sqrt(K); LM head 3/sqrt(K), embedding N(0, 1)): unit-scale
    activations through every layer and logits of O(10), the regime of a trained model - and deep enough for
    rounding to accumulate over 8 residual updates.

The tiny-model tests (test_gpu_llama.py) use fixtures whose values are exact in bf16, so rounding and accumulation
order cannot show there; here every GEMM has K = 4096 / 14336 of non-trivial operands, so an accumulation-order or
rounding regression in any kernel of the path moves the logits.

Metric: max |logit - ref| / max |ref| per step.  Measured on MI355X (printed by the test, quoted in BASELINE.md):
  n002  fp16: HIP vs fp32 oracle 1.7e-3 .. 2.1e-3 (torch fp16 vs fp32: 1.9e-3 .. 2.1e-3); bf16: 1.3e-2 .. 1.6e-2 (1.4e-2 .. 1.6e-2)
  fanin fp16: HIP vs fp32 oracle 1.2e-3 .. 1.4e-3 (torch fp16 vs fp32: 1.2e-3 .. 1.5e-3); bf16: 0.95e-2 .. 1.04e-2 (0.91e-2 .. 1.10e-2)
        (max |logit| 13.5 .. 14.8; greedy tokens equal to the fp32 oracle's on 100 % of rows in fp16, 88 - 100 % in bf16)
Asserted: (1) the HIP path is no further from the fp32 oracle than torch's own 16-bit evaluation of the same
layers + 10 %; (2) absolute bounds at <= 1.15 x the largest measured value.  Where 1e-3 is attainable: at kernel
level (tests/test_gpu_attention.py: <= 0.87 units of fp16 round-off) and for fp16 logits of O(10) over few layers;
it is NOT attainable by any 16-bit evaluation - torch's included - for bf16 (u = 2^-8 = 3.9e-3 per rounding) or for
tiny-logit initialisations, which is what assertion (1) pins instead."""
import pytest
import torch

from oracle import llama as ollama
from oracle import ops

pytestmark = pytest.mark.gpu

# largest measured max|dlogit| / max|logit| over the steps of a case (MI355X): (vs fp32 oracle, vs same-dtype oracle)
MEASURED = {
    ("n002", torch.float16): (2.07e-3, 1.31e-3), ("n002", torch.bfloat16): (1.56e-2, 1.06e-2),
    ("fanin", torch.float16): (1.38e-3, 1.59e-3), ("fanin", torch.bfloat16): (1.04e-2, 1.19e-2),
}
LOOSE = {torch.float16: (6e-3, 6e-3), torch.bfloat16: (5e-2, 5e-2)}     # used only until a case has been measured


def bounds(case, dtype):
    m = MEASURED[(case, dtype)]
    return LOOSE[dtype] if m[0] is None else (1.15 * m[0], 1.15 * m[1])


CASES = {
    # name: (layers, prompt lengths, decode steps)
    "n002": (2, [5, 64, 17, 96, 1, 33, 80, 9, 48, 65, 2, 71, 24, 90, 12, 40], 2),
    "fanin": (8, [3, 40, 17, 1, 29, 8, 33, 12], 2),
}


def init_weights(mr, case, seed):
    if case == "n002":
        return                                      # ModelRunner's own N(0, 0.02) / norm weights 1
    g = torch.Generator(device=mr.device).manual_seed(seed)
    for name, p in mr.model.named_parameters():
        if "norm" in name:
            p.data.fill_(1.0)
        elif "embed_tokens" in name:
            p.data.normal_(0.0, 1.0, generator=g)
        else:
            std = (3.0 if "lm_head" in name else 1.0) / p.shape[1] ** 0.5
            p.data.normal_(0.0, std, generator=g)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", ["n002", "fanin"])
def test_real_width_layers_against_fp32_and_same_dtype_oracle(case, dtype):
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    layers, lens, steps = CASES[case]
    vocab, bs = 32000, len(lens)
    cfg = ModelConfig(4096, 14336, layers, 32, 8, vocab, context_len=256)
    mr = ModelRunner(cfg, ServerArgs(max_total_tokens=4096, max_running_requests=bs, disable_cuda_graph=True),
                     dtype=dtype, seed=3)
    init_weights(mr, case, 5)
    worker = TpModelWorker(mr)
    shape = ollama.LlamaShape(4096, 14336, layers, 32, 8, vocab, False, 500000.0, None, 8192, 1e-5)
    w16 = {k: v.detach().cpu() for k, v in mr.model.named_parameters()}
    w32 = {k: v.float() for k, v in w16.items()}
    gen = torch.Generator().manual_seed(11)
    reqs = [Req(str(i), "", torch.randint(0, vocab, (n,), generator=gen).tolist(), None) for i, n in enumerate(lens)]
    sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=mr.device)
    sb.prepare_for_extend()
    out, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
    hip = [out.next_token_logits.float().cpu()]

    kv32 = ollama.OracleKV(shape, 4096, bs + 1, 260)
    kv16 = ollama.OracleKV(shape, 4096, bs + 1, 260, dtype=dtype)
    ext = torch.tensor(lens, dtype=torch.int32)
    pos, start = ops.compute_position(torch.zeros(bs, dtype=torch.int32), ext)

    def oracle(w, kv, mode, **kw):
        kv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu()[:bs + 1, :260])
        return ollama.forward(shape, w, kv, mode=mode, **kw).float()
    common = dict(input_ids=sb.input_ids.cpu(), positions=pos, req_pool_indices=sb.req_pool_indices.cpu(),
                  seq_lens=sb.seq_lens.cpu(), out_cache_loc=sb.out_cache_loc.cpu(), extend_seq_lens=ext,
                  extend_start_loc=start)
    ref32 = [oracle(w32, kv32, "extend", **common)]
    ref16 = [oracle(w16, kv16, "extend", **common)]
    for step in range(steps):
        sb.output_ids = ref32[-1].argmax(-1).to(mr.device)      # both sides continue from the oracle's tokens
        sb.prepare_for_decode()
        out, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
        hip.append(out.next_token_logits.float().cpu())
        common = dict(input_ids=sb.input_ids.cpu(), positions=ops.clamp_position(sb.seq_lens.cpu()),
                      req_pool_indices=sb.req_pool_indices.cpu(), seq_lens=sb.seq_lens.cpu(),
                      out_cache_loc=sb.out_cache_loc.cpu())
        ref32.append(oracle(w32, kv32, "decode", **common))
        ref16.append(oracle(w16, kv16, "decode", **common))
    b32, b16 = bounds(case, dtype)
    name = {torch.float16: "fp16", torch.bfloat16: "bf16"}[dtype]
    worst = [0.0, 0.0, 0.0]
    for i in range(steps + 1):
        what = f"ragged prefill bs={bs}" if i == 0 else f"decode step {i}"
        scale = float(ref32[i].abs().max())
        d32 = float((hip[i] - ref32[i]).abs().max()) / scale
        d16 = float((hip[i] - ref16[i]).abs().max()) / scale
        dor = float((ref16[i] - ref32[i]).abs().max()) / scale
        worst = [max(worst[0], d32), max(worst[1], d16), max(worst[2], dor)]
        agree = float((hip[i].argmax(-1) == ref32[i].argmax(-1)).float().mean())
        print(f"[parity] real width {case} ({layers} layers, max|logit| {scale:.2f}) {name} {what}: HIP vs fp32 oracle "
              f"{d32:.2e}, HIP vs {name} oracle {d16:.2e} (torch-{name} oracle vs fp32 oracle {dor:.2e}); greedy tokens "
              f"equal on {100 * agree:.0f} % of rows")
        assert d32 <= b32, f"{what}: {d32:.2e} of max|logit| vs the fp32 oracle (bound {b32:.2e})"
        assert d16 <= b16, f"{what}: {d16:.2e} of max|logit| vs the {name} oracle (bound {b16:.2e})"
        # the HIP path (fp32 accumulation inside every kernel, one rounding per op) must not be further from the
        # truth than torch's own 16-bit evaluation of the same layers + 10 %
        assert d32 <= 1.10 * dor + 1e-4, f"{what}: HIP {d32:.2e} vs torch-in-{name} {dor:.2e} from the fp32 oracle"
    print(f"[parity] real width {case} {name} WORST over steps: HIP vs fp32 {worst[0]:.2e}, HIP vs {name} oracle "
          f"{worst[1]:.2e}, torch-{name} vs fp32 {worst[2]:.2e}; north-star target 1.0e-3")
