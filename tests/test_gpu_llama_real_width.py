"""Model-level parity at REAL width: two Llama-3-8B-shaped decoder layers (hidden 4096, 32 / 8 heads of
128, intermediate 14336; vocabulary cut to 32000 so the CPU oracle stays in test time) with random
N(0, 0.02) weights, a ragged prefill of 16 requests and two decode steps through ScheduleBatch ->
ModelRunner -> the HIP kernels, against the CPU oracle evaluated (a) in fp32 and (b) in the same 16-bit
dtype (every op rounding its output, as torch does).

The tiny-model tests (test_gpu_llama.py) use fixtures whose values are exact in bf16, so rounding and
accumulation order cannot show there; here every GEMM has K = 4096 / 14336 of non-trivial operands, so
an accumulation-order or rounding regression in any kernel of the path moves the logits.
Measured on MI355X (max |logit - ref| / max |ref| over prefill + 2 decode steps; printed by the test):
  fp16: HIP vs fp32 oracle 1.7e-3 .. 2.1e-3, HIP vs fp16 oracle 1.2e-3 .. 1.3e-3  (fp16 oracle vs fp32: 1.9e-3 .. 2.1e-3)
  bf16: HIP vs fp32 oracle 1.3e-2 .. 1.6e-2, HIP vs bf16 oracle 0.9e-2 .. 1.1e-2  (bf16 oracle vs fp32: 1.4e-2 .. 1.6e-2)
i.e. with N(0, 0.02) weights (small logits) a 16-bit evaluation of these layers is itself 2e-3 / 1.5e-2
of the logit scale away from fp32, torch's as much as ours; the HIP path is never further from fp32 than
torch's own 16-bit evaluation.  The asserted bounds sit ~1.5x above the measurements."""
import pytest
import torch

from oracle import llama as ollama
from oracle import ops

pytestmark = pytest.mark.gpu

# asserted bounds on max|dlogit| / max|logit|: (vs fp32 oracle, vs same-dtype oracle)
BOUNDS = {torch.float16: (3e-3, 2e-3), torch.bfloat16: (2.5e-2, 1.6e-2)}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_real_width_layers_against_fp32_and_same_dtype_oracle(dtype):
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    vocab, layers = 32000, 2
    cfg = ModelConfig(4096, 14336, layers, 32, 8, vocab, context_len=256)
    mr = ModelRunner(cfg, ServerArgs(max_total_tokens=4096, max_running_requests=16, disable_cuda_graph=True),
                     dtype=dtype, seed=3)
    worker = TpModelWorker(mr)
    shape = ollama.LlamaShape(4096, 14336, layers, 32, 8, vocab, False, 500000.0, None, 8192, 1e-5)
    w16 = {k: v.detach().cpu() for k, v in mr.model.named_parameters()}
    w32 = {k: v.float() for k, v in w16.items()}
    gen = torch.Generator().manual_seed(11)
    lens = [5, 64, 17, 96, 1, 33, 80, 9, 48, 65, 2, 71, 24, 90, 12, 40]          # bs 16, ragged
    reqs = [Req(str(i), torch.randint(0, vocab, (n,), generator=gen).tolist()) for i, n in enumerate(lens)]
    sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, mr.device)
    sb.prepare_for_extend()
    out, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
    hip = [out.next_token_logits.float().cpu()]

    kv32 = ollama.OracleKV(shape, 4096, 17, 260)
    kv16 = ollama.OracleKV(shape, 4096, 17, 260, dtype=dtype)
    ext = torch.tensor(lens, dtype=torch.int32)
    pos, start = ops.compute_position(torch.zeros(16, dtype=torch.int32), ext)

    def oracle(w, kv, mode, **kw):
        kv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu()[:17, :260])
        return ollama.forward(shape, w, kv, mode=mode, **kw).float()
    common = dict(input_ids=sb.input_ids.cpu(), positions=pos, req_pool_indices=sb.req_pool_indices.cpu(),
                  seq_lens=sb.seq_lens.cpu(), out_cache_loc=sb.out_cache_loc.cpu(), extend_seq_lens=ext,
                  extend_start_loc=start)
    ref32 = [oracle(w32, kv32, "extend", **common)]
    ref16 = [oracle(w16, kv16, "extend", **common)]
    for step in range(2):
        sb.output_ids = ref32[-1].argmax(-1).to(mr.device)      # both sides continue from the oracle's tokens
        sb.prepare_for_decode()
        out, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
        hip.append(out.next_token_logits.float().cpu())
        common = dict(input_ids=sb.input_ids.cpu(), positions=ops.clamp_position(sb.seq_lens.cpu()),
                      req_pool_indices=sb.req_pool_indices.cpu(), seq_lens=sb.seq_lens.cpu(),
                      out_cache_loc=sb.out_cache_loc.cpu())
        ref32.append(oracle(w32, kv32, "decode", **common))
        ref16.append(oracle(w16, kv16, "decode", **common))
    b32, b16 = BOUNDS[dtype]
    name = {torch.float16: "fp16", torch.bfloat16: "bf16"}[dtype]
    for i, what in enumerate(("ragged prefill bs=16", "decode step 1", "decode step 2")):
        scale = float(ref32[i].abs().max())
        d32 = float((hip[i] - ref32[i]).abs().max()) / scale
        d16 = float((hip[i] - ref16[i]).abs().max()) / scale
        dor = float((ref16[i] - ref32[i]).abs().max()) / scale
        agree = float((hip[i].argmax(-1) == ref32[i].argmax(-1)).float().mean())
        print(f"[parity] real width {name} {what}: HIP vs fp32 oracle {d32:.2e}, HIP vs {name} oracle {d16:.2e} "
              f"({name} oracle vs fp32 oracle {dor:.2e}); greedy tokens equal on {100 * agree:.0f} % of rows")
        assert d32 <= b32, f"{what}: {d32:.2e} of max|logit| vs the fp32 oracle (bound {b32:.1e})"
        assert d16 <= b16, f"{what}: {d16:.2e} of max|logit| vs the {name} oracle (bound {b16:.1e})"
        # the HIP path (fp32 accumulation inside every kernel, one rounding per op) must not be further
        # from the truth than torch's own 16-bit evaluation by more than a factor
        assert d32 <= 1.25 * dor + 1e-4, f"{what}: HIP {d32:.2e} vs oracle-in-{name} {dor:.2e}"
