"""An understated ``seq_lens_sum`` is reported, not computed around.

The decode split plan and its partial workspace are sized from the HOST's bound on sum(seq_lens)
(``ForwardBatch.seq_lens_sum``; the reference's producer is ScheduleBatch.prepare_for_decode,
scheduler/schedule_batch.py:1230-1308, which keeps it equal to the device-side lengths).  If a caller breaks that
contract the plan kernel lists only the items that fit and the affected rows are wrong: the plan's word 2 carries the
number of items the lengths needed, HipAttnBackend reads it back (at once with ``strict_plan_check``, else one plan
later / in ``check_plans()``) and raises."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _runner():
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs
    cfg = ModelConfig(256, 512, 2, 4, 2, 512, context_len=2048, max_position_embeddings=2048)
    args = ServerArgs(max_total_tokens=4096, max_running_requests=8, disable_cuda_graph=True)
    return ModelRunner(cfg, args, dtype=torch.bfloat16, seed=5)


def _decode_batch(mr, lens, claimed_sum):
    from scratchpad_amd.forward_info import ForwardMode, ModelWorkerBatch
    dev = mr.device
    bs = len(lens)
    table = mr.req_to_token_pool.req_to_token
    g = torch.Generator().manual_seed(1)
    perm = (torch.randperm(4000, generator=g) + 1).to(torch.int32)
    off, loc = 0, []
    for b, n in enumerate(lens):
        table[b, :n] = perm[off:off + n].to(dev)
        loc.append(int(perm[off + n - 1]))
        off += n
    return ModelWorkerBatch(bid=1, forward_mode=ForwardMode.DECODE,
                            input_ids=torch.randint(0, 512, (bs,), generator=g).to(dev),
                            req_pool_indices=torch.arange(bs, device=dev), seq_lens=torch.tensor(lens, device=dev),
                            out_cache_loc=torch.tensor(loc, device=dev), seq_lens_sum=claimed_sum)


def test_understated_seq_lens_sum_raises_instead_of_returning_wrong_logits():
    from scratchpad_amd.model_runner import TpModelWorker
    mr = _runner()
    worker = TpModelWorker(mr)
    backend = mr.attn_backend
    lens = [600, 600, 600, 600]
    # the honest step: no error now or later
    worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))
    backend.check_plans()
    # sum claimed as 400: the launch is sized for 400 / 64 + 4 = 10 items, the (clamped) lengths need 4 x 7
    bad = _decode_batch(mr, lens, 400)
    worker.forward_batch_generation(bad)                 # deferred mode: the step itself does not synchronise ...
    with pytest.raises(RuntimeError, match="split plan overflow.*seq_lens_sum"):
        backend.check_plans()                            # ... the check does
    backend.check_plans()                                # reported once
    # ... and without an explicit check the NEXT step's plan build reports it (after a sync point, e.g. sampling)
    worker.forward_batch_generation(bad)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="split plan overflow"):
        worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))
    backend.check_plans()
    # strict mode: the step that would be wrong raises before any layer runs
    backend.strict_plan_check = True
    with pytest.raises(RuntimeError, match="split plan overflow"):
        worker.forward_batch_generation(bad)
    worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))


def test_fused_split_merge_in_the_backend_gives_the_same_logits(monkeypatch):
    """HipAttnBackend.fused_split_merge = "1" (SP_DECODE_FUSE_MERGE=1; measured slower under graph replay, hence not the
    default): the plans carry arrival counters and every layer's decode launch merges its own splits - the logits of a
    step with ~10 splits per request are the bits of the default (merge launch) path."""
    from scratchpad_amd.attention import HipAttnBackend
    from scratchpad_amd.model_runner import TpModelWorker
    lens = [600, 130, 64, 1999]
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setattr(HipAttnBackend, "fused_split_merge", mode)
        mr = _runner()
        backend = mr.attn_backend
        assert (backend._plan_groups > 0) == (mode == "1") and backend._fuse(len(lens)) == (mode == "1")
        batch = _decode_batch(mr, lens, sum(lens))
        out, _ = TpModelWorker(mr).forward_batch_generation(batch)
        backend.check_plans()
        outs.append(out.next_token_logits.float().cpu())
        if mode == "1":
            plan, slots, _ = backend.forward_metadata[3][0]
            assert int(plan[4 + len(lens) + 2 * slots:].abs().sum()) == 0, "arrival counters back at zero after 2 layers"
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_near_uniform_batches_are_not_split_and_a_wrong_hint_only_costs_speed(monkeypatch):
    """ModelWorkerBatch.seq_lens_max_hint (advisory; ScheduleBatch keeps it like seq_lens_sum): a batch whose longest
    request is within 1.35 x of the mean and that still has a workgroup per CU unsplit gets ONE split per request (no
    partials, nothing to merge); a missing, too-small or absurd hint changes the split size at most, never the logits
    beyond the rounding of another summation order."""
    from scratchpad_amd.attention import HipAttnBackend
    from scratchpad_amd.model_runner import TpModelWorker
    monkeypatch.setattr(HipAttnBackend, "TARGET_ITEMS", 4)          # "one workgroup per CU" scaled down to this 4-request batch
    lens = [600, 610, 620, 605]
    mr = _runner()
    worker = TpModelWorker(mr)
    backend = mr.attn_backend
    outs, chunks = {}, {}
    for name, hint in (("none", None), ("exact", 620), ("low", 100), ("absurd", 10 ** 7)):
        batch = _decode_batch(mr, lens, sum(lens))
        batch.seq_lens_max_hint = hint
        out, _ = worker.forward_batch_generation(batch)
        backend.check_plans()
        outs[name] = out.next_token_logits.float().cpu()
        chunks[name] = int(backend.forward_metadata[3][0][0][1])
    assert chunks["exact"] == 640, "one split covers the longest request (rounded up to 64 keys)"
    assert chunks["none"] == chunks["low"] == chunks["absurd"] <= HipAttnBackend.MAX_CHUNK
    assert torch.equal(outs["none"], outs["low"]) and torch.equal(outs["none"], outs["absurd"])
    scale = float(outs["none"].abs().max())
    assert float((outs["exact"] - outs["none"]).abs().max()) <= 2e-2 * scale      # bf16 model, another summation order
