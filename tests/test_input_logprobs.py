"""Input (prompt) log-probabilities: the scheduler side that decides which token's logprob each prompt position
reports (schedule_batch.py:952-1040) and the logits-processor side that cuts them out (logits_processor.py:148-340).
Golden vectors: tests/golden/input_logprobs.npz, recorded by running the reference's own ScheduleBatch and
LogitsProcessor (gen_golden.py gen_input_logprobs).

CPU part: the oracle restatement and the product's HOST logic (index plans, ScheduleBatch bookkeeping) against the
fixture.  GPU part: the product's LogitsProcessor (library GEMM + the gather) and a full model step."""
import numpy as np
import pytest
import torch

from oracle import logprobs as olp
from oracle import ops
from tests import golden


def _cases():
    g = golden.load("input_logprobs")
    for ci in range(int(g["num_cases"])):
        yield ci, g, {k[len(f"c{ci}_"):]: v for k, v in g.items() if k.startswith(f"c{ci}_")}


def _ragged(c, field, nreq, has_ids=None):
    out = []
    for i in range(nreq):
        if f"{field}_r{i}" not in c:
            out.append(None)
            continue
        rows = int(c[f"{field}_r{i}_rows"])
        flat = c[f"{field}_r{i}"]
        out.append(flat.reshape(rows, -1).tolist() if rows else [])
    return out


def _check_output(res, c, nreq, atol, what):
    """res: dict (oracle) or LogitsProcessorOutput (product)"""
    get = (lambda k: res.get(k)) if isinstance(res, dict) else (lambda k: getattr(res, k))
    nxt = get("next_token_logits")
    assert torch.allclose(nxt.float().cpu(), torch.from_numpy(c["next_token_logits"]), atol=atol), what
    if not bool(c["has_input"]):
        return
    got = get("input_token_logprobs").float().cpu()
    assert torch.allclose(got, torch.from_numpy(c["input_token_logprobs"]), atol=atol), what
    for field in ("input_top_logprobs", "input_token_ids_logprobs"):
        if not bool(c[f"has_{field}_val"]):
            assert get(field + "_val") is None
            continue
        want_v, want_i = _ragged(c, field + "_val", nreq), _ragged(c, field + "_idx", nreq)
        got_v, got_i = get(field + "_val"), get(field + "_idx")
        assert len(got_v) == len(got_i) == nreq
        for r in range(nreq):
            if want_v[r] is None:                     # the reference's output is undefined there (ids is None)
                assert got_v[r] == [] and got_i[r] == []
                continue
            assert len(got_v[r]) == len(want_v[r]), f"{what}: request {r} rows"
            for a, b, ia, ib in zip(got_v[r], want_v[r], got_i[r], want_i[r]):
                assert list(ia) == [int(x) for x in ib], f"{what}: request {r} ids"
                assert np.allclose(np.array(a, np.float64), np.array(b, np.float64), atol=atol), what


def test_oracle_schedule_side_matches_the_reference():
    for ci, g, c in _cases():
        ids, starts = [], []
        for i, (n, pre, chunk_end, start, k) in enumerate(c["spec"].tolist()):
            prompt = c[f"r{i}_prompt"].tolist()
            fill = n if chunk_end < 0 else chunk_end
            lsl = n - 1 if start < 0 else start
            e = olp.extend_logprob_start_len(lsl, pre, fill - pre, n)
            starts.append(e)
            ids += olp.input_logprob_token_ids(prompt, pre, fill, lsl, fill - pre, e)
        assert starts == c["extend_logprob_start_lens"].tolist(), f"case {ci}"
        assert ids == c["extend_input_logprob_token_ids"].tolist(), f"case {ci}"


def test_oracle_logits_side_matches_the_reference():
    for ci, g, c in _cases():
        nreq = len(c["spec"])
        tops = c["spec"][:, 4].tolist()
        ids = [c[f"r{i}_ids"].tolist() if bool(c["has_ids"][i]) else None for i in range(nreq)]
        res = olp.input_logprobs(torch.from_numpy(c["hidden"]), torch.from_numpy(g["head"]), int(g["vocab"]),
                                 c["extend_lens"].tolist(), c["extend_logprob_start_lens"].tolist(),
                                 torch.from_numpy(c["extend_input_logprob_token_ids"]), tops, ids)
        if not bool(c["has_input"]):
            # nobody asked: the reference projects the last token of every request only (182-203)
            last = np.cumsum(c["extend_lens"]) - 1
            want = torch.from_numpy(c["hidden"])[last] @ torch.from_numpy(g["head"]).T
            assert torch.allclose(want[:, :int(g["vocab"])], torch.from_numpy(c["next_token_logits"]), atol=1e-5)
            assert torch.allclose(res["next_token_logits"], torch.from_numpy(c["next_token_logits"]), atol=1e-5)
            continue
        _check_output(res, c, nreq, 2e-5, f"oracle case {ci}")


def test_product_index_plan_equals_oracle_plan():
    from scratchpad_amd.llama import LogitsProcessor
    for ci, g, c in _cases():
        ext, st = c["extend_lens"].tolist(), c["extend_logprob_start_lens"].tolist()
        spans, sample, inputs, pruned = LogitsProcessor.input_logprob_plan(ext, st)
        rows, osample, oinputs, opruned = olp.plan(ext, st)
        assert [r for a, b in spans for r in range(a, b)] == rows
        assert (sample, inputs, pruned) == (osample, oinputs, opruned)
    # chunked prefill whose chunk ends before the logprob start: one sampled row, no input rows (217-222)
    spans, sample, inputs, pruned = LogitsProcessor.input_logprob_plan([4, 3], [4, 1])
    assert spans == [(3, 4), (5, 7)] and sample == [0, 2] and inputs == [1, 2] and pruned == [0, 2]


def _schedule(c, r2t, alloc, device):
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    reqs = []
    for i, (n, pre, chunk_end, start, k) in enumerate(c["spec"].tolist()):
        prompt = c[f"r{i}_prompt"].tolist()
        r = Req(str(i), "", prompt, None, return_logprob=True, top_logprobs_num=k, token_ids_logprob=c[f"r{i}_ids"].tolist() if bool(c["has_ids"][i]) else None)
        r.logprob_start_len = n - 1 if start < 0 else start
        if chunk_end >= 0:
            r.fill_ids = prompt[:chunk_end]
        if pre:
            r.prefix_indices = alloc.alloc(pre)
        reqs.append(r)
    sb = ScheduleBatch(reqs, r2t, alloc, device=device)
    sb.prepare_for_extend()
    return sb


def test_schedule_batch_bookkeeping_matches_the_reference(monkeypatch):
    from scratchpad_amd import _native
    from scratchpad_amd.pool import ReqToTokenPool, TokenToKVPoolAllocator
    monkeypatch.setattr(_native, "write_req_to_token", ops.write_req_to_token)     # host logic only: no GPU here
    for ci, g, c in _cases():
        sb = _schedule(c, ReqToTokenPool(8, 64, "cpu"), TokenToKVPoolAllocator(256, torch.float32, "cpu", None), "cpu")
        assert sb.return_logprob
        assert sb.extend_logprob_start_lens == c["extend_logprob_start_lens"].tolist(), f"case {ci}"
        assert sb.extend_input_logprob_token_ids.tolist() == c["extend_input_logprob_token_ids"].tolist(), f"case {ci}"
        assert sb.extend_lens == c["extend_lens"].tolist() and sb.prefix_lens == c["prefix_lens"].tolist()
        assert sb.top_logprobs_nums == c["spec"][:, 4].tolist()
        mwb = sb.get_model_worker_batch()
        assert mwb.return_logprob and mwb.extend_logprob_start_lens == sb.extend_logprob_start_lens
        assert mwb.extend_input_logprob_token_ids is sb.extend_input_logprob_token_ids
        # filter_batch carries the per-request lists (1346-1352)
        if len(sb.reqs) > 2:
            sb.output_ids = None
            sb.seq_lens = sb.seq_lens.clone()
            sb.filter_batch(keep_indices=[0, 2])
            assert sb.top_logprobs_nums == [c["spec"][0, 4], c["spec"][2, 4]]


def test_prefix_match_stops_at_the_logprob_start():
    """adjust_max_prefix_ids, schedule_batch.py:494-510"""
    from scratchpad_amd.schedule_batch import Req

    class Cache:
        def match_prefix(self, rid, key):
            self.key = key
            return torch.arange(len(key)), None
    cache = Cache()
    r = Req("a", "", list(range(10)), None, return_logprob=True)
    r.logprob_start_len = 3
    r.init_next_round_input(cache)
    assert cache.key == [0, 1, 2] and r.prefix_len == 3
    r = Req("b", "", list(range(10)), None)
    r.init_next_round_input(cache)
    assert len(cache.key) == 9


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_logits_processor_matches_the_reference_outputs():
    """fp32 head + hidden states of the fixture through the product's LogitsProcessor"""
    from types import SimpleNamespace
    from scratchpad_amd.forward_info import ForwardBatch, ForwardMode
    from scratchpad_amd import distributed
    from scratchpad_amd.llama import LogitsProcessor
    distributed.initialize_model_parallel(1)
    dev = torch.device("cuda:0")
    for ci, g, c in _cases():
        nreq = len(c["spec"])
        head = SimpleNamespace(weight=torch.from_numpy(g["head"]).to(dev))
        proc = LogitsProcessor(SimpleNamespace(vocab_size=int(g["vocab"])))
        ext = c["extend_lens"].tolist()
        ids = [c[f"r{i}_ids"].tolist() if bool(c["has_ids"][i]) else None for i in range(nreq)]
        fb = ForwardBatch(forward_mode=ForwardMode.EXTEND, batch_size=nreq, input_ids=None, req_pool_indices=None,
                          seq_lens=None, out_cache_loc=None, seq_lens_sum=0)
        fb.extend_seq_lens = torch.tensor(ext, dtype=torch.int32, device=dev)
        fb.extend_seq_lens_cpu = ext
        fb.return_logprob = True
        fb.top_logprobs_nums = c["spec"][:, 4].tolist()
        fb.token_ids_logprobs = ids
        fb.extend_logprob_start_lens_cpu = c["extend_logprob_start_lens"].tolist()
        fb.extend_input_logprob_token_ids_gpu = torch.from_numpy(c["extend_input_logprob_token_ids"]).to(dev)
        res = proc(None, torch.from_numpy(c["hidden"]).to(dev), head, fb)
        assert (res.input_token_logprobs is not None) == bool(c["has_input"])
        _check_output(res, c, nreq, 5e-5, f"product case {ci}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "f16"])
def test_model_step_with_input_logprobs_against_oracle(dtype):
    """ScheduleBatch -> ModelRunner -> HIP kernels -> LogitsProcessor with prompt logprobs asked for (one request
    with a cached prefix, one from the middle of its prompt, one not at all), against the oracle model + the
    oracle's restatement of the logits processor."""
    from oracle import llama as ollama
    from scratchpad_amd.model_runner import TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    from tests import smoke_impl
    g, pfx, shape, w = smoke_impl.load_case("b")
    mr = smoke_impl.make_runner(shape, w, dtype)
    worker = TpModelWorker(mr)
    gen = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, shape.vocab, (n,), generator=gen).tolist() for n in (9, 6, 5)]
    reqs = [Req("0", "", prompts[0], None, return_logprob=True, top_logprobs_num=3),
            Req("1", "", prompts[1], None, return_logprob=True, token_ids_logprob=[5, 17]),
            Req("2", "", prompts[2], None)]
    reqs[1].logprob_start_len = 2
    reqs[2].logprob_start_len = len(prompts[2]) - 1
    sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=mr.device)
    sb.prepare_for_extend()
    assert sb.extend_logprob_start_lens == [0, 2, 4]
    out, next_ids = worker.forward_batch_generation(sb.get_model_worker_batch())
    wd = {k: v.to(dtype) for k, v in w.items()}
    okv = ollama.OracleKV(shape, 96, 4, 64, dtype=dtype)
    okv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu()[:4, :64])
    ext = torch.tensor(sb.extend_lens, dtype=torch.int32)
    pos, start = ops.compute_position(torch.zeros(3, dtype=torch.int32), ext)
    hidden = ollama.forward(shape, wd, okv, mode="extend", input_ids=sb.input_ids.cpu(), positions=pos,
                            req_pool_indices=sb.req_pool_indices.cpu(), seq_lens=sb.seq_lens.cpu(),
                            out_cache_loc=sb.out_cache_loc.cpu(), extend_seq_lens=ext, extend_start_loc=start,
                            all_hidden=True)
    head = wd["model.embed_tokens.weight"] if shape.tie else wd["lm_head.weight"]
    ref = olp.input_logprobs(hidden, head, shape.vocab, sb.extend_lens, sb.extend_logprob_start_lens,
                             sb.extend_input_logprob_token_ids, sb.top_logprobs_nums, sb.token_ids_logprobs)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    scale = float(ref["next_token_logits"].abs().max())
    assert float((out.next_token_logits.float().cpu() - ref["next_token_logits"]).abs().max()) <= tol * scale
    # 9 + 4 + 1 positions report a logprob (each request's last one is the zero-padded "next token unknown" entry)
    assert out.input_token_logprobs.shape == (9 + 4 + 1,) == ref["input_token_logprobs"].shape
    assert float((out.input_token_logprobs.cpu() - ref["input_token_logprobs"]).abs().max()) <= tol * scale
    assert [len(v) for v in out.input_top_logprobs_val] == [9, 4, 1]
    assert all(len(row) == 3 for row in out.input_top_logprobs_val[0]) and out.input_top_logprobs_val[1] == [[]] * 4
    if dtype == torch.float32:
        assert out.input_top_logprobs_idx[0] == ref["input_top_logprobs_idx"][0]
    assert np.allclose(out.input_top_logprobs_val[0], ref["input_top_logprobs_val"][0], atol=tol * scale)
    assert out.input_token_ids_logprobs_idx[1] == [[5, 17]] * 4 and out.input_token_ids_logprobs_val[0] == []
    assert np.allclose(out.input_token_ids_logprobs_val[1], ref["input_token_ids_logprobs_val"][1], atol=tol * scale)
    # the sampled token is the argmax of the same logits a batch without logprob requests would see
    for r in sb.reqs:
        mr.req_to_token_pool.free(r.req_pool_idx)
    mr.token_to_kv_pool_allocator.free(sb.out_cache_loc)
    plain = ScheduleBatch([Req(str(10 + i), "", p, None) for i, p in enumerate(prompts)], mr.req_to_token_pool,
                          mr.token_to_kv_pool_allocator, device=mr.device)
    plain.prepare_for_extend()
    out2, ids2 = worker.forward_batch_generation(plain.get_model_worker_batch())
    assert out2.input_token_logprobs is None
    assert torch.equal(ids2, next_ids)
    assert torch.allclose(out2.next_token_logits.float(), out.next_token_logits.float(), atol=tol * scale)
