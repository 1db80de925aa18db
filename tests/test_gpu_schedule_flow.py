"""The scheduler-side producers driving the hot path on the GPU: prepare_for_extend (with a cached
prefix), prepare_for_decode over several steps, and a MIXED batch (chunked prefill + running
decodes), checked step by step against the CPU oracle fed with the SAME slots/tables.

This is the continuous-batching trace in miniature: slot allocation, req_to_token writes and the
KV pool contents must agree bit-exactly in their indices; logits within the fp32 bar."""
from types import SimpleNamespace

import pytest
import torch

from oracle import llama as ollama
from oracle import ops
from tests import smoke_impl

pytestmark = pytest.mark.gpu


def oracle_step(shape, w, kv, mode, ids, req, seq, loc, pre=None, ext=None):
    if mode == "decode":
        return ollama.forward(shape, w, kv, mode="decode", input_ids=ids, positions=ops.clamp_position(seq),
                              req_pool_indices=req, seq_lens=seq, out_cache_loc=loc)
    positions, start = ops.compute_position(pre, ext)
    return ollama.forward(shape, w, kv, mode="extend", input_ids=ids, positions=positions, req_pool_indices=req,
                          seq_lens=seq, out_cache_loc=loc, extend_seq_lens=ext, extend_start_loc=start)


def close(got, want, what):
    dev = float((got.float().cpu() - want).abs().max() / want.abs().max())
    assert dev <= 1e-4, f"{what}: relative logit deviation {dev:.2e}"


def test_extend_decode_mixed_trace_matches_oracle():
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.model_runner import TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    g, pfx, shape, w = smoke_impl.load_case("a")
    mr = smoke_impl.make_runner(shape, w, torch.float32)
    worker = TpModelWorker(mr)
    dev = mr.device
    # the accessors the reference's Scheduler calls at start-up (scheduler.py:203-217, 333-336)
    info = worker.get_worker_info()
    assert len(info) == 11 and info[0] == mr.max_total_num_tokens == 96 and info[2] == mr.max_running_requests
    assert info[6] == dev and info[8] == mr.req_to_token_pool.size and info[10] == mr.token_to_kv_pool.size
    assert worker.get_memory_pool() == (mr.req_to_token_pool, mr.token_to_kv_pool_allocator)
    assert worker.get_pad_input_ids_func() is None and worker.get_tp_cpu_group() is None      # plain Llama, TP = 1
    gen = torch.Generator().manual_seed(7)
    okv = ollama.OracleKV(shape, 96, 4, 64)

    def mirror_tables():
        okv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu())

    # ---- 1. prefill of two requests; the second reuses the first 4 tokens' KV of a shared prompt
    shared = torch.randint(0, shape.vocab, (4,), generator=gen).tolist()
    warm = ScheduleBatch([Req("warm", "", shared, None)], mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev)
    warm.prepare_for_extend()
    out, _ = worker.forward_batch_generation(warm.get_model_worker_batch())
    mirror_tables()
    ref = oracle_step(shape, w, okv, "extend", warm.input_ids.cpu(), warm.req_pool_indices.cpu(),
                      warm.seq_lens.cpu(), warm.out_cache_loc.cpu(), torch.tensor([0], dtype=torch.int32),
                      torch.tensor([4], dtype=torch.int32))
    close(out.next_token_logits, ref, "warm prefill")
    prefix_slots = warm.out_cache_loc.clone()          # what RadixCache.match_prefix would return
    mr.req_to_token_pool.free(warm.reqs[0].req_pool_idx)

    r0 = Req("r0", "", torch.randint(0, shape.vocab, (7,), generator=gen).tolist(), None)
    r1 = Req("r1", "", shared + torch.randint(0, shape.vocab, (5,), generator=gen).tolist(), None, prefix_indices=prefix_slots)
    sb = ScheduleBatch([r0, r1], mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev)
    sb.prepare_for_extend()
    assert sb.prefix_lens == [0, 4] and sb.extend_lens == [7, 5] and sb.extend_num_tokens == 12
    table = mr.req_to_token_pool.req_to_token.cpu()
    assert torch.equal(table[r1.req_pool_idx, :4].long(), prefix_slots.cpu()), "cached prefix slots copied"
    assert torch.equal(table[r0.req_pool_idx, :7].long(), sb.out_cache_loc[:7].cpu())
    assert torch.equal(table[r1.req_pool_idx, 4:9].long(), sb.out_cache_loc[7:].cpu())
    out, nxt = worker.forward_batch_generation(sb.get_model_worker_batch())
    mirror_tables()
    ref = oracle_step(shape, w, okv, "extend", sb.input_ids.cpu(), sb.req_pool_indices.cpu(), sb.seq_lens.cpu(),
                      sb.out_cache_loc.cpu(), torch.tensor(sb.prefix_lens, dtype=torch.int32),
                      torch.tensor(sb.extend_lens, dtype=torch.int32))
    close(out.next_token_logits, ref, "prefill with cached prefix")
    assert torch.equal(nxt.cpu(), ref.argmax(-1))

    # ---- 2. three decode steps (seq_lens += 1, alloc(bs), table write) feeding sampled tokens back
    sb.output_ids = nxt
    for step in range(3):
        before = sb.seq_lens.clone()
        sb.prepare_for_decode()
        assert sb.forward_mode == ForwardMode.DECODE and torch.equal(sb.seq_lens, before + 1)
        assert sb.seq_lens_sum == int(sb.seq_lens.sum()), "the host-side sum follows the step"
        out, nxt = worker.forward_batch_generation(sb.get_model_worker_batch())
        mirror_tables()
        ref = oracle_step(shape, w, okv, "decode", sb.input_ids.cpu(), sb.req_pool_indices.cpu(),
                          sb.seq_lens.cpu(), sb.out_cache_loc.cpu())
        close(out.next_token_logits, ref, f"decode step {step}")
        for r, tok in zip(sb.reqs, sb.input_ids.tolist()):
            r.output_ids.append(tok)
        sb.output_ids = nxt

    # ---- 3. MIXED: a new prompt is prefilled in the same batch as the two running decodes
    sb.prepare_for_decode()
    newr = Req("r2", "", torch.randint(0, shape.vocab, (6,), generator=gen).tolist(), None)
    mix = ScheduleBatch([newr], mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev)
    mix.prepare_for_extend()
    for r, tok in zip(sb.reqs, sb.input_ids.tolist()):
        r.output_ids.append(tok)
    mix.mix_with_running(sb)
    assert mix.forward_mode == ForwardMode.MIXED and mix.extend_lens == [6, 1, 1]
    assert mix.prefix_lens[1:] == [int(x) - 1 for x in sb.seq_lens.tolist()]
    out, _ = worker.forward_batch_generation(mix.get_model_worker_batch())
    mirror_tables()
    ref = oracle_step(shape, w, okv, "extend", mix.input_ids.cpu(), mix.req_pool_indices.cpu(), mix.seq_lens.cpu(),
                      mix.out_cache_loc.cpu(), torch.tensor(mix.prefix_lens, dtype=torch.int32),
                      torch.tensor(mix.extend_lens, dtype=torch.int32))
    close(out.next_token_logits, ref, "mixed batch")
    # ---- 4. the same kind of batch with prompt logprobs asked for: one id per kept position - the running
    # rows contribute one each - so the processor's index lists line up; sampled rows and input logprobs = the oracle's
    for r, tok in zip(mix.reqs, out.next_token_logits.argmax(-1).tolist()):
        r.output_ids.append(tok)
    mix.output_ids = out.next_token_logits.argmax(-1)
    mix.prepare_for_decode()
    lp = Req("r3", "", torch.randint(0, shape.vocab, (5,), generator=gen).tolist(), None, return_logprob=True, logprob_start_len=1, top_logprobs_num=2)
    mix2 = ScheduleBatch([lp], mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev)
    mix2.prepare_for_extend()
    assert mix2.extend_input_logprob_token_ids.tolist() == lp.origin_input_ids[2:] + [0]
    mix2.mix_with_running(mix)
    assert mix2.extend_lens == [5, 1, 1, 1] and mix2.extend_logprob_start_lens == [1, 0, 0, 0]
    assert mix2.extend_input_logprob_token_ids.tolist() == lp.origin_input_ids[2:] + [0] + [0, 0, 0]
    out2, _ = worker.forward_batch_generation(mix2.get_model_worker_batch())
    mirror_tables()
    ref2 = oracle_step(shape, w, okv, "extend", mix2.input_ids.cpu(), mix2.req_pool_indices.cpu(), mix2.seq_lens.cpu(),
                       mix2.out_cache_loc.cpu(), torch.tensor(mix2.prefix_lens, dtype=torch.int32),
                       torch.tensor(mix2.extend_lens, dtype=torch.int32))
    close(out2.next_token_logits, ref2, "mixed batch with prompt logprobs: sampled rows")
    assert out2.input_token_logprobs.shape[0] == 4 + 3 and torch.isfinite(out2.input_token_logprobs).all()
    assert [len(v) for v in out2.input_top_logprobs_val] == [4, 1, 1, 1]
    # ---- 5. the other way round: the PREFILL side asks for no logprobs, a RUNNING request does.  The
    # reference's scheduler refuses to build this batch (scheduler.py:944-949); here mix_with_running pads the id list
    # with one zero per kept position of the prefill rows, and the step must give the logits of the oracle for every
    # sampled row with index lists that line up (one input logprob per kept position, top lists per request)
    for r, tok in zip(mix2.reqs, out2.next_token_logits.argmax(-1).tolist()):
        r.output_ids.append(tok)
    mix2.output_ids = out2.next_token_logits.argmax(-1)
    # (the engine has 4 request rows: r0 finishes here and gives its row and slots back, as the scheduler would -
    # filter_batch + cache_finished_req of a ChunkCache, scheduler.py:803-812, chunk_cache.py:37-50)
    from scratchpad_amd.schedule_batch import FINISH_LENGTH
    done = mix2.reqs[2]
    assert done.rid == "r0"
    done.finished_reason = FINISH_LENGTH(length=len(done.output_ids))
    done_len = int(mix2.seq_lens[2])
    mix2.filter_batch()
    assert [r.rid for r in mix2.reqs] == ["r3", "r2", "r1"] and mix2.top_logprobs_nums == [2, 0, 0]
    row = mr.req_to_token_pool.req_to_token[done.req_pool_idx, :done_len]
    mr.token_to_kv_pool_allocator.free(row.to(torch.int64))
    mr.req_to_token_pool.free(done.req_pool_idx)
    mix2.prepare_for_decode()
    assert mix2.return_logprob and mix2.top_logprobs_nums == [2, 0, 0]
    plain = Req("r4", "", torch.randint(0, shape.vocab, (6,), generator=gen).tolist(), None)
    mix3 = ScheduleBatch([plain], mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev)
    mix3.prepare_for_extend()
    assert not mix3.return_logprob and mix3.extend_input_logprob_token_ids is None
    mix3.mix_with_running(mix2)
    assert mix3.return_logprob and mix3.extend_lens == [6, 1, 1, 1] and mix3.extend_logprob_start_lens == [0] * 4
    assert mix3.extend_input_logprob_token_ids.tolist() == [0] * (6 + 3)
    assert mix3.top_logprobs_nums == [0, 2, 0, 0]
    out3, _ = worker.forward_batch_generation(mix3.get_model_worker_batch())
    mirror_tables()
    ref3 = oracle_step(shape, w, okv, "extend", mix3.input_ids.cpu(), mix3.req_pool_indices.cpu(), mix3.seq_lens.cpu(),
                       mix3.out_cache_loc.cpu(), torch.tensor(mix3.prefix_lens, dtype=torch.int32),
                       torch.tensor(mix3.extend_lens, dtype=torch.int32))
    close(out3.next_token_logits, ref3, "mixed batch, logprobs asked by a running request only: sampled rows")
    assert out3.input_token_logprobs.shape[0] == 6 + 3 and torch.isfinite(out3.input_token_logprobs).all()
    assert [len(v) for v in out3.input_top_logprobs_val] == [6, 1, 1, 1]
    assert [len(row) for row in out3.input_top_logprobs_val[1]] == [2] and all(len(row) == 0 for row in out3.input_top_logprobs_val[0])
    # the running request's row: its top-2 input logprobs are the two largest entries of log_softmax of ITS logits row
    want = torch.log_softmax(ref3[1].double(), -1).topk(2).values.float()
    got = torch.tensor(out3.input_top_logprobs_val[1][0])
    assert torch.allclose(got, want, atol=2e-4), (got, want)
    # the two pools hold the same K rows in the same slots
    for layer in range(shape.layers):
        assert torch.allclose(mr.token_to_kv_pool.get_key_buffer(layer).cpu(), okv.k[layer], atol=2e-5)


def test_out_of_memory_is_reported_like_the_reference():
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    g, pfx, shape, w = smoke_impl.load_case("a")
    mr = smoke_impl.make_runner(shape, w, torch.float32)
    big = ScheduleBatch([Req("x", "", list(range(50)), None), Req("y", "", list(range(50)), None)], mr.req_to_token_pool,
                        mr.token_to_kv_pool_allocator, device=mr.device)
    with pytest.raises(RuntimeError, match="Out of memory"):
        big.prepare_for_extend()          # 100 tokens > the 96-slot pool
    many = ScheduleBatch([Req(str(i), "", [1], None) for i in range(9)], mr.req_to_token_pool,
                         mr.token_to_kv_pool_allocator, device=mr.device)
    with pytest.raises(RuntimeError, match="max-running-requests"):
        many.prepare_for_extend()


def test_overlap_worker_matches_synchronous_worker():
    """Forward thread + future-token placeholders (tp_worker_client.py) produce the same greedy
    tokens as the synchronous worker, while results arrive one step late."""
    from scratchpad_amd.model_runner import TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    from scratchpad_amd.tp_worker_client import TpModelWorkerClient, resolve_future_token_ids
    ids = torch.tensor([5, -2, 7, -1], device="cuda")
    fmap = torch.tensor([0, 11, 22, 33], device="cuda")
    resolve_future_token_ids(ids, fmap)
    assert ids.tolist() == [5, 22, 7, 11]

    g, pfx, shape, w = smoke_impl.load_case("a")
    gen = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, shape.vocab, (n,), generator=gen).tolist() for n in (6, 9, 4)]

    def run(overlap):
        mr = smoke_impl.make_runner(shape, w, torch.float32)
        worker = TpModelWorkerClient(mr) if overlap else TpModelWorker(mr)
        sb = ScheduleBatch([Req(str(i), "", list(p), None) for i, p in enumerate(prompts)], mr.req_to_token_pool,
                           mr.token_to_kv_pool_allocator, device=mr.device)
        sb.prepare_for_extend()
        tokens = []
        _, nxt = worker.forward_batch_generation(sb.get_model_worker_batch())
        for step in range(5):
            if overlap:
                assert bool((nxt < 0).all()), "placeholders until the forward thread resolves them"
            sb.output_ids = nxt
            sb.prepare_for_decode()
            _, nxt_new = worker.forward_batch_generation(sb.get_model_worker_batch())
            if overlap:      # the result of the PREVIOUS launch becomes available now
                _, real = worker.resolve_last_batch_result()
                tokens.append(real)
            else:
                tokens.append(nxt.tolist())
            nxt = nxt_new
        if overlap:
            _, real = worker.resolve_last_batch_result()
            tokens.append(real)
            worker.close()
        else:
            tokens.append(nxt.tolist())
        return tokens

    assert run(True) == run(False)


def _fresh_prefill_logits(shape, w, prompt):
    """Oracle logits of an uncached, unchunked prefill of `prompt` (its own empty pool)."""
    okv = ollama.OracleKV(shape, 96, 4, 64)
    n = len(prompt)
    loc = torch.arange(1, n + 1)
    okv.req_to_token[0, :n] = loc.to(torch.int32)
    return oracle_step(shape, w, okv, "extend", torch.tensor(prompt), torch.tensor([0]), torch.tensor([n]), loc,
                       torch.tensor([0], dtype=torch.int32), torch.tensor([n], dtype=torch.int32))


def test_radix_cache_prefix_reuse_chunked_prefill_and_retraction():
    """RadixCache as the producer of prefix_indices: a finished request's KV is reused by a later
    prompt, a chunked prefill continues on the tree's slots, and retraction hands slots back.
    Logits must equal an uncached prefill of the same prompt (the cache may only change WHERE the KV
    lives, never its contents)."""
    from scratchpad_amd.model_runner import TpModelWorker
    from scratchpad_amd.radix_cache import RadixCache
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    g, pfx, shape, w = smoke_impl.load_case("a")
    mr = smoke_impl.make_runner(shape, w, torch.float32)
    worker = TpModelWorker(mr)
    alloc, r2t, dev = mr.token_to_kv_pool_allocator, mr.req_to_token_pool, mr.device
    tree = RadixCache(r2t, alloc)
    gen = torch.Generator().manual_seed(11)
    rand = lambda n: torch.randint(0, shape.vocab, (n,), generator=gen).tolist()

    def batch(reqs):
        return ScheduleBatch(reqs, r2t, alloc, device=dev, tree_cache=tree)

    # A: plain prefill + 2 decode steps, then finish -> the tree owns prompt + first output token
    a = Req("a", "", rand(12), None)
    a.init_next_round_input(tree)
    assert a.prefix_len == 0 and a.last_node is tree.root_node
    tree.inc_lock_ref(a.last_node)
    sa = batch([a])
    sa.prepare_for_extend()
    out, nxt = worker.forward_batch_generation(sa.get_model_worker_batch())
    close(out.next_token_logits, _fresh_prefill_logits(shape, w, a.origin_input_ids), "A prefill")
    for _ in range(2):
        a.output_ids.append(int(nxt[0]))
        sa.output_ids = nxt
        sa.prepare_for_decode()
        out, nxt = worker.forward_batch_generation(sa.get_model_worker_batch())
    a.output_ids.append(int(nxt[0]))
    tree.cache_finished_req(a)
    assert tree.total_size() == 12 + 2 and tree.evictable_size() == 14 and tree.protected_size() == 0
    assert alloc.available_size() + tree.total_size() == 96 and r2t.available_size() == 4

    # B shares A's first 8 prompt tokens; C repeats A's prompt + its first output (full hit - 1)
    b = Req("b", "", a.origin_input_ids[:8] + rand(5), None)
    c = Req("c", "", a.origin_input_ids + a.output_ids[:1], None)
    for r in (b, c):
        r.init_next_round_input(tree)
        tree.inc_lock_ref(r.last_node)
    assert b.prefix_len == 8 and c.prefix_len == 12
    a_slots = tree.match_prefix(a.origin_input_ids)[0]
    assert torch.equal(b.prefix_indices, a_slots[:8]) and torch.equal(c.prefix_indices, a_slots)
    sb = batch([b, c])
    sb.prepare_for_extend()
    assert sb.prefix_lens == [8, 12] and sb.extend_lens == [5, 1]
    out, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
    close(out.next_token_logits[0:1], _fresh_prefill_logits(shape, w, b.origin_input_ids), "B on cached prefix")
    close(out.next_token_logits[1:2], _fresh_prefill_logits(shape, w, c.origin_input_ids), "C on cached prefix")
    assert tree.protected_size() == 12     # the shared path is pinned while B and C run

    # retraction: equal output counts and prompt lengths -> the later request (C) is retracted first
    sb.output_ids = out.next_token_logits.argmax(-1)
    before = alloc.available_size()
    retracted, new_ratio = sb.retract_decode(SimpleNamespace(retract_decode_steps=1, speculative_algorithm=None, page_size=1))
    assert new_ratio == 1.0                            # (no max_new_tokens budget on these requests)
    assert [r.rid for r in retracted] == ["c"] and [r.rid for r in sb.reqs] == ["b"]
    assert c.req_pool_idx is None and len(c.prefix_indices) == 0 and c.is_retracted
    assert alloc.available_size() == before + 1      # C's single own slot; its prefix stays cached
    assert sb.seq_lens.tolist() == [13] and sb.check_decode_mem()

    # D: chunked prefill (7 + 9 tokens) through cache_unfinished_req, sharing B's prompt head
    b.output_ids = [int(sb.output_ids[0])]
    tree.cache_finished_req(b)
    d_ids = b.origin_input_ids[:10] + rand(6)
    d = Req("d", "", d_ids, None)
    d.init_next_round_input(tree)
    assert d.prefix_len == 10
    tree.inc_lock_ref(d.last_node)
    d.fill_ids = d_ids[:12]
    sd = batch([d])
    sd.prepare_for_extend()
    assert sd.prefix_lens == [10] and sd.extend_lens == [2]
    worker.forward_batch_generation(sd.get_model_worker_batch())
    tree.cache_unfinished_req(d)
    assert d.prefix_len == 12 and tree.match_prefix(d_ids[:12])[0].tolist() == d.prefix_indices.tolist()
    d.fill_ids = None
    r2t.free(d.req_pool_idx)       # the scheduler re-allocates the row for the next chunk
    sd = batch([d])
    sd.prepare_for_extend()
    assert sd.prefix_lens == [12] and sd.extend_lens == [4]
    out, _ = worker.forward_batch_generation(sd.get_model_worker_batch())
    close(out.next_token_logits, _fresh_prefill_logits(shape, w, d_ids), "D chunked on cached prefix")

    # eviction under pressure: a prompt that needs more than the free slots evicts unlocked leaves
    d.output_ids = [1]
    tree.cache_finished_req(d)
    cached_before = tree.total_size()
    hog = alloc.alloc(alloc.available_size() - 20)      # leave 20 free slots for a 30-token prompt
    e = Req("e", "", rand(30), None)
    e.init_next_round_input(tree)
    tree.inc_lock_ref(e.last_node)
    se = batch([e])
    se.prepare_for_extend()
    assert tree.total_size() < cached_before, "alloc_token_slots evicted through the tree"
    out, _ = worker.forward_batch_generation(se.get_model_worker_batch())
    close(out.next_token_logits, _fresh_prefill_logits(shape, w, e.origin_input_ids), "E after eviction")
    # conservation: every slot is either free, cached, or held by the running request
    assert alloc.available_size() + tree.total_size() + e.extend_input_len + len(hog) == 96


def test_serve_trace_mode_of_the_bench_runs_and_conserves_slots():
    """bench.py --mode serve at toy size: admission, RadixCache hits on a shared prefix, merge into the
    running batch, graph-replayed decode over changing batch sizes, finish + cache hand-back.  The
    mode asserts slot conservation itself; here the JSON line is checked."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--mode", "serve", "--layers", "2", "--requests", "40",
           "--bs", "12", "--max-input", "300", "--max-output", "24", "--prefix", "64", "--max-prefill-tokens", "1024"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert line["metric"] == "serve_output_tokens_per_sec" and line["value"] > 0
    assert cfg["prefix_cache_hit_tokens"] >= 64 * 20, "later requests reuse the shared prefix"
    assert cfg["extend_steps"] >= 4 and cfg["decode_steps"] >= 23
    assert line["ttft_ms"]["p50"] > 0 and line["tpot_ms"]["p50"] > 0
