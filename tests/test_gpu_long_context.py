"""Graph-mode split geometry is independent of the model's context length.

The reference sizes its static attention scratch by the context length (attn_logits [max_bs, heads,
max_context_len], nn/attention/triton_backend.py:70-80; default context_length 4096, server/args.py:23).  Here
the captured launches cover a slot budget that depends on the batch bucket only and the split size travels in
the per-step plan, so a 131072-token context costs the same scratch as a 4096-token one - and an 8 x 100k-token
decode step replayed from such a graph still equals the oracle."""
import pytest
import torch

from oracle import llama as ollama
from oracle import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("geometry", ["ranges", "items"])
def test_graph_scratch_is_bounded_and_a_100k_token_decode_step_matches_the_oracle(geometry):
    """geometry "ranges": the backend as it is built for a Llama shape (round 6) - plans are the range section alone, the
    scratch is batch size + pieces slots.  "items": the backend planning the (request, split) items too, as it does for a
    model with a soft-cap layer, and the launches sent to them (sp_debug_set("decode_ranges", 0)): the geometry whose
    split size grows with the step so that its items fit the captured launch."""
    from scratchpad_amd import _native
    if geometry == "items":
        _native.debug_set("decode_ranges", 0)
    try:
        _long_context_step(geometry)
    finally:
        _native.debug_set("decode_ranges", -1)


def _long_context_step(geometry):
    from scratchpad_amd import _native
    from scratchpad_amd.forward_info import ForwardMode, ModelWorkerBatch
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    CTX, BS, LEN = 131072, 8, 100_000
    lens = [LEN - 977 * i for i in range(BS)]                       # ragged: 100000 ... 93161
    shape = ollama.LlamaShape(256, 512, 2, 2, 1, 512, False, 500000.0, None, CTX, 1e-5)
    cfg = ModelConfig(shape.hidden, shape.inter, shape.layers, shape.Hq, shape.Hkv, shape.vocab, context_len=CTX,
                      max_position_embeddings=CTX)
    pool = sum(lens) + 4 * BS + 64
    args = ServerArgs(max_total_tokens=pool, max_running_requests=256)       # default bucket list, up to bs 256
    mr = ModelRunner(cfg, args, dtype=torch.float16, seed=3)
    backend = mr.attn_backend
    assert backend.plan_items is False and backend.decode_ranges > 0
    backend.plan_items = geometry == "items"
    mr.init_cuda_graphs()
    scratch = backend.graph_scratch_bytes()
    # what the same slot budget costs with Llama-3-8B's 32 query heads of 128 (the figure DESIGN.md quotes)
    slots = backend._graph_slots(256)
    assert (slots == 0) == (geometry == "ranges")
    r8b = _native.decode_ranges(32, 8, 128, torch.bfloat16) if geometry == "ranges" else 0
    scratch_8b = (_native.decode_workspace_bytes(256, 32, 128, CTX, 64, slots, r8b)
                  + 3 * _native.decode_plan_bytes(256, CTX, 64, slots, r8b))
    static_8b = _native.decode_workspace_bytes(256, 32, 128, CTX, 512)       # round 2: bs x ceil(ctx / 512) splits
    print(f"graph attention scratch at context_len {CTX} ({geometry}): {scratch / 2**20:.1f} MiB for this model, "
          f"{scratch_8b / 2**20:.1f} MiB at Llama-3-8B head counts ({slots or 256 + r8b} slots); the bs-256 bucket alone under "
          f"the round-2 static geometry: {static_8b / 2**20:.0f} MiB")
    assert scratch < 1.5e9 and scratch_8b < 1.5e9 and scratch_8b * 20 < static_8b
    assert len(mr.graph_runner.capture_bs) >= 30 and max(mr.graph_runner.capture_bs) == 256

    # ---- a real step: 8 requests of ~100k cached tokens each, slots a random permutation of the pool
    g = torch.Generator().manual_seed(11)
    dev = mr.device
    kvp = mr.token_to_kv_pool
    for arena in (kvp._k_arena, kvp._v_arena):
        arena.normal_(0.0, 1.0)
    perm = (torch.randperm(pool, generator=g) + 1).to(torch.int32)
    table = mr.req_to_token_pool.req_to_token
    rows = torch.tensor([5, 0, 3, 250, 7, 1, 100, 2])
    off, new_loc = 0, []
    for b in range(BS):
        n = lens[b] + 1                                              # + the token decoded in this step
        table[rows[b], :n] = perm[off:off + n].to(dev)
        new_loc.append(int(perm[off + n - 1]))
        off += n
    seq = torch.tensor([l + 1 for l in lens])
    ids = torch.randint(0, shape.vocab, (BS,), generator=g)
    batch = ModelWorkerBatch(bid=1, forward_mode=ForwardMode.DECODE, input_ids=ids.to(dev),
                             req_pool_indices=rows.to(dev), seq_lens=seq.to(dev),
                             out_cache_loc=torch.tensor(new_loc).to(dev), seq_lens_sum=int(seq.sum()))
    worker = TpModelWorker(mr)
    out, _ = worker.forward_batch_generation(batch)                  # HIP-graph replay (bucket 8)
    got = out.next_token_logits.float().cpu()
    plan = backend.forward_metadata[3][0][0].cpu()
    chunk_used, items = int(plan[1]), int(plan[0])
    if geometry == "items":
        assert _native.debug_get("decode_last_kernel") == 2          # (the launches the graphs captured)
        assert items == sum(-(-int(s) // chunk_used) for s in seq) <= backend._graph_slots(8)
        assert chunk_used >= 512, "the split size grew with the step's sum(seq_lens) instead of the slot count"
    else:       # [0, chunk, 0, 0 | pieces in use, R, ranges, bs | ...]: every piece of the line in use, ~800 k positions cut evenly
        assert _native.debug_get("decode_last_kernel") == 3 and items == 0
        pieces, R, ranges, bs_built = plan[4:8].tolist()
        line = int(seq.sum()) + 16 * BS
        assert (ranges, bs_built) == (backend._graph_ranges, BS)
        assert R == -(-line // ranges) and pieces == -(-line // R)
    mr.graph_runner = None
    out_e, _ = worker.forward_batch_generation(batch)                # the same step, eager launches
    eager = out_e.next_token_logits.float().cpu()

    # ---- the oracle on the same weights / pool (fp32 arithmetic on the fp16 values)
    w = {k: v.detach().float().cpu() for k, v in mr.model.state_dict().items()}
    okv = ollama.OracleKV(shape, pool, 256 + 1, CTX + 4)
    for l in range(shape.layers):
        kb, vb = kvp.get_kv_buffer(l)
        okv.k[l] = kb.float().cpu()
        okv.v[l] = vb.float().cpu()
    okv.req_to_token.copy_(table.cpu())
    ref = ollama.forward(shape, w, okv, mode="decode", input_ids=ids, positions=ops.clamp_position(seq),
                         req_pool_indices=rows, seq_lens=seq, out_cache_loc=torch.tensor(new_loc))
    rel = lambda a: float((a - ref).abs().max() / ref.abs().max())
    print(f"8 x ~100k-token decode step, fp16: graph replay (split size {chunk_used}, {items} items) vs fp32 oracle "
          f"{rel(got):.2e}, eager {rel(eager):.2e}")
    assert rel(got) < 5e-3 and rel(eager) < 5e-3
