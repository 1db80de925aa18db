"""Mllama text model (config 5's cross-attention path) on the GPU against the logits of the
reference's MllamaForCausalLM: cross-attention over the encoder slots, per-head q/k RMSNorm, tanh
gates, the row mask, self-attention behind the encoder slots, encoder K/V reuse in decode - driven
through the scheduler-side encoder bookkeeping (prepare_encoder_info_extend/decode)."""
import numpy as np
import pytest
import torch

from tests import golden

pytestmark = pytest.mark.gpu


def build(dtype, graph_bs=None):
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs
    g = golden.load("tiny_mllama")
    hidden, inter, nl, Hq, Hkv, vocab = (int(x) for x in g["cfg"])
    cfg = ModelConfig(hidden, inter, nl, Hq, Hkv, vocab, context_len=60, rms_norm_eps=1e-5, rope_theta=500000.0,
                      max_position_embeddings=128, cross_attention_layers=[int(x) for x in g["cross_layers"]])
    mr = ModelRunner(cfg, ServerArgs(max_total_tokens=96, max_running_requests=3,
                                     disable_cuda_graph=graph_bs is None, cuda_graph_bs=graph_bs),
                     dtype=dtype, init_weights=False)
    w = {k[3:]: torch.from_numpy(v).to(mr.device) for k, v in g.items() if k.startswith("w::")}
    mr.model.load_full_state_dict(w)
    if graph_bs is not None:
        mr.init_cuda_graphs()
    return g, mr


def test_per_head_rmsnorm_matches_reference():
    from scratchpad_amd.mllama import MllamaTextRMSNorm
    g = golden.load("tiny_mllama")
    n = MllamaTextRMSNorm(64, 1e-5).cuda()
    n.weight.data = torch.from_numpy(g["qnorm_w"]).cuda()
    y = n(torch.from_numpy(g["qnorm_x"]).cuda())
    assert y.shape == (7, 4, 64)
    assert torch.allclose(y.cpu(), torch.from_numpy(g["qnorm_y"]), atol=2e-6, rtol=2e-6)
    k_view = torch.randn(5, 512, device="cuda")[:, 256:384].view(5, 2, 64)      # a strided qkv slice
    assert torch.allclose(n(k_view), n(k_view.contiguous()))


@pytest.mark.parametrize("graph_bs", [None, [4]], ids=["eager", "graph-padded-to-4"])
def test_tiny_mllama_matches_reference_logits(graph_bs):
    """graph variant: the decode step replays a HIP graph captured for bs 4 (one padded row with
    encoder_len 0 and seq_len 1) with encoder_lens as a static graph input (cuda_graph_runner.py:201-208)"""
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.mllama import get_full_text_row_masked_out_mask
    from scratchpad_amd.model_runner import TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    g, mr = build(torch.float32, graph_bs)
    worker = TpModelWorker(mr)
    dev = mr.device
    enc, text = g["encoder_lens"].tolist(), g["text_lens"].tolist()
    ids = g["input_ids"].tolist()
    # the fixture's slot order and request rows: make the allocator and the row pool hand them out
    slots = np.concatenate([np.concatenate([g["encoder_out_cache_loc"][sum(enc[:b]):sum(enc[:b + 1])],
                                            g["out_cache_loc"][sum(text[:b]):sum(text[:b + 1])]]) for b in range(3)]
                           + [g["decode_out_cache_loc"]])
    rest = np.setdiff1d(np.arange(1, 97), slots)
    mr.token_to_kv_pool_allocator.free_slots = torch.from_numpy(np.concatenate([slots, rest])).to(dev)
    mr.req_to_token_pool.free_slots = g["req_pool_indices"].tolist() + [2]
    reqs, off = [], 0
    for b in range(3):
        toks = ids[off:off + text[b]]
        off += text[b]
        if enc[b]:
            reqs.append(Req(str(b), "", [0] * enc[b] + toks, None, num_image_tokens=enc[b]))   # image pad ids first
        else:
            reqs.append(Req(str(b), "", toks, None))
    sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev, is_encoder_decoder=True)
    sb.prepare_for_extend()
    assert sb.extend_lens == text and sb.encoder_lens_cpu == enc and sb.encoder_cached == [False, True, False]
    assert np.array_equal(sb.out_cache_loc.cpu().numpy(), g["out_cache_loc"])
    assert np.array_equal(sb.encoder_out_cache_loc.cpu().numpy(), g["encoder_out_cache_loc"])
    assert np.array_equal(sb.seq_lens.cpu().numpy(), g["text_lens"])
    batch = sb.get_model_worker_batch()
    batch.encoder_states = torch.from_numpy(g["cross_attention_states"]).to(dev)
    out, nxt = worker.forward_batch_generation(batch)
    want = torch.from_numpy(g["prefill_logits"])
    dev1 = float((out.next_token_logits.cpu() - want).abs().max() / want.abs().max())
    assert dev1 <= 1e-4, f"prefill logits deviate {dev1:.2e}"
    assert np.array_equal(nxt.cpu().numpy(), g["next_ids"])
    assert torch.allclose(mr.token_to_kv_pool.get_key_buffer(1).cpu(), torch.from_numpy(g["k_buffer1_after"]), atol=2e-5)
    # decode: encoder K/V come from the pool (cross_attention_states is None from now on)
    sb.output_ids = nxt
    sb.prepare_for_decode()
    assert sb.encoder_cached == [True] * 3
    assert np.array_equal(sb.out_cache_loc.cpu().numpy(), g["decode_out_cache_loc"])
    dec = sb.get_model_worker_batch()
    if graph_bs is not None:
        fb = __import__("scratchpad_amd.forward_info", fromlist=["ForwardBatch"]).ForwardBatch.init_new(dec, mr)
        assert mr.graph_runner is not None and mr.graph_runner.can_run(fb), "the decode step must replay a graph"
    out2, _ = worker.forward_batch_generation(dec)
    want2 = torch.from_numpy(g["decode_logits"])
    dev2 = float((out2.next_token_logits.cpu() - want2).abs().max() / want2.abs().max())
    assert dev2 <= 1e-4, f"decode logits deviate {dev2:.2e}"


def test_row_mask_reproduces_the_reference_quirk():
    from types import SimpleNamespace
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.mllama import flat_encoder_result, get_full_text_row_masked_out_mask
    g = golden.load("tiny_mllama")
    fb = SimpleNamespace(forward_mode=ForwardMode.EXTEND, extend_seq_lens_cpu=g["text_lens"].tolist(),
                         seq_lens=torch.from_numpy(g["text_lens"]).long(), seq_lens_cpu=None,
                         encoder_lens_cpu=g["encoder_lens"].tolist())
    assert np.array_equal(get_full_text_row_masked_out_mask(fb).long().numpy(), g["row_mask_extend"])
    fb2 = SimpleNamespace(forward_mode=ForwardMode.DECODE, encoder_lens=torch.from_numpy(g["encoder_lens"]))
    assert np.array_equal(get_full_text_row_masked_out_mask(fb2).long().numpy(), g["row_mask_decode"])
    states = torch.arange(2 * 5 * 3, dtype=torch.float32).view(2, 5, 3)
    flat = flat_encoder_result(states, [4, 0, 2])
    assert flat.shape == (6, 3) and torch.equal(flat[:4], states[0, :4]) and torch.equal(flat[4:], states[1, :2])


def test_image_request_end_to_end_through_the_vision_tower():
    """An image request through the whole encoder-decoder path on the GPU: pad_input_ids ->
    prepare_for_extend (encoder slots first) -> ModelRunner -> MllamaForConditionalGeneration runs the
    vision tower + projector from forward_batch.mm_inputs -> cross-attention K/V stored at the encoder
    slots -> logits; then a decode step that reads the encoder K/V back from the pool.  Checked
    against the oracle chain (vision oracle -> projector -> text oracle) fed the same slots."""
    from types import SimpleNamespace
    from oracle import llama as ollama, mllama as omllama, mllama_vision as ov, ops
    from scratchpad_amd.mllama import get_full_text_row_masked_out_mask
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    g, gv = golden.load("tiny_mllama"), golden.load("mllama_vision")
    hidden, inter, nl, Hq, Hkv, vocab = (int(x) for x in g["cfg"])
    vsh = ov.VisionShape.from_fixture(gv)
    vis = SimpleNamespace(hidden_size=vsh.hidden, attention_heads=vsh.heads, intermediate_size=vsh.inter,
                          num_hidden_layers=vsh.layers, num_global_layers=vsh.global_layers,
                          image_size=vsh.image_size, patch_size=vsh.patch_size, num_channels=vsh.channels,
                          max_num_tiles=vsh.max_num_tiles, max_aspect_ratio_id=vsh.max_aspect_ratio_id,
                          norm_eps=vsh.norm_eps, intermediate_layers_indices=vsh.intermediate_layers_indices,
                          hidden_act="gelu", vision_output_dim=int(gv["full_out"].shape[-1]))
    cfg = ModelConfig(hidden, inter, nl, Hq, Hkv, vocab, context_len=60, rms_norm_eps=1e-5, rope_theta=500000.0,
                      max_position_embeddings=128, cross_attention_layers=[int(x) for x in g["cross_layers"]],
                      vision_config=vis)
    mr = ModelRunner(cfg, ServerArgs(max_total_tokens=96, max_running_requests=3, disable_cuda_graph=True),
                     dtype=torch.float16, init_weights=False)
    wt = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")}
    wv = {k[2:]: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in gv.items() if k.startswith("w.")}
    gen = torch.Generator().manual_seed(21)
    pw = (torch.randn(hidden, vis.vision_output_dim, generator=gen) * 0.05).half().float()
    pb = (torch.randn(hidden, generator=gen) * 0.05).half().float()
    full = dict(wt)
    full.update({"vision_model." + k: v for k, v in wv.items()})
    full.update({"multi_modal_projector.weight": pw, "multi_modal_projector.bias": pb})
    mr.model.load_full_state_dict({k: v.to(mr.device) for k, v in full.items()})
    worker = TpModelWorker(mr)

    pix = torch.from_numpy(gv["ragged_pixel_values"])[:1]
    ar_ids = torch.from_numpy(gv["ragged_aspect_ratio_ids"])[:1]
    ar_mask = torch.from_numpy(gv["ragged_aspect_ratio_mask"])[:1]
    mm = SimpleNamespace(mm_items=[SimpleNamespace(pixel_values=pix, pad_value=3, aspect_ratio_id=ar_ids,
                                                   aspect_ratio_mask=ar_mask)], num_image_tokens=None)
    text_a = torch.randint(4, vocab, (6,), generator=gen).tolist()
    text_b = torch.randint(4, vocab, (4,), generator=gen).tolist()
    ids_a = mr.model.pad_input_ids(text_a, mm)
    enc = mm.num_image_tokens
    assert enc == vsh.max_num_tiles * vsh.num_patches and ids_a[:enc] == [3] * enc
    reqs = [Req("img", "", ids_a, None, num_image_tokens=enc, multimodal_inputs=mm), Req("txt", "", text_b, None)]
    sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=mr.device, is_encoder_decoder=True)
    sb.prepare_for_extend()
    assert sb.encoder_lens_cpu == [enc, 0] and sb.extend_lens == [6, 4] and sb.encoder_cached == [False, True]
    out, nxt = worker.forward_batch_generation(sb.get_model_worker_batch())

    # ---- oracle chain on the same slots
    wt16 = {k: v.half().float() for k, v in wt.items()}
    vision = ov.forward(vsh, {k: v.half().float() for k, v in wv.items()}, pix, ar_ids, ar_mask)
    states = torch.nn.functional.linear(vision, pw, pb).reshape(enc, hidden)
    shape = ollama.LlamaShape(hidden, inter, nl, Hq, Hkv, vocab, False, 500000.0, None, 128, 1e-5)
    okv = ollama.OracleKV(shape, 96, 4, 64)
    okv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu())
    text = torch.tensor([6, 4], dtype=torch.int32)
    start = torch.tensor([0, 6], dtype=torch.int32)
    fb_like = SimpleNamespace(forward_mode=__import__("scratchpad_amd.forward_info", fromlist=["x"]).ForwardMode.EXTEND,
                              extend_seq_lens_cpu=[6, 4], seq_lens_cpu=torch.tensor([6, 4]),
                              seq_lens=torch.tensor([6, 4]), encoder_lens_cpu=[enc, 0])
    row_mask = get_full_text_row_masked_out_mask(fb_like).float()
    pos, _ = ops.compute_position(torch.zeros(2, dtype=torch.int32), text)
    common = dict(req_pool_indices=sb.req_pool_indices.cpu(), encoder_lens=torch.tensor([enc, 0]))
    ref = omllama.forward(shape, cfg.cross_attention_layers, wt16, okv, mode="extend", input_ids=sb.input_ids.cpu(),
                          positions=pos, seq_lens=sb.seq_lens.cpu(), out_cache_loc=sb.out_cache_loc.cpu(),
                          row_mask=row_mask, extend_seq_lens=text, extend_start_loc=start,
                          cross_attention_states=states, encoder_out_cache_loc=sb.encoder_out_cache_loc.cpu(), **common)
    rel = lambda a, b: float((a.float().cpu() - b.float()).abs().max() / b.float().abs().max())
    assert rel(out.next_token_logits, ref) <= 4e-3, rel(out.next_token_logits, ref)

    sb.output_ids = ref.argmax(-1).to(mr.device)
    sb.prepare_for_decode()
    out2, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
    okv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu())
    ref2 = omllama.forward(shape, cfg.cross_attention_layers, wt16, okv, mode="decode", input_ids=sb.input_ids.cpu(),
                           positions=ops.clamp_position(sb.seq_lens.cpu()), seq_lens=sb.seq_lens.cpu(),
                           out_cache_loc=sb.out_cache_loc.cpu(), row_mask=torch.tensor([[1.0], [0.0]]), **common)
    assert rel(out2.next_token_logits, ref2) <= 4e-3, rel(out2.next_token_logits, ref2)
