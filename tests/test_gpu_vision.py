"""Cache-less (vision) attention and the Mllama vision tower on the HIP extend kernel, against
oracle/mllama_vision.py and the reference-generated fixture tests/golden/mllama_vision.npz.

Tolerances (relative to max|expected|, stated per test): the tower computes in fp16/bf16 with fp32
accumulation; the oracle and the reference run are fp32."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import mllama_vision as ov
from tests import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tp1():
    from scratchpad_amd import distributed as dist_
    if not dist_.model_parallel_is_initialized():
        dist_.initialize_model_parallel(1)


def rel(got, want):
    return float((got.float().cpu() - want.float()).abs().max() / want.float().abs().max())


def _cfg(sh: ov.VisionShape, vision_output_dim=None):
    return SimpleNamespace(hidden_size=sh.hidden, attention_heads=sh.heads, intermediate_size=sh.inter,
                           num_hidden_layers=sh.layers, num_global_layers=sh.global_layers,
                           image_size=sh.image_size, patch_size=sh.patch_size, num_channels=sh.channels,
                           max_num_tiles=sh.max_num_tiles, max_aspect_ratio_id=sh.max_aspect_ratio_id,
                           norm_eps=sh.norm_eps, intermediate_layers_indices=sh.intermediate_layers_indices,
                           hidden_act="gelu",
                           vision_output_dim=vision_output_dim or sh.hidden * (
                               1 + len(sh.intermediate_layers_indices) + (sh.layers - 1 in sh.intermediate_layers_indices)))


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("D", [64, 128])
def test_varlen_attention_matches_dense_softmax(dtype, tol, D):
    from scratchpad_amd.vision import varlen_attention
    gen = torch.Generator().manual_seed(D)
    seq_lens = [5, 130, 64, 1, 257]
    H, total = 4, sum(seq_lens)
    q, k, v = (torch.randn(total, H, D, generator=gen).to(dtype) for _ in range(3))
    got = varlen_attention(q.cuda(), k.cuda(), v.cuda(), seq_lens, D ** -0.5)
    want = torch.zeros(total, H, D)
    lo = 0
    for n in seq_lens:
        sl = slice(lo, lo + n)
        p = torch.softmax(torch.einsum("ihd,jhd->hij", q[sl].float(), k[sl].float()) * D ** -0.5, -1)
        want[sl] = torch.einsum("hij,jhd->ihd", p, v[sl].float())
        lo += n
    assert rel(got, want) <= tol
    # keys chosen through a table: sequence 0's queries over the rows of sequence 2 only
    table = torch.arange(135, 135 + 64, dtype=torch.int32).view(1, -1).cuda()
    got = varlen_attention(q[:5].cuda(), k.cuda(), v.cuda(), [5], D ** -0.5, key_index=table, key_lens=[64])
    p = torch.softmax(torch.einsum("ihd,jhd->hij", q[:5].float(), k[135:199].float()) * D ** -0.5, -1)
    assert rel(got, torch.einsum("hij,jhd->ihd", p, v[135:199].float())) <= tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_causal_varlen_attention_matches_the_reference_prefill_kernel(dtype):
    """varlen_attention(causal=True) - sp_extend_attention with no cached prefix - against the outputs of the
    reference's own cache-less kernel, context_attention_fwd (nn/attention/triton_attn/prefill_attention.py:125-163,
    run under the Triton interpreter: tests/golden/prefill_attention.npz).  Inputs are on a grid exact in fp16 and
    bf16; the bound is one unit of the 16-bit attention error model (helpers.attn_error_units)."""
    from oracle import ops
    from scratchpad_amd.vision import varlen_attention
    from tests.helpers import assert_attn_close
    g = golden.load("prefill_attention")
    for i in range(int(g["num_cases"])):
        q, k, v = (torch.from_numpy(g[f"c{i}_{n}"]) for n in ("q", "k", "v"))
        lens = g[f"c{i}_b_seq_len"].tolist()
        assert g[f"c{i}_b_start_loc"].tolist() == [sum(lens[:b]) for b in range(len(lens))]
        got = varlen_attention(q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda(), lens,
                               q.shape[-1] ** -0.5, causal=True)
        aref = ops.context_attention(q, k, v.abs(), torch.from_numpy(g[f"c{i}_b_start_loc"]),
                                     torch.from_numpy(g[f"c{i}_b_seq_len"]))
        assert_attn_close(got, torch.from_numpy(g[f"c{i}_o"]), aref, dtype, what=f"causal varlen c{i} {dtype}")


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2.5e-2)])
def test_vision_attention_layer_padded_heads_mask_and_cu_seqlens(dtype, tol):
    """Head size 80 (Mllama's) zero-padded to 128 in the weights; tile-mask semantics; ragged rows."""
    from scratchpad_amd.vision import VisionAttention
    gen = torch.Generator().manual_seed(80)
    E, H, b, s = 320, 4, 2, 96
    w = {"a.qkv_proj.weight": torch.randn(3 * E, E, generator=gen) * 0.06,
         "a.qkv_proj.bias": torch.randn(3 * E, generator=gen) * 0.05,
         "a.proj.weight": torch.randn(E, E, generator=gen) * 0.06,
         "a.proj.bias": torch.randn(E, generator=gen) * 0.05}
    w = {k_: v_.to(dtype).float() for k_, v_ in w.items()}
    attn = VisionAttention(E, H, E, bias=True, dtype=dtype).cuda()
    assert attn.head_size == 80 and attn.kernel_head_size == 128
    attn.load_reference_weights(w["a.qkv_proj.weight"], w["a.qkv_proj.bias"], w["a.proj.weight"], w["a.proj.bias"])
    x = (torch.randn(b, s, E, generator=gen)).to(dtype)
    pad = torch.zeros(b, s, dtype=torch.bool)
    pad[0, 40:48] = True
    pad[1, 20:] = True
    for pad_rows in (None, pad):
        got = attn(x.cuda(), pad_rows=pad_rows)
        want = ov.vision_attention(x.float(), w, "a.", H, pad_rows)
        assert rel(got, want) <= tol, pad_rows is None
    cu = [0, 7, 7, 100, 192]
    got = attn(x.reshape(1, b * s, E).cuda(), cu_seqlens=cu)
    want = ov.vision_attention(x.float().reshape(1, b * s, E), w, "a.", H, cu_seqlens=cu)
    assert rel(got, want) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("tag", ["full", "ragged"])
def test_tower_matches_reference_fixture(dtype, tol, tag):
    from scratchpad_amd.mllama_vision import MllamaVisionModel
    g = golden.load("mllama_vision")
    sh = ov.VisionShape.from_fixture(g)
    w = {k[2:]: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in g.items() if k.startswith("w.")}
    model = MllamaVisionModel(_cfg(sh), dtype=dtype).cuda()
    model.load_full_state_dict({k: v.cuda() for k, v in w.items()})
    pixels, ids, mask = (torch.from_numpy(g[f"{tag}_{n}"]) for n in
                         ("pixel_values", "aspect_ratio_ids", "aspect_ratio_mask"))
    got = model(pixels.cuda(), ids.cuda(), mask)
    want = torch.from_numpy(g[f"{tag}_out"])
    assert got.shape == want.shape
    assert rel(got, want) <= tol
    assert rel(got, ov.forward(sh, w, pixels, ids, mask)) <= tol


def test_tower_at_11b_widths_against_oracle(monkeypatch):
    """Two 560x560 tiles -> 2 x 1608 positions, hidden 1280, 16 heads of 80, 1 local + 1 global layer:
    the real per-layer shape (4 tiles halve to keep the fp32 CPU check in seconds)."""
    from scratchpad_amd.mllama_vision import MllamaVisionModel
    sh = ov.VisionShape(1280, 16, 5120, 1, 1, 560, 14, 2, 8, 3, [0])
    gen = torch.Generator().manual_seed(11)
    model = MllamaVisionModel(_cfg(sh), dtype=torch.bfloat16)
    w = {}
    for name, p in model.named_parameters():
        if "self_attn" in name:
            continue
        if p.numel() == 1:
            t = torch.tensor([0.5])
        elif "layernorm" in name:
            t = (1.0 + 0.05 * torch.randn(p.shape, generator=gen)) if name.endswith("weight") else 0.02 * torch.randn(p.shape, generator=gen)
        else:
            t = torch.randn(p.shape, generator=gen) * (0.02 if p.dim() > 1 else 0.01)
        w[name] = t.to(torch.bfloat16).float()
    for pre in ("transformer.layers.0.self_attn.", "global_transformer.layers.0.self_attn."):
        w[pre + "qkv_proj.weight"] = (torch.randn(3 * 1280, 1280, generator=gen) * 0.02).to(torch.bfloat16).float()
        w[pre + "qkv_proj.bias"] = (torch.randn(3 * 1280, generator=gen) * 0.01).to(torch.bfloat16).float()
        w[pre + "proj.weight"] = (torch.randn(1280, 1280, generator=gen) * 0.02).to(torch.bfloat16).float()
        w[pre + "proj.bias"] = (torch.randn(1280, generator=gen) * 0.01).to(torch.bfloat16).float()
    model = model.cuda()
    model.load_full_state_dict({k: v.cuda() for k, v in w.items()})
    pixels = torch.randn(1, 1, 2, 3, 560, 560, generator=gen).to(torch.bfloat16).float()
    ids = torch.tensor([[2]])
    mask = torch.tensor([[[1, 1]]])
    got = model(pixels.cuda(), ids.cuda(), mask)
    want = ov.forward(sh, w, pixels, ids, mask)
    assert got.shape == (1, 1, 2, 1601, 3840)      # final + taps before and after layer 0
    assert rel(got, want) <= 2e-2
    # the switchable row partition (padding + block-spill rows in the side launch) computes the same thing
    from scratchpad_amd.vision import VisionAttnPlan
    monkeypatch.setattr(VisionAttnPlan, "MOVE_SPILL_ROWS", True)
    plan = VisionAttnPlan(1, 2 * 1608, "cuda", pad_rows=__import__("scratchpad_amd.mllama_vision", fromlist=["x"])
                          .padding_positions(mask.reshape(1, 2), 1601, 1608))
    assert plan.main_rows is not None and plan.main.seq_lens == [3200] and plan.side.seq_lens == [14, 2]
    got2 = model(pixels.cuda(), ids.cuda(), mask)
    assert rel(got2, want) <= 2e-2 and rel(got2, got.float().cpu()) <= 1e-2


def test_conditional_generation_computes_cross_attention_states_from_mm_inputs():
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.mllama import MllamaForConditionalGeneration
    g = golden.load("mllama_vision")
    sh = ov.VisionShape.from_fixture(g)
    w = {k[2:]: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in g.items() if k.startswith("w.")}
    text = SimpleNamespace(vocab_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                           num_key_value_heads=1, intermediate_size=64, cross_attention_layers=[1],
                           rms_norm_eps=1e-5, max_position_embeddings=64, rope_theta=10000.0, rope_scaling=None,
                           hidden_act="silu", tie_word_embeddings=False, head_dim=None)
    vis = _cfg(sh)
    try:
        model = MllamaForConditionalGeneration(SimpleNamespace(text_config=text, vision_config=vis),
                                               dtype=torch.float16).cuda()
    except Exception as e:          # the text-side config surface is exercised in test_gpu_mllama.py
        pytest.skip(f"text config surface: {e}")
    model.vision_model.load_full_state_dict({k: v.cuda() for k, v in w.items()})
    gen = torch.Generator().manual_seed(3)
    pw = (torch.randn(64, vis.vision_output_dim, generator=gen) * 0.05).half()
    pb = (torch.randn(64, generator=gen) * 0.05).half()
    model.multi_modal_projector.weight.data.copy_(pw)
    model.multi_modal_projector.bias.data.copy_(pb)
    pix = torch.from_numpy(g["ragged_pixel_values"])
    ids, mask = torch.from_numpy(g["ragged_aspect_ratio_ids"]), torch.from_numpy(g["ragged_aspect_ratio_mask"])
    P = sh.num_patches
    mm = [SimpleNamespace(mm_items=[SimpleNamespace(pixel_values=pix[i:i + 1], pad_value=7,
                                                    aspect_ratio_id=ids[i:i + 1], aspect_ratio_mask=mask[i:i + 1])])
          for i in (0, 1)]                 # the processor always hands over max_num_tiles tiles
    T = sh.max_num_tiles
    assert model.pad_input_ids([1, 2, 3], mm[0]) == [7] * (T * P) + [1, 2, 3] and mm[0].num_image_tokens == T * P
    fb = SimpleNamespace(forward_mode=ForwardMode.EXTEND, encoder_cached=[False, True, False],
                         mm_inputs=[mm[0], None, mm[1]], encoder_lens_cpu=[T * P, 0, T * P],
                         out_cache_loc=torch.zeros(1, device="cuda"))
    images, ar_ids, ar_mask, need = model._batch_image_inputs(fb)
    assert images.shape == (2, 1, 4, 3, 28, 28) and need == [T * P, T * P]
    assert ar_mask.tolist() == mask.tolist() and ar_ids.tolist() == ids.tolist()
    states = model.encode_images(images, ar_ids, ar_mask, need)
    assert states.shape == (2 * T * P, 64)
    want_full = ov.forward(sh, w, pix, ids, mask)
    want = torch.nn.functional.linear(want_full, pw.float(), pb.float()).reshape(2 * T * P, 64)
    assert rel(states, want) <= 6e-3
    fb.encoder_cached = [True, True, True]
    assert model._batch_image_inputs(fb) == (None, None, None, None)
