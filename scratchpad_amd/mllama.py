"""Mllama (Llama-3.2-Vision) TEXT side on the HIP hot path: cross-attention decoder layers.

Re-hosts nn/models/llama/mllama.py: MllamaTextRMSNorm 469-483, MllamaTextCrossAttention 486-571,
MllamaCrossAttentionDecoderLayer 574-634, MllamaTextModel 637-716, MllamaForCausalLM 719-766 and the
batch helpers of MllamaForConditionalGeneration (flat_encoder_result 880-902,
get_full_text_row_masked_out_mask 904-927, forward 929-990).  Same module tree / parameter names.

What runs on HIP: the per-head q/k RMSNorm (sp_rmsnorm over [T*H, D] rows), the cross-attention
itself (sp_extend_attention with causal=0 / sp_decode_attention over the encoder slots
[0, encoder_len) of the request's req_to_token row, K/V stored at encoder_out_cache_loc), and the
decoder self-attention behind the encoder slots (kv_start = encoder_lens).  The tanh gates and the
row mask are the reference's own torch elementwise ops.

The vision tower lives in mllama_vision.py (its attention runs on the same extend kernel).
``MllamaForConditionalGeneration.forward`` either receives the projected, flattened
``cross_attention_states`` or computes them from ``forward_batch.mm_inputs`` exactly where the
reference does (mllama.py:818-877, 961-976).
"""
from typing import List, Optional

import torch
from torch import nn

from . import _native
from .attention import RadixAttention
from .distributed import get_tensor_model_parallel_world_size
from .forward_info import ForwardBatch
from .layers import RMSNorm
from .llama import (LlamaDecoderLayer, LlamaMLP, LogitsProcessor, ParallelLMHead,
                    QKVParallelLinear, RowParallelLinear, VocabParallelEmbedding)


class MllamaTextRMSNorm(nn.Module):
    """mllama.py:469-483: weight * round(x * rsqrt(mean(x^2) + eps)), over the last (head) dim."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        return _native.rmsnorm(hidden_states, self.weight.data, self.variance_epsilon)


class MllamaTextCrossAttention(nn.Module):
    def __init__(self, config, layer_id: int, dtype=None):
        super().__init__()
        self.config = config
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = config.num_attention_heads
        self.num_local_heads = self.num_heads // tp
        self.num_key_value_heads = config.num_key_value_heads
        self.num_local_key_value_heads = self.num_key_value_heads // tp
        self.hidden_size = config.hidden_size
        self.head_dim = config.hidden_size // self.num_heads
        self.layer_id = layer_id
        self.q_local_size = self.num_local_heads * self.head_dim
        self.kv_local_size = self.num_local_key_value_heads * self.head_dim
        self.qkv_proj = QKVParallelLinear(self.hidden_size, self.head_dim, self.num_heads,
                                          self.num_key_value_heads, dtype)
        self.o_proj = RowParallelLinear(self.num_heads * self.head_dim, self.hidden_size, dtype)
        self.q_norm = MllamaTextRMSNorm(self.head_dim, eps=config.rms_norm_eps)
        self.k_norm = MllamaTextRMSNorm(self.head_dim, eps=config.rms_norm_eps)
        self.scaling = self.head_dim ** -0.5
        self.attn = RadixAttention(self.num_local_heads, self.head_dim, self.scaling,
                                   self.num_local_key_value_heads, layer_id=layer_id,
                                   is_cross_attention=True)

    def forward(self, hidden_states, attention_mask, cross_attention_states,
                forward_batch: ForwardBatch) -> torch.Tensor:
        qkv_dec, _ = self.qkv_proj(hidden_states)
        q, _, _ = qkv_dec.split([self.q_local_size, self.kv_local_size, self.kv_local_size], dim=-1)
        if cross_attention_states is None:
            k = v = None          # encoder K/V already in the pool (decode, or a cached encoder)
        else:
            qkv_enc, _ = self.qkv_proj(cross_attention_states)
            _, k, v = qkv_enc.split([self.q_local_size, self.kv_local_size, self.kv_local_size], dim=-1)
            k = k.reshape(-1, self.num_local_key_value_heads, self.head_dim)
            v = v.reshape(-1, self.num_local_key_value_heads, self.head_dim)
            k = self.k_norm(k)
        q = q.reshape(-1, self.num_local_heads, self.head_dim)
        q = self.q_norm(q)
        output = self.attn(q, k, v, forward_batch)
        out, _ = self.o_proj(output)
        return out


class MllamaCrossAttentionDecoderLayer(nn.Module):
    """mllama.py:574-634: cross-attention block with tanh-gated attention and feed-forward."""

    def __init__(self, config, layer_id: int, dtype=None):
        super().__init__()
        self.layer_id = layer_id
        self.cross_attn = MllamaTextCrossAttention(config, layer_id, dtype)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.cross_attn_attn_gate = nn.Parameter(torch.zeros(1), requires_grad=False)
        self.mlp = LlamaMLP(config.hidden_size, config.intermediate_size, config.hidden_act, dtype)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.cross_attn_mlp_gate = nn.Parameter(torch.zeros(1), requires_grad=False)

    def forward(self, hidden_states, cross_attention_states, cross_attention_mask,
                full_text_row_masked_out_mask, forward_batch: ForwardBatch) -> torch.Tensor:
        residual = hidden_states
        hidden_states = self.input_layernorm(hidden_states)
        hidden_states = self.cross_attn(hidden_states=hidden_states, attention_mask=cross_attention_mask,
                                        cross_attention_states=cross_attention_states,
                                        forward_batch=forward_batch)
        hidden_states = full_text_row_masked_out_mask * hidden_states
        hidden_states = residual + self.cross_attn_attn_gate.tanh() * hidden_states
        residual = hidden_states
        hidden_states = self.post_attention_layernorm(hidden_states)
        hidden_states = self.mlp(hidden_states)
        hidden_states = full_text_row_masked_out_mask * hidden_states
        hidden_states = residual + self.cross_attn_mlp_gate.tanh() * hidden_states
        return hidden_states


class MllamaTextModel(nn.Module):
    def __init__(self, config, dtype=None):
        super().__init__()
        self.vocab_size = config.vocab_size
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size + 8, config.hidden_size, dtype)
        self.cross_attention_layers = list(config.cross_attention_layers)
        layers = []
        for layer_id in range(config.num_hidden_layers):
            if layer_id in self.cross_attention_layers:
                layers.append(MllamaCrossAttentionDecoderLayer(config, layer_id, dtype))
            else:
                layers.append(LlamaDecoderLayer(config, layer_id, dtype))
        self.layers = nn.ModuleList(layers)
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, cross_attention_states, cross_attention_mask,
                full_text_row_masked_out_mask, forward_batch: ForwardBatch,
                skip_cross_attention: bool) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        for decoder_layer in self.layers:
            if isinstance(decoder_layer, MllamaCrossAttentionDecoderLayer):
                if not skip_cross_attention:
                    hidden_states = decoder_layer(
                        hidden_states=hidden_states, cross_attention_states=cross_attention_states,
                        cross_attention_mask=cross_attention_mask,
                        full_text_row_masked_out_mask=full_text_row_masked_out_mask,
                        forward_batch=forward_batch)
            else:
                # mllama.py:700-706: self-attention layers are called with residual=None and the
                # residual is added back here (no fused add across layers)
                hidden_states, residual = decoder_layer(positions=positions, hidden_states=hidden_states,
                                                        forward_batch=forward_batch, residual=None)
                hidden_states = hidden_states + residual
        return self.norm(hidden_states)


class MllamaForCausalLM(nn.Module):
    def __init__(self, config, dtype=None):
        super().__init__()
        self.vocab_size = config.vocab_size
        self.model = MllamaTextModel(config, dtype)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, dtype)

    def forward(self, input_ids, positions, cross_attention_states, cross_attention_mask,
                full_text_row_masked_out_mask, forward_batch, skip_cross_attention):
        return self.model(input_ids=input_ids, positions=positions,
                          cross_attention_states=cross_attention_states,
                          cross_attention_mask=cross_attention_mask,
                          full_text_row_masked_out_mask=full_text_row_masked_out_mask,
                          forward_batch=forward_batch, skip_cross_attention=skip_cross_attention)


def flat_encoder_result(cross_attention_states: torch.Tensor, encoder_lens_need: List[int]) -> torch.Tensor:
    """mllama.py:880-902: [n_images, max_len, hidden] -> [sum(encoder_lens_need), hidden]."""
    hidden = cross_attention_states.shape[-1]
    flat = torch.zeros(sum(encoder_lens_need), hidden, device=cross_attention_states.device,
                       dtype=cross_attention_states.dtype)
    i = start = 0
    for n in encoder_lens_need:
        if n == 0:
            continue
        flat[start:start + n] = cross_attention_states[i][:n]
        i += 1
        start += n
    return flat


def get_full_text_row_masked_out_mask(forward_batch: ForwardBatch) -> torch.Tensor:
    """mllama.py:904-927, reproduced as written: in extend mode the row cursor advances by
    ENCODER length (not by the request's text length), so text-only requests mask the rows
    [sum(previous encoder_lens), + seq_len) - a quirk of the reference that parity keeps."""
    if forward_batch.forward_mode.is_decode():
        mask = forward_batch.encoder_lens != 0
    else:
        mask = torch.ones(int(sum(forward_batch.extend_seq_lens_cpu)), dtype=torch.bool)
        start = 0
        seq_lens = (forward_batch.seq_lens_cpu.tolist() if forward_batch.seq_lens_cpu is not None
                    else forward_batch.seq_lens.tolist())
        for seq_len, encoder_len in zip(seq_lens, forward_batch.encoder_lens_cpu):
            if encoder_len == 0:
                mask[start:start + seq_len] = False
            start += encoder_len
        mask = mask.to(forward_batch.seq_lens.device)
    return mask.reshape(-1, 1)


class MllamaForConditionalGeneration(nn.Module):
    """The text side of mllama.py:769-990.  ``cross_attention_states``: projected vision states of
    the requests whose encoder is not cached, flattened ([sum(encoder_lens_need), hidden]) - what
    vision_model + multi_modal_projector + flat_encoder_result produce in the reference."""

    def __init__(self, config, dtype=None):
        super().__init__()
        text_config = getattr(config, "text_config", config)
        self.language_model = MllamaForCausalLM(text_config, dtype)
        self.logits_processor = LogitsProcessor(text_config)
        self.capture_mode = False
        vision_config = getattr(config, "vision_config", None)
        self.vision_model = None
        if vision_config is not None:
            from .mllama_vision import MllamaVisionModel
            self.vision_model = MllamaVisionModel(vision_config, dtype)
            self.multi_modal_projector = nn.Linear(vision_config.vision_output_dim, text_config.hidden_size,
                                                   bias=True, dtype=dtype)
            self.image_size = vision_config.image_size
            self.max_num_tiles = vision_config.max_num_tiles

    def pad_input_ids(self, input_ids: List[int], mm_inputs) -> List[int]:
        """mllama.py:803-816: one pad id per vision position, in front of the text ids."""
        pixel_values = torch.cat([item.pixel_values for item in mm_inputs.mm_items], dim=0)
        pad_values = [item.pad_value for item in mm_inputs.mm_items]
        image_len = pixel_values.shape[1] * pixel_values.shape[2] * self.vision_model.num_patches
        mm_inputs.num_image_tokens = image_len
        reps = (image_len + len(pad_values)) // len(pad_values)
        return (pad_values * reps)[:image_len] + input_ids

    def _batch_image_inputs(self, forward_batch: ForwardBatch):
        """mllama.py:818-877: stack the pixel tiles of the requests whose encoder is not cached."""
        if forward_batch.forward_mode.is_decode() or all(forward_batch.encoder_cached):
            return None, None, None, None
        todo = [(k, mm) for k, mm in enumerate(forward_batch.mm_inputs or [])
                if mm is not None and not forward_batch.encoder_cached[k]]
        if not todo:
            return None, None, None, None
        pix = [torch.cat([item.pixel_values for item in mm.mm_items], dim=0) for _, mm in todo]
        n_img = max(p.shape[1] for p in pix)
        n_tile = max(p.shape[2] for p in pix)
        if n_img * n_tile == 0:
            return None, None, None, None
        dev = forward_batch.out_cache_loc.device
        images = torch.zeros(len(todo), n_img, n_tile, 3, self.image_size, self.image_size,
                             dtype=torch.float32, device=dev)
        ar_ids = torch.ones(len(todo), n_img, dtype=torch.int64, device=dev)
        ar_mask = torch.zeros(len(todo), n_img, n_tile, dtype=torch.int64)
        need = []
        for i, ((k, mm), p) in enumerate(zip(todo, pix)):
            need.append(int(forward_batch.encoder_lens_cpu[k]))
            first = mm.mm_items[0]
            for j in range(p.shape[1]):
                tiles = p[0, j]
                images[i, j, :tiles.shape[0]] = tiles
                ar_ids[i, j] = first.aspect_ratio_id[0, j]
                ar_mask[i, j, :tiles.shape[0]] = first.aspect_ratio_mask[0, j]
        return images, ar_ids, ar_mask, need

    @torch.no_grad()
    def encode_images(self, images, ar_ids, ar_mask, encoder_lens_need: List[int]) -> torch.Tensor:
        """mllama.py:961-976: vision tower -> projector -> flat [sum(encoder_lens_need), hidden]."""
        states = self.multi_modal_projector(self.vision_model(images, ar_ids, ar_mask))
        return flat_encoder_result(states.view(states.shape[0], -1, states.shape[-1]), encoder_lens_need)

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch: ForwardBatch,
                cross_attention_states: Optional[torch.Tensor] = None):
        if self.capture_mode:
            skip_cross_attention = False
        else:
            assert len(forward_batch.encoder_lens_cpu) == len(forward_batch.seq_lens)
            skip_cross_attention = max(forward_batch.encoder_lens_cpu) == 0
        mask = None if skip_cross_attention else get_full_text_row_masked_out_mask(forward_batch)
        needs_encoder = (not forward_batch.forward_mode.is_decode()
                         and not all(forward_batch.encoder_cached))
        if needs_encoder and cross_attention_states is None:
            if self.vision_model is None or not forward_batch.mm_inputs:
                raise RuntimeError("uncached image tokens in the batch but neither cross_attention_states "
                                   "nor (vision_config + forward_batch.mm_inputs) to compute them from")
            images, ar_ids, ar_mask, need = self._batch_image_inputs(forward_batch)
            if images is not None:
                cross_attention_states = self.encode_images(images, ar_ids, ar_mask, need)
        hidden_states = self.language_model(
            input_ids=input_ids, positions=positions, cross_attention_states=cross_attention_states,
            cross_attention_mask=None, full_text_row_masked_out_mask=mask, forward_batch=forward_batch,
            skip_cross_attention=skip_cross_attention)
        return self.logits_processor(input_ids, hidden_states, self.language_model.lm_head, forward_batch)

    def load_full_state_dict(self, full):
        """tp=1 state_dict of the reference's MllamaForCausalLM (keys 'model.*', 'lm_head.*'); keys
        'vision_model.*' / 'multi_modal_projector.*' are loaded too when a vision tower is present."""
        if self.vision_model is not None and any(k.startswith("vision_model.") for k in full):
            self.vision_model.load_full_state_dict(
                {k[len("vision_model."):]: v for k, v in full.items() if k.startswith("vision_model.")})
            for name in ("weight", "bias"):
                key = "multi_modal_projector." + name
                if key in full:
                    getattr(self.multi_modal_projector, name).data.copy_(full[key])
        own = dict(self.language_model.named_parameters())
        mods = dict(self.language_model.named_modules())
        for name, param in own.items():
            src = full[name]
            mod = mods[name.rsplit(".", 1)[0]] if "." in name else None
            if mod is not None and hasattr(mod, "shard_from_full"):
                src = mod.shard_from_full(src)
            param.data.copy_(src.to(param.dtype).reshape(param.shape))
