#!/usr/bin/env python3
"""BASELINE.json config 5: Mllama-11B text shapes, 1 image x 4 tiles (encoder_len 6404), text prompt 64,
decode bs in {1, 32} (eager launches and HIP-graph replay), and the
vision tower (32 + 8 layers, 4 x 1032 positions, 16 heads of 80) on one image.  Random weights."""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def decode_bench(bs, steps=24, warmup=4, enc_len=6404, text_len=64, graph=True):
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    total = warmup + steps + 4
    cfg = ModelConfig.mllama_11b_text(enc_len + text_len + total + 8)
    sargs = ServerArgs(max_total_tokens=bs * (enc_len + text_len + total) + 64, max_running_requests=bs,
                       disable_cuda_graph=not graph, cuda_graph_bs=[bs], cuda_graph_max_bs=bs)
    mr = ModelRunner(cfg, sargs, dtype=torch.bfloat16, seed=0)
    mr.init_cuda_graphs()
    for arena in (mr.token_to_kv_pool._k_arena, mr.token_to_kv_pool._v_arena):
        for layer in range(arena.shape[0]):
            arena[layer].normal_(0.0, 0.5)
    dev = mr.device
    gen = torch.Generator().manual_seed(0)
    alloc = mr.token_to_kv_pool_allocator
    alloc.free_slots = (torch.randperm(alloc.size, generator=gen) + 1).to(torch.int64).to(dev)
    batch = ScheduleBatch([Req(rid=str(i), origin_input_ids=[], num_image_tokens=enc_len) for i in range(bs)],
                          mr.req_to_token_pool, alloc, device=dev, is_encoder_decoder=True)
    rows = batch.alloc_req_slots(bs)
    batch.req_pool_indices = torch.tensor(rows, dtype=torch.int64, device=dev)
    batch.seq_lens = torch.full((bs,), text_len, dtype=torch.int64, device=dev)
    batch.seq_lens_sum = bs * text_len
    batch.encoder_lens_cpu = [enc_len] * bs
    batch.encoder_lens = torch.tensor(batch.encoder_lens_cpu, dtype=torch.int64, device=dev)
    batch.encoder_cached = [True] * bs
    per = enc_len + text_len
    slots = alloc.alloc(bs * per)
    for i in range(bs):
        mr.req_to_token_pool.req_to_token[rows[i], :per] = slots[i * per:(i + 1) * per].to(torch.int32)
    batch.forward_mode = ForwardMode.DECODE
    batch.output_ids = torch.randint(0, cfg.vocab_size, (bs,), generator=gen).to(dev)
    worker = TpModelWorker(mr)

    def step():
        batch.prepare_for_decode()
        _, nxt = worker.forward_batch_generation(batch.get_model_worker_batch())
        batch.output_ids = nxt

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    cross = 8 * bs * enc_len * 2 * 8 * 128 * 2
    selfb = 32 * bs * (text_len + warmup + steps // 2) * 2 * 8 * 128 * 2
    weights = 9.8e9 * 2
    print(f"mllama-11b text decode ({'HIP graph' if graph else 'eager'}) bs={bs:3d} encoder_len={enc_len} text~{text_len}: {ms:7.2f} ms/step  "
          f"{bs / ms * 1e3:8.1f} tok/s   (HBM floor: weights {weights / 8e12 * 1e3:.2f} ms + cross-KV "
          f"{cross / 8e12 * 1e3:.2f} ms + self-KV {selfb / 8e12 * 1e3:.3f} ms)", flush=True)
    del mr, worker, batch
    torch.cuda.empty_cache()


def vision_bench(iters=3):
    from scratchpad_amd import distributed as dist_
    from scratchpad_amd.mllama_vision import MllamaVisionModel
    if not dist_.model_parallel_is_initialized():
        dist_.initialize_model_parallel(1)
    cfg = SimpleNamespace(hidden_size=1280, attention_heads=16, intermediate_size=5120, num_hidden_layers=32,
                          num_global_layers=8, image_size=560, patch_size=14, num_channels=3, max_num_tiles=4,
                          max_aspect_ratio_id=8, norm_eps=1e-5, intermediate_layers_indices=[3, 7, 15, 23, 30],
                          hidden_act="gelu", vision_output_dim=7680)
    model = MllamaVisionModel(cfg, dtype=torch.bfloat16).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, p in model.named_parameters():
        if p.numel() == 1:
            p.data.fill_(0.5)
        elif "layernorm" in name and name.endswith("weight"):
            p.data.fill_(1.0)
        elif name.endswith("bias"):
            p.data.zero_()
        else:
            p.data.normal_(0.0, 0.02, generator=g)
    for mod in model.modules():      # padded head columns must stay zero
        if hasattr(mod, "pack_qkv"):
            D, Dp = mod.head_size, mod.kernel_head_size
            w = mod.qkv_proj.weight.data.view(3, mod.num_heads, Dp, -1)
            w[:, :, D:] = 0
            mod.proj.weight.data.view(-1, mod.num_heads, Dp)[:, :, D:] = 0
    pixels = torch.randn(1, 1, 4, 3, 560, 560, device="cuda", generator=g)
    ids = torch.tensor([[6]], device="cuda")
    for mask in ([[[1, 1, 1, 1]]], [[[1, 1, 0, 0]]]):
        m = torch.tensor(mask)
        model(pixels, ids, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(pixels, ids, m)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        flops_attn = 40 * 4 * 16 * 80 * 4128 * 4128
        print(f"mllama-11b vision tower, 1 image, tiles {mask[0][0]}: {ms:7.2f} ms  out {tuple(out.shape)}  "
              f"(attention {flops_attn / 1e12:.2f} TFLOP useful, GEMMs ~{40 * 4128 * 2 * (4 * 1280 * 1280 + 2 * 1280 * 5120) / 1e12:.2f} TFLOP)",
              flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["decode1", "decode32", "vision"]
    if "vision" in what:
        vision_bench()
    for name, bs in (("decode1", 1), ("decode32", 32)):
        if name in what:
            decode_bench(bs, graph=False)
            decode_bench(bs, graph=True)
