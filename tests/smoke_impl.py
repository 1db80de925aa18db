"""Tiny-Llama end-to-end through the HIP path (shared by __graft_entry__.smoke() and the GPU
tests): build a ModelRunner from the golden fixture's weights, run the ragged prefill and one
decode step as the scheduler would hand them over, compare logits."""
import numpy as np
import torch

from oracle import llama as ollama
from oracle import ops
from tests import golden


def load_case(name):
    g = golden.load("tiny_llama")
    pfx = name + "_"
    hidden, inter, nl, Hq, Hkv, vocab, tie = (int(x) for x in g[pfx + "cfg"])
    sc = tuple(float(x) for x in g[pfx + "rope_scaling"]) if pfx + "rope_scaling" in g else None
    shape = ollama.LlamaShape(hidden, inter, nl, Hq, Hkv, vocab, bool(tie), float(g[pfx + "rope_theta"]),
                              sc, int(g[pfx + "max_pos"]), float(g[pfx + "rms_eps"]))
    w = {k[len(pfx) + 3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pfx + "w::")}
    return g, pfx, shape, w


def make_runner(shape, w, dtype, graph_bs=None):
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs
    scaling = None
    if shape.rope_scaling is not None:
        f = shape.rope_scaling
        scaling = {"rope_type": "llama3", "factor": f[0], "low_freq_factor": f[1],
                   "high_freq_factor": f[2], "original_max_position_embeddings": int(f[3])}
    cfg = ModelConfig(shape.hidden, shape.inter, shape.layers, shape.Hq, shape.Hkv, shape.vocab,
                      context_len=60, rms_norm_eps=shape.rms_eps, rope_theta=shape.rope_theta,
                      rope_scaling=scaling, max_position_embeddings=shape.max_pos,
                      tie_word_embeddings=shape.tie)
    args = ServerArgs(max_total_tokens=96, max_running_requests=3, disable_cuda_graph=graph_bs is None,
                      cuda_graph_bs=graph_bs)
    mr = ModelRunner(cfg, args, dtype=dtype, init_weights=False)
    mr.model.load_full_state_dict({k: v.to(mr.device) for k, v in w.items()})
    if graph_bs is not None:
        mr.init_cuda_graphs()
    return mr


def run_case(name, dtype, graph_bs=None):
    """returns (hip prefill logits, hip decode logits, golden/oracle prefill, decode) as fp32 CPU"""
    from scratchpad_amd import _native
    from scratchpad_amd.forward_info import ForwardMode, ModelWorkerBatch
    from scratchpad_amd.model_runner import TpModelWorker
    g, pfx, shape, w = load_case(name)
    mr = make_runner(shape, w, dtype, graph_bs)
    worker = TpModelWorker(mr)
    dev = mr.device
    ext = torch.from_numpy(g[pfx + "extend_lens"])
    req = torch.from_numpy(g[pfx + "req_pool_indices"]).to(dev)
    loc = torch.from_numpy(g[pfx + "out_cache_loc"]).to(dev)
    ids = torch.from_numpy(g[pfx + "input_ids"]).to(dev)
    seq = ext.long().to(dev)
    zeros = torch.zeros_like(seq)
    _native.write_req_to_token(mr.req_to_token_pool.req_to_token, req, zeros, seq, seq, loc)
    batch = ModelWorkerBatch(bid=1, forward_mode=ForwardMode.EXTEND, input_ids=ids, req_pool_indices=req,
                             seq_lens=seq, out_cache_loc=loc, seq_lens_sum=int(ext.sum()),
                             extend_num_tokens=int(ext.sum()), extend_seq_lens=ext.tolist(),
                             extend_prefix_lens=[0] * len(ext))
    out, nxt = worker.forward_batch_generation(batch)
    prefill = out.next_token_logits.float().cpu()
    # decode step exactly as ScheduleBatch.prepare_for_decode lays it out
    dloc = torch.from_numpy(g[pfx + "decode_out_cache_loc"]).to(dev)
    seq2 = seq + 1
    mr.req_to_token_pool.write((req, seq), dloc.to(torch.int32))
    nxt_in = torch.from_numpy(g[pfx + "next_ids"]).to(dev)       # the reference's sampled ids
    batch2 = ModelWorkerBatch(bid=2, forward_mode=ForwardMode.DECODE, input_ids=nxt_in,
                              req_pool_indices=req, seq_lens=seq2, out_cache_loc=dloc,
                              seq_lens_sum=int(seq2.sum()))
    out2, _ = worker.forward_batch_generation(batch2)
    decode = out2.next_token_logits.float().cpu()
    return (prefill, decode, nxt.cpu(), torch.from_numpy(g[pfx + "prefill_logits"]),
            torch.from_numpy(g[pfx + "decode_logits"]), torch.from_numpy(g[pfx + "next_ids"]), mr)


def oracle_logits(name, dtype):
    """the oracle evaluated in `dtype` (for 16-bit comparisons at equal input rounding)"""
    g, pfx, shape, w = load_case(name)
    w = {k: v.to(dtype) for k, v in w.items()}
    kv = ollama.OracleKV(shape, 96, 4, 64, dtype)
    ext = torch.from_numpy(g[pfx + "extend_lens"])
    req = torch.from_numpy(g[pfx + "req_pool_indices"])
    loc = torch.from_numpy(g[pfx + "out_cache_loc"])
    pre = torch.zeros_like(ext)
    ops.write_req_to_token(kv.req_to_token, req, pre, ext, ext, loc)
    positions, start = ops.compute_position(pre, ext)
    l1 = ollama.forward(shape, w, kv, mode="extend", input_ids=torch.from_numpy(g[pfx + "input_ids"]),
                        positions=positions, req_pool_indices=req, seq_lens=ext.long(), out_cache_loc=loc,
                        extend_seq_lens=ext, extend_start_loc=start)
    seq2 = ext.long() + 1
    dloc = torch.from_numpy(g[pfx + "decode_out_cache_loc"])
    ops.write_req_to_token(kv.req_to_token, req, ext.long(), seq2, torch.ones_like(seq2), dloc)
    l2 = ollama.forward(shape, w, kv, mode="decode", input_ids=torch.from_numpy(g[pfx + "next_ids"]),
                        positions=ops.clamp_position(seq2), req_pool_indices=req, seq_lens=seq2,
                        out_cache_loc=dloc)
    return l1, l2


def run_w64_smoke():
    """One D = 128 launch of the 4-wave x 64-row extend kernels - the ones whose register allocation
    the staged build (build.py compile_w64 + tools/w64_asm.py) allocates in descriptor and metadata - against oracle.ops.extend_attention: a toolchain that
    laid their registers out differently than the kernels' text assumes shows up HERE (the tiny model above has
    D = 64 and never launches them).  Both forms: one workgroup per item, and persistent with a plan."""
    from scratchpad_amd import _native
    from tests.helpers import attn_error_units, cpu, paged_problem
    assert _native.debug_get("w64_descriptor_patched") == 1, "the library was linked without the w64 descriptor patch"
    dtype, Hq, Hkv, D = torch.bfloat16, 8, 2, 128
    pre, ext = [0, 64, 5], [300, 200, 70]
    seq = [a + b for a, b in zip(pre, ext)]
    p = paged_problem(41, len(seq), Hq, Hkv, D, seq, dtype, "cuda")
    g = torch.Generator().manual_seed(42)
    q = torch.randn(sum(ext), Hq, D, generator=g).to(dtype).cuda()
    ext_t = torch.tensor(ext, dtype=torch.int32, device="cuda")
    start = torch.zeros(len(ext), dtype=torch.int32, device="cuda")
    start[1:] = torch.cumsum(ext_t[:-1], 0)
    ws = torch.empty(_native.extend_workspace_bytes(sum(ext), len(ext), Hq, D, dtype), dtype=torch.uint8, device="cuda")
    plan = _native.extend_plan(ext_t, p["seq_lens"], sum(ext), Hq, Hkv, True)
    c = cpu(p)
    fn = lambda v: ops.extend_attention(q.cpu().float(), c["k_buffer"].float(), v, c["req_to_token"], c["req_pool_indices"],
                                        c["seq_lens"], ext_t.cpu(), start.cpu(), D ** -0.5)
    ref, aref = fn(c["v_buffer"].float()), fn(c["v_buffer"].float().abs())
    got = {}
    try:
        for form, persist, use_plan in ((2, 0, None), (3, 2, plan)):
            _native.debug_set("extend_w64", 2)
            _native.debug_set("extend_w64_persist", persist)
            o = torch.full_like(q, float("nan"))
            _native.extend_attention(o, q, p["k_buffer"], p["v_buffer"], p["req_to_token"], p["req_pool_indices"],
                                     p["seq_lens"], ext_t, start, D ** -0.5, 0.0, True, max(ext), max(seq), ws, plan=use_plan)
            torch.cuda.synchronize()
            assert _native.debug_get("extend_last_kernel") == form, (
                f"expected {_native.EXTEND_KERNELS[form]}, the library launched "
                f"{_native.EXTEND_KERNELS[_native.debug_get('extend_last_kernel')]}")
            got[form] = attn_error_units(o, ref, aref, dtype)
            assert got[form] <= 1.0, f"{_native.EXTEND_KERNELS[form]}: {got[form]:.3f} units of bf16 round-off"
    finally:
        _native.debug_set("extend_w64", 1)
        _native.debug_set("extend_w64_persist", 1)
    print(f"smoke w64 ok: extend_w64_kernel {got[2]:.3f}, extend_w64p_kernel {got[3]:.3f} units of bf16 round-off vs the "
          f"oracle (bound 1.0; {sum(ext)} tokens, Hq {Hq} / Hkv {Hkv} / D {D}, descriptor patch in place)")


def run_smoke():
    torch.manual_seed(0)
    prefill, decode, nxt, gp, gd, gn, mr = run_case("a", torch.float32)
    e1 = float((prefill - gp).abs().max() / gp.abs().max())
    e2 = float((decode - gd).abs().max() / gd.abs().max())
    assert e1 < 1e-3 and e2 < 1e-3, (e1, e2)
    assert torch.equal(nxt, gn), "greedy tokens differ from the reference"
    # same step through HIP-graph replay, bf16
    p16, d16, *_ = run_case("a", torch.bfloat16, graph_bs=[4])
    o1, o2 = oracle_logits("a", torch.bfloat16)
    r1 = float((p16 - o1).abs().max() / o1.abs().max())
    r2 = float((d16 - o2).abs().max() / o2.abs().max())
    assert r1 < 3e-2 and r2 < 3e-2, (r1, r2)
    print(f"smoke ok: fp32 rel logit dev prefill {e1:.2e} decode {e2:.2e}; "
          f"bf16+graph vs bf16 oracle {r1:.2e} {r2:.2e}")
    run_w64_smoke()
