"""Call-form compatibility of the host seams with the reference (north_star: "keeping the ForwardBatch /
ScheduleBatch operator API so it drops in under the existing scheduler and managers").

``tests/golden/api_surface.json`` is DATA recorded from the reference in the build container
(``gen_golden.py api_surface``): dataclass field names / order / defaults, the parameter lists of the seam
classes' public methods, and the call forms (positional count + keyword names) the reference's own callers use -
scheduler/scheduler.py:803, 932, 981, 1000, 1711-1720 and the rest.  Here every one of them has to BIND on
``scratchpad_amd``'s classes.  Rules:

* a reference parameter keeps its name and its position; one that has a default upstream has one here;
* extra parameters are allowed only behind the reference's and only with defaults;
* a reference method that is absent here is on ``ABSENT`` with the reason, and nothing else may be missing.
"""
import dataclasses
import enum
import inspect
import json
import os

import pytest

import scratchpad_amd.attention as attention
import scratchpad_amd.custom_op as custom_op
import scratchpad_amd.distributed as distributed
import scratchpad_amd.forward_info as forward_info
import scratchpad_amd.model_runner as model_runner
import scratchpad_amd.pool as pool
import scratchpad_amd.radix_cache as radix_cache
import scratchpad_amd.sampler as sampler
import scratchpad_amd.schedule_batch as schedule_batch
import scratchpad_amd.tp_worker_client as tp_worker_client

SURFACE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_surface.json")))

OURS = {
    "AttentionBackend": attention.AttentionBackend, "RadixAttention": attention.RadixAttention,
    "CustomOp": custom_op.CustomOp, "KVCache": pool.KVCache, "MHATokenToKVPool": pool.MHATokenToKVPool,
    "ReqToTokenPool": pool.ReqToTokenPool, "TokenToKVPoolAllocator": pool.TokenToKVPoolAllocator,
    "ScheduleBatch": schedule_batch.ScheduleBatch, "Req": schedule_batch.Req,
    "ForwardBatch": forward_info.ForwardBatch, "ForwardMode": forward_info.ForwardMode,
    "CaptureHiddenMode": forward_info.CaptureHiddenMode, "ModelWorkerBatch": forward_info.ModelWorkerBatch,
    "RadixCache": radix_cache.RadixCache, "ChunkCache": radix_cache.ChunkCache,
    "GroupCoordinator": distributed.GroupCoordinator,
    "TpModelWorker": model_runner.TpModelWorker, "TpModelWorkerClient": tp_worker_client.TpModelWorkerClient,
}

# reference methods deliberately NOT mirrored, each with its reason (anything else missing fails the test)
ABSENT = {
    ("CustomOp", "forward_native"): "no CPU / torch fallback in the product by design (DESIGN section 1): an op without the "
                                    "HIP library raises; the oracle/ restatement is test infrastructure",
    ("ScheduleBatch", "alloc_paged_token_slots_extend"): "page_size > 1 is refused upstream too (model_runner.py:431-432)",
    ("ScheduleBatch", "alloc_paged_token_slots_decode"): "page_size > 1, as above",
    ("ScheduleBatch", "new_page_count_next_decode"): "page_size > 1, as above",
    ("Req", "init_incremental_detokenize"): "detokenizer side (control plane, out of scope: SURVEY section 8)",
}


def _params(fn):
    return [p for p in inspect.signature(fn).parameters.values() if p.name not in ("self", "cls")]


def _problems(ref_params, ours_params, what):
    """why a call written against `ref_params` might not bind on `ours_params` (empty list: compatible)"""
    out = []
    var_pos = any(p.kind is p.VAR_POSITIONAL for p in ours_params)
    var_kw = any(p.kind is p.VAR_KEYWORD for p in ours_params)
    ours_named = [p for p in ours_params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
    by_name = {p.name: (i, p) for i, p in enumerate(ours_named)}
    ref_names = set()
    for i, r in enumerate(ref_params):
        if r["kind"] in ("VAR_POSITIONAL", "VAR_KEYWORD"):
            if not (var_kw if r["kind"] == "VAR_KEYWORD" else var_pos):
                out.append(f"{what}: the reference takes {r['kind']} '{r['name']}', ours does not")
            continue
        ref_names.add(r["name"])
        if r["name"] not in by_name:
            if not (var_pos and var_kw):
                out.append(f"{what}: parameter '{r['name']}' is missing")
            continue
        j, p = by_name[r["name"]]
        if r["kind"] == "POSITIONAL_OR_KEYWORD":
            if p.kind is p.KEYWORD_ONLY:
                out.append(f"{what}: '{r['name']}' is keyword-only here, positional upstream")
            elif j != i:
                out.append(f"{what}: '{r['name']}' is positional #{j} here, #{i} upstream")
        if not r["required"] and p.default is p.empty:
            out.append(f"{what}: '{r['name']}' has a default upstream, none here")
    for p in ours_named:
        if p.name not in ref_names and p.default is p.empty:
            out.append(f"{what}: extra parameter '{p.name}' has no default")
    return out


def _ours_method(cls_name, name):
    member = inspect.getattr_static(OURS[cls_name], name, None)
    if member is None:
        return None, None
    if isinstance(member, (classmethod, staticmethod)):
        return member.__func__, type(member).__name__
    if isinstance(member, property):
        return member, "property"
    return member, "method"


def test_dataclass_fields_follow_the_reference():
    problems = []
    for name in ("ForwardBatch", "ModelWorkerBatch", "ScheduleBatch"):
        ours = dataclasses.fields(OURS[name])
        ours_params = [inspect.Parameter(f.name, inspect.Parameter.POSITIONAL_OR_KEYWORD,
                                         default=(inspect.Parameter.empty if f.default is dataclasses.MISSING
                                                  and f.default_factory is dataclasses.MISSING else None))
                       for f in ours if f.init]
        ref = [{"name": f["name"], "kind": "POSITIONAL_OR_KEYWORD", "required": f["required"]}
               for f in SURFACE["dataclasses"][name]]
        problems += _problems(ref, ours_params, name)
    assert not problems, "\n".join(problems)


def test_enum_values_follow_the_reference():
    for name, members in SURFACE["enums"].items():
        ours = {m.name: int(m.value) for m in OURS[name]}
        assert ours == members, (name, ours, members)


def test_every_reference_method_exists_with_a_compatible_signature():
    problems = []
    for cls_name, methods in SURFACE["methods"].items():
        for name, rec in methods.items():
            fn, kind = _ours_method(cls_name, name)
            if fn is None:
                if (cls_name, name) not in ABSENT:
                    problems.append(f"{cls_name}.{name}: absent")
                continue
            assert (cls_name, name) not in ABSENT, f"{cls_name}.{name} exists: take it off ABSENT"
            if name == "__init__" and issubclass(OURS[cls_name], enum.Enum):
                continue
            if rec["kind"] == "property" or kind == "property":
                if rec["kind"] != kind:
                    problems.append(f"{cls_name}.{name}: {kind} here, {rec['kind']} upstream")
                continue
            if name == "__init__" and dataclasses.is_dataclass(OURS[cls_name]) and cls_name in SURFACE["dataclasses"]:
                continue                          # compared field by field above
            if rec["kind"] != kind:
                problems.append(f"{cls_name}.{name}: {kind} here, {rec['kind']} upstream")
            problems += _problems(rec["params"], _params(fn), f"{cls_name}.{name}")
    assert not problems, "\n".join(problems)


def test_module_level_functions():
    problems = []
    for name, ref in SURFACE["functions"].items():
        fn = getattr(distributed, name, None)
        if fn is None:
            problems.append(f"distributed.{name}: absent")
            continue
        problems += _problems(ref, _params(fn), name)
    assert not problems, "\n".join(problems)


# ---- the reference's own call sites ------------------------------------------------------------------------------
# receiver name at the call site -> the seam class(es) the object is
RECEIVERS = {
    "batch": ["ScheduleBatch"], "last_batch": ["ScheduleBatch"], "running_batch": ["ScheduleBatch"],
    "new_batch": ["ScheduleBatch"], "idle_batch": ["ScheduleBatch"], "ScheduleBatch": ["ScheduleBatch"],
    "req": ["Req"], "chunked_req": ["Req"], "reqs[]": ["Req"],
    "req_to_token_pool": ["ReqToTokenPool"], "token_to_kv_pool_allocator": ["TokenToKVPoolAllocator"],
    "token_to_kv_pool": ["MHATokenToKVPool"], "tree_cache": ["RadixCache", "ChunkCache"],
    "attn_backend": ["HipAttnBackend"], "ForwardBatch": ["ForwardBatch"], "get_tp_group()": ["GroupCoordinator"],
    "sampling_info": ["SamplingBatchInfo"],
    # scheduler.py holds either worker class in self.tp_worker (183-200): every call has to bind on both
    "tp_worker": ["TpModelWorker", "TpModelWorkerClient"],
}
OURS_CALLABLE = dict(OURS, HipAttnBackend=attention.HipAttnBackend, SamplingBatchInfo=sampler.SamplingBatchInfo,
                     ModelRunner=model_runner.ModelRunner)
# `self.<method>(...)` inside a reference class body: the class is the file's
SELF_CLASSES = {"scheduler/schedule_batch.py": ["ScheduleBatch", "Req"], "memory/radix_cache.py": ["RadixCache"],
                "memory/chunk_cache.py": ["ChunkCache"], "nn/attention/triton_backend.py": ["HipAttnBackend"],
                "nn/attention/flashinfer_backend.py": ["HipAttnBackend"], "model_executor/forward_info.py": ["ForwardBatch"],
                "model_executor/model_runner.py": ["ModelRunner"]}
# the same receiver NAME is a different object in these files
RECEIVERS_BY_FILE = {("memory/chunk_cache.py", "token_to_kv_pool"): ["TokenToKVPoolAllocator"],
                     ("managers/tp_worker_client.py", "worker"): ["TpModelWorker"]}
# (the synchronous worker has no resolve_last_batch_result upstream either: scheduler.py:1080, 1126, 1437 run under enable_overlap)
ONLY_ON = {("tp_worker", "resolve_last_batch_result"): ["TpModelWorkerClient"]}
# receivers that are not seam objects (same method NAME on something out of scope)
NOT_SEAMS = {
    "copy": "the copy module", "grammar": "grammar objects (out of scope)", "grammar_cache": "grammar cache (out of scope)",
    "decode_wrapper": "flashinfer wrapper (third party)", "prefill_wrapper_paged": "flashinfer wrapper (third party)",
    "model": "nn.Module.forward of the model", "model_runner": "ModelRunner.forward (own seam, test_gpu_schedule_flow)",
    "mm_input": "MultimodalInputs (image processor side)", "spec_info": "speculative decoding (out of scope)",
    "distributed": "torch.distributed", "ps": "module alias",
}
# call sites that use a seam object for something the path does not have, with the reason
SKIPPED_SITES = {
    ("scheduler/scheduler.py", "token_to_kv_pool", "available_size"): "stale upstream: the scheduler's token_to_kv_pool is the "
                                                                      "KVCache, which has no available_size (the allocator has)",
    ("scheduler/scheduler.py", "token_to_kv_pool", "clear"): "stale upstream, as above (flush_cache)",
    ("scheduler/schedule_batch.py", "self", "alloc_paged_token_slots_extend"): "page_size > 1",
    ("scheduler/schedule_batch.py", "self", "alloc_paged_token_slots_decode"): "page_size > 1",
    ("scheduler/schedule_batch.py", "self", "new_page_count_next_decode"): "page_size > 1",
    ("scheduler/scheduler.py", "req", "init_incremental_detokenize"): "detokenizer side",
    ("model_executor/forward_info.py", "self", "contains_audio_inputs"): None,    # bound below like any other
    ("managers/tp_worker_client.py", "worker", "get_tp_group"): "upstream calls a method its own TpModelWorker does not define",
}


def _targets(form):
    recv, meth = form["receiver"], form["method"]
    if recv is None:
        if meth in SURFACE["functions"]:
            return [getattr(distributed, meth)]
        return [OURS_CALLABLE[meth]]                       # constructor
    if recv == "self":
        names = SELF_CLASSES.get(form["file"], [])
    else:
        names = ONLY_ON.get((recv, meth)) or RECEIVERS_BY_FILE.get((form["file"], recv), RECEIVERS.get(recv))
    if names is None:
        return None
    found = []
    for n in names:
        cls = OURS_CALLABLE[n]
        if inspect.getattr_static(cls, meth, None) is not None:
            found.append(getattr(cls, meth))
        elif not (recv == "self" and len(names) > 1):
            found.append((n, meth))                        # missing on a class that must have it
    return found


def test_every_reference_call_site_binds():
    problems, bound = [], 0
    for form in SURFACE["call_forms"]:
        recv, meth = form["receiver"], form["method"]
        if recv in NOT_SEAMS:
            continue
        if SKIPPED_SITES.get((form["file"], recv, meth)):
            continue
        where = f"{form['file']}:{form['line']} {recv}.{meth}" if recv else f"{form['file']}:{form['line']} {meth}"
        targets = _targets(form)
        if targets is None:
            problems.append(f"{where}: receiver not mapped to a seam class (extend RECEIVERS or NOT_SEAMS)")
            continue
        if not targets:
            problems.append(f"{where}: no class of this file defines the method")
        for t in targets:
            if isinstance(t, tuple):
                if t not in ABSENT:
                    problems.append(f"{where}: {t[0]}.{t[1]} is absent")
                continue
            sig = inspect.signature(t)
            params = list(sig.parameters.values())
            if params and params[0].name in ("self",) and not inspect.isclass(t):
                sig = sig.replace(parameters=params[1:])
            if form["star_args"] or form["star_kwargs"]:
                continue                                   # *args / **kwargs at the call site: nothing to pin
            try:
                sig.bind(*([None] * form["n_positional"]), **{k: None for k in form["keywords"]})
                bound += 1
            except TypeError as e:
                problems.append(f"{where}: does not bind on ours {sig}: {e}")
    assert not problems, "\n".join(problems)
    assert bound > 150, bound


def test_behaviour_of_the_drifted_call_forms():
    """the five call sites VERDICT r4 found raising, executed on host objects (no device work)."""
    import torch
    from types import SimpleNamespace
    r2t = pool.ReqToTokenPool(8, 16, "cpu", False)
    alloc = pool.TokenToKVPoolAllocator(64, torch.float32, "cpu", None)
    Req, ScheduleBatch = schedule_batch.Req, schedule_batch.ScheduleBatch
    reqs = [Req(str(i), "", [1, 2, 3 + i], sampler.SamplingParams(max_new_tokens=8)) for i in range(3)]
    b = ScheduleBatch.init_new(reqs, r2t, alloc, None, SimpleNamespace(is_encoder_decoder=False, vocab_size=32),
                               False, None, False)
    assert b.device == "cpu" and not b.is_empty() and b.batch_size() == 3 and not b.return_logprob
    # filter_batch(chunked_req_to_exclude=...) scheduler.py:803
    b.req_pool_indices = torch.arange(3)
    b.seq_lens = torch.tensor([3, 3, 3])
    b.output_ids = torch.tensor([7, 8, 9])
    b.filter_batch(chunked_req_to_exclude=reqs[1])
    assert [r.rid for r in b.reqs] == ["0", "2"] and b.output_ids.tolist() == [7, 9] and b.seq_lens_sum == 6
    b.filter_batch()                                       # nothing finished: unchanged
    assert b.batch_size() == 2
    reqs[0].finished_reason = schedule_batch.FINISH_LENGTH(length=8)
    b.filter_batch()
    assert [r.rid for r in b.reqs] == ["2"]
    # copy() scheduler.py:425
    c = b.copy()
    assert c.reqs is b.reqs and c.forward_mode == b.forward_mode and c.return_logprob == b.return_logprob
    # prepare_for_idle scheduler.py:1711-1720
    idle = ScheduleBatch.init_new([], r2t, alloc, None, SimpleNamespace(is_encoder_decoder=False, vocab_size=32),
                                  False, None, False)
    idle.prepare_for_idle()
    assert idle.forward_mode.is_idle() and idle.is_empty() and idle.input_ids.numel() == 0 and idle.extend_num_tokens == 0
    # alloc_token_slots(n, backup_state=True) schedule_batch.py:728
    before = alloc.available_size()
    out, state = b.alloc_token_slots(4, backup_state=True)
    assert out.numel() == 4 and state.numel() == before
    alloc.restore_state(state)
    assert alloc.available_size() == before
    # Req.check_finished schedule_batch.py:525
    r = Req("x", "", [1, 2], sampler.SamplingParams(max_new_tokens=2), eos_token_ids={5})
    r.output_ids = [9]
    r.check_finished()
    assert not r.finished()
    r.output_ids = [9, 9]
    r.check_finished()
    assert isinstance(r.finished_reason, schedule_batch.FINISH_LENGTH)
    r = Req("y", "", [1, 2], sampler.SamplingParams(max_new_tokens=9), eos_token_ids={5})
    r.output_ids = [5]
    r.check_finished()
    assert isinstance(r.finished_reason, schedule_batch.FINISH_MATCHED_TOKEN) and r.finished_reason.matched == 5
    r = Req("z", "", [1, 2], sampler.SamplingParams(max_new_tokens=9, ignore_eos=True), eos_token_ids={5})
    r.output_ids = [5]
    r.check_finished()
    assert not r.finished()
    r.to_abort = True
    r.check_finished()
    assert isinstance(r.finished_reason, schedule_batch.FINISH_ABORT)
    # ReqToTokenPool write variants memory/pool.py:58-63
    rec = pool.ReqToTokenPool(4, 8, "cpu", True)
    rec.write((0, slice(0, 2)), torch.tensor([5, 6], dtype=torch.int32))
    assert len(rec.get_write_records()) == 1
    rec.write_without_records((1, slice(0, 1)), torch.tensor([9], dtype=torch.int32))
    assert rec.get_write_records() == [] and rec.req_to_token[1, 0] == 9
    plain = pool.ReqToTokenPool(4, 8, "cpu", False)
    plain.write_with_records((0, slice(0, 1)), torch.tensor([3], dtype=torch.int32))
    assert len(plain.get_write_records()) == 1
