"""Prefix-cache parity: replay the op trace recorded from the reference's RadixCache
(tests/golden/gen_golden.py::gen_radix_cache) through scratchpad_amd.radix_cache on CPU tensors.
Every returned slot list, prefix length, size counter and the allocator's free list must match
exactly (integer work: bit-exact)."""
import json
import os
import random
from types import SimpleNamespace

import pytest
import torch

from scratchpad_amd.pool import ReqToTokenPool, TokenToKVPoolAllocator
from scratchpad_amd.radix_cache import ChunkCache, RadixCache, common_prefix_len

HERE = os.path.dirname(os.path.abspath(__file__))


def _snap(cache, alloc):
    return {"evictable": cache.evictable_size(), "protected": cache.protected_size(),
            "total": cache.total_size(), "available": alloc.available_size(),
            "values_sorted": sorted(int(x) for x in cache.all_values_flatten().tolist())}


def test_replay_reference_trace():
    with open(os.path.join(HERE, "golden", "radix_cache.json")) as f:
        trace = json.load(f)
    r2t = ReqToTokenPool(trace["max_reqs"], trace["context_len"], "cpu")
    alloc = TokenToKVPoolAllocator(trace["pool_size"], torch.float32, "cpu", None)
    cache = RadixCache(r2t, alloc, page_size=1)
    live = {}
    for i, op in enumerate(trace["ops"]):
        kind = op["op"]
        where = f"op {i} ({kind})"
        if kind == "insert":
            slots = alloc.alloc(len(op["key"]))
            assert slots.tolist() == op["slots"], where
            n = cache.insert(op["key"], slots.clone())
            assert n == op["ret"], where
            alloc.free(slots[:n])
        elif kind == "match":
            val, node = cache.match_prefix(op["key"])
            assert val.tolist() == op["value"], where
            assert list(node.key) == op["node_key"], where
        elif kind == "req_begin":
            ids = op["ids"]
            req = SimpleNamespace(rid=op["rid"], origin_input_ids=ids, output_ids=[],
                                  fill_ids=ids[:op["fill_len"]], prefix_indices=[], last_node=None,
                                  req_pool_idx=None)
            prefix, node = cache.match_prefix(req.fill_ids[:max(len(req.fill_ids) - 1, 0)])
            assert prefix.tolist() == op["prefix"], where
            need = len(req.fill_ids) - len(prefix)
            if alloc.available_size() < need:
                cache.evict(need)
            req.prefix_indices, req.last_node = prefix, node
            cache.inc_lock_ref(node)
            req.req_pool_idx = r2t.alloc(1)[0]
            assert req.req_pool_idx == op["req_pool_idx"], where
            new = alloc.alloc(need)
            assert new.tolist() == op["new_slots"], where
            r2t.write((req.req_pool_idx, slice(0, len(prefix))), prefix.to(torch.int32))
            r2t.write((req.req_pool_idx, slice(len(prefix), len(req.fill_ids))), new.to(torch.int32))
            live[req.rid] = req
        elif kind == "req_chunk":
            req = live[op["rid"]]
            cache.cache_unfinished_req(req)
            assert req.prefix_indices.tolist() == op["prefix_after"], where
            old, grow = len(req.fill_ids), op["grow"]
            assert grow > 0, "the recorded trace never ran out of slots mid-request"
            if alloc.available_size() < grow:
                cache.evict(grow)
            if grow:
                req.fill_ids = req.origin_input_ids[:old + grow]
                new = alloc.alloc(grow)
                assert new.tolist() == op["new_slots"], where
                r2t.write((req.req_pool_idx, slice(old, old + grow)), new.to(torch.int32))
            assert r2t.req_to_token[req.req_pool_idx, :old + grow].tolist() == op["row"], where
        elif kind == "req_finish":
            req = live.pop(op["rid"])
            outs = op["output_ids"]
            if alloc.available_size() < len(outs):
                cache.evict(len(outs))
            base = len(req.origin_input_ids)
            if len(outs) > 1:
                dec = alloc.alloc(len(outs) - 1)
                assert dec.tolist() == op["decode_slots"], where
                r2t.write((req.req_pool_idx, slice(base, base + len(outs) - 1)), dec.to(torch.int32))
            req.output_ids = outs
            cache.cache_finished_req(req)
        elif kind == "evict":
            cache.evict(op["n"])
            assert alloc.free_slots.tolist() == op["free_slots"], where
        assert _snap(cache, alloc) == op["after"], where


def test_common_prefix_len_matches_scalar_loop():
    rnd = random.Random(7)
    for _ in range(500):
        a = [rnd.randrange(3) for _ in range(rnd.randrange(0, 40))]
        b = list(a[:rnd.randrange(0, len(a) + 1)]) + [rnd.randrange(3) for _ in range(rnd.randrange(0, 10))]
        want = 0
        for x, y in zip(a, b):
            if x != y:
                break
            want += 1
        assert common_prefix_len(tuple(a), tuple(b)) == want


def test_lock_protects_from_eviction_and_sizes_balance():
    alloc = TokenToKVPoolAllocator(64, torch.float32, "cpu", None)
    cache = RadixCache(None, alloc)
    a = alloc.alloc(6)
    assert cache.insert([1, 2, 3, 4, 5, 6], a) == 0
    b = alloc.alloc(5)
    assert cache.insert([1, 2, 3, 9, 9], b) == 3      # splits [1,2,3] | [4,5,6]
    alloc.free(b[:3])
    val, node = cache.match_prefix([1, 2, 3, 4, 5, 6, 7])
    assert val.tolist() == a.tolist()
    assert cache.inc_lock_ref(node) == -6
    assert (cache.evictable_size(), cache.protected_size()) == (2, 6)
    cache.evict(100)                                   # only the unlocked [9,9] leaf can go
    assert cache.total_size() == 6 and cache.evictable_size() == 0
    assert cache.dec_lock_ref(node) == 6
    cache.evict(100)
    assert cache.total_size() == 0 and alloc.available_size() == 64
    assert sorted(alloc.free_slots.tolist()) == list(range(1, 65))


def test_disabled_cache_and_empty_key():
    alloc = TokenToKVPoolAllocator(8, torch.float32, "cpu", None)
    cache = RadixCache(None, alloc, disable=True)
    assert cache.insert([1, 2], alloc.alloc(2)) == 0
    val, node = cache.match_prefix([1, 2])
    assert val.numel() == 0 and node is cache.root_node
    cache = RadixCache(None, alloc)
    val, node = cache.match_prefix([])
    assert val.numel() == 0 and node is cache.root_node
    with pytest.raises(NotImplementedError):
        RadixCache(None, alloc, page_size=16)


def test_chunk_cache_keeps_only_the_requests_own_slots():
    r2t = ReqToTokenPool(4, 16, "cpu")
    alloc = TokenToKVPoolAllocator(32, torch.float32, "cpu", None)
    cache = ChunkCache(r2t, alloc)
    req = SimpleNamespace(rid="a", origin_input_ids=[5, 6, 7, 8], output_ids=[], fill_ids=[5, 6],
                          prefix_indices=[], last_node=None, req_pool_idx=r2t.alloc(1)[0])
    assert cache.match_prefix(rid="a", key=[5, 6]) == ([], None)
    s = alloc.alloc(2)
    r2t.write((req.req_pool_idx, slice(0, 2)), s.to(torch.int32))
    cache.cache_unfinished_req(req)
    assert req.prefix_indices.tolist() == s.tolist()
    val, entry = cache.match_prefix(rid="a", key=[5])
    assert val.tolist() == s[:1].tolist() and entry is req.last_node
    s2 = alloc.alloc(2)
    r2t.write((req.req_pool_idx, slice(2, 4)), s2.to(torch.int32))
    req.fill_ids = [5, 6, 7, 8]
    req.output_ids = [9]
    cache.cache_finished_req(req)
    assert alloc.available_size() == 32 and r2t.available_size() == 4
    assert cache.evictable_size() == 0 and cache.entries == {}


def test_random_traces_against_a_brute_force_prefix_model():
    """200 seeded operations per trace, 20 traces: the tree's answer must equal a brute-force model
    (a list of every inserted (key, slots) pair): the longest common prefix over all stored keys,
    with the slots of whichever entry owns that prefix first (first writer wins, as insert() only
    adds the unseen tail).  Size counters and allocator conservation hold throughout."""
    for seed in range(20):
        rnd = random.Random(seed)
        alloc = TokenToKVPoolAllocator(4000, torch.float32, "cpu", None)
        cache = RadixCache(None, alloc)
        owner = {}            # prefix tuple -> slot of its last token (first writer)
        locked = []
        for step in range(200):
            op = rnd.random()
            key = [rnd.randrange(4) for _ in range(rnd.randrange(1, 12))]
            if op < 0.45:
                slots = alloc.alloc(len(key))
                if slots is None:
                    continue
                n = cache.insert(key, slots.clone())
                alloc.free(slots[:n])
                want_n = 0
                while want_n < len(key) and tuple(key[:want_n + 1]) in owner:
                    want_n += 1
                assert n == want_n, (seed, step)
                for i in range(n, len(key)):
                    owner[tuple(key[:i + 1])] = int(slots[i])
            elif op < 0.85:
                val, node = cache.match_prefix(key)
                want = []
                for i in range(len(key)):
                    slot = owner.get(tuple(key[:i + 1]))
                    if slot is None:
                        break
                    want.append(slot)
                assert val.tolist() == want, (seed, step)
                if want and rnd.random() < 0.3:
                    cache.inc_lock_ref(node)
                    locked.append(node)
            elif op < 0.93 and locked:
                cache.dec_lock_ref(locked.pop(rnd.randrange(len(locked))))
            else:
                before = set(cache.all_values_flatten().tolist())
                cache.evict(rnd.randrange(1, 30))
                after = set(cache.all_values_flatten().tolist())
                gone = before - after
                owner = {k: v for k, v in owner.items() if v not in gone}
                # a locked path is never evicted
                for node in locked:
                    n = node
                    while n is not cache.root_node:
                        assert set(n.value.tolist()) <= after
                        n = n.parent
            assert cache.total_size() == len(owner) == cache.evictable_size() + cache.protected_size()
            assert alloc.available_size() + cache.total_size() == 4000
