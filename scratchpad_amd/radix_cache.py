"""Prefix caches: the producers of ``Req.prefix_indices`` for the extend path.

Same public surface and observable behaviour as memory/radix_cache.py:66-420 (``RadixCache``),
memory/chunk_cache.py:16-83 (``ChunkCache``) and memory/base_prefix_cache.py:5-48, page_size = 1
(the only page size the reference's KV pool accepts, model_runner.py:431-432):

* ``match_prefix(key)`` returns the cached KV slots of the longest cached prefix of ``key`` and the
  tree node that ends it, splitting an edge when the match ends inside it (radix_cache.py:105-135,
  312-336);
* ``insert(key, value)`` returns how many leading tokens were already cached (137-144, 351-382);
* ``cache_finished_req`` / ``cache_unfinished_req`` hand the request's slots to the tree and free
  the duplicates (146-221);
* ``evict(n)`` frees least-recently-used unlocked leaves until ``n`` slots are back (231-254);
* ``inc_lock_ref`` / ``dec_lock_ref`` pin the path to the root and keep evictable/protected sizes
  (256-282).

Built differently from the reference: edges hold their tokens as tuples and edge matching is a
galloping slice comparison (C speed on long prompts) instead of a per-token Python loop, recency
is a logical clock (deterministic, no ``time.time()`` ties).
"""
import abc
import heapq
import itertools
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .pool import ReqToTokenPool, TokenToKVPoolAllocator


class BasePrefixCache(abc.ABC):
    """memory/base_prefix_cache.py:5-48."""

    @abc.abstractmethod
    def reset(self): ...

    @abc.abstractmethod
    def match_prefix(self, **kwargs): ...

    @abc.abstractmethod
    def insert(self, **kwargs): ...

    @abc.abstractmethod
    def cache_finished_req(self, **kwargs): ...

    @abc.abstractmethod
    def cache_unfinished_req(self, **kwargs): ...

    @abc.abstractmethod
    def evict(self, num_tokens: int, evict_callback=None): ...

    @abc.abstractmethod
    def inc_lock_ref(self, node): ...

    @abc.abstractmethod
    def dec_lock_ref(self, node): ...

    @abc.abstractmethod
    def evictable_size(self): ...

    def total_size(self):
        raise NotImplementedError()

    def pretty_print(self):
        raise NotImplementedError()


def common_prefix_len(a: Sequence[int], b: Sequence[int]) -> int:
    """Length of the longest common prefix (what _key_match_page_size_1, radix_cache.py:47-53,
    computes token by token).  Whole-slice equality first, then bisection on slices."""
    n = min(len(a), len(b))
    if a[:n] == b[:n]:
        return n
    lo, hi = 0, n            # a[:lo] == b[:lo] holds, a[:hi] == b[:hi] does not
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if a[lo:mid] == b[lo:mid]:
            lo = mid
        else:
            hi = mid
    return lo


def _prefix_len(req) -> int:
    return 0 if req.prefix_indices is None else len(req.prefix_indices)


class TreeNode:
    """One edge of the tree: ``key`` tokens and their KV slots ``value`` (radix_cache.py:15-44)."""
    __slots__ = ("children", "parent", "key", "value", "lock_ref", "last_access_time", "id")
    _ids = itertools.count()

    def __init__(self):
        self.children: Dict[int, "TreeNode"] = {}
        self.parent: Optional["TreeNode"] = None
        self.key: Tuple[int, ...] = ()
        self.value: Optional[torch.Tensor] = None
        self.lock_ref = 0
        self.last_access_time = 0
        self.id = next(TreeNode._ids)

    @property
    def evicted(self):
        return self.value is None

    def __lt__(self, other: "TreeNode"):
        return self.last_access_time < other.last_access_time


class RadixCache(BasePrefixCache):
    def __init__(self, req_to_token_pool: Optional[ReqToTokenPool],
                 token_to_kv_pool_allocator: Optional[TokenToKVPoolAllocator],
                 page_size: int = 1, disable: bool = False):
        if page_size != 1:
            raise NotImplementedError("page_size > 1 (reference: model_runner.py:431-432)")
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.page_size = page_size
        self.disable = disable
        self.device = (token_to_kv_pool_allocator.device if token_to_kv_pool_allocator is not None
                       else torch.device("cpu"))
        self.reset()

    # ---- public API -------------------------------------------------------------------------
    def reset(self):
        self._clock = itertools.count(1)
        self.root_node = TreeNode()
        self.root_node.value = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.root_node.lock_ref = 1
        self.evictable_size_ = 0
        self.protected_size_ = 0

    def match_prefix(self, key: List[int], **kwargs) -> Tuple[torch.Tensor, TreeNode]:
        empty = torch.empty((0,), dtype=torch.int64, device=self.device)
        if self.disable or len(key) == 0:
            return empty, self.root_node
        key = tuple(key)
        node = self.root_node
        self._touch(node)
        pieces = []
        pos = 0
        while pos < len(key):
            child = node.children.get(key[pos])
            if child is None:
                break
            self._touch(child)
            m = common_prefix_len(child.key, key[pos:])
            if m < len(child.key):
                node = self._split(child, m)
                pieces.append(node.value)
                break
            pieces.append(child.value)
            node = child
            pos += m
        return (torch.cat(pieces) if pieces else empty), node

    def insert(self, key: List[int], value=None) -> int:
        if self.disable:
            return 0
        if value is None:
            value = torch.tensor(list(key), dtype=torch.int64, device=self.device)
        key = tuple(key)
        node = self.root_node
        self._touch(node)
        pos = 0
        while pos < len(key):
            child = node.children.get(key[pos])
            if child is None:
                break
            self._touch(child)
            m = common_prefix_len(child.key, key[pos:])
            pos += m
            if m < len(child.key):
                node = self._split(child, m)   # the rest of `key` diverges below the cut
                break
            node = child
        if pos < len(key):
            leaf = TreeNode()
            leaf.parent = node
            leaf.key = key[pos:]
            leaf.value = value[pos:]
            self._touch(leaf)
            node.children[key[pos]] = leaf
            self.evictable_size_ += len(leaf.key)
        return pos

    def cache_finished_req(self, req):
        """radix_cache.py:146-178: all tokens but the last sampled one (its KV was never written)."""
        n_tok = len(req.origin_input_ids) + len(req.output_ids) - 1
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, :n_tok]
        if self.disable:
            self.token_to_kv_pool_allocator.free(kv_indices.to(torch.int64))
            self.req_to_token_pool.free(req.req_pool_idx)
            return
        token_ids = (req.origin_input_ids + req.output_ids)[:-1]
        kv_indices = kv_indices.to(torch.int64)
        new_prefix_len = self.insert(token_ids, kv_indices.clone())
        # slots in [old prefix, new prefix) duplicate ones the tree already owned
        self.token_to_kv_pool_allocator.free(kv_indices[_prefix_len(req):new_prefix_len])
        self.req_to_token_pool.free(req.req_pool_idx)
        self.dec_lock_ref(req.last_node)

    def cache_unfinished_req(self, req):
        """radix_cache.py:180-221: chunked prefill; the request continues on the tree's slots."""
        if self.disable:
            return
        token_ids = req.fill_ids
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, :len(token_ids)].to(torch.int64)
        new_prefix_len = self.insert(token_ids, kv_indices.clone())
        self.token_to_kv_pool_allocator.free(kv_indices[_prefix_len(req):new_prefix_len])
        new_indices, new_last_node = self.match_prefix(token_ids)
        self.req_to_token_pool.write(
            (req.req_pool_idx, slice(_prefix_len(req), len(new_indices))),
            new_indices[_prefix_len(req):].to(torch.int32))
        self.dec_lock_ref(req.last_node)
        self.inc_lock_ref(new_last_node)
        req.prefix_indices = new_indices
        req.last_node = new_last_node

    def evict(self, num_tokens: int, evict_callback=None):
        if self.disable:
            return
        heap = self._collect_leaves()
        heapq.heapify(heap)
        freed = 0
        while freed < num_tokens and heap:
            node = heapq.heappop(heap)
            if node is self.root_node:
                break
            if node.lock_ref > 0:
                continue
            self.token_to_kv_pool_allocator.free(node.value)
            freed += len(node.key)
            parent = node.parent
            del parent.children[node.key[0]]
            self.evictable_size_ -= len(node.key)
            if not parent.children:
                heapq.heappush(heap, parent)

    def inc_lock_ref(self, node: TreeNode) -> int:
        if self.disable:
            return 0
        delta = 0
        while node is not self.root_node:
            if node.lock_ref == 0:
                n = len(node.key)
                self.evictable_size_ -= n
                self.protected_size_ += n
                delta -= n
            node.lock_ref += 1
            node = node.parent
        return delta

    def dec_lock_ref(self, node: TreeNode) -> int:
        if self.disable:
            return 0
        delta = 0
        while node is not self.root_node:
            if node.lock_ref == 1:
                n = len(node.key)
                self.evictable_size_ += n
                self.protected_size_ -= n
                delta += n
            node.lock_ref -= 1
            node = node.parent
        return delta

    def evictable_size(self):
        return self.evictable_size_

    def protected_size(self):
        return self.protected_size_

    def total_size(self):
        total, stack = 0, [self.root_node]
        while stack:
            node = stack.pop()
            total += len(node.key)
            stack.extend(node.children.values())
        return total

    def all_values_flatten(self) -> torch.Tensor:
        values, stack = [], [self.root_node]
        while stack:
            node = stack.pop()
            for child in node.children.values():
                values.append(child.value)
                stack.append(child)
        return torch.cat(values) if values else torch.empty((0,), dtype=torch.int64, device=self.device)

    def pretty_print(self):
        stack = [(self.root_node, 0)]
        while stack:
            node, depth = stack.pop()
            print(" " * depth, len(node.key), list(node.key[:10]), f"r={node.lock_ref}")
            stack.extend((c, depth + 2) for c in node.children.values())
        print(f"#tokens: {self.total_size()}")

    # ---- internals --------------------------------------------------------------------------
    def _touch(self, node: TreeNode):
        node.last_access_time = next(self._clock)

    def _split(self, child: TreeNode, at: int) -> TreeNode:
        """Cut the edge into child.parent -> head(key[:at]) -> child(key[at:]); head inherits the
        lock count of the edge (radix_cache.py:338-349)."""
        head = TreeNode()
        head.parent = child.parent
        head.key, head.value = child.key[:at], child.value[:at]
        head.lock_ref = child.lock_ref
        head.last_access_time = next(self._clock)
        child.parent.children[head.key[0]] = head
        child.key, child.value = child.key[at:], child.value[at:]
        child.parent = head
        head.children[child.key[0]] = child
        return head

    def _collect_leaves(self) -> List[TreeNode]:
        leaves, stack = [], [self.root_node]
        while stack:
            node = stack.pop()
            if node.children:
                stack.extend(node.children.values())
            else:
                leaves.append(node)
        return leaves


class ChunkCacheEntry:
    def __init__(self, rid, value):
        self.rid = rid
        self.value = value


class ChunkCache(BasePrefixCache):
    """memory/chunk_cache.py:16-83: no sharing; only remembers a chunked request's own slots."""

    def __init__(self, req_to_token_pool: ReqToTokenPool, token_to_kv_pool: Optional[TokenToKVPoolAllocator] = None,
                 token_to_kv_pool_allocator: Optional[TokenToKVPoolAllocator] = None):
        # the slot allocator under either name: chunk_cache.py:17-24 declares `token_to_kv_pool`, the reference's
        # own scheduler passes `token_to_kv_pool_allocator=` (scheduler.py:342-345)
        self.disable = True
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator if token_to_kv_pool_allocator is not None else token_to_kv_pool
        if self.token_to_kv_pool_allocator is None:
            raise TypeError("ChunkCache needs the KV slot allocator (token_to_kv_pool / token_to_kv_pool_allocator)")
        self.reset()

    def reset(self):
        self.entries: Dict[str, ChunkCacheEntry] = {}

    def match_prefix(self, rid=None, key: List[int] = (), **kwargs):
        entry = self.entries.get(rid)
        if entry is None:
            return [], None
        return entry.value[:len(key)], entry

    def cache_finished_req(self, req, token_ids: Optional[List[int]] = None):
        n = (len(req.origin_input_ids) + len(req.output_ids) - 1) if token_ids is None else len(token_ids)
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, :n]
        self.req_to_token_pool.free(req.req_pool_idx)
        self.token_to_kv_pool_allocator.free(kv_indices.to(torch.int64))
        self.entries.pop(req.rid, None)

    def cache_unfinished_req(self, req, token_ids: Optional[List[int]] = None):
        n = len(req.fill_ids) if token_ids is None else len(token_ids)
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, :n].to(torch.int64)
        entry = self.entries.setdefault(req.rid, ChunkCacheEntry(req.rid, kv_indices))
        entry.value = kv_indices
        req.prefix_indices = kv_indices
        req.last_node = entry

    def insert(self, **kwargs):
        raise NotImplementedError()

    def evict(self, num_tokens: int, evict_callback=None):
        pass

    def inc_lock_ref(self, node):
        return 0

    def dec_lock_ref(self, node):
        return 0

    def evictable_size(self):
        return 0
