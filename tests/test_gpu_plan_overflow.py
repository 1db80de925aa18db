"""An understated ``seq_lens_sum`` is reported, not computed around.

The decode split plan and its partial workspace are sized from the HOST's bound on sum(seq_lens)
(``ForwardBatch.seq_lens_sum``; the reference's producer is ScheduleBatch.prepare_for_decode,
scheduler/schedule_batch.py:1230-1308, which keeps it equal to the device-side lengths).  If a caller breaks that
contract the plan kernel lists only the items that fit and the affected rows are wrong: the plan's word 2 carries the
number of items the lengths needed, HipAttnBackend reads it back (at once with ``strict_plan_check``, else one plan
later / in ``check_plans()``) and raises."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _runner(items=True):
    """items=True: the backend plans the (request, split) items as it does for a model with a soft-cap layer or a shape
    the range kernel refuses, and its launches are sent to them (sp_debug_set("decode_ranges", 0) in the tests below) -
    the geometry whose slots the host's bound sizes.  items=False: the backend as it is built for Llama (round 6): plans
    are the range section alone."""
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs
    cfg = ModelConfig(256, 512, 2, 4, 2, 512, context_len=2048, max_position_embeddings=2048)
    args = ServerArgs(max_total_tokens=4096, max_running_requests=8, disable_cuda_graph=True)
    mr = ModelRunner(cfg, args, dtype=torch.bfloat16, seed=5)
    assert mr.attn_backend.decode_ranges > 0 and mr.attn_backend.plan_items is False, "a Llama shape plans no items"
    mr.attn_backend.plan_items = items
    return mr


@pytest.fixture
def item_geometry():
    from scratchpad_amd import _native
    _native.debug_set("decode_ranges", 0)
    yield
    _native.debug_set("decode_ranges", -1)


def _decode_batch(mr, lens, claimed_sum):
    from scratchpad_amd.forward_info import ForwardMode, ModelWorkerBatch
    dev = mr.device
    bs = len(lens)
    table = mr.req_to_token_pool.req_to_token
    g = torch.Generator().manual_seed(1)
    perm = (torch.randperm(4000, generator=g) + 1).to(torch.int32)
    off, loc = 0, []
    for b, n in enumerate(lens):
        table[b, :n] = perm[off:off + n].to(dev)
        loc.append(int(perm[off + n - 1]))
        off += n
    return ModelWorkerBatch(bid=1, forward_mode=ForwardMode.DECODE,
                            input_ids=torch.randint(0, 512, (bs,), generator=g).to(dev),
                            req_pool_indices=torch.arange(bs, device=dev), seq_lens=torch.tensor(lens, device=dev),
                            out_cache_loc=torch.tensor(loc, device=dev), seq_lens_sum=claimed_sum)


def test_understated_seq_lens_sum_raises_instead_of_returning_wrong_logits(item_geometry):
    from scratchpad_amd.model_runner import TpModelWorker
    mr = _runner()
    worker = TpModelWorker(mr)
    backend = mr.attn_backend
    lens = [600, 600, 600, 600]
    # the honest step: no error now or later
    worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))
    backend.check_plans()
    # sum claimed as 400: the launch is sized for 400 / 64 + 4 = 10 items, the (clamped) lengths need 4 x 7
    bad = _decode_batch(mr, lens, 400)
    worker.forward_batch_generation(bad)                 # deferred mode: the step itself does not synchronise ...
    with pytest.raises(RuntimeError, match="split plan overflow.*seq_lens_sum"):
        backend.check_plans()                            # ... the check does
    backend.check_plans()                                # reported once
    # ... and without an explicit check the NEXT step's plan build reports it (after a sync point, e.g. sampling)
    worker.forward_batch_generation(bad)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="split plan overflow"):
        worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))
    backend.check_plans()
    # strict mode: the step that would be wrong raises before any layer runs
    backend.strict_plan_check = True
    with pytest.raises(RuntimeError, match="split plan overflow"):
        worker.forward_batch_generation(bad)
    worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))


def test_overlap_worker_reports_the_overflow_with_the_same_steps_results(item_geometry):
    """The overlap worker (tp_worker_client.py) synchronises on a step's results one step later anyway: the step's plan
    headers, copied out ahead of its forward on the same stream, have landed by then, so a cut plan raises when THAT
    step's token ids are resolved - before they reach the scheduler - and not with the following step."""
    from scratchpad_amd.tp_worker_client import TpModelWorkerClient
    mr = _runner()
    client = TpModelWorkerClient(mr)
    try:
        lens = [600, 600, 600, 600]
        client.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))
        _, ids = client.resolve_last_batch_result()
        assert len(ids) == 4
        client.forward_batch_generation(_decode_batch(mr, lens, 400))          # understated: 10 items for 28
        with pytest.raises(RuntimeError, match="split plan overflow.*seq_lens_sum"):
            client.resolve_last_batch_result()
        client.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))   # the engine goes on; reported once
        _, ids = client.resolve_last_batch_result()
        assert len(ids) == 4
    finally:
        client.close()


def test_overlap_worker_reports_step_n_even_when_step_n_plus_1_is_already_enqueued(item_geometry):
    """under real overlap scheduling step N + 1 is enqueued BEFORE step N is resolved.  The forward thread,
    building step N + 1's plans, used to find step N's landed header, raise inside the thread and end it: resolve(N)
    returned N's ids with no error and every later call failed with "forward thread failed".  Plans are tagged with their
    step now; the forward thread never raises them and resolve(N) checks exactly step N's: the bad step is reported
    with its own results, the good one behind it is not, and the engine goes on."""
    from scratchpad_amd.tp_worker_client import TpModelWorkerClient
    mr = _runner()
    client = TpModelWorkerClient(mr)
    try:
        lens = [600, 600, 600, 600]
        good = lambda: _decode_batch(mr, lens, sum(lens))
        client.forward_batch_generation(good())                               # step 1
        client.forward_batch_generation(_decode_batch(mr, lens, 400))         # step 2: understated - enqueued before 1 resolves
        _, ids = client.resolve_last_batch_result()                           # step 1: fine
        assert len(ids) == 4
        client.forward_batch_generation(good())                               # step 3 enqueued before 2 resolves
        torch.cuda.synchronize()                                              # (every header has landed: the old failure window)
        with pytest.raises(RuntimeError, match="split plan overflow.*seq_lens_sum"):
            client.resolve_last_batch_result()                                # step 2: reported with its own results
        client.forward_batch_generation(good())                               # step 4: the forward thread is alive
        _, ids = client.resolve_last_batch_result()                           # step 3: not blamed for step 2
        assert len(ids) == 4
        _, ids = client.resolve_last_batch_result()                           # step 4
        assert len(ids) == 4 and client.error is None
    finally:
        client.close()


def test_range_plans_do_not_size_anything_from_the_hosts_sum():
    """The backend as it is built for Llama / Mllama since round 6: plans carry the range section alone, whose slots
    (batch size + pieces) and cuts come from the device-side lengths.  A seq_lens_sum understated so far that the item
    geometry would drop splits (2000 claimed for 4 x 600: 35 slots for 40 items - the report of the tests above) changes
    NOTHING here: the same bits as the honest step, nothing to report, no header copy per step.  (What the host's sum
    still bounds in an eager step is the longest request, max_len = min(context, sum - (bs - 1)): lengths are clamped to
    it, as on every path; 1997 >= 600 here.)"""
    from scratchpad_amd import _native
    from scratchpad_amd.model_runner import TpModelWorker
    mr = _runner(items=False)
    worker = TpModelWorker(mr)
    lens = [600, 600, 600, 600]
    honest, _ = worker.forward_batch_generation(_decode_batch(mr, lens, sum(lens)))
    assert _native.debug_get("decode_last_kernel") == 3, "the range kernel ran"
    honest = honest.next_token_logits.clone()
    assert _native.decode_plan_slots(4, 1997, 64, 2000) < 4 * 10, "the item geometry would have overflowed"
    mr.attn_backend.strict_plan_check = True
    bad, _ = worker.forward_batch_generation(_decode_batch(mr, lens, 2000))
    assert torch.equal(bad.next_token_logits, honest)
    assert not mr.attn_backend._plan_checks, "no item section: nothing is watched"
    mr.attn_backend.check_plans()
