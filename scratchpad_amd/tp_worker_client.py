"""Overlap worker: the forward pass runs on its own thread and HIP stream while the scheduler
prepares the next batch.

Mirrors managers/tp_worker_client.py:43-228: a forward thread pinned to ``forward_stream``, an
input/output queue pair, and the *future token id* protocol - ``forward_batch_generation`` returns
negative placeholder ids at once; the forward thread writes the sampled ids into
``future_token_ids_map`` and resolves placeholders found in the next batch's ``input_ids``
(``resolve_future_token_ids``, 34-40) before running it.  Every kernel of the hot path launches on
``torch.cuda.current_stream()``, which is ``forward_stream`` inside the thread.
"""
import threading
from queue import Queue
from typing import Optional

import torch

from .forward_info import ModelWorkerBatch
from .model_runner import ModelRunner, TpModelWorker


def resolve_future_token_ids(input_ids: torch.Tensor, future_token_ids_map: torch.Tensor) -> None:
    """tp_worker_client.py:34-40 (in place): ids < 0 are placeholders -k for map[k]."""
    input_ids[:] = torch.where(input_ids < 0, future_token_ids_map[torch.clamp(-input_ids, min=0)],
                               input_ids)


class TpModelWorkerClient:
    def __init__(self, model_runner: ModelRunner):
        self.worker = TpModelWorker(model_runner)
        self.max_running_requests = model_runner.max_running_requests
        self.device = model_runner.device
        self.future_token_ids_ct = 0
        self.future_token_ids_limit = self.max_running_requests * 3
        self.future_token_ids_map = torch.empty((self.max_running_requests * 5,), dtype=torch.int64,
                                                device=self.device)
        self.input_queue: Queue = Queue()
        self.output_queue: Queue = Queue()
        self.forward_stream = torch.cuda.Stream(device=self.device)
        self.scheduler_stream = torch.cuda.current_stream(self.device)
        self.error: Optional[BaseException] = None
        self.step = 0                  # steps enqueued so far: tags a step's decode plans (HipAttnBackend.plan_step)
        self.forward_thread = threading.Thread(target=self.forward_thread_func, daemon=True)
        self.forward_thread.start()

    @property
    def model_runner(self):
        return self.worker.model_runner

    # tp_worker_client.py:86-108: the scheduler's start-up accessors go through to the worker
    def get_worker_info(self):
        return self.worker.get_worker_info()

    def get_pad_input_ids_func(self):
        return self.worker.get_pad_input_ids_func()

    def get_tp_cpu_group(self):
        return self.worker.get_tp_cpu_group()

    def get_memory_pool(self):
        return self.worker.get_memory_pool()

    def forward_thread_func(self):
        try:
            with torch.cuda.stream(self.forward_stream):
                self.forward_thread_func_()
        except BaseException as e:  # the reference SIGQUITs its parent (110-116); we surface it
            self.error = e
            self.output_queue.put((None, None, None, None))

    @torch.inference_mode()
    def forward_thread_func_(self):
        """The forward thread's loop (protocol of tp_worker_client.py:118-168): take a batch, patch the
        placeholder ids the scheduler put into it, run it, publish this batch's sampled ids for the NEXT
        batch's placeholders, and hand the results back without waiting for the GPU."""
        import collections
        in_flight = collections.deque(maxlen=2)      # the GPU may still be reading the previous batch's tensors
        for batch, first_slot, step in iter(self.input_queue.get, (None, None, None)):
            in_flight.append(batch)
            self.output_queue.put(self._run_one(batch, first_slot, step))

    def _run_one(self, batch: ModelWorkerBatch, first_slot: int, step: int):
        resolve_future_token_ids(batch.input_ids, self.future_token_ids_map)
        # the step's decode plans are tagged with it and checked by the scheduler's thread when THIS step is resolved;
        # the forward thread itself never raises a plan overflow (it may already be running step N + 1 when step N's
        # header lands: raising here ended the thread and step N's ids went out unreported - ADVICE r5)
        backend = getattr(self.model_runner, "attn_backend", None)
        if hasattr(backend, "plan_step"):
            backend.plan_step = step
        logits_output, ids = self.worker.forward_batch_generation(batch)
        n = len(batch.seq_lens)
        self.future_token_ids_map[first_slot + 1:first_slot + 1 + n] = ids       # placeholder -k reads map[k]
        host_ids = ids.to("cpu", non_blocking=True)
        done = torch.cuda.Event()
        done.record()                                 # after the D2H copy on forward_stream
        return done, logits_output, host_ids, step

    def resolve_last_batch_result(self, launch_done: Optional[threading.Event] = None):
        """tp_worker_client.py:170-190: wait for the previous batch's results (one step later)."""
        copy_done, logits_output, next_token_ids, step = self.output_queue.get()
        if copy_done is None:
            raise RuntimeError("forward thread failed") from self.error
        if launch_done is not None:
            launch_done.wait()
        copy_done.synchronize()
        # the step's split-plan headers were copied out ahead of its forward on the same stream: they have landed.  A
        # plan that was cut short (the scheduler understated seq_lens_sum) raises HERE, in the scheduler's thread, before
        # this step's token ids are handed out - exactly this step's plans: the forward thread may
        # already have built step N + 1's, which are reported with step N + 1
        backend = getattr(self.model_runner, "attn_backend", None)
        if hasattr(backend, "check_plans"):
            backend.check_plans(wait=True, upto=step)
        return logits_output, next_token_ids.tolist()

    def forward_batch_generation(self, model_worker_batch: ModelWorkerBatch):
        """Enqueue the batch and return placeholder ids (-ct-1 .. -ct-bs) at once (192-220)."""
        if self.error is not None:
            raise RuntimeError("forward thread failed") from self.error
        # the scheduler's writes (req_to_token, allocator slices, input ids) must be visible to
        # the forward stream before it reads them (203-204)
        self.scheduler_stream.synchronize()
        self.step += 1
        self.input_queue.put((model_worker_batch, self.future_token_ids_ct, self.step))
        bs = len(model_worker_batch.seq_lens)
        future_next_token_ids = torch.arange(-(self.future_token_ids_ct + 1),
                                             -(self.future_token_ids_ct + 1 + bs), -1,
                                             dtype=torch.int64, device=self.device)
        self.future_token_ids_ct = (self.future_token_ids_ct + bs) % self.future_token_ids_limit
        return None, future_next_token_ids

    def close(self):
        self.input_queue.put((None, None, None))
        self.forward_thread.join(timeout=30)
        backend = getattr(self.model_runner, "attn_backend", None)
        if hasattr(backend, "plan_step"):
            backend.plan_step = None          # the runner's later (non-overlapped) steps check their plans themselves
