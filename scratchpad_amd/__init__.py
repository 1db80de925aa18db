"""scratchpad_amd - MI355X-native batched prefill/decode attention hot path for Scratchpad.

Host-side mirror of the reference's operator seams (ForwardBatch / ModelWorkerBatch,
AttentionBackend, CustomOp, KVCache, GroupCoordinator) over hand-written HIP kernels reached
through a C ABI (``include/scratchpad_hip.h``).  There is no torch/CPU fallback for any op."""

__version__ = "0.1.0"
