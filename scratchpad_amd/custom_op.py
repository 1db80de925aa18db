"""CustomOp seam (model_executor/custom_op.py:5-28): an op is a module whose ``forward`` runs
whichever implementation ``_forward_method`` points at.

The reference always binds ``forward_cuda`` and lets the graph runner re-point the attribute
(cuda_graph_runner.py:25-42).  Here the single device implementation is ``forward_hip``;
``forward_cuda`` resolves to it so reference-style call sites and overrides keep working, and
there is deliberately no ``forward_native``: without the HIP library, or on host tensors, an op
raises instead of falling back."""
from torch import nn

_DEVICE_IMPL = "forward_hip"


class CustomOp(nn.Module):
    def __init__(self, *_args, **_kwargs):
        super().__init__()
        self.bind(_DEVICE_IMPL)

    def bind(self, impl: str) -> None:
        """Point ``forward`` at the method called ``impl``."""
        self._forward_method = getattr(self, impl)

    def dispatch_forward(self):
        """The reference's hook name for choosing the implementation."""
        return getattr(self, _DEVICE_IMPL)

    def forward_hip(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} has no HIP implementation")

    def forward_cuda(self, *args, **kwargs):
        return self.forward_hip(*args, **kwargs)

    def forward(self, *args, **kwargs):
        return self._forward_method(*args, **kwargs)
