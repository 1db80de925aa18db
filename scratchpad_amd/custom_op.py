"""CustomOp seam - mirrors model_executor/custom_op.py:5-28.

The reference dispatches every op to ``forward_cuda`` unconditionally (custom_op.py:25-28); here
the one device path is ``forward_hip`` (``forward_cuda`` is kept as an alias so reference-style
call sites keep working).  There is deliberately no ``forward_native``: an op called without the
HIP library or with host tensors raises."""
import torch.nn as nn


class CustomOp(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self._forward_method = self.dispatch_forward()

    def forward(self, *args, **kwargs):
        return self._forward_method(*args, **kwargs)

    def forward_hip(self, *args, **kwargs):
        raise NotImplementedError

    def forward_cuda(self, *args, **kwargs):
        return self.forward_hip(*args, **kwargs)

    def dispatch_forward(self):
        return self.forward_hip
