"""`ganloss` on the HIP reduction kernel.

Same class name, constructor arguments and call signature as /root/reference/traiNNer/losses/gan_loss.py:13-126:
`GANLoss(loss_weight, gan_type="vanilla", real_label_val=1.0, fake_label_val=0.0)`, `forward(input, target_is_real, is_disc=False)`.
Like the reference, the returned value is NOT multiplied by `loss_weight` (the trainer applies it, sr_model.py:641,763-768), and
for `hinge` the generator side is `-mean(pred)` regardless of `target_is_real`.  `gan_type="r3gan"` is rewritten to R3GANLoss by the
reference's builder (losses/__init__.py:69-76), which is outside this path: it raises.  Pinned by tests/golden/pixel_gan_losses.pt.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import _lib as L
from .. import ops
from ..utils.registry import LOSS_REGISTRY


class _GanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gan_type, target_is_real, is_disc, real_label, fake_label):
        x = logits.contiguous().float()
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        ops.gan_loss(gan_type, x, target_is_real, is_disc, out, grad=grad, real_label=real_label, fake_label=fake_label)
        ctx.save_for_backward(grad)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


@LOSS_REGISTRY.register()
class GANLoss(nn.Module):
    def __init__(self, loss_weight: float, gan_type: str = "vanilla", real_label_val: float = 1.0, fake_label_val: float = 0.0) -> None:
        super().__init__()
        if gan_type not in L.GAN_TYPES:
            raise NotImplementedError(f"GAN type {gan_type} is not implemented.")
        self.gan_type = gan_type
        self.loss_weight = loss_weight
        self.real_label_val = real_label_val
        self.fake_label_val = fake_label_val

    def forward(self, input: Tensor, target_is_real: bool, is_disc: bool = False, **kwargs) -> Tensor:  # noqa: A002
        return _GanFn.apply(input, self.gan_type, bool(target_is_real), bool(is_disc), self.real_label_val, self.fake_label_val)

    def hip_value_and_grad(self, logits: Tensor, target_is_real: bool, is_disc: bool, *, value_weight: float, value_out: Tensor,
                           total_weight: float = 0.0, total_out: Tensor | None = None, grad_weight: float = 1.0,
                           grad: Tensor | None = None) -> None:
        """value_out[0] = value_weight * loss; total_out[0] += total_weight * loss; grad = grad_weight * dloss/dlogits."""
        ops.gan_loss(self.gan_type, logits, target_is_real, is_disc, value_out, value_weight=value_weight, total_out=total_out,
                     total_weight=total_weight, grad=grad, grad_weight=grad_weight, real_label=self.real_label_val,
                     fake_label=self.fake_label_val)
