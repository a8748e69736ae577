"""``models.losses.AdaLoss`` of the reference (models/losses.py:15-84), same constructor and
return value.  Operates on the [B,C] logits and the [B,12,196,1] mask tensor the model returns
(host-side glue on tiny tensors for the module-API path; the fused step path evaluates the same
loss inside libdyt_hip's loss kernel, csrc/rowops.hip: loss_kernel)."""
import torch
import torch.nn as nn


class AdaLoss(nn.Module):
    def __init__(self, base_criterion, layer_target_ratio=0.5, layer_loss_ratio=2., layer_diverse_ratio=0.1,
                 layer_entropy_weight=0.1, layer_minimal_weight=0., layer_minimal=0., token_target_ratio=0.5,
                 token_loss_ratio=2., token_minimal=0.1, token_minimal_weight=1.):
        super().__init__()
        self.base_criterion = base_criterion
        self.token_target_ratio = token_target_ratio
        self.token_loss_ratio = token_loss_ratio
        self.token_minimal = token_minimal
        self.token_minimal_weight = token_minimal_weight

    def forward(self, outputs, y):
        x, token_select = outputs["prediction"], outputs["token_select"]
        base_loss = self.base_criterion(x, y)
        token_loss = self._get_token_loss(x, token_select)
        loss = base_loss + self.token_loss_ratio * token_loss
        return loss, dict(base_loss=base_loss, token_loss=self.token_loss_ratio * token_loss)

    def _get_token_loss(self, x, token_select):
        if token_select is None:
            return x.new_zeros(1).mean()
        token_flops_loss = ((token_select.mean() - self.token_target_ratio) ** 2).mean()
        if self.token_minimal_weight > 0:
            token_minimal_loss = (self.token_minimal - token_select.mean(-1)).clamp(min=0.).sum()
        else:
            token_minimal_loss = 0
        return token_flops_loss + self.token_minimal_weight * token_minimal_loss
