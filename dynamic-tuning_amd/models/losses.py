"""``models.losses.AdaLoss`` of the reference (models/losses.py:15-84): same constructor signature, same attributes, same return
value.  Host-side glue on the [B,C] logits and the [B,12,196,1] mask tensor the model returns (the module-API path); the fused step
evaluates the same loss inside libdyt_hip (csrc/rowops.hip: loss_rows_kernel / loss_final_kernel)."""
import torch.nn as nn


class AdaLoss(nn.Module):
    # the token-ratio terms the DyT path uses; the layer_* arguments are accepted and unused, as in the reference (models/losses.py:19-43)
    _TOKEN_FIELDS = ("token_target_ratio", "token_loss_ratio", "token_minimal", "token_minimal_weight")

    def __init__(self, base_criterion, layer_target_ratio=0.5, layer_loss_ratio=2., layer_diverse_ratio=0.1,
                 layer_entropy_weight=0.1, layer_minimal_weight=0., layer_minimal=0., token_target_ratio=0.5,
                 token_loss_ratio=2., token_minimal=0.1, token_minimal_weight=1.):
        super().__init__()
        self.base_criterion = base_criterion
        for name, value in zip(self._TOKEN_FIELDS, (token_target_ratio, token_loss_ratio, token_minimal, token_minimal_weight)):
            setattr(self, name, value)

    def forward(self, outputs, y):
        """(loss, {"base_loss", "token_loss"}) with loss = criterion(prediction, y) + token_loss_ratio * token term (reference :47-60)."""
        logits, mask = outputs["prediction"], outputs["token_select"]
        base = self.base_criterion(logits, y)
        scaled = self.token_loss_ratio * self._get_token_loss(logits, mask)
        return base + scaled, {"base_loss": base, "token_loss": scaled}

    def _get_token_loss(self, logits, mask):
        """(mean kept fraction over all blocks, images and tokens - target)^2, plus -- when weighted -- the per-(image, block) shortfall below
        ``token_minimal`` summed (reference :62-82)."""
        if mask is None:
            return logits.new_zeros(1).mean()
        ratio_term = ((mask.mean() - self.token_target_ratio) ** 2).mean()
        if not self.token_minimal_weight > 0:
            return ratio_term + self.token_minimal_weight * 0
        shortfall = (self.token_minimal - mask.mean(-1)).clamp(min=0.).sum()
        return ratio_term + self.token_minimal_weight * shortfall
