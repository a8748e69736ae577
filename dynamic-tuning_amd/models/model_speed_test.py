"""``models.model_speed_test`` of the reference on MI355X (reference models/model_speed_test.py:235-310,493-496,525-532): the
inference-only twin that the reference's ``speed.py`` (:36,224,261) times.

There the twin exists because only it evaluates the MLP on the gathered kept tokens (``Block.batch_forward`` :274-310,
``single_forward`` :243-271) with the deterministic gate (:27-37); here every eval-mode forward of
``models.vision_transformer_IN21K`` already is that compacted path (gate kernel -> kept-row index lists -> fc1 / fc2 on the
gathered rows -> scatter-add), so the twin is the same model with the twin's call contract: ``forward(x) -> logits`` only,
no gradient, no training mode.
"""
import torch

from .vision_transformer_IN21K import Attention, Block, Mlp, PatchEmbed, VisionTransformer as _DyTVisionTransformer  # noqa: F401
from .dynamic_adapter import Adapter, TokenSelect  # noqa: F401


class VisionTransformer(_DyTVisionTransformer):
    def train(self, mode=True):
        if mode:
            raise NotImplementedError("models.model_speed_test is the inference twin (deterministic gate, reference :27-37); "
                                      "train models.vision_transformer_IN21K")
        return super().train(False)

    @torch.no_grad()
    def forward(self, x):
        logits, _ = super().forward(x)
        return logits


def vit_base_patch16_224_in21k(**kwargs):
    """Reference models/model_speed_test.py:525-532."""
    model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, **kwargs)
    torch.nn.Module.train(model, False)
    return model
