"""FLOPs accounting of the reference (block_flops_dict.py:33-83,209-227) without fvcore.

The reference measures, with fvcore on a CUDA device, the multiply-accumulates of one DyT block
as a function of how many tokens go through the MLP (``Block.forward_count_flops``,
models/vision_transformer_IN21K.py:167-185) and adds a fixed base for patch-embed + head.  Here the
same table is computed in closed form (dense contractions only, MACs, in units of 1e9 like the
reference's "GFlops"); ``select_flops`` / ``batch_select_flops`` keep the reference's signatures.
"""
import torch

N, D, H, HD, DM = 197, 768, 12, 64, 3072
BASE_FLOPS_IN21K = 0.116438784  # reference block_flops_dict.py:223 (patch-embed + head, 100 classes)


def block_gmacs(mlp_tokens, ffn_num=64):
    """GMACs of one block with the MLP evaluated on ``mlp_tokens`` of the 197 tokens."""
    attn = N * D * 3 * D + 2 * H * N * N * HD + N * D * D      # qkv + QK^T + AV + proj
    gate = (N - 1) * D                                          # TokenSelect Linear(768,1) on the patch tokens
    adapter = 2 * N * D * ffn_num                               # down + up on all tokens
    mlp = mlp_tokens * 2 * D * DM                               # fc1 + fc2 on the kept tokens
    return (attn + gate + adapter + mlp) / 1e9


def get_block_flops(args=None, ffn_num=None):
    """Table indexed by the number of MLP tokens 0..197 (index 0 unused), block_flops_dict.py:33-55."""
    r = ffn_num if ffn_num is not None else (args.tuning_config.ffn_num if args is not None else 64)
    return torch.tensor([0.0] + [block_gmacs(t, r) for t in range(1, N + 1)])


def get_base_flops(args=None):
    return BASE_FLOPS_IN21K


def select_flops(flops_dict, token_select, block_num, base_flops=0.33):
    """block_flops_dict.py:57-71: token_select [layers, tokens] of {0,1} for one image."""
    t = token_select.shape[1]
    ada_t = token_select.shape[0]
    counts = [t] * (block_num - ada_t) + token_select.sum(-1).int().tolist()
    flops = base_flops
    for c in counts:
        flops += float(flops_dict[c + 1])  # + cls token
    return flops


def batch_select_flops(bs, flops_dict, token_select, block_num=12, base_flops=0.116):
    """block_flops_dict.py:73-83: token_select [N, layers, tokens, 1]."""
    token_select = token_select.squeeze(-1)
    return torch.tensor([select_flops(flops_dict, t, block_num, base_flops) for t in token_select])
