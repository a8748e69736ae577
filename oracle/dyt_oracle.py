"""CPU oracle for the DyT ViT-B/16 fine-tune hot path -- TEST INFRASTRUCTURE ONLY.

A torch-fp32 restatement of the reference's algorithm for the path BASELINE.json's
``north_star`` names.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product (``dynamic-tuning_amd/``) never
does and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (behind stand-ins for the absent third-party packages timm/easydict)
in the build container, runs its model, its ``AdaLoss`` and its own
``engine_finetune.train_one_epoch`` / ``evaluate`` on seeded inputs with recorded Gumbel
noise and dropout masks, and commits the results as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those vectors.
The reference itself holds no tests or golden vectors for this path.

Third-party arithmetic restated here because it is not under /root/reference:
timm==0.9.12 ``PatchEmbed`` (Conv2d k=s=16 + flatten + transpose), ``Mlp``
(fc1 -> exact-erf GELU -> fc2) and ``DropPath`` (per-sample Bernoulli / keep, ``drop_path_scales``;
pinned by tests/golden/drop_path_step.npz = the reference model stepped with drop_path_rate 0.3),
torch ``LayerNorm`` / ``scaled_dot_product_attention`` / ``AdamW``.  Every function cites the reference file:line it follows (paths relative to
/root/reference).

Floating point: everything is fp32 on CPU, exactly like the reference's CPU path
(``torch.cuda.amp.autocast`` is a no-op there).
"""
import math

import torch
import torch.nn.functional as F

DEPTH, DIM, HEADS, HEAD_DIM, NTOK = 12, 768, 12, 64, 197
LN_EPS = 1e-6  # models/vision_transformer_IN21K.py:262


def patch_embed(sd, x):
    """timm PatchEmbed as used at models/vision_transformer_IN21K.py:272-278,344."""
    y = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
    return y.flatten(2).transpose(1, 2)


def embed(sd, x):
    """cls/pos prologue, models/vision_transformer_IN21K.py:344-352 (all dropouts are p=0)."""
    t = patch_embed(sd, x)
    t = torch.cat((sd["cls_token"].expand(t.shape[0], -1, -1), t), dim=1)
    return t + sd["pos_embed"]


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def attention(sd, p, x):
    """Attention.forward, models/vision_transformer_IN21K.py:54-75 (explicit-softmax branch
    :66-70; the fused SDPA branch :60-64 computes the same function)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, HEADS, HEAD_DIM).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q = q * (HEAD_DIM ** -0.5)
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])


def gumbel_sigmoid(logits, g1, g2, tau=5.0, threshold=0.5, training=True):
    """_gumbel_sigmoid, models/dynamic_adapter.py:25-54, with the two Gumbel draws
    (:30-39) passed in instead of drawn.  Returns (straight-through select, y_soft)."""
    if training:
        y_soft = ((logits + g1 - g2) / tau).sigmoid()  # :41-42
    else:
        y_soft = logits.sigmoid()  # :44
    y_hard = torch.zeros_like(logits).masked_fill(y_soft > threshold, 1.0)  # :47-50
    return y_hard - y_soft.detach() + y_soft, y_soft  # :51


def token_select(sd, p, x, g1, g2, training, tau=5.0, threshold=0.5):
    """TokenSelect.forward, models/dynamic_adapter.py:70-77."""
    logits = F.linear(x[:, 1:, :], sd[p + "mlp_token_select.mlp_head.weight"],
                      sd[p + "mlp_token_select.mlp_head.bias"])  # :72
    sel, _ = gumbel_sigmoid(logits, g1, g2, tau, threshold, training)
    sel = torch.cat([sel.new_ones(x.shape[0], 1, 1), sel], dim=1)  # :75
    return sel, logits


ADAPTER_LN_KEY = "oracle.adapter_layernorm_option"   # int tensor in the state dict handed to the oracle: 1 = "in", 2 = "out" (absent: "none")


def adapter(sd, p, x, scale, keep_mask=None, drop_p=0.1):
    """Adapter.forward with add_residual=False, models/dynamic_adapter.py:120-140.  ``keep_mask`` (0/1, same shape as the bottleneck
    activation) replaces the Bernoulli draw of F.dropout (:127); None = eval / p=0.  Layernorm option "none" unless the state dict
    carries ``ADAPTER_LN_KEY`` and the block's ``adaptmlp.adapter_layer_norm_before.{weight,bias}`` (nn.LayerNorm(768), default eps 1e-5,
    :95-98): "in" normalises the adapter's input (:121-122), "out" its scaled output (:132-133)."""
    ln = int(sd[ADAPTER_LN_KEY]) if ADAPTER_LN_KEY in sd else 0
    lw, lb = sd.get(p + "adaptmlp.adapter_layer_norm_before.weight"), sd.get(p + "adaptmlp.adapter_layer_norm_before.bias")
    if ln == 1:
        x = F.layer_norm(x, (x.shape[-1],), lw, lb, 1e-5)
    down = F.relu(F.linear(x, sd[p + "adaptmlp.down_proj.weight"], sd[p + "adaptmlp.down_proj.bias"]))
    if keep_mask is not None:
        down = down * keep_mask.to(down.dtype) * (1.0 / (1.0 - drop_p))
    up = F.linear(down, sd[p + "adaptmlp.up_proj.weight"], sd[p + "adaptmlp.up_proj.bias"])
    if p + "adaptmlp.scale" in sd:   # adapter_scalar == "learnable_scalar": nn.Parameter(torch.ones(1)), :101-102 (trainable: an "adaptmlp." tensor)
        scale = sd[p + "adaptmlp.scale"]
    up = up * scale  # :130
    if ln == 2:
        up = F.layer_norm(up, (up.shape[-1],), lw, lb, 1e-5)
    return up


def mlp(sd, p, x):
    """timm Mlp as used at models/vision_transformer_IN21K.py:124-129,159: fc2(gelu_erf(fc1 x))."""
    h = F.gelu(F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def drop_path_scales(keep_draws, rate, depth=DEPTH):
    """timm==0.9.12 ``drop_path`` (timm/layers/drop.py; not under /root/reference) as the reference's blocks use it: block i holds
    DropPath(dpr[i]) on each branch with dpr = linspace(0, rate, depth) (models/vision_transformer_IN21K.py:285,121,131; Identity where
    dpr[i] == 0), and in training mode multiplies the branch by ``bernoulli(keep) / keep`` drawn per SAMPLE (shape [B,1,1], scale_by_keep).
    ``keep_draws``: [2, depth, B] uniforms in [0,1) standing for the Bernoulli draws (branch 0 = attention, 1 = MLP) -> factors [2, depth, B]."""
    dpr = torch.linspace(0, rate, depth)
    keep = (1.0 - dpr).reshape(1, depth, 1)
    s = (keep_draws < keep).float() / keep
    s[:, dpr == 0] = 1.0
    return s


def block(sd, i, x, g1, g2, keep_mask, scale, complete_model, training, mode="masked",
          tau=5.0, threshold=0.5, drop_p=0.1, count_flops_tokens=0, dp1=None, dp2=None):
    """Block.forward, models/vision_transformer_IN21K.py:144-165.

    dp1 / dp2: stochastic-depth factors [B] of this block's attention / MLP branch (drop_path1 :148, drop_path2 :159; None = Identity).

    mode="masked":   the reference's training semantics -- MLP on every token, multiplied by
                     the straight-through mask (:159-162).
    mode="compact":  forward value identical, but the MLP output of dropped tokens is treated
                     as never computed, so the gate gradient <dL/dx', h> exists only for kept
                     tokens ("compact" training mode).
    mode="gather":   real gather / scatter like models/model_speed_test.py:274-310
                     (forward-only use; same values as "masked").
    """
    p = "blocks.%d." % i
    att = attention(sd, p, layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]))
    if dp1 is not None:
        att = att * dp1.reshape(-1, 1, 1)
    x = x + att  # :148
    sel, logits = token_select(sd, p, x, g1, g2, training, tau, threshold)  # :150-152
    adapt = adapter(sd, p, x, scale, keep_mask, drop_p)  # :157
    if count_flops_tokens:  # Block.forward_count_flops :167-185: MLP on the first n tokens, gate result unused
        n = count_flops_tokens
        out = x + adapt
        out[:, :n, :] = out[:, :n, :] + mlp(sd, p, layer_norm(x[:, :n, :], sd[p + "norm2.weight"], sd[p + "norm2.bias"]))
        return out, sel, logits
    if complete_model or mode != "gather":
        h = mlp(sd, p, layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"]))  # :159
        if dp2 is not None:
            h = h * dp2.reshape(-1, 1, 1)
        if not complete_model:
            if mode == "compact":
                h = h * sel.detach()
            h = sel * h  # :161-162
    else:
        B, N, C = x.shape
        flat = x.reshape(B * N, C)
        idx = sel.reshape(-1).nonzero()[:, 0]  # model_speed_test.py:300
        hk = mlp(sd, p, layer_norm(flat[idx], sd[p + "norm2.weight"], sd[p + "norm2.bias"]))
        assert dp2 is None
        h = torch.zeros_like(flat)
        h[idx] = hk  # :302-304
        h = h.reshape(B, N, C)
    return x + h + adapt, sel, logits  # :163


def attentive_pool(sd, t, frames):
    """Pooling tail of the video model: VisionTransformer.forward
    (video_models/video_vision_transformer_IN21K.py:463-483) -> AttentiveBlock.forward (:44-51) ->
    CrossAttention.forward (:79-110).  ``t``: final-norm tokens [b*frames, 197, 768] -> [b, 768]."""
    bt, n, c = t.shape
    b = bt // frames
    x = t.reshape(b, frames * n, c)                                   # "(b t) tokens c -> b (t tokens) c"  :476
    p = "attentive_blocks."
    xq = layer_norm(sd["query_token"].expand(b, -1, -1), sd[p + "norm_q.weight"], sd[p + "norm_q.bias"])  # :45
    xk = layer_norm(x, sd[p + "norm_k.weight"], sd[p + "norm_k.bias"])   # :46
    xv = layer_norm(x, sd[p + "norm_v.weight"], sd[p + "norm_v.bias"])   # :47
    ca = p + "cross_attn."
    q = F.linear(xq, sd[ca + "q.weight"], sd[ca + "q_bias"])            # :92  (k_bias is zeros, :89)
    k = F.linear(xk, sd[ca + "k.weight"])                               # :95
    v = F.linear(xv, sd[ca + "v.weight"], sd[ca + "v_bias"])            # :98
    q = q.reshape(b, 1, HEADS, HEAD_DIM).transpose(1, 2) * (HEAD_DIM ** -0.5)   # :93,:101
    k = k.reshape(b, frames * n, HEADS, HEAD_DIM).transpose(1, 2)
    v = v.reshape(b, frames * n, HEADS, HEAD_DIM).transpose(1, 2)
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)                        # :102-104
    o = (a @ v).transpose(1, 2).reshape(b, 1, c)                         # :107
    o = F.linear(o, sd[ca + "proj.weight"], sd[ca + "proj.bias"])        # :108
    return o[:, 0, :]                                                    # :479


def forward(sd, x, g1=None, g2=None, keep_masks=None, scale=0.1, complete_model=False,
            training=True, mode="masked", tau=5.0, threshold=0.5, drop_p=0.1, depth=DEPTH,
            return_blocks=False, frames=1, count_flops_tokens=0, drop_scales=None):
    """VisionTransformer.forward, models/vision_transformer_IN21K.py:343-385.

    drop_scales: [2, depth, B] stochastic-depth factors of this pass (drop_path_scales(); training only) or None.

    g1, g2: [depth, B, 196] Gumbel draws (None in eval); keep_masks: [depth, B*197, r]
    uint8/bool adapter-dropout keep masks or None.
    Returns logits [B,C] and {"token_select","token_logits"} each [B,depth,196,1].

    frames > 1: the video model (video_models/video_vision_transformer_IN21K.py:435-483): ``x`` is the
    clip tensor already folded to [(b t),3,224,224] ("b c t h w -> (b t) c h w", :437); the trunk is
    identical, the head is the attentive pooling over the t*197 final-norm tokens of each clip."""
    B = x.shape[0]
    t = embed(sd, x)
    sels, logs, xs = [], [], [t]
    for i in range(depth):
        a = g1[i].reshape(B, NTOK - 1, 1) if (training and g1 is not None) else 0.0
        b = g2[i].reshape(B, NTOK - 1, 1) if (training and g2 is not None) else 0.0
        km = None
        if training and keep_masks is not None:
            km = keep_masks[i].reshape(B, NTOK, -1)
        ds = drop_scales if (training and drop_scales is not None and i > 0) else None
        t, sel, lg = block(sd, i, t, a, b, km, scale, complete_model, training, mode,
                           tau, threshold, drop_p, count_flops_tokens,
                           None if ds is None else ds[0, i], None if ds is None else ds[1, i])
        sels.append(sel)
        logs.append(lg)
        xs.append(t)
    ts = torch.stack(sels, dim=1)[:, :, 1:, :]  # :367 (drop the cls column)
    tl = torch.stack(logs, dim=1)  # :368
    t = layer_norm(t, sd["norm.weight"], sd["norm.bias"])  # :370
    if frames > 1:
        logits = F.linear(attentive_pool(sd, t, frames), sd["head.weight"], sd["head.bias"])  # video :480
    else:
        logits = F.linear(t[:, 0], sd["head.weight"], sd["head.bias"])  # :375-380 (cls pooling)
    out = dict(token_select=ts, token_logits=tl)
    if return_blocks:
        out["blocks"] = xs
    return logits, out


def ada_loss(logits, token_sel, y, token_target_ratio=0.5, token_loss_ratio=2.0,
             token_minimal=0.0, token_minimal_weight=0.0):
    """AdaLoss.forward / _get_token_loss, models/losses.py:48-84."""
    base = F.cross_entropy(logits, y)
    tok = ((token_sel.mean() - token_target_ratio) ** 2).mean()  # :69-72
    if token_minimal_weight > 0:
        tok = tok + token_minimal_weight * (token_minimal - token_sel.mean(-1)).clamp(min=0.).sum()  # :74-78
    return base + token_loss_ratio * tok, dict(base_loss=base, token_loss=token_loss_ratio * tok)


def step_loss(sd, x, y, g1, g2, keep_masks, scale=0.1, mode="masked", token_target_ratio=0.5,
              token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0, depth=DEPTH,
              drop_p=0.1, frames=1, drop_scales=None):
    """Loss of one fine-tune step, engine_finetune.py:47-65: student + teacher forward,
    (drop_scales: [2 passes, 2, depth, B] stochastic-depth factors, each forward call of the reference draws its own; None = drop_path 0)
    KL(student || teacher.detach()), teacher CE, AdaLoss(student)  (video: the identical body of
    train_video_one_epoch, engine_finetune.py:138-155, with frames > 1).

    g1/g2: [2, depth, B, 196] (pass 0 = student, 1 = teacher; the teacher pass also draws
    gate noise although its mask is discarded, :152,161); keep_masks [2, depth, B*197, r]."""
    km = (None, None) if keep_masks is None else (keep_masks[0], keep_masks[1])
    dps = (None, None) if drop_scales is None else (drop_scales[0], drop_scales[1])
    out_s, tok = forward(sd, x, g1[0], g2[0], km[0], scale, False, True, mode, depth=depth, drop_p=drop_p, frames=frames, drop_scales=dps[0])
    out_t, _ = forward(sd, x, g1[1], g2[1], km[1], scale, True, True, mode, depth=depth, drop_p=drop_p, frames=frames, drop_scales=dps[1])
    kl = F.kl_div(F.log_softmax(out_s, dim=-1), F.log_softmax(out_t.detach(), dim=-1),
                  reduction="batchmean", log_target=True)  # :52-57
    teacher = F.cross_entropy(out_t, y)  # :60
    loss, d = ada_loss(out_s, tok["token_select"], y, token_target_ratio, token_loss_ratio,
                       token_minimal, token_minimal_weight)  # :61-62
    loss = loss + teacher + kl  # :63
    d = dict(d, teacher_loss=teacher, distillation_loss=kl, loss=loss)
    return loss, d, (out_s, out_t, tok)


def trainable_names(sd):
    """Freeze rule, main_image.py:250-256."""
    return [k for k in sd if ("adaptmlp." in k) or ("mlp_token_select." in k) or k.startswith("head.") or
            k == "query_token" or k.startswith("attentive_blocks.")]   # video: the pooling head is not in the checkpoint


def step_grads(sd, x, y, g1, g2, keep_masks, **kw):
    """Loss components and gradients of every trainable tensor for one step."""
    names = trainable_names(sd)
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in names else v.detach()) for k, v in sd.items()}
    loss, d, outs = step_loss(leaf, x, y, g1, g2, keep_masks, **kw)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names])
    return {k: v.detach() for k, v in d.items()}, dict(zip(names, grads)), outs


def step_grads_chunked(sd, x, y, g1, g2, keep_masks, chunk, frames=1, scale=0.1, mode="masked", token_target_ratio=0.5,
                       token_loss_ratio=2.0, depth=DEPTH, drop_p=0.1):
    """step_grads() for batches whose autograd graph does not fit the host (configs[4]: 16 clips x 8 frames), exact, in
    chunks of ``chunk`` samples (clips when frames > 1).  Samples interact only through batch means (engine_finetune.py:52-63:
    CE / KL are means over the samples, models/losses.py:69-72 the mean over ALL gate decisions), so
      dL/dtheta = sum_chunks d/dtheta [ sum_{samples in chunk}(CE_s + CE_t + KL) / n  +  c * sum_{chunk} token_select ],
      c = token_loss_ratio * 2 (mean(token_select) - target) / numel(token_select),
    with the global mean taken from a first gradient-free student pass.  (No minimal-token term.)  Pinned to step_grads by
    tests/test_oracle_golden.py::test_chunked_step_grads_equal_step_grads."""
    names = trainable_names(sd)
    n = y.shape[0]
    B = x.shape[0]
    assert B == n * frames and n % chunk == 0
    fr = chunk * frames

    def sl(c, per):   # rows of chunk c in a tensor with `per` rows per sample-frame
        return slice(c * fr * per, (c + 1) * fr * per)

    def km(p, c):
        return None if keep_masks is None else keep_masks[p][:, sl(c, NTOK)]

    kept, total = 0.0, 0
    with torch.no_grad():
        for c in range(n // chunk):
            _, tok = forward(sd, x[sl(c, 1)], g1[0][:, sl(c, 1)], g2[0][:, sl(c, 1)], km(0, c), scale, False, True, mode,
                             depth=depth, drop_p=drop_p, frames=frames)
            kept += float(tok["token_select"].double().sum())
            total += tok["token_select"].numel()
    mean = kept / total
    coef = token_loss_ratio * 2.0 * (mean - token_target_ratio) / total
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in names else v.detach()) for k, v in sd.items()}
    grads = [torch.zeros_like(sd[k]) for k in names]
    ce_s = ce_t = klsum = 0.0
    outs_s, outs_t, sels = [], [], []
    for c in range(n // chunk):
        xs, ys = x[sl(c, 1)], y[c * chunk:(c + 1) * chunk]
        out_s, tok = forward(leaf, xs, g1[0][:, sl(c, 1)], g2[0][:, sl(c, 1)], km(0, c), scale, False, True, mode,
                             depth=depth, drop_p=drop_p, frames=frames)
        out_t, _ = forward(leaf, xs, g1[1][:, sl(c, 1)], g2[1][:, sl(c, 1)], km(1, c), scale, True, True, mode,
                           depth=depth, drop_p=drop_p, frames=frames)
        a = F.cross_entropy(out_s, ys, reduction="sum")
        b = F.cross_entropy(out_t, ys, reduction="sum")
        k = F.kl_div(F.log_softmax(out_s, dim=-1), F.log_softmax(out_t.detach(), dim=-1), reduction="sum", log_target=True)
        sur = (a + b + k) / n + coef * tok["token_select"].sum()
        for acc, g in zip(grads, torch.autograd.grad(sur, [leaf[kk] for kk in names])):
            acc += g
        ce_s += float(a.detach()); ce_t += float(b.detach()); klsum += float(k.detach())
        outs_s.append(out_s.detach()); outs_t.append(out_t.detach()); sels.append(tok["token_select"].detach())
    tokl = token_loss_ratio * (mean - token_target_ratio) ** 2
    d = dict(base_loss=ce_s / n, token_loss=tokl, teacher_loss=ce_t / n, distillation_loss=klsum / n,
             loss=ce_s / n + tokl + ce_t / n + klsum / n)
    return ({k: torch.tensor(v) for k, v in d.items()}, dict(zip(names, grads)),
            (torch.cat(outs_s), torch.cat(outs_t), dict(token_select=torch.cat(sels))))


def adamw_update(p, g, m, v, step, lr, wd=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (decoupled decay on every trainable tensor,
    main_image.py:285).  ``step`` is 1-based.  Returns new (p, m, v)."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def lr_at(epoch_float, lr, min_lr, warmup_epochs, epochs):
    """util/lr_sched.py:9-21 (per-iteration warm-up + half cosine)."""
    if epoch_float < warmup_epochs:
        return lr * epoch_float / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch_float - warmup_epochs) / (epochs - warmup_epochs)))


def accuracy(output, target, topk=(1,)):
    """util/metrics.py:4-11."""
    maxk = min(max(topk), output.size()[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / target.size(0) for k in topk]


def train_step(sd, opt_state, x, y, g1, g2, keep_masks, lr, wd=0.01, **kw):
    """One full step on CPU: losses, grads, AdamW on the trainable tensors (in place in ``sd``).
    ``opt_state``: dict name -> (m, v), plus "step".  Used as bench.py's cpu_baseline."""
    d, grads, _ = step_grads(sd, x, y, g1, g2, keep_masks, **kw)
    opt_state["step"] = opt_state.get("step", 0) + 1
    for k, g in grads.items():
        m, v = opt_state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
        sd[k], m, v = adamw_update(sd[k], g, m, v, opt_state["step"], lr, wd)
        opt_state[k] = (m, v)
    return d
