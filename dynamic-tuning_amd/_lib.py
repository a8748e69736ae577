"""ctypes binding of libdyt_hip.so (C ABI: include/dyt_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load this
module raises, and every model forward / step goes through it.  PyTorch is used only for
device memory, streams and torch.distributed.
"""
import ctypes
import os

# dyt_step_fwd_bwd runs its two passes on two streams; with DDP-style training there are also the gradient-sum stream, the
# all-reduce stream and RCCL's.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share one are serialised (measured: 30.6 vs 25.9 ms/step in the 1-rank RCCL path with 4 vs 8 queues).  Effective only if this
# module is imported before the HIP runtime initialises (first torch.cuda call); launch scripts should export it themselves.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.environ.get("DYT_LIB_DIR", _HERE)   # development knob: A/B against another build of the two libraries
LIB_PATH = os.path.join(_LIB_DIR, "libdyt_hip.so")
LIB_PATH_F16 = os.path.join(_LIB_DIR, "libdyt_hip_f16.so")   # the same sources with IEEE-half operands (precision "fp16")

PREC_FP32, PREC_BF16 = 0, 1
PREC_FP16 = 2   # host-side name only: libdyt_hip_f16.so with its 16-bit mode (DYT_PREC_BF16 = 1 inside that library)
PREC_FP16X3F = 4  # host-side name only: PREC_FP16X3 with the gradient products as the hi * hi term alone (DYT_OPT_F32_SPLIT16 = 2)
PREC_FP16X3H = 5  # host-side name only: the PREC_FP16X3 forward bit for bit, the backward pass on 16-bit operands with the fp16 mode's kernels (DYT_OPT_F32_SPLIT16 = 3)
PREC_FP16F8 = 6   # host-side name only: PREC_FP16X3H with the forward GEMMs' correction products on the fp8 matrix cores (DYT_OPT_F32_SPLIT16 = 4)
PREC_FP16X3Q = 7  # host-side name only: PREC_FP16X3H with the attention branch's GEMMs (qkv, proj) in the fp8-correction form, the MLP three-part (DYT_OPT_F32_SPLIT16 = 5)
PREC_FP16X3 = 3   # host-side name only: libdyt_hip_f16.so in its fp32 mode with DYT_OPT_F32_SPLIT16 (frozen-weight GEMMs as three IEEE-half products)
F_TRAINING, F_COMPLETE, F_SAVE, F_MASKED_DENSE, F_GATE_ALWAYS, F_ACCUM_GRAD, F_DEVICE_SEED, F_TOKENS_IN, F_TOKENS_OUT = 1, 2, 4, 8, 16, 32, 64, 128, 256
OPT_STREAM_OVERLAP, OPT_CLS_TAIL, OPT_SHARE_BLOCK0, OPT_COUNT_FLOPS_TOKENS, OPT_GRAD_SCALE_LOG2, OPT_FC2_CAT, OPT_ATTN_BWD_FUSED, OPT_F32_SPLIT16, OPT_ATTN_V2, OPT_GEMM_SPLITK, OPT_LEARNABLE_SCALE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11

# enum dyt_param (include/dyt_hip.h)
(P_CLS, P_POS, P_PE_W, P_PE_B, P_LN1_W, P_LN1_B, P_QKV_W, P_QKV_B, P_PROJ_W, P_PROJ_B, P_LN2_W, P_LN2_B,
 P_FC1_W, P_FC1_B, P_FC2_W, P_FC2_B, P_NORM_W, P_NORM_B, P_AD_DOWN_W, P_AD_DOWN_B, P_AD_UP_W, P_AD_UP_B,
 P_GATE_W, P_GATE_B, P_HEAD_W, P_HEAD_B,
 P_POOL_QUERY, P_POOL_NQ_W, P_POOL_NQ_B, P_POOL_NK_W, P_POOL_NK_B, P_POOL_NV_W, P_POOL_NV_B, P_POOL_Q_W, P_POOL_K_W,
 P_POOL_V_W, P_POOL_Q_BIAS, P_POOL_V_BIAS, P_POOL_PROJ_W, P_POOL_PROJ_B, P_AD_SCALE, P_AD_LN_W, P_AD_LN_B, P_COUNT) = range(44)

# reference state_dict key suffix -> param id  
GLOBAL_KEYS = {
    "cls_token": P_CLS, "pos_embed": P_POS, "patch_embed.proj.weight": P_PE_W, "patch_embed.proj.bias": P_PE_B,
    "norm.weight": P_NORM_W, "norm.bias": P_NORM_B, "head.weight": P_HEAD_W, "head.bias": P_HEAD_B,
}
# video model only: the attentive pooling head (video_models/video_vision_transformer_IN21K.py:407-410)
POOL_KEYS = {
    "query_token": P_POOL_QUERY,
    "attentive_blocks.norm_q.weight": P_POOL_NQ_W, "attentive_blocks.norm_q.bias": P_POOL_NQ_B,
    "attentive_blocks.norm_k.weight": P_POOL_NK_W, "attentive_blocks.norm_k.bias": P_POOL_NK_B,
    "attentive_blocks.norm_v.weight": P_POOL_NV_W, "attentive_blocks.norm_v.bias": P_POOL_NV_B,
    "attentive_blocks.cross_attn.q.weight": P_POOL_Q_W, "attentive_blocks.cross_attn.k.weight": P_POOL_K_W,
    "attentive_blocks.cross_attn.v.weight": P_POOL_V_W, "attentive_blocks.cross_attn.q_bias": P_POOL_Q_BIAS,
    "attentive_blocks.cross_attn.v_bias": P_POOL_V_BIAS, "attentive_blocks.cross_attn.proj.weight": P_POOL_PROJ_W,
    "attentive_blocks.cross_attn.proj.bias": P_POOL_PROJ_B,
}
GLOBAL_KEYS.update(POOL_KEYS)
BLOCK_KEYS = {
    "norm1.weight": P_LN1_W, "norm1.bias": P_LN1_B, "attn.qkv.weight": P_QKV_W, "attn.qkv.bias": P_QKV_B,
    "attn.proj.weight": P_PROJ_W, "attn.proj.bias": P_PROJ_B, "norm2.weight": P_LN2_W, "norm2.bias": P_LN2_B,
    "mlp.fc1.weight": P_FC1_W, "mlp.fc1.bias": P_FC1_B, "mlp.fc2.weight": P_FC2_W, "mlp.fc2.bias": P_FC2_B,
    "adaptmlp.down_proj.weight": P_AD_DOWN_W, "adaptmlp.down_proj.bias": P_AD_DOWN_B,
    "adaptmlp.up_proj.weight": P_AD_UP_W, "adaptmlp.up_proj.bias": P_AD_UP_B,
    "mlp_token_select.mlp_head.weight": P_GATE_W, "mlp_token_select.mlp_head.bias": P_GATE_B,
    "adaptmlp.scale": P_AD_SCALE,   # only with ffn_adapter_scalar == "learnable_scalar"
    "adaptmlp.adapter_layer_norm_before.weight": P_AD_LN_W,   # only with ffn_adapter_layernorm_option "in" / "out" (dyt_config.adapter_ln)
    "adaptmlp.adapter_layer_norm_before.bias": P_AD_LN_B,
}


def key_to_param(name):
    """'blocks.3.attn.qkv.weight' -> (P_QKV_W, 3); global keys -> (id, 0)."""
    if name in GLOBAL_KEYS:
        return GLOBAL_KEYS[name], 0
    if name.startswith("blocks."):
        _, idx, rest = name.split(".", 2)
        if rest in BLOCK_KEYS:
            return BLOCK_KEYS[rest], int(idx)
    raise KeyError("not a DyT ViT parameter: %s" % name)


def is_trainable_param(pid):
    return pid >= P_AD_DOWN_W


class Config(ctypes.Structure):
    _fields_ = [("num_classes", ctypes.c_int32), ("ffn_num", ctypes.c_int32), ("depth", ctypes.c_int32),
                ("precision", ctypes.c_int32), ("max_batch", ctypes.c_int32), ("slots", ctypes.c_int32),
                ("adapter_scale", ctypes.c_float), ("adapter_dropout", ctypes.c_float), ("tau", ctypes.c_float),
                ("threshold", ctypes.c_float), ("frames", ctypes.c_int32), ("adapter_ln", ctypes.c_int32)]


class DyTError(RuntimeError):
    pass


_lib = None

_vp, _i, _i64, _f, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64

# every symbol include/dyt_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dyt_last_error": (ctypes.c_char_p, []),
    "dyt_version": (_i, []),
    "dyt_operand_type": (_i, []),
    "dyt_ctx_create": (_i, [ctypes.POINTER(Config), ctypes.POINTER(_vp)]),
    "dyt_ctx_destroy": (_i, [_vp]),
    "dyt_ctx_bytes": (_i, [_vp, ctypes.POINTER(_i64)]),
    "dyt_ctx_set_option": (_i, [_vp, _i, _i]),
    "dyt_set_global_option": (_i, [_i, _i]),
    "dyt_set_drop_path": (_i, [_vp, _f]),
    "dyt_set_drop_path_scales": (_i, [_vp, _i, _vp]),
    "dyt_set_soft_targets": (_i, [_vp, _vp, _i]),
    "dyt_set_frozen": (_i, [_vp, _i, _i, _vp, _vp]),
    "dyt_trainable_numel": (_i, [_vp, ctypes.POINTER(_i64)]),
    "dyt_trainable_offset": (_i, [_vp, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dyt_forward": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "dyt_backward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dyt_loss": (_i, [_vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "dyt_adamw": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _f, _f, _f, _f, _vp]),
    "dyt_adamw_guarded": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _vp]),
    "dyt_step_fwd_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _u64, _f, _f, _f, _f, _vp, _vp, _vp, _vp,
                              _vp, _vp]),
    "dyt_seed": (_i, [_vp, _u64, _vp]),
    "dyt_grad_part": (_i, [_vp, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dyt_stream_wait_grads": (_i, [_vp, _i, _vp]),
    "dyt_allreduce_grads": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "dyt_clip_grad_norm": (_i, [_vp, _vp, _i64, _f, _f, _vp, _vp]),
    "dyt_debug_dispatch": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dyt_debug_dact": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "dyt_debug_drop_path": (_i, [_vp, _i, _vp, _vp]),
    "dyt_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "dyt_linear": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dyt_linear_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dyt_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "dyt_adapter_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _u64, _i, _vp]),
    "dyt_adapter_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _u64, _i, _vp]),
    "dyt_mlp_gathered_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "dyt_mlp_gathered_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "dyt_gate_compact": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dyt_gemm_bf16_raw": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dyt_gemm_f32_raw": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dyt_wgrad_scratch_floats": (ctypes.c_int64, [_i]),
    "dyt_wgrad_raw": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dyt_debug_counters": (_i, [ctypes.POINTER(ctypes.c_uint64), _i]),
    "dyt_debug_checksums": (_i, [_i, ctypes.POINTER(ctypes.c_uint64), _i, ctypes.POINTER(_i)]),
    "dyt_debug_checksum_label": (ctypes.c_char_p, [_i, _i]),
    "dyt_debug_dump_read": (ctypes.c_int64, [_vp, _i64]),
    "dyt_profile_enable": (_i, [_vp, _i]),
    "dyt_profile_read": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64),
                              ctypes.POINTER(ctypes.c_double)]),
}


_lib16 = None


def lib(fp16=False):
    """Load libdyt_hip.so (once; fp16=True: libdyt_hip_f16.so).  Raises DyTError when it is absent -- there is no CPU path."""
    global _lib, _lib16
    if fp16:
        if _lib16 is None:
            lib()   # HIP runtime / RCCL promoted to the global scope, error text shared
            if not os.path.exists(LIB_PATH_F16):
                raise DyTError("%s not found: build it with `make -C dynamic-tuning_amd/csrc`" % LIB_PATH_F16)
            L = ctypes.CDLL(LIB_PATH_F16)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            if L.dyt_operand_type() != 1:
                raise DyTError("%s was not built with -DDYT_FP16" % LIB_PATH_F16)
            _lib16 = L
        return _lib16
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DyTError("%s not found: build it with `make -C dynamic-tuning_amd/csrc` "
                           "(or __graft_entry__.build()); the DyT path has no fallback" % LIB_PATH)
        # libdyt_hip.so is linked WITHOUT a HIP runtime (csrc/Makefile: -no-hip-rt): it must bind to the
        # runtime PyTorch already loaded, so that torch's stream handles, allocations and synchronisation
        # are the ones our launches see.  Promote that runtime to the global symbol scope first.
        import torch
        rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if not os.path.exists(rt):
            raise DyTError("PyTorch-ROCm's HIP runtime not found at %s" % rt)
        ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(rccl):   # dyt_allreduce_grads binds ncclAllReduce (a weak reference) to the RCCL torch.distributed uses
            ctypes.CDLL(rccl, mode=ctypes.RTLD_GLOBAL)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, L=None):
    if rc != 0:
        msg = (L or lib()).dyt_last_error().decode()
        if not msg and _lib16 is not None and L is None:
            msg = _lib16.dyt_last_error().decode()
        raise DyTError("libdyt_hip: %s (code %d)" % (msg, rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libdyt_hip needs contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- RCCL communicator for dyt_allreduce_grads: created natively (ncclCommInitRank), the 128-byte unique id travels over the
# ---- already initialised torch.distributed group (plumbing only) ----------------------------------------------------------------
class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


_rccl = None


def rccl():
    global _rccl
    if _rccl is None:
        import torch
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise DyTError("PyTorch-ROCm's librccl.so not found at %s" % path)
        R = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        R.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
        R.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        R.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        R.ncclGetErrorString.restype = ctypes.c_char_p
        R.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _rccl = R
    return _rccl


def rccl_comm_create(device, group=None):
    """One ncclComm_t per process over the ranks of the (initialised) torch.distributed group."""
    import torch
    import torch.distributed as dist
    R = rccl()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = _NcclUniqueId()
    if rank == 0:
        rc = R.ncclGetUniqueId(ctypes.byref(uid))
        if rc != 0:
            raise DyTError("ncclGetUniqueId: %s" % R.ncclGetErrorString(rc).decode())
    t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().tolist()), 128)
    comm = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = R.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
    if rc != 0:
        raise DyTError("ncclCommInitRank: %s" % R.ncclGetErrorString(rc).decode())
    return comm


_shared_comms = {}


def rccl_comm_shared(device):
    """The process's communicator for `device` over the default group: created on first use (a re-created engine -- a larger eval
    batch -- reuses it instead of running another ncclCommInitRank mid-training), destroyed at interpreter exit."""
    import atexit
    import torch
    key = torch.device(device).index or 0
    if key not in _shared_comms:
        _shared_comms[key] = rccl_comm_create(device)
        if len(_shared_comms) == 1:
            atexit.register(rccl_comm_destroy_all)
    return _shared_comms[key]


def rccl_comm_ranks(comm):
    """Number of ranks the library's communicator spans (ncclCommCount): what a multi-GPU record should show next to torch's world size."""
    n = ctypes.c_int(0)
    rc = rccl().ncclCommCount(comm, ctypes.byref(n))
    if rc != 0:
        raise DyTError("ncclCommCount: %s" % rccl().ncclGetErrorString(rc).decode())
    return n.value


def rccl_comm_destroy_all():
    for comm in list(_shared_comms.values()):
        try:
            rccl().ncclCommDestroy(comm)
        except Exception:
            pass
    _shared_comms.clear()
