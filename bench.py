#!/usr/bin/env python3
"""Benchmark of the DyT ViT-B/16 fine-tune step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: starts N ranks itself, or runs as one rank
                                                         when launched by torch.distributed.run)

One "step" = engine_finetune.py:47-79 of the reference: student forward (compacted MLP, Gumbel
gate, adapter dropout), teacher forward (complete model), CE + 2*token-ratio + teacher CE + KL,
one backward over both passes, (all-reduce of the 5 MB trainable gradients), AdamW -- on a batch
of synthetic N(0,1) images resident in HBM, B=128 per GPU (configs[1] of BASELINE.json, the
train_IN21K.sh shape), r=64, C=100, gate biases calibrated so the measured keep ratio is ~0.70.

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline     -- the dominant kernel (the bf16 MFMA GEMM family, csrc/gemm.hip): algorithmic FLOPs of
                  its launches in one step / their summed duration, measured with HIP events on the
                  launch stream inside this process, against the dense bf16 MFMA peak (2.5 PFLOP/s);
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference, pinned to golden vectors)
                  running the same step at B=16 on the host cores (N=1, rank 0 only).
"""
import argparse
import json
import math
import os
import sys
import time

# The step keeps two streams busy (student / teacher pass) next to torch's current stream, the gradient-sum stream, the
# all-reduce stream and RCCL's own.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin, and two
# streams that share a queue run one after the other: measured with the 1-rank RCCL path, 30.6 ms/step with 4 queues vs
# 25.9 with 8 (without RCCL: 25.9 either way).  Must be set before the HIP runtime initialises, i.e. before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dynamic-tuning_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth  # noqa: E402

STEP_GFLOP_AT_07 = 126.851   # SURVEY.md section 8d / BASELINE.md section 3, compact mode, r=64, C=100
STEP_GFLOP_SLOPE = 42.542    # d(GFLOP)/d(keep ratio)
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3,   # TFLOP/s dense MFMA, MI355X_MICROARCH.md
        "fp16x3": 2500.0 / 3,   # three half-precision products per fp32-class product: the USEFUL-FLOP ceiling of the split form
        "fp16x3f": 2500.0 / 2,  # forward GEMMs three products, gradient GEMMs one (equal useful FLOPs either side): two on average
        "fp16x3h": 2500.0 / 2,
        "fp16x3q": 2500.0 / 1.75,  # qkv / proj (both passes) and the teacher pass's MLP in the fp8-correction form (two f16-equivalents), the student's MLP three-part
        "fp16f8": 2500.0 / 1.5}  # forward: one f16 product + two fp8 products at twice the rate = two f16-equivalents; backward one  # the same product counts; the backward runs on 16-bit operands with the fp16 mode's kernels
TRAFFIC_JSON = os.path.join("round6", "gemm_traffic.json")
SUSTAINED_MFMA_TFLOPS = 1670.0   # measured, see roofline.sustained_mfma_measured


class Cfg(dict):
    def __getattr__(self, k):   # AttributeError (not KeyError) for a missing key: copy / pickle probe attributes
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def build_model(args, device):
    video = args.video_frames > 1
    if video:
        from video_models.video_vision_transformer_IN21K import vit_base_patch16_224_in21k
    else:
        from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    sd = synth.make_state_dict(args.classes, args.ffn_num, seed=0, kind="bench", gate_bias=math.log(0.7 / 0.3), video=video)
    tuning = Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none",
                 ffn_adapter_init_option="lora", ffn_adapter_scalar="0.1", ffn_num=args.ffn_num, d_model=768)
    model = vit_base_patch16_224_in21k(num_classes=args.classes, drop_path_rate=0.0, tuning_config=tuning,
                                       select_config=Cfg(open=True, keep_layers=0), precision=args.precision,
                                       max_batch=args.batch, train_mode=args.mode)
    model.load_state_dict(sd)
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    return model.to(device)


def calibrate_gates(model, x, target, iters=8):
    """Shift each block's gate bias until the measured training-mode keep ratio is `target`
    (keep ratio is learned in the reference; benchmarks pin it)."""
    model.train()
    keep = None
    with torch.no_grad():
        for _ in range(iters):
            _, aux = model(x)
            keep = aux["token_select"].float().mean(dim=(0, 2, 3)).clamp(1e-3, 1 - 1e-3)  # [depth]
            delta = (math.log(target / (1 - target)) - torch.log(keep / (1 - keep)))
            for i, blk in enumerate(model.blocks):
                blk.mlp_token_select.mlp_head.bias.add_(delta[i])
    return float(keep.mean())


def cpu_baseline(args):
    """The oracle's full step (losses, autograd, AdamW) at B=16 on the host cores: 1 warm-up + 3 timed
    (bounded: stops early if the sample exceeds ~60 s)."""
    from oracle import dyt_oracle as O
    B = 16
    ncores = synth.available_cores()
    torch.set_num_threads(ncores)
    video = args.video_frames > 1
    sd = synth.make_state_dict(args.classes, args.ffn_num, seed=0, kind="bench", gate_bias=math.log(0.7 / 0.3), video=video)
    x, y = synth.make_batch(B, args.classes, seed=0)
    if video:
        y = y[: B // args.video_frames].contiguous()
    g1, g2 = synth.make_noise(B, seed=2)
    keep = synth.make_dropout_masks(B, args.ffn_num, seed=3)
    opt = {}
    times = []
    t_all = time.time()
    for i in range(4):
        t0 = time.time()
        O.train_step(sd, opt, x, y, g1, g2, keep, lr=1e-3, wd=0.01, scale=0.1, mode="masked", token_target_ratio=0.5,
                     frames=args.video_frames if video else 1)
        times.append(time.time() - t0)
        log("cpu baseline step %d: %.2f s" % (i, times[-1]))
        if time.time() - t_all > 60 and len(times) >= 2:
            break
    dt = sum(times[1:]) / len(times[1:])
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": round(B / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/dyt_oracle.py train_step (reference-as-written masked step, fp32%s), B=16, %d timed steps "
                      "after 1 warm-up, %.2f s/step, %s" % (", video model: %d clips x %d frames" % (B // args.video_frames, args.video_frames)
                                                            if video else "", len(times) - 1, dt, model)}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "fp32", "fp16x3", "fp16x3f", "fp16x3h", "fp16x3q", "fp16f8"],
                    help="16-bit operand type of the fast kernels (bf16, or fp16 = the reference's own autocast dtype, same MFMA rate) or the exact-fp32 parity mode")
    ap.add_argument("--mode", default="compact", choices=["compact", "masked"])
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--ffn_num", type=int, default=64)
    ap.add_argument("--keep", type=float, default=0.7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the second (fp32 parity mode) measurement at N=1")
    ap.add_argument("--hip-graph", type=int, default=0,
                    help="1: replay the step's forward+backward from a captured hipGraph.  Off by default: on ROCm 7.2 a replay "
                         "costs the host as much as the eager launches (DESIGN.md section 7) and the capture has to serialise "
                         "the teacher pass's adapter branch")
    ap.add_argument("--host-batches", type=int, default=-1,
                    help="N > 0: after the resident-data measurement, feed N DISTINCT pinned host batches through engine_finetune.train_one_epoch "
                         "(the reference's loop: host -> device copy of every batch, reference engine_finetune.py:34-42; here prefetched on a copy "
                         "stream) for --steps steps and report images/s beside `value` as `host_fed`.  -1 (default): 8 at N=1, 0 otherwise")
    ap.add_argument("--video-frames", type=int, default=0,
                    help="T > 1: BASELINE.json configs[4] shape instead of the headline one -- the video model, "
                         "--batch frames per GPU = batch/T clips of T frames (train_video.sh: 16 clips x 8 frames, 400 classes)")
    args = ap.parse_args()
    if args.precision is None:
        # headline mode: IEEE-half operands (the reference's own GPU dtype: it trains under fp16 autocast, engine_finetune.py:47) --
        # at the bf16 mode's speed it is ~7x closer to the fp32 reference (worst of five draws at B=16: logits 4.8e-3 vs 0.019, 4 vs 20 of 37 632 decisions)
        args.precision = "fp16"
    if args.video_frames > 1:
        assert args.batch % args.video_frames == 0, "--batch must be a multiple of --video-frames"
    if args.host_batches < 0:
        args.host_batches = 8 if (args.gpus == 1 and args.video_frames <= 1) else 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start N ranks ourselves (one process per GPU over RCCL), exactly the command the
        # reference's train_IN21K.sh:10-16 launch line maps to; rank 0 of the children prints the one JSON line.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DYT_BENCH_SHARE_GPU=1 (test rig, 1-GPU boxes): every rank on cuda:0, torch.distributed over gloo, the torch.distributed form of the gradient
    # all-reduce (RCCL refuses two ranks on one device).  Exercises the N > 1 launch line, barriers, broadcast, all-reduce, max-over-ranks timing
    # and the rank-0 JSON line with real processes; the number it prints is NOT a multi-GPU measurement and says so (`config.parallelism`).
    share_gpu = bool(os.environ.get("DYT_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
        if os.environ["DYT_BENCH_SHARE_GPU"] != "native":   # "native": let the ranks TRY the library's own RCCL communicator (it cannot exist on a shared
            os.environ["DYT_NATIVE_RCCL"] = "0"             # device) and fall back by the collective agreement of engine_finetune.allreduce_grads
        if os.environ.get("DYT_BENCH_WATCHDOG"):   # dump every thread's stack and exit if the rig hangs
            import faulthandler
            faulthandler.dump_traceback_later(int(os.environ["DYT_BENCH_WATCHDOG"]), exit=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the DyT path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # DYT_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL process group, barriers, broadcast, chunked all-reduce on the
    # comm stream, max-over-ranks timing) even with one rank -- how the N > 1 path is exercised on a 1-GPU box
    if world > 1 or os.environ.get("DYT_BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    head = measure(args, args.precision, args.mode, args.steps, args.warmup, device, world, rank, host_batches=args.host_batches)
    side_errors = {}

    def side(name, *a, **kw):
        """A secondary measurement (other modes, A/B): a failure is recorded in the JSON line and must not cost the headline."""
        try:
            return measure(*a, **kw)
        except Exception as e:   # noqa: BLE001
            side_errors[name] = "%s: %s" % (type(e).__name__, str(e)[:300])
            log("side measurement %s failed: %s" % (name, side_errors[name]))
            try:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            except Exception:   # noqa: BLE001
                pass
            return None
    parity = None
    other = None
    exact = None
    extras = world == 1 and args.video_frames <= 1 and args.precision in ("fp16", "bf16") and not args.no_parity_mode
    if extras:
        # the other 16-bit operand type on the same workload (same kernels, same MFMA rate)
        torch.cuda.empty_cache()
        o = "bf16" if args.precision == "fp16" else "fp16"
        om = side("other_fast_mode", args, o, args.mode, max(2, min(args.steps, 10)), 2, device, world, rank)
        if om is not None:
            other = {"dtype": o, "value": om["value"], "unit": "images/s", "ms_per_step": om["ms_per_step"], "steps": om["steps"],
                     "roofline_frac": om["roofline"]["frac"] if om["roofline"] else None,
                     "parity": "vs the CPU oracle at B=16, WORST of five draws (tests/test_gpu_round5.py::test_fast_modes_vs_oracle_over_seeds; round 6 run, "
                           "profiles/round6/r6_gpu_tests_full.txt): fp16 logits 4.4e-3 student / 1.9e-3 teacher, up to 4 of 37 632 token-keep decisions differ, "
                           "gradients gate 7.7e-3 / down_proj 8.2e-2 / up_proj 1.3e-3 / head 8.2e-4 (round 5: 2.6e-2 / 8.2e-2 / 1.1e-3 / 8e-4; the gate and down_proj "
                           "figures are decisions that come out the other way, not round-off) -- outside north_star's 1e-3 / bit-exact bar (that is "
                           "parity_mode's); bf16 logits 1.9e-2 / 20 decisions / gate 6e-2, down_proj 0.10"}
            # A/B: the headline mode with LayerNorm-2 as its own kernel (DYT_LN_FOLD=0; the default folds it into the fc1 GEMM)
            torch.cuda.empty_cache()
            os.environ["DYT_LN_FOLD"] = "0"
            try:
                fv = side("headline_with_ln2_as_kernel", args, args.precision, args.mode, max(2, min(args.steps, 10)), 2, device, world, rank)
            finally:
                os.environ.pop("DYT_LN_FOLD", None)
            if fv is not None:
                other["headline_with_ln2_as_kernel"] = {"dtype": args.precision, "value": fv["value"], "unit": "images/s", "ms_per_step": fv["ms_per_step"],
                                                        "steps": fv["steps"]}
        # the same step in the fastest mode that meets north_star's parity bars (logits <= 1e-3, gate masks bit-exact).
        # "fp16x3q" = fp32 data flow; MLP GEMMs, patch embedding and attention as three IEEE-half products per fp32-class product; the
        # attention branch's GEMMs (qkv, proj) as hi * hi in half + two fp8 (e4m3) correction products -- and the teacher
        # (complete_model) pass, whose gate output is discarded so that no token-keep decision depends on it, its MLP GEMMs too, and its
        # attention forward as the hi * hi product alone (IEEE-half attention on the hi planes); backward
        # pass on 16-bit operands with the exact forward's masks (DYT_OPT_F32_SPLIT16 = 5); tests/test_gpu_round4.py
        torch.cuda.empty_cache()
        pm = side("parity_mode", args, "fp16x3q", args.mode, max(2, min(args.steps, 6)), 1, device, world, rank)
        if pm is not None:
            parity = {"dtype": "fp16x3q", "train_mode": args.mode, "value": pm["value"], "unit": "images/s", "ms_per_step": pm["ms_per_step"],
                      "steps": pm["steps"], "step_gflop_per_image": pm["step_gflop_per_image"], "step_mfma_frac": pm["step_mfma_frac"],
                      "keep_ratio_measured": pm["keep_ratio_measured"], "roofline": pm["roofline"],
                      "parity": "vs the CPU oracle at B=16 over five seeds (tests/test_gpu_round4.py::test_parity_modes_vs_oracle_over_seeds): logits "
                            "max abs err <= 2.5e-5 student / 1.0e-4 teacher (bar 1e-3), 0 of 5 x 37 632 token-keep decisions differ, losses 1e-5; 74 gradients rel-L2 "
                            "<= 1.5e-3 worst over the seeds (round 6: the backward's gradient stream is carried in 16 bits between its row kernels, typical 7e-4 -> 1.1e-3, "
                            "worst 1.40e-3 -> 1.53e-3; DYT_G16_B16=0 restores the fp32 stream at +0.75 ms per step; one draw with an adapter unit on the other side of the "
                            "ReLU: 4e-3 in that layer's two tensors, 6e-4 without that row -- over 24 draws 22 steps need no such exclusion, 2 need one layer: "
                            "profiles/round6/r6_relu_side_events.txt); at B=128 vs the exact-fp32 mode: logits 1.4e-5 / 3.0e-4, 0 of 301 056 decisions; "
                            "also run on the reference goldens, the VTAB shapes and the video model (tests/gpu_diag.py, test_gpu_round2.py); "
                            "`roofline.peak` = useful-FLOP ceiling of the product counts, `roofline.frac_of_mfma_peak` = useful FLOP/s / 2500 TFLOP/s"}
            torch.cuda.empty_cache()
            # "fp16x3h": every forward GEMM three-part -- the fp16x3 forward to 1.2e-6 (bit for bit with DYT_OPT_FC2_CAT = 0; logits 6.9e-6 from the oracle)
            hm = side("forward_all_three_part", args, "fp16x3h", args.mode, max(2, min(args.steps, 4)), 1, device, world, rank)
            if hm is not None:
                parity["forward_all_three_part"] = {
                    "dtype": "fp16x3h", "value": hm["value"], "unit": "images/s", "ms_per_step": hm["ms_per_step"], "steps": hm["steps"],
                    "roofline": hm["roofline"],
                    "parity": "logits <= 6.9e-6, 0 of 5 x 37 632 decisions, 74 gradients <= 1.5e-3 over five seeds"}
            torch.cuda.empty_cache()
            # "fp16f8": the fp8-correction form for the MLP GEMMs as well (every forward GEMM 2K- instead of 3K-equivalent): meets the logit
            # bar, NOT the bit-exact-mask bar (gate logits ~5e-5 from the reference: near-ties flip)
            qm = side("fp8_corrections", args, "fp16f8", args.mode, max(2, min(args.steps, 6)), 1, device, world, rank)
            if qm is not None:
                parity["fp8_corrections"] = {
                    "dtype": "fp16f8", "value": qm["value"], "unit": "images/s", "ms_per_step": qm["ms_per_step"], "steps": qm["steps"],
                    "roofline": qm["roofline"],
                    "parity": "vs the CPU oracle at B=16 over five seeds: logits max abs err <= 6.2e-5 (bar 1e-3; one draw with a flipped decision 5.9e-4).  "
                      "Token-keep decisions under the ONE tie rule of tests/parity_rules.py (a decision may differ only inside the reference's own "
                      "fp32 tie band, 4 x its measured gate-logit round-off vs float64: 0.5 - 1.9e-6 in (logit + g) / tau): the first differing decision "
                      "of two of the five draws lies inside the band (margins 3.6e-7 / <1.4e-6), and ONE of 301 056 decisions at B=128 lies just outside "
                      "(margin 9.8e-7, band 9.5e-7).  With gate logits ~5e-5 from the reference (25x the band; fp16x3q: ~1e-5) fp16f8 keeps its masks by "
                      "the draw, not by construction, and a flipped token moves the student logits by up to 1.7e-3: not the parity mode.  74 gradients "
                      "rel-L2 <= 2.0e-3 on the draws without a flip; per GEMM 1.5-2.5e-5 of max|C| vs fp64 (three-part: 1-2e-6, plain half: 4e-4)"}
            torch.cuda.empty_cache()
            fm = side("all_products_three_part", args, "fp16x3", args.mode, max(2, min(args.steps, 4)), 1, device, world, rank)
            if fm is not None:
                parity["all_products_three_part"] = {
                    "dtype": "fp16x3", "value": fm["value"], "unit": "images/s", "ms_per_step": fm["ms_per_step"], "steps": fm["steps"],
                    "roofline_frac": fm["roofline"]["frac"] if fm["roofline"] else None, "roofline_peak": PEAK["fp16x3"],
                    "roofline_frac_of_mfma_peak": fm["roofline"]["frac_of_mfma_peak"] if fm["roofline"] else None,
                    "parity": "fp32-width backward as well: same logits / decisions / losses as fp16x3h with DYT_OPT_FC2_CAT = 0; 74 gradients rel-L2 <= 1.5e-4 (71 of them <= 6e-6); "
                              "fp16x3f (gradient products hi * hi in the fp32 data flow, 49 ms/step in round 3) is still built and tested"}
        # ... and in the exact-fp32 mode (fp32 operands on the matrix cores, v_mfma_f32_32x32x2_f32: the reference arithmetic)
        torch.cuda.empty_cache()
        em = side("exact_mode", args, "fp32", args.mode, max(2, min(args.steps, 3)), 1, device, world, rank)
        if em is not None:
            exact = {"dtype": "fp32", "train_mode": args.mode, "value": em["value"], "unit": "images/s", "ms_per_step": em["ms_per_step"],
                     "steps": em["steps"], "roofline": em["roofline"],
                     "parity": "fp32 mode vs reference goldens on MI355X: logits max abs err 4e-6, token-keep masks bit-exact, "
                               "74 gradients rel-L2 < 2e-3 (tests/test_gpu_parity.py, tests/gpu_diag.py)"}
    dist_info = None
    if dist.is_initialized():
        cores = [None] * world
        dist.all_gather_object(cores, synth.available_cores())
        dist_info = {"torch_world_size": world, "rccl_ranks": head.get("rccl_ranks"), "host_cores_per_rank": cores,
                     "comm_stream_concurrent": head.get("comm_stream_concurrent"),
                     "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "images/sec DyT ViT-B/16 fine-tune step (student+teacher fwd, bwd, AdamW) @ keep~0.7",
            "value": head["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("ViT-B/16 DyT on CIFAR-100 shape, batch=128/GPU, 1xMI355X per rank, keep-ratio target 0.7 "
                                    "(BASELINE.json configs[1]; N>1 = configs[3] shape: global batch 128*N, DP over RCCL)")
                       if args.video_frames <= 1 else
                       ("NOT the headline config: video DyT ViT-B/16 (BASELINE.json configs[4] shape), %d clips x %d frames per GPU, "
                        "%d classes; `value` counts frames/s" % (args.batch // args.video_frames, args.video_frames, args.classes)),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "ffn_num": args.ffn_num,
                       "num_classes": args.classes, "train_mode": args.mode, "parallelism": ("dp%d" % world) + (" (TEST RIG: all ranks share ONE GPU over gloo, DYT_BENCH_SHARE_GPU)" if os.environ.get("DYT_BENCH_SHARE_GPU") else ""),
                       "keep_ratio_measured": head["keep_ratio_measured"], "keep_ratio_calibrated": head["keep_ratio_calibrated"],
                       "hip_graph": head["hip_graph"]},
            "images_per_s_per_gpu": round(head["value"] / world, 2),
            "step_gflop_per_image": head["step_gflop_per_image"],
            "step_mfma_frac": head["step_mfma_frac"],
            "step_gflop_executed_per_image": head["step_gflop_executed_per_image"],
            "step_mfma_frac_executed": head["step_mfma_frac_executed"],
            "loss": head["loss"],
            "host_enqueue_ms_per_step": head["host_enqueue_ms_per_step"],
            "roofline": head["roofline"],
        }
        if head.get("host_fed") is not None:
            out["host_fed"] = head["host_fed"]
        if dist_info is not None:
            out["distributed"] = dist_info
        if parity is not None:
            out["parity_mode"] = parity
        if exact is not None:
            out["exact_mode"] = exact
        if other is not None:
            out["other_fast_mode"] = other
        if side_errors:
            out["side_measurement_errors"] = side_errors
        log("roofline", head["roofline"])
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:   # noqa: BLE001 -- reported, never fatal for the line
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def host_fed(args, model, opt, steps, warmup, device, world, resident_ms):
    """The same step fed from the HOST the way the reference's loop is (engine_finetune.py:34-42): `n` distinct pinned batches cycled
    through engine_finetune.train_one_epoch -- per-iteration lr schedule, host -> device copy of every batch (prefetched on a copy stream
    into a second device buffer, DevicePrefetcher), the fused step, the loop's own bookkeeping and its sync every 20 steps."""
    import types
    from engine_finetune import train_one_epoch
    from models.losses import AdaLoss
    n = args.host_batches
    batches = []
    for i in range(n):
        x, y = synth.make_batch(args.batch, args.classes, seed=500 + i)
        batches.append((x.pin_memory(), y.pin_memory()))
    crit = AdaLoss(torch.nn.CrossEntropyLoss(), token_target_ratio=args.keep, token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0)
    lr = opt.param_groups[0]["lr"]
    la = types.SimpleNamespace(accum_iter=1, lr=lr, min_lr=lr, warmup_epochs=0, epochs=10 ** 6, hip_graph=bool(args.hip_graph))
    out = {}
    # one timed EPOCH: its fixed costs (first batch copied ahead of the first step, optimizer state published for the checkpoint, the closing
    # sync) belong to the loop, but over 20 steps they weigh 20x what they do in the shortest epoch of the reference's scripts (CIFAR-100 at
    # B = 128 per GPU: 390 steps) -- the epoch here is 4 x the headline's step count, at least 80
    steps = max(4 * steps, 80)
    resident = [(x.to(device), y.to(device)) for x, y in batches[:2]]
    for name, env in (("prefetched", "1"), ("copy_on_compute_stream", "0"), ("loop_on_resident_batches", "1")):
        os.environ["DYT_PREFETCH"] = env
        src = resident if name == "loop_on_resident_batches" else batches
        try:
            train_one_epoch(model, crit, [src[i % len(src)] for i in range(max(warmup, 2))], opt, device, 0, None, args=la)
            torch.cuda.synchronize()
            if dist.is_initialized():
                dist.barrier()
            t0 = time.perf_counter()
            train_one_epoch(model, crit, [src[i % len(src)] for i in range(steps)], opt, device, 1, None, args=la)
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()
            out[name] = (time.perf_counter() - t0) / steps * 1e3
        finally:
            os.environ.pop("DYT_PREFETCH", None)
    ms = out["prefetched"]
    return {"value": round(args.batch * world / ms * 1e3, 2), "unit": "images/s", "ms_per_step": round(ms, 3), "steps": steps,
            "distinct_batches": n, "pinned": True, "h2d_mb_per_step": round(args.batch * 3 * 224 * 224 * 4 / 1e6, 1),
            "resident_ms_per_step": round(resident_ms, 3), "vs_resident": round(resident_ms / ms, 4),
            "copy_on_compute_stream_ms_per_step": round(out["copy_on_compute_stream"], 3),
            "loop_on_resident_batches_ms_per_step": round(out["loop_on_resident_batches"], 3),
            "copy_stream_probe": getattr(model._engine, "_copy_stream_probe", (None, None))[1],
            "loop": "engine_finetune.train_one_epoch (reference engine_finetune.py:16-106): lr schedule per iteration, H2D of every batch on a copy "
                    "stream into the other of two device buffers (event hand-over), fused step, one host sync per 20 steps; one timed epoch of `steps` steps; "
                    "`copy_on_compute_stream` = the reference's placement of the copy (DYT_PREFETCH=0)"}


def measure(args, precision, mode, steps, warmup, device, world, rank, host_batches=0):
    """Build the model in one arithmetic / training mode, calibrate the keep ratio, time `steps` fused steps (barrier +
    synchronize on both sides, max over ranks) and take the dominant-kernel roofline of one extra event-profiled step."""
    from engine_finetune import FusedAdamW, train_step
    margs = argparse.Namespace(**vars(args))
    margs.precision, margs.mode = precision, mode
    torch.manual_seed(1234 + rank)
    model = build_model(margs, device)
    x, y = synth.make_batch(args.batch, args.classes, seed=100 + rank)  # each rank its own shard of the global batch
    if args.video_frames > 1:   # [b,c,t,h,w] clips, one target per clip
        T = args.video_frames
        x = x.reshape(args.batch // T, T, 3, 224, 224).permute(0, 2, 1, 3, 4).contiguous()
        y = y[: args.batch // T].contiguous()
    x, y = x.to(device), y.to(device)
    keep_cal = calibrate_gates(model, x, args.keep)
    log("[%s/%s] calibrated keep ratio %.4f, ctx %.1f GB" % (precision, mode, keep_cal, model._engine.bytes / 1e9))
    model.train()
    opt = FusedAdamW(model, lr=1e-3 * args.batch * world / 256, weight_decay=0.01)   # broadcasts rank 0's trainables (DDP ctor)
    eng = model._engine
    if os.environ.get("DYT_NO_OVERLAP"):   # profiling aid: serial launches give clean per-kernel durations
        import _lib
        eng.set_option(_lib.OPT_STREAM_OVERLAP, 0)
    use_graph = bool(args.hip_graph) and not os.environ.get("DYT_NO_OVERLAP")
    losses = torch.zeros(8, device=device)
    acc = torch.zeros(8, device=device)

    def one_step(i, update=True):
        train_step(model, x, y, opt, losses_out=losses, seed=1000 + i, target_ratio=args.keep, token_minimal=0.0,
                   token_minimal_weight=0.0, graph=use_graph, update=update)

    for i in range(warmup):
        one_step(i)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(warmup + i)
        acc += losses
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # host cost of a step: enqueue two steps into EMPTY queues (inside the timed loop the host runs ahead of the GPU until the
    # hardware queues are full and then only measures back-pressure -- round 1's "13.6 ms" was that)
    th0 = time.perf_counter()
    for i in range(2):
        one_step(10 ** 5 + i)
    t_host = (time.perf_counter() - th0) / 2 * steps
    torch.cuda.synchronize()
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    host = (acc / steps).tolist()
    log("[%s/%s] timed %d steps: %.2f ms/step (host enqueue %.2f ms/step)" % (precision, mode, steps, dt / steps * 1e3, t_host / steps * 1e3))
    keep_meas = host[5]

    # dominant-kernel roofline: HIP events around every GEMM launch of ONE step (same stream)
    roof = None
    traffic = None
    if precision in ("bf16", "fp16"):
        try:  # HBM bytes per GEMM launch from the committed PMC passes of this same command (tools/pmc_traffic.py)
            with open(os.path.join(ROOT, "profiles", TRAFFIC_JSON)) as f:
                traffic = json.load(f)
        except Exception:
            pass
    if rank == 0:
        use_graph = False   # the event-profiled step runs eagerly, one stream
        eng.profile(True)
        # N > 1: forward + backward only -- this step runs on rank 0 ALONE, and the update leg holds the gradient all-reduce -- a collective the other
        # ranks would never join (found with two real processes, DYT_BENCH_SHARE_GPU: rank 0 waited in all_reduce, rank 1 in all_gather_object)
        one_step(10 ** 6, update=(world == 1))
        ms, n, fl = eng.profile_read(0)
        ms_a, n_a, fl_a = eng.profile_read(1)
        ms_o, n_o, _ = eng.profile_read(2)
        _, n_kern, _ = eng.profile_read(3)   # kernel launches behind the n GEMMs (PMC traffic is per kernel launch)
        eng.profile(False)
        ach = fl / (ms * 1e-3) / 1e12
        kern = {"fp32": "gemm_f32_mfma_nt_kernel (exact-fp32 MFMA 32x32x2, all epilogues)",
                "fp16x3": "split3_a_kernel + gemm_bf16_nt_kernel (fp32 operands as IEEE-half hi / lo parts, three f16 MFMA 16x16x32 products per "
                          "fp32-class product, fp32 epilogues; adapter-sized GEMMs on gemm_f32_mfma_nt_kernel); achieved = USEFUL FLOPs"
                          ,
                "fp16x3f": "gemm_bf16_nt_kernel on IEEE-half hi / lo parts of fp32 operands: forward GEMMs three f16 MFMA 16x16x32 products per "
                           "fp32-class product, gradient GEMMs the hi * hi product alone, fp32 epilogues; achieved = USEFUL FLOPs",
                "fp16f8": "forward: gemm_bf16_nt_kernel<F8> on [hi16 | e4m3 hi | e4m3 lo] images of fp32 operands (hi * hi as f16 MFMA 16x16x32, the two "
                          "correction products as v_mfma_scale_f32_16x16x128_f8f6f4, fp32 epilogues); backward: the fp16 mode's kernels on 16-bit "
                          "operands; achieved = USEFUL FLOPs",
                "fp16x3h": "forward: gemm_bf16_nt_kernel on IEEE-half hi / lo parts of fp32 operands (three f16 MFMA 16x16x32 products per fp32-class "
                           "product, fp32 epilogues); backward: the fp16 mode's gemm_bf16_nt_kernel + gemm_bf16_bpre_kernel on 16-bit operands; "
                           "achieved = USEFUL FLOPs"
                }.get(precision, "gemm_bf16_nt_kernel + gemm_bf16_bpre_kernel (%s MFMA 16x16x32, all epilogues)" % precision)
        roof = {"bound": "mfma", "kernel": kern,
                "achieved": round(ach, 2), "peak": PEAK[precision], "unit": "TFLOP/s", "frac": round(ach / PEAK[precision], 4),
                "frac_of_mfma_peak": round(ach / (PEAK["fp32"] if precision == "fp32" else 2500.0), 4),
                "traffic": round(traffic["hbm_bytes_per_launch"] * max(n_kern, 1) / max(n, 1)) if traffic else None,
                "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 "
                                   "gfx950 correction): bytes per kernel launch x kernel launches per GEMM (%d / %d this step)" % (TRAFFIC_JSON, n_kern, n)) if traffic else None,
                "timing_note": "per-kernel durations are taken with the multi-stream overlap switched off (serial launches), "
                               "so they are clean but pessimistic w.r.t. the overlapped schedule that `value` is measured on", "launches_per_step": n, "gflop_per_launch": round(fl / max(n, 1) / 1e9, 3),
                "avg_launch_ms": round(ms / max(n, 1), 4), "gemm_ms_per_step": round(ms, 3),
                "attention_ms_per_step": round(ms_a, 3), "attention_tflops": round(fl_a / (ms_a * 1e-3) / 1e12, 2) if ms_a else None,
                "other_kernels_ms_per_step": round(ms_o, 3), "other_launches": n_o}
        if precision != "fp32":
            # `peak` is the guide's dense figure (2.4 GHz).  What the matrix cores SUSTAIN on this board with non-zero operands is set by its power cap:
            # a register-only v_mfma_f32_{16x16x32,32x32x16}_f16 loop (no LDS, no memory) runs 2 475 TFLOP/s on zeros (2.40 GHz, 0.9 kW) and
            # 1 670 TFLOP/s on random data (1.75 GHz at the 1.3 kW cap) -- tools/probes/r5/mfma_f16_peak.hip, profiles/round5/r5_mfma_f16_peak.txt
            roof["sustained_mfma_measured"] = {"value": SUSTAINED_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach * 2500.0 / PEAK[precision] / SUSTAINED_MFMA_TFLOPS, 4),
                                               "source": "profiles/round5/r5_mfma_f16_peak.txt: register-only f16 MFMA loop, random operands, 5 s, power-capped at 1.3 kW / 1.75 GHz (zeros: 2475 at 2.4 GHz); informational -- `frac` above is against `peak`"}
    ips = args.batch * world * steps / dt
    # algorithmic FLOPs per image.  compact: MLP forward AND backward on the kept tokens of the student pass;
    # masked: the student forward is dense (mask-multiply, as the reference), its backward is compacted all the same
    # (rows of mask * dL/dx' of dropped tokens are zero) -> 133.510 G at k = 0.7 with the backward's share of the slope (11 of 23
    # block-passes: the forward's 12 are dense, block 0 has no MLP backward).  (Round 2 charged the reference-as-written 139.614.)
    if mode == "compact":
        gflop = STEP_GFLOP_AT_07 + STEP_GFLOP_SLOPE * (keep_meas - 0.7)
    else:
        gflop = 133.510 + STEP_GFLOP_SLOPE * 11.0 / 23.0 * (keep_meas - 0.7)
    # what the library EXECUTES is less: the teacher pass reuses the student's patch embedding + block-0 attention branch
    # (DYT_OPT_SHARE_BLOCK0) and the last block's MLP / adapter run on the cls rows only, forward and backward (DYT_OPT_CLS_TAIL)
    mlp_tok = 9.437184e-3            # GFLOP per token of one MLP pass (fc1 + fc2)
    k_fwd = keep_meas if mode == "compact" else 1.0
    skipped = (0.231211 + 0.697172 + 2 * 0.059609 + 0.232391) + mlp_tok * 196 * (k_fwd + 1.0 + keep_meas + 1.0) \
        + 2 * 3 * 0.038732 * 196 / 197
    gflop_exec = gflop - skipped
    if args.video_frames > 1:   # + k/v projections of the pooling head per frame: fwd + dgrad + wgrad, two passes
        gflop += 2 * 3 * 2 * (2 * 197 * 768 * 768) / 1e9
        gflop_exec = gflop - (0.231211 + 0.697172 + 2 * 0.059609 + 0.232391)   # no cls-only tail in the video model
    res = {"value": round(ips, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "host_enqueue_ms_per_step": round(t_host / steps * 1e3, 3), "keep_ratio_measured": round(keep_meas, 4),
           "keep_ratio_calibrated": round(keep_cal, 4), "loss": round(host[0], 4), "step_gflop_per_image": round(gflop, 3),
           "step_mfma_frac": round(ips / world * gflop * 1e9 / (PEAK[precision] * 1e12), 4),
           "step_gflop_executed_per_image": round(gflop_exec, 3),
           "step_mfma_frac_executed": round(ips / world * gflop_exec * 1e9 / (PEAK[precision] * 1e12), 4), "roofline": roof,
           "hip_graph": bool(args.hip_graph) and not os.environ.get("DYT_NO_OVERLAP")}
    if dist.is_initialized():   # evidence for a multi-GPU record: the ranks RCCL itself saw, and that the all-reduce stream has a hardware queue of its own
        import _lib
        comm = getattr(eng, "_rccl_comm", None)
        res["rccl_ranks"] = _lib.rccl_comm_ranks(comm) if comm is not None else None
        res["comm_stream_concurrent"] = bool(eng.streams_concurrent(torch.cuda.current_stream(device), eng.comm_stream()))
    if host_batches > 0:
        try:   # a side measurement: it must never take the headline down with it (pinned allocations can be refused by a container's memlock limit)
            res["host_fed"] = host_fed(args, model, opt, steps, warmup, device, world, dt / steps * 1e3)
            log("[%s/%s] host-fed loop: %.2f ms/step prefetched, %.2f with the copy on the compute stream (resident %.2f)" % (
                precision, mode, res["host_fed"]["ms_per_step"], res["host_fed"]["copy_on_compute_stream_ms_per_step"], dt / steps * 1e3))
        except Exception as e:   # noqa: BLE001
            res["host_fed"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            log("[%s/%s] host-fed loop failed: %s" % (precision, mode, res["host_fed"]["error"]))
            torch.cuda.synchronize()
    del opt, model, eng
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
