import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
if os.environ.get("DYT_LIB_PATH"):
    _lib.LIB_PATH = os.environ["DYT_LIB_PATH"]
    _lib.SYMBOLS = {k: v for k, v in _lib.SYMBOLS.items() if k not in ("dyt_seed", "dyt_grad_part", "dyt_stream_wait_grads", "dyt_clip_grad_norm", "dyt_debug_dispatch", "dyt_gemm_f32_raw")}
import test_gpu_round2 as T
B = int(os.environ.get("PB", "4"))
prec = os.environ.get("PPREC", "bf16")
def run(overlap):
    m, _ = T._bench_model(prec, "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    x, y = x.cuda(), y.cuda()
    eng = m.engine(B, x.device)
    eng.set_option(_lib.OPT_STREAM_OVERLAP, overlap)
    if os.environ.get('PSHARE'): eng.set_option(_lib.OPT_SHARE_BLOCK0, int(os.environ['PSHARE']))
    if os.environ.get('PTAIL'): eng.set_option(_lib.OPT_CLS_TAIL, int(os.environ['PTAIL']))
    out = []
    for i in range(2):
        eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900 + i)
        torch.cuda.synchronize()
        out.append(eng.grad.clone())
    return out
for overlap in [int(v) for v in os.environ.get('POVERLAP', '1').split(',')]:
    runs = [run(overlap) for _ in range(4)]
    for i in (1, 2, 3):
        d = [(runs[0][k] - runs[i][k]).abs() for k in range(2)]
        print("overlap", overlap, "run0 vs run%d:" % i, [bool(torch.equal(runs[0][k], runs[i][k])) for k in range(2)],
              [float(x.max()) for x in d], [int((x > 0).sum()) for x in d], [int(x.argmax()) for x in d])
if os.environ.get("PDETAIL"):
    m, _ = T._bench_model(prec, "compact", B, 0.85)
    eng = m.engine(B, torch.device("cuda", 0))
    names = [n for n, p in m.named_parameters() if synth.is_trainable(n)]
    for i in (1, 2, 3):
        bad = []
        for n in names:
            off, num = eng.trainable_slice(n)
            d = (runs[0][0][off:off + num] - runs[i][0][off:off + num]).abs()
            if float(d.max()) > 0:
                bad.append("%s:%d/%d" % (n.replace("blocks.", "b").replace("adaptmlp.", "").replace("mlp_token_select.mlp_head", "gate"), int((d > 0).sum()), num))
        print("run0 vs run%d step0 differing tensors:" % i, bad[-14:])
