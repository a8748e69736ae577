"""Probe: which scheduling option breaks hipGraph capture of the step.  usage: graph_probe.py overlap share tail"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
import test_gpu_round2 as T
overlap, share, tail = (int(v) for v in sys.argv[1:4])
B = 4
m, _ = T._bench_model("bf16", "compact", B, 0.85)
m.train()
x, y = synth.make_batch(B, 100, seed=61)
x, y = x.cuda(), y.cuda()
eng = m.engine(B, x.device)
eng.set_option(_lib.OPT_STREAM_OVERLAP, overlap)
eng.set_option(_lib.OPT_SHARE_BLOCK0, share)
eng.set_option(_lib.OPT_CLS_TAIL, tail)
out = eng.step_graph(x, y, seed=5)
torch.cuda.synchronize()
print("captured + replayed ok", sys.argv[1:4], out.tolist()[:3], flush=True)
out = eng.step_graph(x, y, seed=5)
torch.cuda.synchronize()
print("second replay ok", out.tolist()[:3], flush=True)
