import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/dynamic-tuning_amd"); sys.path.insert(0, "/root/repo/tests")
import torch, synth
from test_gpu_round2 import _bench_model
B=int(os.environ.get("BB","4")); what=os.environ.get("WHAT","step"); prec=os.environ.get("PP","fp16")
m,_=_bench_model(prec,"compact",B,0.85,kind="test")
eng=m.engine(B, torch.device("cuda",0))
x,y=synth.make_batch(B,100,seed=1); x=x.cuda(); y=y.cuda()
if what=="eval":
    m.eval()
    with torch.no_grad(): out,_=m(x)
    torch.cuda.synchronize(); print("eval ok", float(out.abs().sum()))
elif what=="fwd":
    m.train()
    lg,ts,tl=eng.forward(x, slot=0, training=True, save=True, seed=3)
    torch.cuda.synchronize(); print("fwd ok", float(lg.abs().sum()))
elif what=="fwdbwd":
    m.train()
    lg,ts,tl=eng.forward(x, slot=0, training=True, save=True, seed=3)
    torch.cuda.synchronize(); print("fwd ok", float(lg.abs().sum()), flush=True)
    g=torch.zeros_like(eng.flat); eng.backward(0, torch.randn_like(lg)*0.01, g)
    torch.cuda.synchronize(); print("bwd ok", float(g.abs().sum()))
else:
    m.train()
    out=eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=3)
    torch.cuda.synchronize()
    print("step ok", out.tolist()[:3], float(eng.grad.abs().sum()), bool(torch.isfinite(eng.grad).all()))
