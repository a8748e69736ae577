# split-K for the last partial round of the pre-shuffled-weight kernel's N = 768 dgrads: results vs the unsplit launch, run-to-run bits, step A/B
python - <<'P'
import os, sys, subprocess
code = r'''
import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "dynamic-tuning_amd")
from test_gpu_round3 import _step
r = _step(sys.argv[2], "compact", 1, B=128)
torch.save({k: v for k, v in r.items()}, sys.argv[1])
'''
import torch
for prec in ("fp16", "bf16"):
    for tag, v in (("off", "0"), ("on1", "1"), ("on2", "1")):
        subprocess.check_call([sys.executable, "-c", code, "/tmp/sk_%s.pt" % tag, prec], env=dict(os.environ, DYT_BPRE_SK=v))
    a, b, b2 = torch.load("/tmp/sk_off.pt"), torch.load("/tmp/sk_on1.pt"), torch.load("/tmp/sk_on2.pt")
    for k in a:
        if torch.is_tensor(a[k]):
            print("%s %-8s on vs off: %s (max|d| %.2e); on run-to-run: %s" % (prec, k, "bitwise" if torch.equal(a[k], b[k]) else "differs", float((a[k].float() - b[k].float()).abs().max()),
                                                                            "bitwise" if torch.equal(b[k], b2[k]) else "DIFFERS"))
    worst = max(float((a["grads"][n] - b["grads"][n]).norm() / a["grads"][n].norm().clamp_min(1e-30)) for n in a["grads"])
    rr = sum(torch.equal(b["grads"][n], b2["grads"][n]) for n in b["grads"])
    print("%s 74 gradients: on vs off worst rel-L2 %.2e; on run-to-run bitwise equal: %d of %d" % (prec, worst, rr, len(b["grads"])))
P
for i in 1 2 3; do
for v in 0 1; do
DYT_BPRE_SK=$v python bench.py --no-cpu-baseline --no-parity-mode --steps 20 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/bpre_sk=$v /"
done; done
for v in 0 1; do
DYT_BPRE_SK=$v python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/bpre_sk=$v /"
done
