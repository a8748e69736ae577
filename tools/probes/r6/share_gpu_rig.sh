# The driver's N > 1 launch line with real processes on a 1-GPU box (DYT_BENCH_SHARE_GPU: all ranks on cuda:0, gloo).  Not a measurement:
# a check that the multi-rank control flow (barriers, broadcast, all-reduce, rank-0-only sections) completes and prints ONE JSON line.
port=29700
for n in 2 4 8; do
port=$((port+1))
echo "=== N=$n"
DYT_BENCH_WATCHDOG=240 DYT_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 3 --warmup 1 > /tmp/rig_$n.out 2> /tmp/rig_$n.err
echo "rc=$? json_lines=$(grep -c '^{' /tmp/rig_$n.out)"; tail -1 /tmp/rig_$n.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['global_batch'], d['config']['parallelism'][:40], d.get('distributed'))"
grep -n "Timeout\|Error\|FAILED" /tmp/rig_$n.err | head -5
done
echo "=== N=2, ranks try the native RCCL communicator first (must fall back collectively)"
DYT_BENCH_WATCHDOG=240 DYT_BENCH_SHARE_GPU=native timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29710 bench.py --gpus 2 --steps 3 --warmup 1 > /tmp/rig_nat.out 2> /tmp/rig_nat.err
echo "rc=$? json_lines=$(grep -c '^{' /tmp/rig_nat.out)"; grep -n "Timeout\|Error\|FAILED\|fall\|RCCL\|rccl" /tmp/rig_nat.err | head -8
