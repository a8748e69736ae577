for i in 1 2 3; do
python -m pytest tests/test_gpu_round2.py -x -q -s -k "adapter_submodule" 2>&1 | grep -E "adapter_bwd|passed|failed" | grep -E "197, 768|passed|failed|, 768\)" | tr '\n' ';' ; echo
done
