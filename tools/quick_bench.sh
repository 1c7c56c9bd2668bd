python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_bf16.py tests/test_gpu_ops.py -q -x 2>&1 | tail -3
for a in "--dtype bf16s --batch 1024 --steps 40" "--dtype bf16 --batch 1024 --steps 40" "--dtype bf16s --batch 256 --steps 100" "--dtype f32 --steps 100"; do
echo "== bench $a"; python bench.py --no-cpu-baseline --no-kernel-rooflines $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
