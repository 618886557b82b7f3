export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r05j
F="--no-cpu-baseline --no-other-configs --no-kernel-profile --no-dre-extra --steps 20 --warmup 5"
for r in 1 2 3; do
  (cd variants/r04tree && python bench.py $F 2>/dev/null | python -c "import json,sys; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-4 end tree (fc6fd84): %.3f ms per step = %.1f clips/s' % (p['ms_per_step'], p['value']))")
  python bench.py $F --no-host-boundary 2>/dev/null | python -c "import json,sys; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-5 tree:                %.3f ms per step = %.1f clips/s' % (p['ms_per_step'], p['value']))"
done | tee gpurun_out/r05j/r04_vs_r05.txt
