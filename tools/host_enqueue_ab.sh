# same-box A/B of the host's enqueue time per step: round-4 end tree (fc6fd84, ab_old/r04tree) vs this tree, alternating.
# host_enqueue_ms_per_step rose 13.4 -> 16.2 -> 23.1 across the builder's round-5 profiles (different boxes).
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r06a
F="--no-cpu-baseline --no-other-configs --no-kernel-profile --no-dre-extra --no-host-boundary --steps 20 --warmup 5"
P='import json,sys; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s: %.3f ms per step, host enqueue %.2f ms per step" % (sys.argv[1], p["ms_per_step"], p["host_enqueue_ms_per_step"]))'
nproc; grep -m1 "model name" /proc/cpuinfo; uptime
for r in 1 2 3; do
  (cd ab_old/r04tree && python bench.py $F 2>/dev/null | python -c "$P" "round-4 end tree (fc6fd84)")
  python bench.py $F 2>/dev/null | python -c "$P" "this tree                 "
done 2>&1 | tee gpurun_out/r06a/host_enqueue_r04_vs_head.txt
python tools/host_profile.py > gpurun_out/r06a/host_profile_head.txt 2>&1
(cd ab_old/r04tree && python tools/host_profile.py > ../../gpurun_out/r06a/host_profile_r04.txt 2>&1)
tail -5 gpurun_out/r06a/host_profile_head.txt
