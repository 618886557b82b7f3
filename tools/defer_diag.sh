# needs the prototype: git apply tools/gemm_defer_prototype.patch && python 2023-tifs-istvt_amd/build.py, then
#   for d in 2 3; do tools/build_variant.sh ab_old/lib_defer_d$d.so gemm.hip -DISTVT_Q_DEFER_DIAG=$d; done
# deferred-epilogue prototype (gemm256q.h DEFER): the A/B of the shipped build, then the timing-diagnostic variants
# (ab_old/lib_defer_d{1..4}.so: -DISTVT_Q_DEFER_DIAG=n; their results are wrong by construction)
export PYTHONUNBUFFERED=1 GB_WHICH=none GB_DEFER_AB=1 GB_REPS=10
mkdir -p gpurun_out/r06d
timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
for d in 2 3; do
  echo "== DEFER_DIAG=$d (1 passes empty, 2 LDS round trip only, 3 stores only, 4 no wait states behind the stores)"
  GB_LIB=ab_old/lib_defer_d$d.so timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^plain"
done
