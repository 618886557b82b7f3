#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#     tools/collect_profiles.sh r03_a
# -> gpurun_out/<tag>/{bench.json, kernel_stats.csv, overlap_kernel_stats.csv, pmc_whole_step_summary.json, *.log}
# Copy what is to be judged into profiles/ afterwards (profiles/README.md lists the commands).
# Counter passes are separate runs with --kernel-trace only (no sys / hip / hsa trace domains), the program directly after `--`.
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_single -- python3 bench.py --no-cpu-baseline --no-dre-extra --no-other-configs --no-host-boundary --no-step-graphs --no-wgrad-overlap > $out/bench_prof_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_overlap -- python3 bench.py --no-cpu-baseline --no-dre-extra --no-other-configs --no-host-boundary --no-step-graphs > $out/bench_prof_overlap.log 2>&1
cp $(ls $out/trace_single/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
cp $(ls $out/trace_overlap/*/*_kernel_stats.csv | head -1) $out/overlap_kernel_stats.csv
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-dre-extra --no-other-configs --no-wgrad-overlap --no-host-boundary --no-step-graphs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- $B > $out/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
python3 tools/pmc_summary.py --steps 3 $out/pmc_whole_step_summary.json $out/pmc_fetch/*/*counter_collection.csv $out/pmc_write/*/*counter_collection.csv $out/pmc_mfma/*/*counter_collection.csv $out/pmc_sq/*/*counter_collection.csv
rm -rf $out/trace_single $out/trace_overlap $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_sq
ls -la $out
