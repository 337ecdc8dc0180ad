#!/bin/bash
# One GPU-box round trip: GPU parity tests, smoke, a short bench and a rocprofv3 kernel trace.
# Everything is logged under gpurun_out/ (merged back by gpurun).  Usage:
#   gpurun --timeout 1800 -- 'bash tools/gpu_check.sh [tests|bench|prof|all]'
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx9" | sort | uniq -c; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket") > gpurun_out/box.txt 2>&1
if [[ $what == all || $what == tests ]]; then
  rm -f gpurun_out/test_report.txt
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -n 40 gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
  tail -n 4 gpurun_out/smoke.log
fi
if [[ $what == all || $what == tune ]]; then
  timeout 600 python tools/tune_conv.py > gpurun_out/tune_conv.log 2>&1; echo "tune rc=$?"; head -n 50 gpurun_out/tune_conv.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench rc=$?"; cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
fi
if [[ $what == all || $what == prof ]]; then
  rm -rf gpurun_out/prof
  export DVC_AUTOTUNE_CACHE=$PWD/gpurun_out/autotune.json
  timeout 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-speed-leg --refs 0 --clips 0 > /dev/null 2>&1   # fills the autotune cache
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-speed-leg --refs 0 --clips 0 > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
  echo "prof rc=$?"; cat gpurun_out/prof_bench.json
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [[ -n "$f" ]] && head -n 40 "$f"
  # keep only the summaries (the raw trace is large)
  find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
fi
unset DVC_AUTOTUNE_CACHE
if [[ $what == pmc || $what == all2 ]]; then
  rm -rf gpurun_out/pmc*; 
  sha256sum deep-exemplar-based-video-colorization_amd/csrc/corr.hip | cut -c1-16 > gpurun_out/corr_hip_sha.txt
  rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc1 -o p -- python tools/prof_kernels.py > gpurun_out/pmc1.log 2>&1; echo "pmc1 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc2 -o p -- python tools/prof_kernels.py > gpurun_out/pmc2.log 2>&1; echo "pmc2 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc3 -o p -- python tools/prof_kernels.py > gpurun_out/pmc3.log 2>&1; echo "pmc3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc4 -o p -- python tools/prof_kernels.py > gpurun_out/pmc4.log 2>&1; echo "pmc4 rc=$?"
  ls gpurun_out/pmc*/ ; tail -n 3 gpurun_out/pmc1.log; tail -n 3 gpurun_out/pmc4.log
fi
if [[ $what == pmcr ]]; then
  # the same counters on ColorVidNet's layer shapes at batch 4 under the batch-aware plan (multi-reference pass)
  rm -rf gpurun_out/pmcr*
  export DVC_PROF_R=4
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcr1 -o p -- python tools/prof_kernels.py > gpurun_out/pmcr1.log 2>&1; echo "pmcr1 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcr2 -o p -- python tools/prof_kernels.py > gpurun_out/pmcr2.log 2>&1; echo "pmcr2 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcr3 -o p -- python tools/prof_kernels.py > gpurun_out/pmcr3.log 2>&1; echo "pmcr3 rc=$?"
  unset DVC_PROF_R
  tail -n 3 gpurun_out/pmcr1.log
fi
