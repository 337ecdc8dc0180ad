#!/bin/bash
# Round-end artefacts in one GPU-box round trip: bench lines (default, 432x768, bf16 correlation, torchrun single rank),
# per-layer direct-vs-Winograd sweep, rocprofv3 kernel stats of the bench command, the per-frame busy probe, the PMC passes
# (batch 1 and the multi-reference pass's batch 4), and the r04 probes (correlation stage alone under the kernel trace, where the
# multi-reference pass spends its time, training-side numbers).  `python tools/summarize_profiles.py rNN` then copies the judged
# summaries into profiles/.
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_check.sh bench
timeout 300 python bench.py --steps 30 --warmup 5 --hw 432x768 --refs 0 --clips 0 > gpurun_out/bench_432x768.json 2> gpurun_out/bench_432x768.err; echo "bench 432x768 rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --corr bf16 --no-cpu-baseline > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "bench bf16 rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --refs 0 --clips 0 > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "bench torchrun rc=$?"
timeout 400 python tools/conv_algo_sweep.py > gpurun_out/conv_algo_sweep.txt 2>&1; echo "sweep rc=$?"; tail -1 gpurun_out/conv_algo_sweep.txt
bash tools/gpu_check.sh prof > gpurun_out/prof_stdout.txt 2>&1; echo "prof done"
bash tools/busy_probe.sh > gpurun_out/busy_probe.txt 2>&1; tail -2 gpurun_out/busy_probe.txt
bash tools/gpu_check.sh pmc > gpurun_out/pmc_stdout.txt 2>&1; echo "pmc done"
bash tools/gpu_check.sh pmcr > gpurun_out/pmcr_stdout.txt 2>&1; echo "pmcr done"
timeout 200 python tools/corr_roofline_probe.py > gpurun_out/corr_roofline_probe.txt 2>&1; tail -1 gpurun_out/corr_roofline_probe.txt
rm -rf gpurun_out/corrprof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/corrprof -o t -- python tools/corr_roofline_probe.py > /dev/null 2>&1
timeout 300 python tools/refs_chain_probe.py > gpurun_out/refs_chain_probe.txt 2>&1; tail -3 gpurun_out/refs_chain_probe.txt
(echo "# default: plain batched GEMMs through the vendor library (ops.bmm)"; timeout 600 python tools/training_side_probe.py; echo; echo "# DVC_GEMM_LIB=0: the same products on the 1x1-convolution engine"; DVC_GEMM_LIB=0 timeout 600 python tools/training_side_probe.py) > gpurun_out/training_side_probe.txt 2>&1; tail -3 gpurun_out/training_side_probe.txt
timeout 200 python tools/gemm_lib_probe.py > gpurun_out/gemm_lib_probe.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +6M -delete
# r05: the tail probe (FGS scan solver), the bf16 bench line's own kernel trace (its roofline block cites it)
timeout 300 python tools/tail_probe.py > gpurun_out/tail_probe.txt 2>&1; tail -9 gpurun_out/tail_probe.txt
rm -rf gpurun_out/prof_bf16; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bf16 -o trace -- python bench.py --steps 20 --warmup 3 --corr bf16 --no-cpu-baseline --no-speed-leg --refs 0 --clips 0 > gpurun_out/prof_bf16_bench.json 2> gpurun_out/prof_bf16.err; echo "prof bf16 rc=$?"
find gpurun_out -name "*kernel_trace.csv" -size +6M -delete
rm -rf gpurun_out/tailprof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tailprof -o t -- python tools/tail_trace.py > /dev/null 2>&1; echo "tail trace rc=$?"
timeout 300 python tools/tail_insitu_probe.py 2>&1 | grep round > gpurun_out/tail_insitu_probe.txt; tail -3 gpurun_out/tail_insitu_probe.txt
find gpurun_out -name "*kernel_trace.csv" -size +6M -delete
