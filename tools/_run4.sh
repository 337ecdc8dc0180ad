mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "conv2d_ws or routes_direct" > gpurun_out/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_sel.log
timeout 400 python tools/conv_algo_sweep.py > gpurun_out/conv_algo_sweep.txt 2>&1; echo "sweep rc=$?"; grep -E "ws|per frame" gpurun_out/conv_algo_sweep.txt
