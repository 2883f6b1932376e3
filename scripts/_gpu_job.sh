cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "projfunc or nmfsc" 2>&1 | tail -3
python bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -c 900
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2_c5 -o c5 -- python bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r2_c5.log 2>&1
ls gpurun_out/prof_r2_c5 | head
