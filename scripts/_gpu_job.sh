cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -x -q -k "cnmf" 2>&1 | tail -3
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['phases_ms_per_step'])"
python bench.py --workload c4kl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['phases_ms_per_step'])"
NMFX_CNMF_NO_QGEMM=1 python bench.py --workload c4kl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['phases_ms_per_step'])"
