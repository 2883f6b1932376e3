cd $GRAFT_REPO_ROOT
(time python -m pytest tests -m gpu -q 2>&1) > gpurun_out/r2_08_gputests_all.log 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/r2_08_gputests_all.log | head
for path in 0 1; do python bench.py --workload c2is --path $path --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['config']['path'][:30], d['value'], d['roofline']['frac'], d['roofline']['phases_ms_per_step'])"; done
