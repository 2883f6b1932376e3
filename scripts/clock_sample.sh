#!/bin/bash
# sample the shader clock / power while a workload runs: scripts/clock_sample.sh <workload> <steps>   -> prints the sclk histogram seen by rocm-smi during the timed region
W=$1; STEPS=${2:-1500}
python bench.py --workload $W --steps $STEPS --warmup 50 --no-cpu-baseline --no-profile > /tmp/cs_$W.json 2>/dev/null &
pid=$!
: > /tmp/cs_$W.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" >> /tmp/cs_$W.txt
  sleep 0.1
done
python - <<PY
import re, json, collections
t=open("/tmp/cs_$W.txt").read()
clk=[int(x) for x in re.findall(r"sclk.*?\((\d+)Mhz\)", t)]
clk=[c for c in clk if c > 1000]   # the samples taken while the kernels ran
pw=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", t)]
d=json.loads(open("/tmp/cs_$W.json").read().strip().splitlines()[-1])
print("$W", "it/s", d["value"], "samples", len(clk), "sclk MHz min/median/max", (min(clk), sorted(clk)[len(clk)//2], max(clk)) if clk else None, "power W median", sorted(pw)[len(pw)//2] if pw else None)
print("   histogram", sorted(collections.Counter(clk).items()))
PY
