#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06d}
timeout 900 python -m pytest tests -m gpu -q -x -k "f16x1 or abi or range" 2>&1 | tail -15 > $OUT/pytest_$TAG.log
tail -5 $OUT/pytest_$TAG.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-other-configs --no-sweep --no-pmc > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
python - $OUT/bench_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps(d.get("f16x1"), indent=1))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
