#!/bin/bash
# consensus cache: parity tests + synth256_p9 with and without the cache.  usage: tools/cache_check.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_large_golden.py -q -m gpu -x -k "cache or consensus_part or large_case" > gpurun_out/${tag}_tests.txt 2>&1
tail -5 gpurun_out/${tag}_tests.txt
for mode in auto 0; do
  PPP_CONS_CACHE=$mode timeout 900 python bench.py --workload synth256_p9 --steps 1 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/${tag}_256_cache_$mode.json 2> gpurun_out/${tag}_256_cache_$mode.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_256_cache_$mode.json").read().strip().splitlines()[-1])
    print("cache=$mode", d["ms_per_step"], d.get("instances_crc32"), d.get("stage_wall_ms"), d["config"].get("tiles"), d.get("notes", {}).get("cons_cache_gb"))
except Exception as e:
    print("cache=$mode failed", e); print(open("gpurun_out/${tag}_256_cache_$mode.err").read()[-1500:])
PY
done
