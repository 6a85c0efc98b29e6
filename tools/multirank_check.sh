#!/bin/bash
# Development check of the multi-rank bench paths on a ONE-GPU box: the ranks share device 0
# (PPP_BENCH_ONE_GPU=1, gloo through the host) -- timings mean nothing, the instance checksums
# must equal the 1-rank run's.  usage: tools/multirank_check.sh [workload] [ranks...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=${1:-synth256_p9_provider}; shift
ranks=${@:-2 4}
export PPP_BENCH_RANK_HBM_GB=${PPP_BENCH_RANK_HBM_GB:-50}
python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/mr_${wl}_1.json 2> gpurun_out/mr_${wl}_1.err
for n in $ranks; do
  PPP_BENCH_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n --workload $wl --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/mr_${wl}_$n.json 2> gpurun_out/mr_${wl}_$n.err
done
python3 - "$wl" 1 $ranks <<'PY'
import json, sys
wl = sys.argv[1]
for n in sys.argv[2:]:
    try:
        txt = [l for l in open("gpurun_out/mr_%s_%s.json" % (wl, n)) if l.startswith("{")][-1]
        d = json.loads(txt); c = d["config"]
        print(n, "ranks:", round(d["ms_per_step"]), "ms", c["instances_found"], c["instances_crc32"], "slice-crc", c.get("instances_slice_crc32"), c["parallelism"], c["per_rank_peak_hbm_gb"], c.get("result"))
    except Exception as e:
        print(n, "ranks: ERR", e)
PY
