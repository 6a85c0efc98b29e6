#!/bin/bash
# the 512^3 step under a few environment settings on ONE box (boxes differ by up to 1.4 x in S2):
# usage: tools/bench_ab.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   (an empty string = the defaults)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/${tag}.txt
: > $out
for setting in "$@"; do
  env $setting PPP_BENCH_STAGES=0 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['kernel_ms']
print(json.dumps({'env': '''$setting''', 'ms_per_step': round(d['ms_per_step']), 'crc': d['config']['instances_crc32'],
                  'consensus': round(k.get('consensus', 0)), 'rank_patches': round(k.get('rank_patches', 0)), 'patch_graph': round(k.get('patch_graph', 0)),
                  'cover': round(k.get('cover', 0)), 'launches': {n: len(v) if isinstance(v, list) else None for n, v in {}.items()},
                  'rank_group': d['workload_stats'].get('rank_group'), 'rank_tile': d['workload_stats'].get('rank_tile'), 'trial': d['workload_stats'].get('rank_tile_trial_ns_per_centre'), 'ring_z_scores': d['workload_stats'].get('ring_z_scores'),
                  's1_launch_ms': round(d['roofline']['avg_ms'], 1), 's2_launch_ms': round(d['roofline_other_kernels']['rank_patches']['avg_ms'], 1),
                  's2_launches': d['roofline_other_kernels']['rank_patches']['launches']}))" >> $out
done
cat $out
