cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 2 4; do
PPP_BENCH_ONE_GPU=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2965$n bench.py --gpus $n --workload synth256_p9 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03t_s256_${n}ranks.json 2> gpurun_out/r03t_s256_${n}ranks.err
done
timeout 600 python -m pytest tests/test_cli_gpu.py -q -m gpu > gpurun_out/r03t_tests.txt 2>&1; tail -3 gpurun_out/r03t_tests.txt
python3 - <<'PY'
import json
for f in ("r03t_s256_2ranks","r03t_s256_4ranks"):
    try:
        txt=[l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1]
        d=json.loads(txt); c=d["config"]
        print(f, round(d["ms_per_step"],1), d["n_gpus"], d["scaling"], c["instances_found"], c["instances_crc32"], c["parallelism"], c["per_rank_peak_hbm_gb"])
        print("   ", {k: round(v) for k,v in d["stage_wall_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
grep -v "^\[W\|amdgpu.ids\|^$" gpurun_out/r03t_s256_4ranks.err | grep -i "error\|Traceback" | head -5
