cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r03s_all_gpu_tests.txt 2>&1
tail -4 gpurun_out/r03s_all_gpu_tests.txt
timeout 600 python3 bench.py --workload synth256_p9 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03s_s256_1rank.json 2> gpurun_out/r03s_s256_1rank.err
for n in 2 4; do
PPP_BENCH_ONE_GPU=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2964$n bench.py --gpus $n --workload synth256_p9 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03s_s256_${n}ranks.json 2> gpurun_out/r03s_s256_${n}ranks.err
done
python3 - <<'PY'
import json
for f in ("r03s_s256_1rank","r03s_s256_2ranks","r03s_s256_4ranks"):
    try:
        txt=[l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1]
        d=json.loads(txt); c=d["config"]
        print(f, round(d["ms_per_step"],1), d["n_gpus"], d["scaling"], c["instances_found"], c["instances_crc32"], c["parallelism"], c["per_rank_peak_hbm_gb"])
        print("   ", {k: round(v) for k,v in d["stage_wall_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r03s_s256_4ranks.err
