cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --workload synth128_p7_provider --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03g_prov128.json 2> gpurun_out/r03g_prov128.err
tail -c 1500 gpurun_out/r03g_prov128.json; echo; tail -3 gpurun_out/r03g_prov128.err
PPP_BENCH_ONE_GPU=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --workload synth128_p7_provider --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03g_prov128_2ranks.json 2> gpurun_out/r03g_prov128_2ranks.err
tail -c 1200 gpurun_out/r03g_prov128_2ranks.json; echo; tail -3 gpurun_out/r03g_prov128_2ranks.err
PPP_BENCH_ONE_GPU=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --workload synth96_p7 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03g_res96_2ranks.json 2> gpurun_out/r03g_res96_2ranks.err
tail -c 800 gpurun_out/r03g_res96_2ranks.json; echo; tail -3 gpurun_out/r03g_res96_2ranks.err
timeout 600 python3 bench.py --workload synth96_p7 --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-north-star > gpurun_out/r03g_res96_1rank.json 2> gpurun_out/r03g_res96_1rank.err
python3 - <<'PY'
import json
for f in ("r03g_prov128","r03g_prov128_2ranks","r03g_res96_2ranks","r03g_res96_1rank"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], c["instances_found"], c["instances_crc32"], c["parallelism"], c.get("per_rank_peak_hbm_gb"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python3 bench.py --workload synth256_p9_provider --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03g_prov256.json 2> gpurun_out/r03g_prov256.err
tail -c 1500 gpurun_out/r03g_prov256.json; echo; tail -3 gpurun_out/r03g_prov256.err
timeout 900 python3 bench.py --workload synth256_p9 --steps 1 --warmup 0 --no-cpu-baseline --no-variants --no-north-star > gpurun_out/r03g_res256.json 2> gpurun_out/r03g_res256.err
python3 - <<'PY'
import json
for f in ("r03g_prov256","r03g_res256"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], c["instances_found"], c["instances_crc32"], c["parallelism"], c.get("per_rank_peak_hbm_gb"), d["stage_wall_ms"])
    except Exception as e: print(f, "ERR", e)
PY
