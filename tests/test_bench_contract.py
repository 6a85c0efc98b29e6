"""bench.py's bookkeeping that does not need a GPU: the roofline traffic is taken from a committed
PMC profile only when it was taken from THIS tree's kernel sources and workload, the calibration
kernels' known byte counts correct the raw counters, and the workload table covers the
BASELINE.json configurations."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.fixture
def fake_profiles(tmp_path, monkeypatch):
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_sha16", lambda: "feedfacecafebeef")

    def write(tag, sha, workload, rows, **meta):
        base = tmp_path / "profiles" / (tag + "_bench_" + workload)
        (tmp_path / "profiles" / (tag + "_bench_" + workload + ".meta.json")).write_text(
            json.dumps(dict(src_sha16=sha, workload=workload, **meta)))
        (tmp_path / "profiles" / (tag + "_bench_" + workload + "_pmc_fetch_write.txt")).write_text(
            "\n".join("%-60s %-10s %20.1f  (n=%d)" % r for r in rows) + "\n")
        return base
    return write


def test_traffic_needs_a_profile_of_these_sources_and_this_workload(fake_profiles):
    rows = [("void ppp::consensus_v3_kernel<__half, 9, true>(__half const*", "FETCH_SIZE", 2.0e6, 4),
            ("void ppp::consensus_v3_kernel<__half, 9, true>(__half const*", "WRITE_SIZE", 1.0e6, 4)]
    fake_profiles("r09_a", "0123456789abcdef", "synth512_p9", rows)                 # other sources
    fake_profiles("r09_b", "feedfacecafebeef", "flylight140_p7", rows)              # other workload
    t = bench.pmc_traffic("consensus_v3_kernel", "synth512_p9")
    assert t["traffic"] is None and "1 from other sources" in t["traffic_note"]
    fake_profiles("r09_c", "feedfacecafebeef", "synth512_p9", rows)
    t = bench.pmc_traffic("consensus_v3_kernel", "synth512_p9")
    assert t["traffic_read"] == 2.0e6 * 1024 and t["traffic_write"] == 1.0e6 * 1024      # KiB -> bytes
    assert t["traffic"] == 3.0e6 * 1024 and "counter_over_true_bytes" not in t
    assert bench.pmc_traffic("rank_wg_kernel", "synth512_p9")["traffic"] is None         # kernel not in it


def test_calibration_kernels_correct_the_counters(fake_profiles):
    rows = [("void ppp::consensus_v3_kernel<__half, 9, true>(__half const*", "FETCH_SIZE", 4.0e6, 4),
            ("void ppp::consensus_v3_kernel<__half, 9, true>(__half const*", "WRITE_SIZE", 3.0e6, 4),
            ("void ppp::calib_read_kernel<__half>(__half const*, long long", "FETCH_SIZE", 2.0e6, 1),
            ("void ppp::calib_read_kernel<__half>(__half const*, long long", "WRITE_SIZE", 0.0, 1),
            ("ppp::calib_write_kernel(float*, long long)", "FETCH_SIZE", 1.0, 1),
            ("ppp::calib_write_kernel(float*, long long)", "WRITE_SIZE", 1.5e6, 1)]
    fake_profiles("r09_d", "feedfacecafebeef", "synth512_p9", rows,
                  calib_read_bytes=4.0e6 * 1024, calib_write_bytes=1.0e6 * 1024)
    t = bench.pmc_traffic("consensus_v3_kernel", "synth512_p9")
    assert t["counter_over_true_bytes"]["read"] == 0.5 and t["counter_over_true_bytes"]["write"] == 1.5
    # 4e6 KiB counted at half the true bytes, 3e6 KiB counted at 1.5x
    assert t["traffic_corrected"] == pytest.approx((8.0e6 + 2.0e6) * 1024)


def test_workloads_name_the_baseline_configurations():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5
    # configs [0]..[4] -> worm2d_p25, flylight140_p7, synth512_p9 (default), synth1024_p9, dec256_p7
    for name, shape, ps in (("worm2d_p25", (1, 520, 696), (1, 25, 25)), ("flylight140_p7", (140,) * 3, (7,) * 3),
                            ("synth512_p9", (512,) * 3, (9,) * 3), ("synth1024_p9", (1024,) * 3, (9,) * 3),
                            ("dec256_p7", (256,) * 3, (7,) * 3)):
        assert bench.WORKLOADS[name][0] == shape and bench.WORKLOADS[name][1] == ps
    assert bench.DEFAULT_WORKLOAD == "synth512_p9" and bench.FALLBACK_WORKLOAD == "flylight140_p7"
    assert "synth1024_p9" in bench.PROVIDER_WORKLOADS and "dec256_p7" in bench.DECODE_WORKLOADS


def _run_bench(args, env=None, timeout=600):
    import subprocess
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, timeout=timeout,
                          capture_output=True, text=True)


def test_gpus_n_must_match_the_ranks_that_run():
    """`--gpus 8` inside a 1-rank environment must not print a line that claims 8 GPUs."""
    r = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_plain_gpus_2_launches_two_ranks_and_fails_loudly_without_devices():
    """`python bench.py --gpus 2` with no launcher starts its ranks itself (a child
    torch.distributed.run); on a box with fewer than 2 devices every rank refuses and the exit code
    of the child is passed on -- no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices: the run would succeed")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "synth64_p5",
                    "--no-cpu-baseline"])
    assert "launching 2 ranks" in r.stderr and "--nproc-per-node 2" in r.stderr
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "needs a GPU" in r.stderr or "device(s) visible" in r.stderr


@pytest.mark.gpu
def test_self_launched_ranks_reproduce_the_one_rank_checksum():
    """`python bench.py --gpus 2` (ranks sharing device 0, gloo: PPP_BENCH_ONE_GPU=1) = the 1-rank
    run: same split-independent checksum, and the line says how many ranks the collectives span."""
    common = ["--workload", "synth64x2_p5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = _run_bench(common)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    two = _run_bench(["--gpus", "2"] + common, env={"PPP_BENCH_ONE_GPU": "1", "PPP_BENCH_RANK_HBM_GB": "40"})
    assert two.returncode == 0, two.stderr[-2000:]
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["config"]["collective_ranks"] == 2 and b["config"]["ranks"] == 2
    assert len(b["config"]["plan"]) == 2 and b["config"]["plan"][0]["own_z"][0] == 0
    assert a["config"]["instances_slice_crc32"] is not None
    assert a["config"]["instances_slice_crc32"] == b["config"]["instances_slice_crc32"]
    assert a["config"]["instances_found"] == b["config"]["instances_found"]
