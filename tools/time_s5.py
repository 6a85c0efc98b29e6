"""One bench step of a workload (default synth256_p9) and the time of the patch-graph kernel in it.
Usage: python3 tools/time_s5.py [workload [flags]]   (PPP_LIB selects the library)"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "synth256_p9"
flags = sys.argv[2] if len(sys.argv) > 2 else "shipped"
sys.argv = ["bench.py", "--workload", wl, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--flags", flags]
import bench  # noqa: E402

buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
k = d.get("kernel_ms", {})
print("%s %s %s: step %.0f ms, patch_graph %.1f ms, masks beforehand %.1f ms, crc %s" % (
    os.path.basename(os.environ.get("PPP_LIB", "shipped")), wl, flags, d["ms_per_step"],
    k.get("patch_graph", float("nan")), k.get("patch_graph_lcg", 0.0),
    d.get("config", {}).get("instances_slice_crc32", d.get("instances_slice_crc32"))))
