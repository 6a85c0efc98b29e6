"""Run the default bench step with the instrumented per-patch S5 kernel (build the library with
PPP_EXTRA_FLAGS=-DPA_STATS first) and print its work counters."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + sys.argv[1:]
import bench  # noqa: E402
from patchperpix_amd import backend  # noqa: E402

bench.main()
out = (ctypes.c_ulonglong * 8)()
backend.lib().ppp_pa_stats(out)
steps, planes, rows, useful, lcg, live = [int(v) for v in out[:6]]
print("wave-steps %d  planes/step %.2f  rows/step %.2f  lcg rows/step %.2f" %
      (steps, planes / steps, rows / steps, lcg / steps))
wl = [a for a in sys.argv if a in bench.WORKLOADS]
px = bench.WORKLOADS[wl[0]][1][2] if wl else 7          # patch width of the workload (7: the default one)
print("useful slots / (rows*64*px) = %.3f   live lanes = %.3f" %
      (useful / (rows * 64.0 * px), live / (steps * 64.0)))
