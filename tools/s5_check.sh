#!/bin/bash
# parity of the per-patch S5 kernel (goldens, fresh inputs, the 96^3 dense list) and its time with
# the thinning masks made beforehand / inside the kernel.  usage: tools/s5_check.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-s5}
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "goldens or dense or per_patch or oracle" 2>&1 | tail -5
out=gpurun_out/${tag}.txt; : > $out
for wl in "flylight140_p7 shipped" "flylight140_p7 nothin_cc" "synth64_p5 shipped" "synth256_p9 shipped" "synth128_p9 nothin_cc"; do
  python3 tools/time_s5.py $wl 2>/dev/null | tail -1 >> $out
  PPP_PA_LCG_BYTES=0 python3 tools/time_s5.py $wl 2>/dev/null | tail -1 | sed 's/^/generator in the kernel: /' >> $out
done
cat $out
