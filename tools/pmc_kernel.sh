#!/bin/bash
# PMC counters (three passes) of one kernel.
# usage: tools/pmc_kernel.sh <kernel regex> <tag> [program args...]
# The profiled program is python3 with the given arguments (default: one step of bench.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
k=$1; tag=$2; shift 2
if [ $# -eq 0 ]; then set -- bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-variants; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_1 -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/${tag}_2 -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_3 -- python3 "$@" > /dev/null 2>&1
for i in 1 2 3; do python3 tools/summarize_prof.py gpurun_out/${tag}_$i gpurun_out/${tag}_$i.txt | grep -E "$k"; rm -rf gpurun_out/${tag}_$i; done
