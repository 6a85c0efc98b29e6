#!/bin/bash
# PMC passes of the S2 kernel alone (tools/time_s2.py).  usage: tools/pmc_s2.sh <tag> [case]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; case=${2:-128p9}
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "TC[CP]_[A-Z0-9_]*\(sum\)\?" | sort -u | tr '\n' ' ' > $out/avail_tc.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 tools/time_s2.py --case $case --reps 1 > /dev/null 2>$out/err$i.txt
  python3 tools/summarize_prof.py $out/p$i $out/p$i.txt | grep rank_wg_kernel >> $out/summary.txt
  rm -rf $out/p$i
done
cat $out/summary.txt
