#!/bin/bash
# usage (on the GPU box): tools/pmc_multi.sh <tag> [program args...]   -- several rocprofv3 --pmc passes (kernel trace only) of
# `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline` (or the given program), one counter group per pass; the per-kernel
# means of the finest-level dispatches land in gpurun_out/<tag>_pmc_table.txt (tools/pmc_multi_table.py).
# At most FOUR counters of the TCP and of the TA block per pass: seven in one pass is more than the block can collect
# (rocprofiler_create_counter_config: error 38, "Request exceeds the capabilities of the hardware to collect" -> rocprofv3
# aborts with signal 6; round 3 misread that as a limitation of the pool).  A pass that fails stops the script.
# Round 4: with four per pass the TCP_* groups collect (profiles/r04_legs_pmc_multi.txt); the TA_* group still aborts rocprofv3
# (signal 6 in the counter set-up, then a hung tool: gpurun_out/r4legs/p7.err) even at four counters -- left out.
tag=${1:-pmc}; shift
prog=("$@"); [ ${#prog[@]} -eq 0 ] && prog=(python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline)
out=gpurun_out/$tag
mkdir -p $out
# (the GPU box exports GRAFT_REPO_ROOT; `cd ""` would leave every pass in /tmp with relative paths that do not exist there)
[ -n "$GRAFT_REPO_ROOT" ] && [ -d "$GRAFT_REPO_ROOT" ] || { echo "pmc_multi.sh: GRAFT_REPO_ROOT is not set (run this through gpurun)"; exit 2; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/p$i -o p -- "${prog[@]}" > /dev/null 2> $out/p$i.err || { echo "pass $i ($group) failed:"; tail -5 $out/p$i.err; exit 1; }
  ls $out/p$i/*counter_collection.csv > /dev/null 2>&1 || ls $out/p$i/*/*counter_collection.csv > /dev/null 2>&1 || { echo "pass $i ($group): no counter_collection.csv"; exit 1; }
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM
SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INST_CYCLES_VMEM_RD
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_avr
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum TCC_EA0_RDREQ_32B_sum
GROUPS
python3 tools/pmc_multi_table.py $out > gpurun_out/${tag}_pmc_table.txt 2>&1
find $out -name "*.csv" ! -name "*counter_collection.csv" -delete
tail -50 gpurun_out/${tag}_pmc_table.txt
