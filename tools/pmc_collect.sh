#!/bin/bash
# Dev tool (GPU box): per-kernel PMC counters of an experiment library's fused launch, several passes of <= 4 counters.
#   tools/pmc_collect.sh NAME [WIDTH | nc3d | plate | plate70 | conf]     (library build/exp/NAME/libpinn_hip.so; output gpurun_out/pmc_NAME[_WIDTH]/)
# WIDTH (80, 100): the launches of tools/wide_time.py WIDTH NAME (1,000,000 points of the 8 x WIDTH net) instead of tools/exp_run.py (8x64, 2 M).
NAME=$1
WIDTH=$2
OUT=$PWD/gpurun_out/pmc_$NAME${WIDTH:+_$WIDTH}
if [ "$WIDTH" = "nc3d" ]; then CMD="$GRAFT_REPO_ROOT/tools/nc3d_time.py $NAME 6"; WHAT="fused 3-D kernel Fused<OpF16,3,128,10,5,false,4>, 1,000,000 points per launch (tools/nc3d_time.py)"
elif [ "$WIDTH" = "plate70" ]; then CMD="$GRAFT_REPO_ROOT/tools/plate_time.py 70 $NAME"; WHAT="plate collocation kernel of the reference 8 x 70 net, Fused<OpF16,3,96,8,5>, 1,000,000 points per launch (tools/plate_time.py 70)"
elif [ "$WIDTH" = "conf" ]; then CMD="$GRAFT_REPO_ROOT/tools/conf_time.py"; WHAT="the reference confined-domain net 6 x 140, Fused<OpF16,3,160,6,4>, 1,000,000 points per launch (tools/conf_time.py; the fused launches only)"
elif [ "$WIDTH" = "plate" ]; then CMD="$GRAFT_REPO_ROOT/tools/plate_time.py 64 $NAME"; WHAT="plate collocation kernel Fused<OpF16,3,64,8,5>, 1,000,000 points per launch (tools/plate_time.py)"
elif [ -n "$WIDTH" ]; then CMD="$GRAFT_REPO_ROOT/tools/wide_time.py $WIDTH $NAME"; WHAT="8x$WIDTH net, 1,000,000 points per launch (tools/wide_time.py)"
else CMD="$GRAFT_REPO_ROOT/tools/exp_run.py $NAME"; WHAT="fused_wave_kernel<OpF16,3,64,8,4>, 2,000,000 points per launch (tools/exp_run.py)"; fi
PTS=1000000; [ -z "$WIDTH" ] && PTS=2000000
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_IFETCH" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $CMD > $OUT/p$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py "$OUT" $PTS "$WHAT"
