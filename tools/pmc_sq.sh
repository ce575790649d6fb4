# SQ counters for one kernel family (run on the GPU box): bash tools/pmc_sq.sh <tag>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-sq}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$i -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${TAG}_$i.log 2>&1
done
