# SQ counters per kernel (run on the GPU box): bash tools/pmc_sq.sh <tag> [bench.py arguments ...]
#   default arguments: the C2 step, one frame in flight.  C3 (the 9-7 kernels): bash tools/pmc_sq.sh sqc3 --config c3 --inflight 1 --batch 1
# The program itself follows `--` (no env / bash -c hop: the profiler's preload has initialised the GPU by then), variables are exported HERE.
cd /tmp; export TMPDIR=/tmp; export J2K_TUNING=1
R=$GRAFT_REPO_ROOT
TAG=${1:-sq}; shift
ARGS=${@:---inflight 1}
case "$ARGS" in *c3*|*c1gpu*) export GPU_MAX_HW_QUEUES=32;; esac
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $ARGS > $R/gpurun_out/${TAG}_$i.log 2>&1
  echo "set $i ($set): $(tail -c 200 $R/gpurun_out/${TAG}_$i.log | tr '\n' ' ' | cut -c1-160)"
done
