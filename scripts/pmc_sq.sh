set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timing > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"; tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log
done
