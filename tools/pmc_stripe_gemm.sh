set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3y; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -f csv -d $R/$O/p1 -- python $R/tools/run_prefill_once.py 1024 > $R/$O/p1.txt 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace -f csv -d $R/$O/p2 -- python $R/tools/run_prefill_once.py 1024 > $R/$O/p2.txt 2>&1
cd $R
C1=$(find $O/p1 -name "*counter_collection.csv" | head -1); C2=$(find $O/p2 -name "*counter_collection.csv" | head -1)
PMC_KERNEL=stripe_gemm_kernel PMC_NOTE="M = 1024, K = N = 4096, tools/run_prefill_once.py 1024" python tools/pmc_gemm.py $O/stripe_gemm_pmc.json "$C1" "$C2" | tail -28
rm -rf $O/p1 $O/p2
