# PMC passes over single launches of conv_gemm_v3 (tools/prof_one.py): gpurun -- bash tools/_pmc3.sh [names...]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/pmc3; mkdir -p $O
NAMES="${@:-dec0.dg enc4.fwd enc3.fwd dec0.fwd0 dec0.fwd1}"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmcA -o run -- python3 $R/tools/prof_one.py $NAMES --reps 3 > $O/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmcB -o run -- python3 $R/tools/prof_one.py $NAMES --reps 3 > $O/b.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmcC -o run -- python3 $R/tools/prof_one.py $NAMES --reps 3 > $O/c.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcD -o run -- python3 $R/tools/prof_one.py $NAMES --reps 3 > $O/d.log 2>&1
cd $R; for p in A B C D; do python tools/pmc_summary.py $O/pmc$p conv_gemm_v > $O/sum$p.txt 2>&1; rm -rf $O/pmc$p; done; tail -2 $O/a.log; cat $O/sumA.txt $O/sumB.txt $O/sumC.txt $O/sumD.txt
