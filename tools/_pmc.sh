cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; mkdir -p $O
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmcA -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/a.log 2>&1
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmcB -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/b.log 2>&1
SEHIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmcC -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/c.log 2>&1
cd $R; for p in A B C; do python tools/pmc_summary.py $O/pmc$p conv_gemm_v2 > $O/sum$p.txt 2>&1; rm -f $O/pmc$p/*kernel_trace.csv $O/pmc$p/*/*kernel_trace.csv; done; tail -3 $O/a.log; cat $O/sumA.txt $O/sumB.txt $O/sumC.txt
