set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
python bench.py > $O/r02a_bench.json 2> $O/bench.err
tail -c 600 $O/r02a_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-trained-bias > $O/prof_stats.log 2>&1
cp $O/prof_stats/*/*_kernel_stats.csv $O/r02a_kernel_stats.csv
rm -rf $O/prof_stats
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-trained-bias > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-trained-bias > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc3 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-trained-bias > $O/pmc3.log 2>&1
python tools/pmc_summary.py "k_fused_tile<64, 64, 1>" $O/pmc1 $O/pmc2 $O/pmc3 > $O/r02a_pmc_summary.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
cat $O/r02a_pmc_summary.txt
