# Round evidence: bench lines + rocprofv3 kernel stats + PMC passes (separate passes, --kernel-trace only), summaries into gpurun_out/.
# usage: bash tools/evidence.sh <tag>     e.g. r02b
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03a}
python bench.py > $O/${T}_bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu-baseline --no-trained-bias --no-c4 --no-rollout > $O/prof_stats.log 2>&1
cp $O/prof_stats/*/*_kernel_stats.csv $O/${T}_kernel_stats.csv
rm -rf $O/prof_stats
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-trained-bias --no-c4 --no-rollout > $O/pmc1.log 2>&1
echo "pmc1 done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-trained-bias --no-c4 --no-rollout > $O/pmc2.log 2>&1
echo "pmc2 done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc3 -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-trained-bias --no-c4 --no-rollout > $O/pmc3.log 2>&1
echo "pmc3 done"
python tools/pmc_summary.py "k_fused_ws<1>" $O/pmc1 $O/pmc2 $O/pmc3 > $O/${T}_pmc_summary.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
cat $O/${T}_pmc_summary.txt
# d = 128 (the reference's default width): bench + FETCH / WRITE passes
python bench.py --embed 128 --layers 2 --no-cpu-baseline > $O/${T}_d128_bench.json 2>> $O/bench.err
echo "d128 bench done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc4 -- python3 bench.py --embed 128 --layers 2 --steps 20 --warmup 10 --no-cpu-baseline --no-trained-bias > $O/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc5 -- python3 bench.py --embed 128 --layers 2 --steps 20 --warmup 10 --no-cpu-baseline --no-trained-bias > $O/pmc5.log 2>&1
python tools/pmc_summary.py "k_fused_cs<128, 128, 128" $O/pmc4 $O/pmc5 > $O/${T}_d128_pmc_summary.txt
rm -rf $O/pmc4 $O/pmc5
cat $O/${T}_d128_pmc_summary.txt
python bench.py --autoregressive --no-cpu-baseline --no-trained-bias --no-c4 > $O/${T}_autoregressive_bench.json 2>> $O/bench.err
python bench.py --workload c5 --steps 20 --warmup 5 > $O/${T}_c5_bench.json 2>> $O/bench.err
python bench.py --workload c4 > $O/${T}_c4_bench.json 2>> $O/bench.err
python bench.py --workload small > $O/${T}_small_bench.json 2>> $O/bench.err
echo "all done"
