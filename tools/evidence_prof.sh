# rocprofv3 passes of the headline launches only (no c4 / rollout / trained-bias legs): kernel stats + PMC, summaries into gpurun_out/.
# usage: bash tools/evidence_prof.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03b}
F="--no-cpu-baseline --no-trained-bias --no-c4 --no-rollout"
python bench.py $F > $O/${T}_bench_headline_only.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py $F > $O/prof_stats.log 2>&1
cp $O/prof_stats/*/*_kernel_stats.csv $O/${T}_kernel_stats.csv
rm -rf $O/prof_stats
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 bench.py --steps 20 --warmup 20 $F > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -- python3 bench.py --steps 20 --warmup 20 $F > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc3 -- python3 bench.py --steps 20 --warmup 20 $F > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmc4 -- python3 bench.py --steps 20 --warmup 20 $F > $O/pmc4.log 2>&1 || echo "pmc4 failed"
python tools/pmc_summary.py "k_fused_ws<1>" $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/${T}_pmc_summary.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
cat $O/${T}_pmc_summary.txt
head -3 $O/${T}_kernel_stats.csv
