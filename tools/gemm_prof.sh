# rocprofv3 recipe for k_remainder_gemm (kernel stats + three --pmc passes; TCP_TCC_READ_REQ_sum is NOT collectable on this image: it aborts rocprofv3)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
cat > /tmp/gm.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gnn_uds_amd import _lib
dev = torch.device('cuda:0')
N, E, S = 10000, 12000, 60
rest = (torch.randn(N, E) * 0.01).to(dev)
x = torch.randn(S, E, 32).to(dev)
packed = _lib.remainder_pack(rest)
for _ in range(4):
    _lib.remainder_forward(packed, (N, E), x)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/g0 -- python3 /tmp/gm.py > $O/g0.log 2>&1
grep -h "remainder\|split" $O/g0/*/*_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $O/g1 -- python3 /tmp/gm.py > $O/g1.log 2>&1 || tail -5 $O/g1.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/g2 -- python3 /tmp/gm.py > $O/g2.log 2>&1 || tail -5 $O/g2.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/g3 -- python3 /tmp/gm.py > $O/g3.log 2>&1 || tail -5 $O/g3.log
python tools/pmc_summary.py k_remainder_gemm $O/g1 $O/g2 $O/g3
rm -rf $O/g0 $O/g1 $O/g2 $O/g3
