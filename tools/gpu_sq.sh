#!/bin/bash
# SQ counters of the conv kernels on the real layers of one step (tools/conv_bench.py) -> gpurun_out/<tag>/sq.txt
tag=${1:-sq}
shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/sq_$tag -o sq -- python /root/repo/tools/${BENCH:-conv_bench.py} "$@" > /root/repo/gpurun_out/$tag/conv_bench.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq2_$tag -o sq -- python /root/repo/tools/${BENCH:-conv_bench.py} "$@" > /root/repo/gpurun_out/$tag/conv_bench2.txt 2>&1
cd /root/repo
python tools/sq_counters.py /tmp/sq_$tag > gpurun_out/$tag/sq.txt 2>&1
python tools/sq_counters.py /tmp/sq2_$tag > gpurun_out/$tag/sq2.txt 2>&1
tail -3 gpurun_out/$tag/conv_bench.txt
head -40 gpurun_out/$tag/sq.txt
head -40 gpurun_out/$tag/sq2.txt
