#!/bin/bash
# Hardware-counter passes for the dominant kernel, as MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc runs (SQ block, then
# FETCH_SIZE, then WRITE_SIZE -- the two TCC counters do not fit one pass), each with --kernel-trace only. Run on the GPU box:
#   bash tools/pmc_passes.sh <tag>     -> gpurun_out/pmc_<tag>_{sq,fetch,write}/ and gpurun_out/pmc_<tag>.txt (summary)
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD="python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$1 -o p -- $CMD > $ROOT/gpurun_out/pmc_${TAG}_$1.json 2> $ROOT/gpurun_out/pmc_${TAG}_$1.err || true; }
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
cd $ROOT
python tools/pmc_summary.py $TAG > gpurun_out/pmc_${TAG}.txt
cat gpurun_out/pmc_${TAG}.txt
