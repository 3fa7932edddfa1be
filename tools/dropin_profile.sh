#!/bin/bash
# the slot-shaped path's two profiles (profiles/README.md): host wall time by library call, and the device's timeline of the same leg
#   bash tools/dropin_profile.sh <tag>   -> gpurun_out/<tag>_dropin_host_profile.txt, gpurun_out/<tag>_dropin_timeline.txt
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
T2GPU_RX_PROF=1 T2GPU_DEMOD_PROF=1 timeout 300 python bench.py --only-drop-in --saturate 2> gpurun_out/${TAG}_dropin_host_profile.txt | cut -c1-120
T2GPU_RX_PROF=1 T2GPU_DEMOD_PROF=1 timeout 300 python bench.py --only-drop-in 2> gpurun_out/${TAG}_dropin_host_profile_reference_cast.txt | cut -c1-120
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/dropin_tl
T2GPU_DROPIN_WRAPPER="rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/dropin_tl --" timeout 400 python $R/bench.py --only-drop-in 2>/dev/null | cut -c1-120
cd $R
python tools/dropin_timeline.py gpurun_out/dropin_tl > gpurun_out/${TAG}_dropin_timeline.txt 2>&1
rm -rf gpurun_out/dropin_tl
