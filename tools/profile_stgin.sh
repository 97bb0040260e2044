#!/bin/bash
# rocprofv3 kernel stats of the ST-GIN train step (diagnostic): bash tools/profile_stgin.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_stgin
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/time_engines.py stgin > $O/log.txt 2>&1
tail -2 $O/log.txt
F=$(ls $O/trace/*/*_kernel_stats.csv | tail -1)
head -16 $F | cut -c1-200
