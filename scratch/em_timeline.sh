#!/bin/bash
# device timeline of one steady EM iteration of BSC config 2 (kernels + copies, gaps between them)
R=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o tl -- python3 $R/scratch/em_loop.py > /tmp/tl.log 2>&1 || tail -3 /tmp/tl.log
python3 $R/scratch/timeline.py /tmp/tl | tail -40
