#!/bin/bash
# run scratch/mca_T_sweep.py under rocprofv3 with each ablation library in place of the shipped one (restored afterwards)
cd /root/repo
cp prosper_amd/libprosper_hip.so /tmp/lib_keep.so
for a in 1 2 3; do
  cp scratch/libabl$a.so prosper_amd/libprosper_hip.so
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abl$a -o mca -- python3 /root/repo/scratch/mca_T_sweep.py > /tmp/log.txt 2>&1
  cd /root/repo; echo "ablation $a"; python3 scratch/prof_stats.py /tmp/prof_abl$a 24 | grep -i "scatter"
done
cp /tmp/lib_keep.so prosper_amd/libprosper_hip.so
