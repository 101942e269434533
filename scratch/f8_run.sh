#!/bin/bash
# run the harness over the variants present in scratch/libs (on the GPU box)
cd "$(dirname "$0")/.."
OUT=gpurun_out/f8_run.log
: > $OUT
B=scratch/libs/f8_bench
for v in "$@"; do
  name=${v%%:*}; rest=${v#*:}; tile=${rest%%:*}; ms=0; st=0
  [ "$rest" != "$tile" ] && ms=${rest#*:}
  case "$name" in stamps*) st=1;; esac
  timeout 120 $B scratch/libs/libpm_$name.so $tile ${F8_N:-196608} 20 $st $ms ${F8_PART:-0} 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
