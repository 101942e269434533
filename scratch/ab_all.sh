#!/bin/bash
# same-box A/B of two builds over the model benches: scratch/libA = old, in-tree = new
cp prosper_amd/libprosper_hip.so /tmp/new.so
for r in 1 2; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/new.so prosper_amd/libprosper_hip.so; else cp scratch/libA/libprosper_hip.so prosper_amd/libprosper_hip.so; fi
    echo "== $v"
    python scratch/em_loop.py 2>&1 | tail -1
    python scratch/bench_mca.py 2>&1 | tail -1
    python scratch/bench_dsc.py 2>&1 | tail -2 | head -1
    python scratch/bench_gsc_estep.py 2>&1 | tail -2 | head -1
  done
done
cp /tmp/new.so prosper_amd/libprosper_hip.so
