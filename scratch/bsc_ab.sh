#!/bin/bash
# same-box A/B of the BSC EM loop under environment toggles: bsc_ab.sh "<env A>" "<env B>" ...
for rep in 1 2; do
for e in "$@"; do
  echo "== $e"; env $e python scratch/bsc_loop_prof.py 0 | tail -1
done
done
