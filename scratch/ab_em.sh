#!/bin/bash
# same-box A/B of the BSC EM loop: base copy (scratch/ab_base) vs the working tree and its toggles
for rep in 1 2; do
  echo "base:         $(python scratch/em_ab.py scratch/ab_base 2>/dev/null | grep ms/iter)"
  echo "new:          $(python scratch/em_ab.py 2>/dev/null | grep ms/iter)"
  echo "new gram_old: $(PM_AB_GRAM_OLD=1 python scratch/em_ab.py 2>/dev/null | grep ms/iter)"
  echo "new dl_main:  $(PM_AB_DL_MAIN=1 python scratch/em_ab.py 2>/dev/null | grep ms/iter)"
done
