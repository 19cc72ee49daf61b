#!/bin/bash
# same-box A/B of engine builds: tools/ab.sh <lib_a.so> <lib_b.so> [rounds]   (box-to-box variance is ~3 %)
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for L in $A $B; do TLD_LIB=$PWD/$L python tools/classes.py 2>/dev/null | tail -1; done
done
