#!/bin/bash
# same-box A/B of engine builds: tools/ab.sh <rounds> <lib_a.so> <lib_b.so> [...]   (box-to-box variance is ~3 %)
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do TLD_LIB=$PWD/$L python tools/classes.py 2>/dev/null | tail -1; done
done
