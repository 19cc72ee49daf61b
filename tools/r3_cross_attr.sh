for dbg in 0 1 3 0 1 3; do echo "== cross dbg=$dbg"; TLD_CROSS_DBG=$dbg timeout 300 python tools/classes.py 2>/dev/null | tail -1; done
