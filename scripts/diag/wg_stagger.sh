for a in 0 0.03 0.06 0.1 0.15; do echo "stagger $a"; FGS_WGRAD_STAGGER=$a python scripts/wgrad_bench.py 2>/dev/null | grep "wgrad"; done
