for i in 1 2 3 4 5 6 7 8 9 10 11 12; do python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_feature and bf16x3" 2>&1 | grep -E "AssertionError|passed|failed" | head -2; done
