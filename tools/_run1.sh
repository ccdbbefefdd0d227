set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_glue.py -m gpu -x -q -k "multi" 2>&1 | tail -5
