set -e
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unordered" 2>&1 | tail -2
timeout -k 10 200 python tools/sk_probe.py 3e9 31 3 2>&1 | tail -1 | cut -c1-900
