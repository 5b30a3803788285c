timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -o timeout_method=thread -k "split" 2>&1 | tail -40
