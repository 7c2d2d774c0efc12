timeout 600 python -m pytest tests/test_c_abi.py tests/test_gpu_fusion.py tests/test_gpu_model.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -4
