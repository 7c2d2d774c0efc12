mkdir -p gpurun_out
timeout 600 python tools/hazard/product_check.py 1500 > gpurun_out/r03_product_check.txt 2>&1; echo "rc $?"
tail -8 gpurun_out/r03_product_check.txt | cut -c1-300
