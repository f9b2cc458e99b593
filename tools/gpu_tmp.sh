timeout 300 python tools/exp_power_gemm.py 2>&1 | grep -a "^{" 
