for x in 0 1 2 4 7; do
 echo "== XPF2=$x"
 OTAL_PREC=1 OTAL_CONV_DIRECT_XPF2=$x python tools/micro_conv.py 2c,3c_b1b,4f_b1b 20 fwd,dgrad 2>&1 | grep -v amdgpu
 OTAL_CONV_DIRECT_XPF2=$x python tools/micro_planes6.py 20 3b_b1b,4b_b1b,4c_b1b,4d_b1b,4e_b1b 2>&1 | grep -v amdgpu | cut -c1-75
done
OTAL_CONV_DIRECT_XPF2=7 python -m pytest tests/test_ops_gpu.py -x -q -k "direct or bf16_operands" 2>&1 | tail -2
