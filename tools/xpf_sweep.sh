# direct 3x3x3 kernel: A/B of two library builds on one box.  usage: tools/xpf_sweep.sh libA.so libB.so
for lib in "$@"; do
 echo "== $lib"
 env OTAL_PREC=1 OTAL_LIB_PATH=$lib python tools/micro_conv.py 2c,3c_b1b,4f_b1b 20 fwd,dgrad 2>&1 | grep -v amdgpu
 env OTAL_LIB_PATH=$lib python tools/micro_planes6.py 20 3b_b1b,3c_b2b,4b_b1b,4c_b1b,4d_b1b,4e_b1b,4f_b2b 2>&1 | grep -v amdgpu | cut -c1-75
done
