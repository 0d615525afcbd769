for n in 0 1 2 3 4 8 16 32; do
  if [ $n = 0 ]; then lib=cer-mvs_amd/csrc/libcermvs.so; else lib=cer-mvs_amd/csrc/variants/libcermvs_sxabl$n.so; fi
  echo "== SX_ABL=$n"; CER_MVS_LIB=$PWD/$lib timeout 120 python tools/bench_conv_s16.py --mt 4 --rounds 3 --reps 5 --only "z|r,q gru,delta" 2>&1 | grep -v amdgpu.ids
done
