for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export RSI_HOT_LIB=$PWD/rsicnv_amd/librsi_hot_base.so; else unset RSI_HOT_LIB; fi
  for cfg in 2 3; do echo "== $lib cfg$cfg"; timeout -k 10 120 python tools/lone_phases.py $cfg 2>&1 | grep -v amdgpu.ids | cut -c1-1500; done
done; done
