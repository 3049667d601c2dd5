export DSS_LPCNET_SYNTHETIC=1
for m in "0,1,3,2,5,4" "1,0,5,4,2,3" "0,1,5,4,2,3" "0,1,4,5,2,3" "0,1,5,4,3,2" "1,0,4,5,3,2"; do
  echo "== rank_wave_h $m"; DSS_RANK_WAVE_H=$m timeout -k 10 120 python tools/ab_time.py --child delayed-speech-synthesis_amd/libdss_hip.so 2>&1 | grep -v amdgpu.ids
done
for lib in gb_80_96_128 gb_96_96_112 gb_80_104_104; do
  for m in "0,1,3,2,5,4" "1,0,5,4,2,3"; do echo "== $lib rank_wave_h $m"; DSS_RANK_WAVE_H=$m timeout -k 10 120 python tools/ab_time.py --child tools/ab/$lib.so 2>&1 | grep -v amdgpu.ids; done
done
