# timings of the ablation builds tools/splitk_ablate.sh left in flute_amd/csrc (run on the GPU box)
for lib in shipped $(ls flute_amd/csrc/libflute_amd_abl*.so | sort -V); do
  if [ $lib = shipped ]; then unset FLUTE_AMD_LIB; else export FLUTE_AMD_LIB=$lib; fi
  python tools/splitk_lab.py time_abl 2>/dev/null | grep '"time"'
done
