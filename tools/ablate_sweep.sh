# (PEDONI_ABLATE exists in the diagnostics build of the library only)
export PEDONI_HIP_LIB=$(pwd)/pedoni_amd/lib/libpedoni_hip_diag.so
for a in 0 1 2 3; do
  echo "ABLATE=$a" >> gpurun_out/ablate_r02.txt
  PEDONI_ABLATE=$a python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fast-leg --no-profile >> gpurun_out/ablate_r02.txt 2>&1 || exit 1
done
