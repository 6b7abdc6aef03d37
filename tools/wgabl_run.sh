set -e
mkdir -p gpurun_out/wgabl
for v in base abl11 abl12 abl13; do
  if [ $v = base ]; then L=""; else L="$PWD/fluidnet_amd/lib/$v/libfluidnet_hip.so"; fi
  echo "== $v" >> gpurun_out/wgabl/log.txt
  FNET_LIB=$L timeout -k 10 120 python -u tools/conv_bench.py run >> gpurun_out/wgabl/log.txt 2>&1
done
echo "== native-vs-oracle test" >> gpurun_out/wgabl/log.txt
timeout -k 10 200 python -u -m pytest tests/test_gpu_native_rollout.py -x -q --timeout 120 --timeout-method thread -k oracle -s >> gpurun_out/wgabl/log.txt 2>&1
cat gpurun_out/wgabl/log.txt
