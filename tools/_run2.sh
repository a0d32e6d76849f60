set -e
for pad in 0 2800 6600; do
  HFZ_DECODER_LDS_PAD=$pad bash tools/profile.sh r03pad$pad > gpurun_out/r03pad$pad.log 2>&1
  HFZ_DECODER_LDS_PAD=$pad bash tools/profile.sh r03pad${pad}_2g --total-mib 2048 --frame-kib 1024 > gpurun_out/r03pad${pad}_2g.log 2>&1
done
bash tools/profile.sh r03bits12 --decoder-table-bits 12 > gpurun_out/r03bits12.log 2>&1
