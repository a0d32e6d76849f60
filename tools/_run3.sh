set -e
for cfg in "224 0" "224 2800" "192 0" "192 6600" "160 0" "160 9000"; do
  set -- $cfg
  HFZ_DECODER_LDS_PAD=$2 bash tools/profile.sh r03occ_$1_$2 --total-mib $1 > gpurun_out/r03occ_$1_$2.log 2>&1
done
