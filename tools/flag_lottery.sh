#!/bin/bash
# Build libltxmi variants that differ in ONE backend option for ONE source (attention_pipe.hip or gemm.hip): tools/_ab/lot/*.so.
# hipcc's placement of the same hand-scheduled loop differs by a few per cent between instances (profiles/r04_attn_q_prescaled.log):
# this tries the options that move it.   bash tools/flag_lottery.sh ; then on the GPU: bash tools/flag_lottery_run.sh
set -e
OLDPWD=$(cd "$(dirname "$0")/.." && pwd)
cd "$OLDPWD/ltx-video-gpupoor_amd/csrc"
make -s
OUT=../../tools/_ab/lot
mkdir -p $OUT
rm -f $OUT/*.so $OUT/*.o $OUT/index.txt
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -Wall -Wno-unused-function -Wno-unused-variable"
ATT="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fhonor-infinities"
OBJS="api.o gemm.o attention.o attention_pipe.o attention_pipe128.o attention_cross.o rowops.o pointwise.o upsampler.o conv_direct.o"
i=0
while read -r opt; do
  [ -z "$opt" ] && continue
  for src in ${LOTTERY_SRCS:-attention_pipe gemm}; do
    extra=""; [ $src = attention_pipe ] && extra="$ATT"
    tag=$(printf "%s_%02d" $src $i)
    if /opt/rocm/bin/hipcc $BASE $extra $opt -Rpass-analysis=kernel-resource-usage -c $src.hip -o $OUT/$tag.o 2> $OUT/$tag.log; then
      spills=$(grep -E "VGPRs Spill" $OUT/$tag.log | sed 's/.*Spill: //; s/ .*//' | sort -n | tail -1)
      sum=$(md5sum < $OUT/$tag.o | cut -c1-12)
      if grep -q "$src.*md5 $sum" $OUT/index.txt 2>/dev/null; then
        echo "$tag | $opt | same object as $(grep "$src.*md5 $sum" $OUT/index.txt | head -1 | cut -d' ' -f1)" >> $OUT/index.txt
        rm -f $OUT/$tag.o $OUT/$tag.log
        continue
      fi
      objs=""; for o in $OBJS; do if [ $o = $src.o ]; then objs="$objs $OUT/$tag.o"; else objs="$objs $o"; fi; done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$tag.so $objs
      echo "$tag | $opt | max spills $spills | md5 $sum" >> $OUT/index.txt
    else
      echo "$tag | $opt | BUILD FAILED" >> $OUT/index.txt
    fi
    rm -f $OUT/$tag.o $OUT/$tag.log
  done
  i=$((i+1))
done < "${LOTTERY_OPTS:-$OLDPWD/tools/flag_lottery_opts.txt}"
cat $OUT/index.txt
