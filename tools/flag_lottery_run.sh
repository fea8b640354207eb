#!/bin/bash
# On the GPU box: time every variant of tools/_ab/lot (one process each, same device) -> gpurun_out/flag_lottery.log
LOT=tools/_ab/lot
LOG=gpurun_out/flag_lottery.log
mkdir -p gpurun_out
: > $LOG
for round in 1 2 3; do
  while IFS='|' read -r tag opt rest; do
    tag=$(echo $tag); [ -f $LOT/$tag.so ] || continue
    what=gemm; case $tag in attention*) what=attention;; esac
    echo "round $round $tag |$opt| $(LTXMI_LIB=$PWD/$LOT/$tag.so timeout -k 5 120 python3 tools/flag_lottery_time.py $what 2>&1 | grep -v amdgpu.ids | tail -1)" >> $LOG
  done < $LOT/index.txt
done
