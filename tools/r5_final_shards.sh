#!/bin/bash
# ON THE GPU BOX: the emulated scaling table of DESIGN.md section 7 (every rank's shard rendered one after the other on one device)
mkdir -p gpurun_out; : > gpurun_out/r5_shards.txt
bash tools/r4_shard_ratio.sh 20 2>&1 | tee -a gpurun_out/r5_shards.txt
bash tools/r4_shard_ratio.sh 128 2>&1 | tee -a gpurun_out/r5_shards.txt
for w in 2 4; do for k in 20 128; do echo "== W=$w steps=$k" | tee -a gpurun_out/r5_shards.txt; bash tools/shards.sh $w $k "" 2>&1 | tail -1 | tee -a gpurun_out/r5_shards.txt; done; done
