#!/bin/bash
# usage: tools/tile_sweep.sh  (on the GPU box) -- times k_level_image per level for candidate tiles
for spec in "2:32:30" "2:64:14" "2:64:16" "2:32:14" "2:16:30" "3:16:22" "3:32:14" "3:32:6" "3:16:14" "3:8:30" "3:64:6" "4:4:26" "4:16:6" "4:16:4" "4:8:10" "4:8:6" "4:32:2" "4:16:2" "5:4:6" "5:8:2" "5:8:4" "5:4:4" "5:16:2" "5:2:8"; do
  lvl=${spec%%:*}
  t=$(TF_IMG_TILES=$spec python tools/kprof.py 4k 8 2>&1 | grep "fb_level_image.k$lvl" | awk '{print $(NF-1)}')
  echo "$spec $t us/launch"
done
