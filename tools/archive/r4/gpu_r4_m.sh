#!/bin/bash
# round 4, call M: serial trace + PMC traffic of BASELINE config 4 (R101 800x800 B=8 bf16) and config 2 (R50 keypoint subnet 480x480 B=16 fp32)
bash tools/gpu_profile_config.sh r4_cfg4 "--layers 101 --size 800 --batch 8 --dtype bf16"
bash tools/gpu_profile_config.sh r4_cfg2 "--layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet"
