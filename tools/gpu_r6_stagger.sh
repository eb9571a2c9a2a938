#!/bin/bash
# round 6: start-up stagger of the ping-pong tiles on multi-round launches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=seed-story_amd/lib/libseedstory_hip.so
U=tools/bin/gemm_ubench
{
for st in 0 1 2 3 4; do
echo "== gemm_pp_stagger=$st =="
UBENCH_KNOB=gemm_pp_stagger=$st timeout 300 $U $L 16384,10240,1280,16:56/0,58/0 16384,3840,1280:56/4 65536,5120,640,16:55/4,56/0 65536,640,2560,0,1:56/4 c16,64,64,640,640,1,0,1:56/8 c16,128,128,320,320,1,0,1:56/4 8192,8192,8192:54/8 | grep -v "max|diff| 0.000e+00.*max 0 elements"
done
} > gpurun_out/r6_stagger.txt 2>&1
grep -E "==|min" gpurun_out/r6_stagger.txt
