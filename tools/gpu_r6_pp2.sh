#!/bin/bash
# round 6, step 2: persistent ping-pong tiles (cfg 53, 55)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=seed-story_amd/lib/libseedstory_hip.so
U=tools/bin/gemm_ubench
{
echo "== race screen =="
UBENCH_SCREEN=30 timeout 300 $U $L 256,256,256:60,53,55 512,768,512:60,53,55 256,256,64:60,53 1024,1024,128:60,53 4096,4096,4096:60,53,55 1000,700,320:60,53 16384,10240,1280,16:69,53,55 7304,12288,4096:60,53
echo "== timing =="
UBENCH_WSCALE=1.0 timeout 300 $U $L 8192,8192,8192:60,54,55,53
timeout 600 $U $L 16384,10240,1280,16:69/0,52/0,54/0,53/0,55/0,55/4,55/8 16384,3840,1280:60/4,54/4,55/4,55/8,55/0,53/4 \
   65536,5120,640,16:60/4,54/4,55/4,55/0 8192,10240,1280,16:60/8,54/4,55/4,55/0 16384,1280,1280,0,1:62/8,54/4,55/4 7304,12288,4096:60/4,54/4,55/4,55/0 7304,22016,4096:60/4,55/4,55/0 7304,4096,11008,0,1:60/4,55/4
} > gpurun_out/r6_pp2.txt 2>&1
grep -v "max|diff| 0.000e+00.*max 0 elements" gpurun_out/r6_pp2.txt | tail -80
