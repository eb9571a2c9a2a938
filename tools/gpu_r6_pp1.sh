#!/bin/bash
# round 6, step 1: ping-pong 256x256 tile (cfg 50-55) vs the one-barrier tiles: race screen + sustained timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=seed-story_amd/lib/libseedstory_hip.so
U=tools/bin/gemm_ubench
{
echo "== race screen (bit-equality vs cfg 60, 30 runs each) =="
UBENCH_SCREEN=30 timeout 300 $U $L 256,256,256:60,50,51,52,54 512,512,512:60,50,52 256,256,64:60,50,52 256,256,128:60,50,52 256,256,192:60,50,52 \
   4096,4096,4096:60,50,51,52,54 1000,700,320:60,50,52 16384,10240,1280,16:69,52,54
echo "== uniform [-1,1) both operands =="
UBENCH_WSCALE=1.0 timeout 300 $U $L 4096,4096,4096:60,50,51,52,54 8192,8192,8192:60,50,51,52,54
echo "== default operand class (W x0.05) =="
timeout 600 $U $L 8192,8192,8192:60,50,52 16384,10240,1280,16:69/0,60/8,52/0,52/4,52/8,54/4 16384,3840,1280:60/4,52/4,52/8,54/4 \
   16384,1280,1280,0,1:61/4,62/8,52/4,52/8 16384,1280,5120,0,1:62/4,52/4,52/8 65536,5120,640,16:60/4,52/4,52/8 65536,640,2560,0,1:62/8,52/4 \
   8192,10240,1280,16:60/8,52/4,52/8 4096,10240,1280,16:60/8,52/4
} > gpurun_out/r6_pp1.txt 2>&1
tail -80 gpurun_out/r6_pp1.txt
