#!/bin/bash
# round 6: bias segment through LDS (DMA'd by the prologue) + GEGLU roundings as the (value, gate) pairs they come in — equality screen against the
# one-barrier tiles and timing of the previous commit's library (lib/ab/libseedstory_hip_base.so) vs this one, interleaved twice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=seed-story_amd/lib/libseedstory_hip.so
LB=seed-story_amd/lib/ab/libseedstory_hip_base.so
U=tools/bin/gemm_ubench
O=gpurun_out/r6_epipipe2.txt
T="16384,10240,1280,16:58/0 16384,3840,1280:58/4 16384,1280,1280,0,1:58/8 16384,1280,1280:58/8 16384,1280,5120,0,1:58/8 65536,640,2560,0,1:58/4 65536,640,640,0,1:56/4 65536,5120,640,16:58/8 65536,1920,640:58/8 c16,32,32,1280,1280,1,0,1:56/8 c16,64,64,640,640,1,0,0,1:56/8"
{
echo "== equality screen (new lib; first cfg of a list is the reference) =="
UBENCH_SCREEN=20 timeout 600 $U $L 256,256,256:60,54,55,57 256,320,256:60,56,58 512,640,512,0,1:60,54,55,57,56,58 256,320,64,16:60,56,58 256,640,128,2:60,56,58,54 256,640,192,16:60,54,57,58 \
   1000,512,256,0,1:60,54,55,57 1000,520,256:60,54,55,57 1000,520,320,16:60,54,57 1000,512,320,2:60,55,54 264,328,64,0,1:60,54,57 \
   4096,1280,4096:60,57,58,54,55,56 16384,1280,1280,0,1:62,58,56,54 16384,10240,1280,16:69,57,58,54,56 16384,3840,1280,2:69,58,57 7304,12288,4096:60,57,55 7304,4096,11008,0,1:60,55,57 \
   c1,16,16,64,320,1,0:63,56,54 c2,32,32,128,640,1,0,1,1:63,56,54,57 c16,32,32,1280,1280,1,0,1:63,56 c16,32,32,1280,1280,1,0,0,1:63,56,54 c4,128,128,320,320,1,0,1:63,56,54,55 c1,8,32,64,256,1,0,1:63,54,57 c2,64,64,640,640,1,0,0,1:63,56,57
for rep in 1 2; do
echo "== timing: BASE lib (pass $rep) =="
timeout 600 $U $LB $T
echo "== timing: NEW lib (pass $rep) =="
timeout 600 $U $L $T
done
} > $O 2>&1
grep -c "max 0 elements, 0 of" $O; grep "max|diff|" $O | grep -v "max|diff| 0.000e+00.*max 0 elements, 0 of" | head
grep -E "^== timing|case|min" $O | grep -A400 "== timing" | awk '/^==/{print} /case/{c=$2} /min/{print "   ", c, $4, $5}' | head -120
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_sdxl_gpu.py tests/test_fp16_gpu.py -m gpu -x -q -k "pipelined or pingpong or tile_config or fp16" > gpurun_out/r6_epipipe2_pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r6_epipipe2_pytest.txt
