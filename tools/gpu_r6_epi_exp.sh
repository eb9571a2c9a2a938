#!/bin/bash
# round 6 EXPERIMENT (profiles/round6_gemm_epilogue_cost.txt) — needs a patched library, not the shipped one:
#   epilogue compiled out:  in gemm_pp_kernel, in front of `if constexpr (STAGED) {`:
#       if (g.epi & (1 << 28)) { if (!has_next) break; vid = vnext; continue; }
#   non-temporal stores:    in gemm_epilogue_staged's readback, `if (g.epi & (1 << 27)) __builtin_nontemporal_store(...) else` the uint4 store
# the ubench passes the bit through the epilogue argument of ss_gemm (case = M,N,K,<epi bits>)
cd "$GRAFT_REPO_ROOT" || exit 1
L=seed-story_amd/lib/libseedstory_hip.so
U=tools/bin/gemm_ubench
NT=134217728
{
timeout 600 $U $L 16384,5120,64:56/0 16384,5120,64,$NT:56/0 16384,5120,1280:56/0 16384,5120,1280,$NT:56/0 16384,1280,1280:56/8 16384,1280,1280,$NT:56/8 16384,10240,1280,16:56/0 16384,10240,1280,$((NT+16)):56/0 16384,3840,1280:56/4 16384,3840,1280,$NT:56/4
} 2>&1 | grep -E "case|min" > gpurun_out/r6_epi_exp3.txt
cat gpurun_out/r6_epi_exp3.txt
