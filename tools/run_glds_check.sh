mkdir -p gpurun_out
for c in 8 10 15; do python - <<PY
import sys; sys.path.insert(0,'seed-story_amd'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, synth
from seedstory import _lib, ops
_lib.set_tuning("gemm_cfg", $c)
def rel(a,b): a,b=a.float().cpu(),b.float().cpu(); return float((a-b).norm()/(b.norm()+1e-30))
worst=0
for (M,N,K) in [(1,64,64),(37,100,256),(65,4096,4096),(343,768,512),(130,4992,1664),(256,1664,608),(300,256,8192),(1024,512,1664)]:
    a=synth.normal_like(14,(M,K),1.0,dtype=torch.bfloat16); w=synth.normal_like(15,(N,K),0.05,dtype=torch.bfloat16)
    bias=synth.normal_like(16,(N,),0.5,dtype=torch.bfloat16); res=synth.normal_like(17,(M,N),1.0,dtype=torch.bfloat16)
    ref=a.float()@w.float().t()
    y=ops.gemm(a.cuda(),w.cuda(),bias=bias.cuda(),residual=res.cuda())
    r=rel(y,((ref+bias.float()).bfloat16()+res).float()); worst=max(worst,r)
print("cfg",$c,"gemm worst rel",worst)
PY
done
timeout 300 python -m pytest tests/test_sdxl_gpu.py -m gpu -q -k "conv3x3" 2>&1 | tail -2
