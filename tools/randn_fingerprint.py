"""Does this host's torch CPU generator draw the same LLaMA-7B test weights as the host that wrote tests/golden/llama7b_truth.safetensors?
Prints a fingerprint of the first two tensors of tests/test_fulldim_gpu.py::_llama7b_weights (seed 4242) — compare across boxes."""
import hashlib
import torch
g = torch.Generator().manual_seed(4242)
h = hashlib.sha256()
for shape in ((32066, 4096), (32066, 4096)):
    t = (torch.randn(*shape, generator=g) * 0.02).to(torch.bfloat16)
    f = t.flatten()
    h.update(f[:64].float().numpy().tobytes())
    h.update(f[:: max(1, f.numel() // 4096)].float().numpy().tobytes())
    h.update(f.view(torch.int16)[::997].to(torch.int64).sum().numpy().tobytes())
print("randn fingerprint", h.hexdigest()[:32], torch.__version__, torch.get_num_threads())
