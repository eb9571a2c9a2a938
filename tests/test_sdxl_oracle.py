"""CPU: structural checks of the SDXL oracle restatement (parity unpinned at the diffusers boundary)."""
import torch

import sdxl_oracle as S
import synth


def test_unet_program_matches_published_sdxl_base():
    """Structural pins of the oracle's own derivation (a program built from the SURVEY Appendix-B stage table):
    diffusers' SDXL-base UNet has 2,567,463,684 parameters (SURVEY's enumeration says 2,566,942,084: it leaves out
    521,600 = 0.02 %), 70 transformer blocks (4 + 20 + 10 + 30 + 6), 140 attention calls, 17 ResBlocks."""
    st = S.unet_stats(S.SDXL_BASE_UNET)
    assert st["params"] == 2_567_463_684
    assert abs(st["params"] - 2_566_942_084) / 2_566_942_084 < 5e-4
    assert st["transformer_blocks"] == 70 and st["attention_calls"] == 140 and st["resnets"] == 17
    prog = S.unet_program(S.SDXL_BASE_UNET)
    cat = [(o["cin"] - o["skip"], o["skip"], o["cout"]) for o in prog if o["op"] == "res" and o["skip"]]
    # Appendix B up path: 2560->1280 x2, 1920->1280 | 1920->640, 1280->640, 960->640 | 960->320, 640->320 x2
    assert [a + b for a, b, _ in cat] == [2560, 2560, 1920, 1920, 1280, 960, 960, 640, 640]
    assert [c for _, _, c in cat] == [1280] * 3 + [640] * 3 + [320] * 3


def test_product_module_key_set_equals_oracle_key_set():
    """Two independent derivations of the diffusers parameter tree (the oracle's op program vs the product's module
    builder) must give the same names and shapes — tiny and SDXL-base configurations."""
    from seedstory import diffusion as D
    u = D.UNet2DConditionModel(S.TINY_UNET)
    assert {k: tuple(v.shape) for k, v in u.state_dict().items()} == S.unet_shapes(S.TINY_UNET)
    v = D.AutoencoderKL(S.TINY_VAE)
    assert {k: tuple(t.shape) for k, t in v.state_dict().items()} == S.vae_decoder_shapes(S.TINY_VAE)
    assert D._unet_shapes(D.SDXL_BASE_UNET) == S.unet_shapes(S.SDXL_BASE_UNET)
    assert D._vae_shapes(D.SDXL_BASE_VAE) == S.vae_decoder_shapes(S.SDXL_BASE_VAE)


def test_euler_schedule_literals_and_product_agreement():
    """30-step SDXL-base schedule as literals (public config: scaled-linear 0.00085..0.012, 1000 train steps, leading
    spacing, offset 1), and the product's float32 scheduler (diffusers computes in float32) against the oracle's
    float64 closed form."""
    ts, sig, init = S.euler_schedule(30)
    assert ts == [958, 925, 892, 859, 826, 793, 760, 727, 694, 661, 628, 595, 562, 529, 496, 463, 430, 397, 364, 331,
                  298, 265, 232, 199, 166, 133, 100, 67, 34, 1]
    assert abs(sig[0] - 11.476846458) < 1e-7 and abs(sig[1] - 9.543582567) < 1e-7 and abs(sig[29] - 0.041314412) < 1e-8
    assert sig[30] == 0.0 and abs(init - 11.520330057) < 1e-7
    assert all(a > b for a, b in zip(sig[:-1], sig[1:]))                  # strictly decreasing
    ts50, sig50, init50 = S.euler_schedule(50)
    assert ts50[:3] == [981, 961, 941] and ts50[-1] == 1 and abs(sig50[0] - 13.120410743) < 1e-7
    from seedstory.diffusion import EulerDiscreteScheduler
    sch = EulerDiscreteScheduler()
    for n, (t_ref, s_ref, i_ref) in ((30, (ts, sig, init)), (50, (ts50, sig50, init50))):
        sch.set_timesteps(n)
        assert [int(t) for t in sch.timesteps] == t_ref
        rel = max(abs(float(a) - b) / max(b, 1e-9) for a, b in zip(sch.sigmas[:-1], s_ref[:-1]))
        assert rel < 3e-6 and float(sch.sigmas[-1]) == 0.0
        assert abs(sch.init_noise_sigma - i_ref) / i_ref < 3e-6


def test_unet_cfg_linearity_and_shapes():
    c = S.TINY_UNET
    wd = S.synth_weights(S.unet_shapes(c), 1)
    x = synth.normal_like(5, (2, 4, 8, 8), 1.0)
    ctx = synth.normal_like(6, (2, 8, 128), 1.0)
    pooled = synth.normal_like(7, (2, 80), 1.0)
    tid = torch.tensor([[64, 64, 0, 0, 64, 64]] * 2, dtype=torch.float32)
    y = S.unet_forward(wd, c, x, torch.tensor(801.0), ctx, pooled, tid)
    assert y.shape == x.shape and torch.isfinite(y).all()
    # batch elements are independent (no cross-batch leakage through GroupNorm / attention)
    y0 = S.unet_forward(wd, c, x[:1], torch.tensor(801.0), ctx[:1], pooled[:1], tid[:1])
    assert (y0 - y[:1]).abs().max() < 1e-4
