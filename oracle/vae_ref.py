"""CPU oracle of the VAE either side of the hot path: diffusers 0.32.2 `AutoencoderKL` (encoder / decoder of FLUX.1 and SD3.5), as the
reference calls it at src/UniGenPipeline.py:635-636 (`vae.encode(control_image).latent_dist.sample()`, then `(z - shift) * scale`) and
:797-798 (`z / scale + shift`, `vae.decode(z)`); train.py:527,572 encode the same way.

TEST INFRASTRUCTURE ONLY (see oracle/unigen_ref.py). PARITY UNPINNED: diffusers is not importable here and no VAE weights or vectors
exist on disk; this restates the published module graph of the pinned release on a flat state dict under diffusers' key names:
  Encoder: conv_in -> DownEncoderBlock2D x len(block_out_channels) [ResnetBlock2D x layers_per_block, Downsample2D(padding=0) except last]
           -> UNetMidBlock2D [resnet, Attention(heads = 1, group_norm), resnet] -> GroupNorm -> SiLU -> conv_out (2 x latent channels)
  DiagonalGaussianDistribution: mean, logvar = chunk(2, dim=1); logvar.clamp(-30, 20); sample = mean + exp(0.5 logvar) * randn
  Decoder: conv_in -> UNetMidBlock2D -> UpDecoderBlock2D x n [ResnetBlock2D x (layers_per_block + 1), Upsample2D (nearest 2x + conv) except last]
           -> GroupNorm -> SiLU -> conv_out
FLUX / SD3.5 VAEs have no quant_conv / post_quant_conv (use_quant_conv = use_post_quant_conv = False); both are honoured when present.
`dtype` = torch.bfloat16 gives the reference's eager rounding points, torch.float32 the truth on the same (bf16-representable) weights.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.3611
    shift_factor: float = 0.1159
    use_quant_conv: bool = False
    use_post_quant_conv: bool = False
    mid_block_add_attention: bool = True


def _conv(st: State, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    return F.conv2d(x, st[p + ".weight"].to(x.dtype), st[p + ".bias"].to(x.dtype), stride=stride, padding=padding)


def _gn(st: State, p: str, x: torch.Tensor, groups: int) -> torch.Tensor:
    return F.group_norm(x, groups, st[p + ".weight"].to(x.dtype), st[p + ".bias"].to(x.dtype), eps=1e-6)


def resnet_block(st: State, p: str, cfg: VAEConfig, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1): x + conv2(silu(norm2(conv1(silu(norm1(x)))))), 1x1 conv shortcut."""
    h = _conv(st, p + ".conv1", F.silu(_gn(st, p + ".norm1", x, cfg.norm_num_groups)))
    h = _conv(st, p + ".conv2", F.silu(_gn(st, p + ".norm2", h, cfg.norm_num_groups)))
    if p + ".conv_shortcut.weight" in st:
        x = _conv(st, p + ".conv_shortcut", x, padding=0)
    return x + h


def mid_attention(st: State, p: str, cfg: VAEConfig, x: torch.Tensor) -> torch.Tensor:
    """Attention(heads = 1, dim_head = C, norm_num_groups, residual_connection=True, _from_deprecated_attn_block=True) + AttnProcessor2_0."""
    B, C, H, W = x.shape
    res = x
    h = x.view(B, C, H * W)
    h = F.group_norm(h, cfg.norm_num_groups, st[p + ".group_norm.weight"].to(x.dtype), st[p + ".group_norm.bias"].to(x.dtype), eps=1e-6).transpose(1, 2)
    lin = lambda n, t: F.linear(t, st[f"{p}.{n}.weight"].to(t.dtype), st[f"{p}.{n}.bias"].to(t.dtype))
    q, k, v = lin("to_q", h), lin("to_k", h), lin("to_v", h)
    o = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1), dropout_p=0.0, is_causal=False).squeeze(1).to(q.dtype)
    o = lin("to_out.0", o).transpose(1, 2).reshape(B, C, H, W)
    return o + res


def mid_block(st: State, p: str, cfg: VAEConfig, x: torch.Tensor) -> torch.Tensor:
    x = resnet_block(st, p + ".resnets.0", cfg, x)
    if cfg.mid_block_add_attention:
        x = mid_attention(st, p + ".attentions.0", cfg, x)
    return resnet_block(st, p + ".resnets.1", cfg, x)


def encode_moments(st: State, cfg: VAEConfig, image: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """AutoencoderKL.encode up to the moments [B, 2 * latent, H/8, W/8] (Encoder.forward [+ quant_conv])."""
    x = _conv(st, "encoder.conv_in", image.to(dtype))
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            x = resnet_block(st, f"encoder.down_blocks.{i}.resnets.{j}", cfg, x)
        if i < n - 1:            # Downsample2D(use_conv=True, padding=0): F.pad(x, (0, 1, 0, 1)) then conv k=3 s=2
            x = _conv(st, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(x, (0, 1, 0, 1), mode="constant", value=0), stride=2, padding=0)
    x = mid_block(st, "encoder.mid_block", cfg, x)
    x = _conv(st, "encoder.conv_out", F.silu(_gn(st, "encoder.conv_norm_out", x, cfg.norm_num_groups)))
    if cfg.use_quant_conv:
        x = _conv(st, "quant_conv", x, padding=0)
    return x


def gaussian_sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution(moments).sample() with the randn draw passed in."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    std = torch.exp(0.5 * logvar)
    return mean + std * noise.to(moments.dtype)


def decode(st: State, cfg: VAEConfig, z: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """AutoencoderKL.decode(z, return_dict=False)[0]: [B, latent, h, w] -> [B, 3, 8h, 8w]."""
    z = z.to(dtype)
    if cfg.use_post_quant_conv:
        z = _conv(st, "post_quant_conv", z, padding=0)
    x = _conv(st, "decoder.conv_in", z)
    x = mid_block(st, "decoder.mid_block", cfg, x)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = resnet_block(st, f"decoder.up_blocks.{i}.resnets.{j}", cfg, x)
        if i < n - 1:            # Upsample2D: nearest 2x, then conv 3x3
            x = _conv(st, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(x.dtype))
    return _conv(st, "decoder.conv_out", F.silu(_gn(st, "decoder.conv_norm_out", x, cfg.norm_num_groups)))


def encode_condition(st: State, cfg: VAEConfig, image: torch.Tensor, noise: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """src/UniGenPipeline.py:635-636: (vae.encode(image).latent_dist.sample() - shift_factor) * scaling_factor."""
    z = gaussian_sample(encode_moments(st, cfg, image, dtype), noise)
    return (z - cfg.shift_factor) * cfg.scaling_factor


def decode_latents(st: State, cfg: VAEConfig, latents: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """src/UniGenPipeline.py:797-798: vae.decode(latents / scaling_factor + shift_factor)."""
    return decode(st, cfg, (latents.to(dtype) / cfg.scaling_factor) + cfg.shift_factor, dtype)


# ---------------------------------------------------------------------------------------------------------------------
# parameter shapes under diffusers' AutoencoderKL state-dict key names, and seeded synthetic weights
# ---------------------------------------------------------------------------------------------------------------------

def _resnet_shapes(s, p, cin, cout):
    s[p + ".norm1.weight"] = (cin,); s[p + ".norm1.bias"] = (cin,)
    s[p + ".conv1.weight"] = (cout, cin, 3, 3); s[p + ".conv1.bias"] = (cout,)
    s[p + ".norm2.weight"] = (cout,); s[p + ".norm2.bias"] = (cout,)
    s[p + ".conv2.weight"] = (cout, cout, 3, 3); s[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + ".conv_shortcut.bias"] = (cout,)


def _mid_shapes(s, p, c, attn):
    _resnet_shapes(s, p + ".resnets.0", c, c)
    if attn:
        a = p + ".attentions.0"
        s[a + ".group_norm.weight"] = (c,); s[a + ".group_norm.bias"] = (c,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[f"{a}.{n}.weight"] = (c, c); s[f"{a}.{n}.bias"] = (c,)
    _resnet_shapes(s, p + ".resnets.1", c, c)


def vae_state_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    ch = cfg.block_out_channels
    n = len(ch)
    s["encoder.conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3); s["encoder.conv_in.bias"] = (ch[0],)
    cin = ch[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            _resnet_shapes(s, f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else ch[i], ch[i])
        cin = ch[i]
        if i < n - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (ch[i], ch[i], 3, 3); s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (ch[i],)
    _mid_shapes(s, "encoder.mid_block", ch[-1], cfg.mid_block_add_attention)
    s["encoder.conv_norm_out.weight"] = (ch[-1],); s["encoder.conv_norm_out.bias"] = (ch[-1],)
    s["encoder.conv_out.weight"] = (2 * cfg.latent_channels, ch[-1], 3, 3); s["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    if cfg.use_quant_conv:
        s["quant_conv.weight"] = (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1); s["quant_conv.bias"] = (2 * cfg.latent_channels,)
    if cfg.use_post_quant_conv:
        s["post_quant_conv.weight"] = (cfg.latent_channels, cfg.latent_channels, 1, 1); s["post_quant_conv.bias"] = (cfg.latent_channels,)
    s["decoder.conv_in.weight"] = (ch[-1], cfg.latent_channels, 3, 3); s["decoder.conv_in.bias"] = (ch[-1],)
    _mid_shapes(s, "decoder.mid_block", ch[-1], cfg.mid_block_add_attention)
    rev = list(reversed(ch))
    cin = rev[0]
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            _resnet_shapes(s, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else rev[i], rev[i])
        cin = rev[i]
        if i < n - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (rev[i], rev[i], 3, 3); s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (rev[i],)
    s["decoder.conv_norm_out.weight"] = (ch[0],); s["decoder.conv_norm_out.bias"] = (ch[0],)
    s["decoder.conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3); s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def make_vae_state(cfg: VAEConfig, seed: int = 0, dtype=torch.bfloat16) -> State:
    """Seeded weights: conv / linear ~ N(0, 1 / fan_in) (activations stay O(1) through the depth), norm weights 1 + 0.1 N, biases 0.02 N."""
    g = torch.Generator().manual_seed(seed)
    st: State = {}
    for name, shape in vae_state_shapes(cfg).items():
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = int(torch.Size(shape[1:]).numel())
            t = torch.randn(shape, generator=g) / fan_in ** 0.5
        st[name] = t.to(dtype)
    return st
