"""AutoencoderKL on MI355X: the VAE either side of the hot path.

Reference call sites: src/UniGenPipeline.py:635-636 (`self.vae.encode(control_image).latent_dist.sample()`, then `(z - shift_factor) *
scaling_factor`) and :797-798 (`latents / scaling_factor + shift_factor`, `self.vae.decode(latents, return_dict=False)[0]`); train.py:527,572.
The module is diffusers 0.32.2 `AutoencoderKL` (FLUX.1 / SD3.5 VAE: 16 latent channels, block_out_channels (128, 256, 512, 512), 2 layers per
block, GroupNorm(32, eps 1e-6), one mid-block attention head of dim 512, no quant convs). Same class name, `from_pretrained` layout
(config.json + diffusion_pytorch_model.safetensors), state-dict keys, `encode(x).latent_dist.sample(generator)`, `decode(z, return_dict=False)[0]`,
`config.scaling_factor / shift_factor`, so the pipeline (and the reference's own) can hold it as `pipe.vae`.

Nothing here computes in torch: activations are NHWC 2-D tensors [B*H*W, C] and every op is a C-ABI call (unigen_amd/ops.py):
ug_conv2d_nhwc (implicit GEMM on MFMA; nearest-2x upsampling and Downsample2D's one-sided padding folded into its gather), ug_groupnorm_nhwc
(+ SiLU), ug_gemm_bf16 (1x1 shortcuts, attention projections, scores, P.V), ug_softmax_rows, ug_vae_sample, ug_nchw_to_nhwc / ug_nhwc_to_nchw.
fp32 parameters select the verification twins, as for the transformer.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import lib as L
from . import ops
from .engine import _Holder, _register, _Workspace

BF = torch.bfloat16
FLUX_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                       scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False, use_post_quant_conv=False, mid_block_add_attention=True)


def _pad_to(n: int, q: int) -> int:
    return (n + q - 1) // q * q


def vae_param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """diffusers AutoencoderKL parameter names and (torch) shapes."""
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(p, cout, cin, k=3):
        s[p + ".weight"] = (cout, cin, k, k); s[p + ".bias"] = (cout,)

    def norm(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); conv(p + ".conv1", cout, cin); norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout)
        if cin != cout:
            conv(p + ".conv_shortcut", cout, cin, 1)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c)
        if cfg.mid_block_add_attention:
            a = p + ".attentions.0"
            norm(a + ".group_norm", c)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                s[f"{a}.{n}.weight"] = (c, c); s[f"{a}.{n}.bias"] = (c,)
        resnet(p + ".resnets.1", c, c)

    ch = tuple(cfg.block_out_channels)
    n = len(ch)
    conv("encoder.conv_in", ch[0], cfg.in_channels)
    cin = ch[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else ch[i], ch[i])
        cin = ch[i]
        if i < n - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch[i], ch[i])
    mid("encoder.mid_block", ch[-1])
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out", 2 * cfg.latent_channels, ch[-1])
    if cfg.use_quant_conv:
        conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    if cfg.use_post_quant_conv:
        conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    conv("decoder.conv_in", ch[-1], cfg.latent_channels)
    mid("decoder.mid_block", ch[-1])
    rev = ch[::-1]
    cin = rev[0]
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else rev[i], rev[i])
        cin = rev[i]
        if i < n - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", rev[i], rev[i])
    norm("decoder.conv_norm_out", ch[0])
    conv("decoder.conv_out", cfg.out_channels, ch[0])
    return s


class DiagonalGaussianDistribution:
    """diffusers DiagonalGaussianDistribution over NHWC moments held on the device; `sample(generator)` draws the noise with torch's RNG (as
    the reference does through randn_tensor) and evaluates mean + std * noise in ug_vae_sample."""

    def __init__(self, moments: torch.Tensor, B: int, latent: int, H: int, W: int):
        self._m, self._shape = moments, (B, latent, H, W)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None, shift: float = 0.0, scale: float = 1.0) -> torch.Tensor:
        B, Lc, H, W = self._shape
        dev, dt = self._m.device, self._m.dtype
        if noise is None:
            gdev = generator.device if generator is not None else dev
            noise = torch.randn(self._shape, generator=generator, device=gdev, dtype=torch.float32)
        return ops.vae_sample(self._m, noise.to(dev, dt).contiguous(), B=B, latent=Lc, H=H, W=W, shift=shift, scale=scale)

    def mode(self) -> torch.Tensor:
        B, Lc, H, W = self._shape
        return ops.nhwc_to_nchw(self._m, B, Lc, H, W)


class AutoencoderKL(nn.Module):
    """Drop-in for diffusers `AutoencoderKL` (encode / decode surface used by the reference's pipelines)."""

    def __init__(self, config: Optional[dict] = None, device=None, dtype=BF, **kwargs):
        super().__init__()
        c = dict(FLUX_VAE_CONFIG)
        c.update(config or {})
        c.update(kwargs)
        c["block_out_channels"] = tuple(c["block_out_channels"])
        self.config = SimpleNamespace(**c)
        self._ws = _Workspace()
        self._packed: Dict[str, Tuple] = {}
        for name, shape in vae_param_shapes(self.config).items():
            _register(self, name, shape, device, dtype)

    # ------------------------------------------------------------------ construction ---------------------------------
    @classmethod
    def from_config(cls, config: dict, **kw) -> "AutoencoderKL":
        return cls(config, **kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=BF, device=None, **kw) -> "AutoencoderKL":
        """Local directory only: config.json + *.safetensors (diffusers layout: <model>/vae/)."""
        path = os.fspath(pretrained_model_name_or_path)
        if subfolder:
            path = os.path.join(path, subfolder)
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (unigen_amd loads checkpoints from disk only)")
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        model = cls({k: raw[k] for k in FLUX_VAE_CONFIG if k in raw}, device=device, dtype=torch_dtype)
        from safetensors.torch import load_file
        sd = {}
        for fn in sorted(f for f in os.listdir(path) if f.endswith(".safetensors")):
            sd.update(load_file(os.path.join(path, fn)))
        if not sd:
            raise OSError(f"no *.safetensors weights under {path}")
        res = model.load_state_dict(sd, strict=False)
        if res.missing_keys:
            raise RuntimeError(f"VAE checkpoint misses parameters: {res.missing_keys[:8]} ...")
        return model

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def init_synthetic_(self, seed: int = 0) -> "AutoencoderKL":
        """Seeded weights ~ N(0, 1 / fan_in), norm weights 1 + 0.1 N, biases 0.02 N (activations stay O(1) through the depth)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                r = torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
                if "norm" in name and name.endswith(".weight"):
                    p.copy_(1.0 + 0.1 * r)
                elif name.endswith(".bias"):
                    p.copy_(0.02 * r)
                else:
                    p.copy_(r / float(p[0].numel()) ** 0.5)
        return self

    # ------------------------------------------------------------------ weight packing --------------------------------
    def _P(self, name: str) -> torch.Tensor:
        return self.get_parameter(name).data

    def _conv_w(self, name: str):
        """[Cout, Cin, KH, KW] -> contiguous [Cout_p, KH, KW, Cin_p] (+ bias [Cout_p]): Cin zero-padded to the conv's K granularity (64), Cout to 8.
        Packed once and re-packed when the parameter is replaced or written in place."""
        w, b = self.get_parameter(name + ".weight"), self.get_parameter(name + ".bias")
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version, w.dtype, w.device)
        hit = self._packed.get(name)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        cout, cin, kh, kw = w.shape
        cin_p, cout_p = _pad_to(cin, 64), _pad_to(cout, 8)
        wp = torch.zeros(cout_p, kh, kw, cin_p, device=w.device, dtype=w.dtype)
        wp[:cout, :, :, :cin] = w.data.permute(0, 2, 3, 1)                  # layout only
        bp = torch.zeros(cout_p, device=w.device, dtype=w.dtype)
        bp[:cout] = b.data
        self._packed[name] = (key, wp, bp)
        return wp, bp

    def _w(self, name, shape, dtype=None):
        return self._ws.get(name, shape, dtype if dtype is not None else self.dtype, self.device)

    # ------------------------------------------------------------------ blocks ----------------------------------------
    def _conv(self, name: str, x: torch.Tensor, B: int, H: int, W: int, out_tag: str, *, stride: int = 1, pad: int = 1, up: int = 0, down: bool = False,
              residual: Optional[torch.Tensor] = None):
        """3x3 convolution on NHWC rows; returns (out [B*Ho*Wo, Cout_p], Ho, Wo). down = Downsample2D(padding=0): F.pad (0, 1, 0, 1) + stride 2."""
        wp, bp = self._conv_w(name)
        kh, kw = wp.shape[1], wp.shape[2]
        if down:
            Ho, Wo, stride, pt, pl = H // 2, W // 2, 2, 0, 0
        else:
            Hv, Wv = H << up, W << up
            Ho, Wo, pt, pl = (Hv + 2 * pad - kh) // stride + 1, (Wv + 2 * pad - kw) // stride + 1, pad, pad
        assert x.shape[1] == wp.shape[3], (name, x.shape, wp.shape)
        out = self._w(out_tag, (B * Ho * Wo, wp.shape[0]))
        ops.conv2d_nhwc(x, wp, bp, out, B=B, H=H, W=W, Ho=Ho, Wo=Wo, KH=kh, KW=kw, stride=stride, pad_t=pt, pad_l=pl, up=up, residual=residual)
        return out, Ho, Wo

    def _gn(self, name: str, x: torch.Tensor, B: int, HW: int, tag: str, silu: bool) -> torch.Tensor:
        out = self._w(tag, tuple(x.shape))
        return ops.groupnorm_nhwc(x, self._P(name + ".weight"), self._P(name + ".bias"), out, B=B, HW=HW, groups=self.config.norm_num_groups, eps=1e-6, silu=silu)

    def _resnet(self, p: str, x: torch.Tensor, B: int, H: int, W: int, out_tag: str) -> torch.Tensor:
        """ResnetBlock2D: x + conv2(silu(norm2(conv1(silu(norm1(x)))))), 1x1 conv shortcut when the channel count changes."""
        h = self._gn(p + ".norm1", x, B, H * W, "rn_a", True)
        h, _, _ = self._conv(p + ".conv1", h, B, H, W, "rn_b")
        h = self._gn(p + ".norm2", h, B, H * W, "rn_c", True)
        sc = x
        if (p + ".conv_shortcut.weight") in self._names():
            wsc = self._P(p + ".conv_shortcut.weight")
            sc = self._w("rn_sc", (x.shape[0], wsc.shape[0]))
            ops.gemm(x, wsc.view(wsc.shape[0], wsc.shape[1]), self._P(p + ".conv_shortcut.bias"), sc, M=x.shape[0])
        out, _, _ = self._conv(p + ".conv2", h, B, H, W, out_tag, residual=sc)
        return out

    def _names(self):
        n = getattr(self, "_name_cache", None)
        if n is None:
            n = {k for k, _ in self.named_parameters()}
            self._name_cache = n
        return n

    def _attention(self, p: str, x: torch.Tensor, B: int, HW: int, out_tag: str) -> torch.Tensor:
        """Attention(heads = 1, dim_head = C, group_norm, residual_connection): x + to_out(softmax(q k^T / sqrt(C)) v)."""
        Cc = x.shape[1]
        if HW % 64 != 0:
            raise L.UniGenHipError(f"VAE mid-block attention needs H*W at the latent resolution to be a multiple of 64 (got {HW})")
        h = self._gn(p + ".group_norm", x, B, HW, "at_n", False)
        q, k = self._w("at_q", (B * HW, Cc)), self._w("at_k", (B * HW, Cc))
        ops.gemm(h, self._P(p + ".to_q.weight"), self._P(p + ".to_q.bias"), q, M=B * HW)
        ops.gemm(h, self._P(p + ".to_k.weight"), self._P(p + ".to_k.bias"), k, M=B * HW)
        scores, probs = self._w("at_s", (HW, HW), torch.float32), self._w("at_p", (HW, HW))
        vT, o = self._w("at_vT", (Cc, HW)), self._w("at_o", (B * HW, Cc))
        for b in range(B):
            sl = slice(b * HW, (b + 1) * HW)
            ops.gemm(q[sl], k[sl], None, scores, M=HW, epilogue=L.EPI_F32)                                 # scores = q k^T (fp32)
            ops.softmax_rows(scores, probs, float(Cc) ** -0.5)
            ops.gemm(self._P(p + ".to_v.weight"), h[sl], None, vT, M=Cc)                                    # v^T = W_v h^T  [C, HW]
            ops.gemm(probs, vT, self._P(p + ".to_v.bias"), o[sl], M=HW)                                     # P v (+ b_v: rows of P sum to 1)
        out = self._w(out_tag, tuple(x.shape))
        ops.gemm(o, self._P(p + ".to_out.0.weight"), self._P(p + ".to_out.0.bias"), out, M=B * HW, epilogue=L.EPI_RES_SCALE, residual=x, alpha=1.0)
        return out

    def _mid(self, p: str, x: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        x = self._resnet(p + ".resnets.0", x, B, H, W, "mid_a")
        if self.config.mid_block_add_attention:
            x = self._attention(p + ".attentions.0", x, B, H * W, "mid_b")
        return self._resnet(p + ".resnets.1", x, B, H, W, "mid_c")

    # ------------------------------------------------------------------ encode / decode -------------------------------
    @torch.no_grad()
    def _encode_moments(self, image: torch.Tensor):
        cfg, dt = self.config, self.dtype
        B, Cc, H, W = image.shape
        if Cc != cfg.in_channels or H % (2 ** (len(cfg.block_out_channels) - 1)) or W % (2 ** (len(cfg.block_out_channels) - 1)):
            raise ValueError(f"image must be [B, {cfg.in_channels}, H, W] with H, W multiples of {2 ** (len(cfg.block_out_channels) - 1)}")
        x = ops.nchw_to_nhwc(image.to(self.device, dt), _pad_to(Cc, 64))
        x, H, W = self._conv("encoder.conv_in", x, B, H, W, "e_in")
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block):
                x = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", x, B, H, W, f"e_r{(i * 4 + j) % 2}")
            if i < n - 1:
                x, H, W = self._conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", x, B, H, W, "e_ds", down=True)
        x = self._mid("encoder.mid_block", x, B, H, W)
        x = self._gn("encoder.conv_norm_out", x, B, H * W, "e_no", True)
        m, _, _ = self._conv("encoder.conv_out", x, B, H, W, "e_out")
        if cfg.use_quant_conv:
            wq = self._P("quant_conv.weight")
            mq = self._w("e_q", (m.shape[0], wq.shape[0]))
            if m.shape[1] != wq.shape[1]:
                raise L.UniGenHipError("quant_conv needs 2 * latent_channels to be a multiple of 64 in this build")
            m = ops.gemm(m, wq.view(wq.shape[0], wq.shape[1]), self._P("quant_conv.bias"), mq, M=m.shape[0])
        return m, B, H, W

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        m, B, H, W = self._encode_moments(x)
        # the moments live in a reused workspace: the distribution owns a copy (2 * latent channels at latent resolution - tiny), so a second
        # encode() before .sample() / .mode() cannot overwrite it (diffusers semantics)
        dist = DiagonalGaussianDistribution(m.clone(), B, self.config.latent_channels, H, W)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def encode_scaled(self, image: torch.Tensor, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """src/UniGenPipeline.py:635-636 in one go: (vae.encode(image).latent_dist.sample() - shift_factor) * scaling_factor -> [B, L, H/8, W/8]."""
        return self.encode(image).latent_dist.sample(generator, noise, shift=self.config.shift_factor, scale=self.config.scaling_factor)

    @torch.no_grad()
    def _decode(self, z: torch.Tensor, div: float, add: float) -> torch.Tensor:
        cfg, dt = self.config, self.dtype
        B, Lc, H, W = z.shape
        if Lc != cfg.latent_channels:
            raise ValueError(f"latents must be [B, {cfg.latent_channels}, h, w]")
        if cfg.use_post_quant_conv:
            raise L.UniGenHipError("post_quant_conv is not supported (FLUX / SD3.5 VAEs have use_post_quant_conv = False)")
        x = ops.nchw_to_nhwc(z.to(self.device, dt), _pad_to(Lc, 64), div, add)
        x, H, W = self._conv("decoder.conv_in", x, B, H, W, "d_in")
        x = self._mid("decoder.mid_block", x, B, H, W)
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, B, H, W, f"d_r{j % 2}")
            if i < n - 1:           # Upsample2D: nearest 2x folded into the convolution's gather
                x, H, W = self._conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x, B, H, W, "d_us", up=1)
        x = self._gn("decoder.conv_norm_out", x, B, H * W, "d_no", True)
        y, _, _ = self._conv("decoder.conv_out", x, B, H, W, "d_out")
        return ops.nhwc_to_nchw(y, B, cfg.out_channels, H, W)

    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        img = self._decode(z, 0.0, 0.0)
        return SimpleNamespace(sample=img) if return_dict else (img,)

    def decode_scaled(self, latents: torch.Tensor) -> torch.Tensor:
        """src/UniGenPipeline.py:797-798 in one go: vae.decode(latents / scaling_factor + shift_factor)."""
        return self._decode(latents, self.config.scaling_factor, self.config.shift_factor)
