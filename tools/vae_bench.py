"""AutoencoderKL (FLUX geometry) at 1024^2: decode and encode latency, per-call conv shapes. usage: python tools/vae_bench.py [--size 1024] [--batch 1]"""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd.vae import AutoencoderKL

ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=1024); ap.add_argument("--batch", type=int, default=1); a = ap.parse_args()
dev = torch.device("cuda:0")
vae = AutoencoderKL(device=dev, dtype=torch.bfloat16)
vae.init_synthetic_(seed=0)
g = torch.Generator(device=dev).manual_seed(0)
img = torch.randn(a.batch, 3, a.size, a.size, generator=g, device=dev).to(torch.bfloat16)
lat = torch.randn(a.batch, 16, a.size // 8, a.size // 8, generator=g, device=dev).to(torch.bfloat16)
from unigen_amd import ops
_flops = [0.0]
_conv = ops.conv2d_nhwc
def _counting_conv(x, w, bias, out, *, B, H, W, Ho, Wo, KH, KW, **kw):
    _flops[0] += 2.0 * B * Ho * Wo * w.shape[0] * KH * KW * w.shape[-1]
    return _conv(x, w, bias, out, B=B, H=H, W=W, Ho=Ho, Wo=Wo, KH=KH, KW=KW, **kw)
import unigen_amd.vae as _v
_v.ops.conv2d_nhwc = _counting_conv
def count(f):
    _flops[0] = 0.0; f(); return _flops[0]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
dec = timeit(lambda: vae.decode(lat))
enc = timeit(lambda: vae.encode(img))
fd, fe = count(lambda: vae.decode(lat)), count(lambda: vae.encode(img))
print("VAE_BENCH", json.dumps(dict(size=a.size, batch=a.batch, decode_ms=round(dec, 2), encode_ms=round(enc, 2), decode_conv_tflop=round(fd / 1e12, 2),
      encode_conv_tflop=round(fe / 1e12, 2), decode_tflops_effective=round(fd / dec / 1e9, 1), encode_tflops_effective=round(fe / enc / 1e9, 1))))
