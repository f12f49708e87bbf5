"""Denoise loop of UniGenFLUXPipeline on MI355X (reference: src/UniGenPipeline.py:452-1133, loop :721-789 / :1048-1116).

Scope: the timestep schedule, the transformer call with the reference's kwargs and the Euler step.
Text encoders (CLIP/T5) are the row before the hot path and are NOT part of this package: the pipeline accepts what they produce
- prompt embeds, pooled embeds - exactly as the reference's `__call__` does through its `prompt_embeds=`, `pooled_prompt_embeds=`,
`condition_pooled_prompt_embeds=` arguments, or DELEGATES to an injected `encode_prompt` callable with the reference's own keyword
arguments (src/UniGenPipeline.py:575-619), so `infer.py:204`'s call shape `pipe(prompt=..., condition_prompt=..., control_image=<image>)`
runs once the caller attaches its encoders. The VAE either side (:635-636, :797-798) is likewise an attribute: any object with the
diffusers AutoencoderKL surface (`encode(x).latent_dist.sample()`, `decode(z, return_dict=False)[0]`, `config.scaling_factor /
shift_factor`) - the native one of unigen_amd/vae.py or the caller's. The reference defines `__call__` twice (single- and
multi-condition; the second shadows the first); here one `__call__` serves both: lists select the multi-condition path.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

import torch

from . import ops

BF = torch.bfloat16


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15) -> float:
    """diffusers pipeline_flux.calculate_shift (src/UniGenPipeline.py:664-670)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


def flow_match_sigmas(num_inference_steps: int, *, sigmas: Optional[Sequence[float]] = None, shift: float = 1.0,
                      use_dynamic_shifting: bool = False, mu: Optional[float] = None) -> List[float]:
    """FlowMatchEulerDiscreteScheduler.set_timesteps: sigma' = shift*s/(1+(shift-1)s), or exp(mu)/(exp(mu)+(1/s-1))
    with dynamic shifting (FLUX-dev); a final 0 is appended. FLUX-schnell: shift = 1, no dynamic shifting."""
    if sigmas is None:
        sigmas = [1.0 - i * (1.0 - 1.0 / num_inference_steps) / max(num_inference_steps - 1, 1) for i in range(num_inference_steps)]
    out = []
    for s in sigmas:
        s = float(s)
        if use_dynamic_shifting:
            s = math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))
        else:
            s = shift * s / (1.0 + (shift - 1.0) * s)
        out.append(s)
    return out + [0.0]


def sd3_default_sigmas(num_inference_steps: int, shift: float = 3.0, num_train_timesteps: int = 1000) -> List[float]:
    """The un-shifted sigmas FlowMatchEulerDiscreteScheduler.set_timesteps(num_inference_steps) starts from when the caller passes none
    (diffusers 0.32.2, the release pinned in the reference's environment.yaml; UniGenSD3Pipeline, src/UniGenPipeline.py:355-357):
    `linspace(sigma_to_t(sigma_max), sigma_to_t(sigma_min), n) / num_train_timesteps`, where the scheduler's __init__ has ALREADY applied the
    static shift to its training sigmas, so sigma_max = 1 and sigma_min = shift * (1/T) / (1 + (shift - 1) / T) (0.002994 for shift 3) - and
    set_timesteps then applies the shift a second time (flow_match_sigmas)."""
    s_min = shift * (1.0 / num_train_timesteps) / (1.0 + (shift - 1.0) / num_train_timesteps)
    s_max = shift * 1.0 / (1.0 + (shift - 1.0) * 1.0)
    n = num_inference_steps
    return [s_max + i * (s_min - s_max) / max(n - 1, 1) for i in range(n)]


def _step32(sigma: float, sigma_next: float) -> float:
    """`sigma_next - sigma` as the scheduler computes it: both are elements of an fp32 tensor, the difference an fp32 subtraction."""
    s = torch.tensor([sigma, sigma_next], dtype=torch.float32)
    return float(s[1] - s[0])


def _t32(sigma: float) -> float:
    """A timestep as the scheduler holds it: `sigmas * num_train_timesteps` on the fp32 sigma tensor (one fp32 product)."""
    return float(torch.tensor(sigma, dtype=torch.float32) * 1000.0)


def prepare_latent_image_ids(height: int, width: int, device, dtype) -> torch.Tensor:
    """FluxPipeline._prepare_latent_image_ids: [h*w, 3], [:,1] = row, [:,2] = col."""
    ids = torch.zeros(height, width, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(height)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(width)[None, :]
    return ids.reshape(height * width, 3).to(device=device, dtype=dtype)


def pack_latents(latents: torch.Tensor) -> torch.Tensor:
    """FluxPipeline._pack_latents: [B, C, H, W] -> [B, H/2*W/2, 4C]. GPU tensors go through ug_pack_latents (C ABI); the view/permute
    formula below is the definition (and what host-side tensors take)."""
    if latents.is_cuda:
        return ops.pack_latents(latents)
    B, C, H, W = latents.shape
    return latents.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 2) * (W // 2), C * 4)


def unpack_latents(latents: torch.Tensor, height: int, width: int, vae_scale_factor: int = 8) -> torch.Tensor:
    """FluxPipeline._unpack_latents (height/width in pixels)."""
    B, _, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    if latents.is_cuda:
        return ops.unpack_latents(latents, h, w)
    return latents.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, ch // 4, h, w)


@torch.no_grad()
def denoise_loop(transformer, *, latents: torch.Tensor, control_tokens, prompt_embeds: torch.Tensor, pooled_prompt_embeds: torch.Tensor,
                 condition_pooled_prompt_embeds, text_ids: torch.Tensor, latent_image_ids: torch.Tensor, condition_ids,
                 num_inference_steps: int = 4, sigmas: Optional[Sequence[float]] = None, guidance_scale: float = 3.5,
                 conditioning_scale: float = 1.0, shift: float = 1.0, use_dynamic_shifting: bool = False, gate_uniforms=None,
                 true_cfg_scale: float = 1.0, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 negative_pooled_prompt_embeds: Optional[torch.Tensor] = None, negative_text_ids: Optional[torch.Tensor] = None,
                 negative_gate_uniforms=None, callback_on_step_end=None, callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",),
                 pipeline=None, shift_params: Optional[dict] = None) -> torch.Tensor:
    """The hot loop (src/UniGenPipeline.py:721-789): per step `timestep = t.expand(B).to(latents.dtype)`, transformer(timestep / 1000)[0],
    latents = latents + (sigma_next - sigma) * noise_pred evaluated in fp32 and cast back. Updates and returns `latents` in place.
    True classifier-free guidance (`:748-763`: `true_cfg_scale > 1` with negative embeds): a second forward on the negative prompt - called, as
    the reference does, WITHOUT `conditioning_scale` (the forward's default) - and `neg + true_cfg_scale * (pred - neg)` in the latents' dtype
    (ug_cfg_combine: the three bf16 tensor ops' roundings). `callback_on_step_end(pipeline, i, t, {name: tensor})` (`:774-781`) may return
    replacements for `latents` and `prompt_embeds`."""
    # :663-670: mu from the token count and the SCHEDULER's base / max sequence lengths and shifts (the diffusers defaults when it names none)
    sp = shift_params or {}
    mu = calculate_shift(latents.shape[1], sp.get("base_image_seq_len", 256), sp.get("max_image_seq_len", 4096), sp.get("base_shift", 0.5),
                         sp.get("max_shift", 1.15)) if use_dynamic_shifting else None
    sig = flow_match_sigmas(num_inference_steps, sigmas=sigmas, shift=shift, use_dynamic_shifting=use_dynamic_shifting, mu=mu)
    B = latents.shape[0]
    guidance = None
    if transformer.config.guidance_embeds:
        guidance = torch.full([B], guidance_scale, device=latents.device, dtype=torch.float32)
    latents = latents.contiguous()
    do_true_cfg = true_cfg_scale > 1 and negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None
    if do_true_cfg and negative_text_ids is None:
        negative_text_ids = torch.zeros(negative_prompt_embeds.shape[1], 3, device=latents.device, dtype=text_ids.dtype)
    for i in range(num_inference_steps):
        # `t.expand(B).to(latents.dtype)`: built on the device (a fill kernel, no host copy -> the loop is HIP-graph capturable)
        timestep = torch.full((B,), _t32(sig[i]), dtype=torch.float32, device=latents.device).to(latents.dtype)
        uni = None if gate_uniforms is None else gate_uniforms[i]
        noise_pred = transformer(hidden_states=latents, condition_hidden_states=control_tokens, conditioning_scale=conditioning_scale,
                                 encoder_hidden_states=prompt_embeds, pooled_projections=pooled_prompt_embeds,
                                 condition_pooled_projections=condition_pooled_prompt_embeds, timestep=timestep / 1000, txt_ids=text_ids,
                                 img_ids=latent_image_ids, guidance=guidance, condition_ids=condition_ids, gate_uniform=uni)[0]
        if do_true_cfg:
            nuni = None if negative_gate_uniforms is None else negative_gate_uniforms[i]
            neg = transformer(hidden_states=latents, condition_hidden_states=control_tokens, encoder_hidden_states=negative_prompt_embeds,
                              pooled_projections=negative_pooled_prompt_embeds, condition_pooled_projections=condition_pooled_prompt_embeds,
                              timestep=timestep / 1000, txt_ids=negative_text_ids, img_ids=latent_image_ids, guidance=guidance,
                              condition_ids=condition_ids, gate_uniform=nuni)[0]
            noise_pred = ops.cfg_combine(neg.contiguous(), noise_pred.contiguous(), float(true_cfg_scale), torch.empty_like(neg, memory_format=torch.contiguous_format))
        ops.euler_step(latents, noise_pred, _step32(sig[i], sig[i + 1]))
        if callback_on_step_end is not None:
            t = torch.tensor(_t32(sig[i]), dtype=torch.float32, device=latents.device)
            have = dict(latents=latents, prompt_embeds=prompt_embeds, noise_pred=noise_pred, timestep=timestep)
            outs = callback_on_step_end(pipeline, i, t, {k: have[k] for k in callback_on_step_end_tensor_inputs})
            new_latents = outs.pop("latents", latents)
            if new_latents is not latents:
                latents = new_latents.to(latents.dtype).contiguous()
            prompt_embeds = outs.pop("prompt_embeds", prompt_embeds)
    return latents


class UniGenFLUXPipeline:
    """Call-surface twin of the reference `UniGenFLUXPipeline(FluxPipeline)`: the denoise loop runs here; text encoding and the VAE are
    delegated to attributes the caller attaches (`encode_prompt`, `vae`, `image_processor`), with the reference's keyword arguments."""

    def __init__(self, transformer=None, scheduler_config: Optional[dict] = None, vae_scale_factor: int = 8, encode_prompt=None, vae=None,
                 image_processor=None):
        self.transformer = transformer
        self.vae_scale_factor = vae_scale_factor
        self.default_sample_size = 128
        sc = dict(shift=1.0, use_dynamic_shifting=False, base_image_seq_len=256, max_image_seq_len=4096, base_shift=0.5, max_shift=1.15)
        sc.update(scheduler_config or {})
        self.scheduler = SimpleNamespace(config=sc)
        self._device, self._dtype = None, BF
        self.encode_prompt = encode_prompt      # callable(prompt=, prompt_2=, prompt_embeds=, pooled_prompt_embeds=, device=, ...) -> (embeds, pooled, text_ids)
        self.vae = vae                          # AutoencoderKL surface
        self.image_processor = image_processor  # .preprocess(image, height=, width=) / .postprocess(image, output_type=)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, transformer=None, **kwargs) -> "UniGenFLUXPipeline":
        """infer.py:146-149 builds the pipeline with `transformer=None` and assigns `.transformer` afterwards. The scheduler config is read
        from disk (model_index-style layout); `vae/` is loaded into the native VAE when present; text encoders are attached by the caller."""
        import json, os
        sc, vae = {}, kwargs.get("vae")
        if pretrained_model_name_or_path is not None:
            root = os.fspath(pretrained_model_name_or_path)
            p = os.path.join(root, "scheduler", "scheduler_config.json")
            if os.path.exists(p):
                with open(p) as f:
                    raw = json.load(f)
                sc = {k: raw[k] for k in ("shift", "use_dynamic_shifting", "base_image_seq_len", "max_image_seq_len", "base_shift", "max_shift") if k in raw}
            if vae is None and os.path.exists(os.path.join(root, "vae", "config.json")):
                from .vae import AutoencoderKL
                vae = AutoencoderKL.from_pretrained(os.path.join(root, "vae"))
        return cls(transformer=transformer, scheduler_config=sc, encode_prompt=kwargs.get("encode_prompt"), vae=vae,
                   image_processor=kwargs.get("image_processor"))

    def to(self, device=None, dtype=None):
        if device is not None:
            self._device = torch.device(device)
        if dtype is not None:
            self._dtype = dtype
        if self.transformer is not None:
            self.transformer.to(device=device, dtype=dtype)
        if self.vae is not None and hasattr(self.vae, "to"):
            self.vae.to(device=device, dtype=dtype)
        return self

    # ---- the two delegated stages -----------------------------------------------------------------------------------
    def _encode(self, what: str, prompt, prompt_2, prompt_embeds, pooled_prompt_embeds, device, num_images_per_prompt, max_sequence_length):
        """self.encode_prompt(...) with the reference's kwargs (src/UniGenPipeline.py:575-619). Pre-computed embeds pass through."""
        if prompt is None:
            return prompt_embeds, pooled_prompt_embeds
        if self.encode_prompt is None:
            raise NotImplementedError(f"`{what}` given as text but no text encoder is attached: set `pipe.encode_prompt` to a callable with "
                                      "FluxPipeline.encode_prompt's signature (CLIP/T5 are outside this package), or pass the embeds")
        out = self.encode_prompt(prompt=prompt, prompt_2=prompt_2, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds, device=device,
                                 num_images_per_prompt=num_images_per_prompt, max_sequence_length=max_sequence_length, lora_scale=None)
        return out[0], out[1]

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype, guess_mode=False):
        """The reference's method of the same name (src/UniGenPipeline.py:457-483): a tensor passes through untouched, anything else goes through
        `image_processor.preprocess`; ONE control image serves the whole batch (repeated `batch_size` times), otherwise every image is repeated
        `num_images_per_prompt` times next to its prompt (repeat_interleave), then device / dtype."""
        if not isinstance(image, torch.Tensor):
            if self.image_processor is None:
                raise NotImplementedError("control_image is not a tensor and no image processor is attached: set `pipe.image_processor` (.preprocess(image, height=, width=))")
            image = self.image_processor.preprocess(image, height=height, width=width)
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        return image.repeat_interleave(repeat_by, dim=0).to(device=device, dtype=dtype)

    def _encode_control(self, image: torch.Tensor, dtype, generator):
        """vae.encode -> (x - shift) * scale -> _pack_latents of a prepared control image (:634-647). Packed latents [B, N, 4C] pass through."""
        if image.ndim == 3:
            return image
        if image.ndim != 4:
            raise ValueError("control_image must be an image batch [B, 3, H, W] or packed condition latents [B, N, 4*C]")
        if self.vae is None:
            raise NotImplementedError("control_image given as pixels but no VAE is attached: set `pipe.vae` (AutoencoderKL surface) or pass packed latents")
        image = image.to(device=self.transformer.device, dtype=getattr(self.vae, "dtype", dtype))
        if hasattr(self.vae, "encode_scaled"):           # native VAE (unigen_amd/vae.py): sampling and the affine run in ug_vae_sample
            z = self.vae.encode_scaled(image, generator=generator)
        else:
            z = self.vae.encode(image).latent_dist.sample(generator=generator)
            z = (z - self.vae.config.shift_factor) * self.vae.config.scaling_factor
        return pack_latents(z.to(dtype).contiguous())

    def _decode(self, latents: torch.Tensor, height, width, output_type):
        """_unpack_latents -> z / scale + shift -> vae.decode -> postprocess (:796-799)."""
        if self.vae is None:
            raise NotImplementedError("output_type other than 'latent' needs a VAE: set `pipe.vae` (AutoencoderKL surface)")
        z = unpack_latents(latents, height, width, self.vae_scale_factor)
        if hasattr(self.vae, "decode_scaled"):           # native VAE: the un-scaling is folded into the NCHW -> NHWC conversion
            image = self.vae.decode_scaled(z.to(getattr(self.vae, "dtype", z.dtype)))
        else:
            z = (z / self.vae.config.scaling_factor) + self.vae.config.shift_factor
            image = self.vae.decode(z.to(getattr(self.vae, "dtype", z.dtype)), return_dict=False)[0]
        if self.image_processor is not None:
            return self.image_processor.postprocess(image, output_type=output_type)
        return image

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, condition_prompt=None, control_image=None, conditioning_scale: float = 1.0,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 28, sigmas=None,
                 guidance_scale: float = 3.5, num_images_per_prompt: int = 1, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, pooled_prompt_embeds: Optional[torch.Tensor] = None,
                 condition_prompt_embeds=None, condition_pooled_prompt_embeds=None, condition_ids=None, output_type: str = "latent",
                 return_dict: bool = True, max_sequence_length: int = 512, dtype: torch.dtype = BF, gate_uniforms=None,
                 true_cfg_scale: float = 1.0, negative_prompt=None, negative_prompt_2=None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 negative_pooled_prompt_embeds: Optional[torch.Tensor] = None, negative_gate_uniforms=None, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",), **kwargs):
        tr = self.transformer
        dev = tr.device
        for unsupported in ("ip_adapter_image", "ip_adapter_image_embeds", "negative_ip_adapter_image", "negative_ip_adapter_image_embeds"):
            if kwargs.get(unsupported) is not None:
                raise NotImplementedError(f"{unsupported}: IP-Adapter image embeds are outside this package's scope (SURVEY section 8)")
        if control_image is None:
            raise ValueError("control_image is required (pixels [B, 3, H, W] with a VAE attached, or packed condition latents [B, N, 4*C])")
        multi = isinstance(control_image, (list, tuple))
        prompt_embeds, pooled_prompt_embeds = self._encode("prompt", prompt, prompt_2, prompt_embeds, pooled_prompt_embeds, dev, num_images_per_prompt,
                                                           max_sequence_length)
        if multi:       # one condition prompt (or pooled embed) per condition (second __call__ of the reference, :927-946)
            cps = condition_prompt if isinstance(condition_prompt, (list, tuple)) else [condition_prompt] * len(control_image)
            cpe = condition_pooled_prompt_embeds if isinstance(condition_pooled_prompt_embeds, (list, tuple)) else [condition_pooled_prompt_embeds] * len(control_image)
            condition_pooled_prompt_embeds = [self._encode("condition_prompt", cp, None, None, ce, dev, num_images_per_prompt, max_sequence_length)[1]
                                              for cp, ce in zip(cps, cpe)]
        else:
            condition_pooled_prompt_embeds = self._encode("condition_prompt", condition_prompt, None, condition_prompt_embeds, condition_pooled_prompt_embeds,
                                                          dev, num_images_per_prompt, max_sequence_length)[1]
        if prompt_embeds is None or pooled_prompt_embeds is None or condition_pooled_prompt_embeds is None or \
                (multi and any(c is None for c in condition_pooled_prompt_embeds)):
            raise ValueError("prompt (or prompt_embeds + pooled_prompt_embeds) and condition_prompt (or condition_pooled_prompt_embeds) are required")
        # true classifier-free guidance (src/UniGenPipeline.py:567-570, 594-607): a negative prompt (or its embeds) and true_cfg_scale > 1
        has_neg = negative_prompt is not None or (negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None)
        do_true_cfg = true_cfg_scale > 1 and has_neg
        if do_true_cfg:
            negative_prompt_embeds, negative_pooled_prompt_embeds = self._encode("negative_prompt", negative_prompt, negative_prompt_2, negative_prompt_embeds,
                                                                                 negative_pooled_prompt_embeds, dev, num_images_per_prompt, max_sequence_length)
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        hl, wl = height // (self.vae_scale_factor * 2), width // (self.vae_scale_factor * 2)
        B = prompt_embeds.shape[0]                       # batch_size * num_images_per_prompt: the encoders repeat per image
        vdt = getattr(self.vae, "dtype", dtype) if self.vae is not None else dtype
        control = []
        for c in (control_image if multi else [control_image]):
            c = self.prepare_image(image=c, width=width, height=height, batch_size=B, num_images_per_prompt=num_images_per_prompt, device=dev, dtype=vdt)
            if c.ndim == 4:                              # the latents take the prepared control image's size (:631, :957), not the arguments'
                height, width = c.shape[-2:]
                hl, wl = height // (self.vae_scale_factor * 2), width // (self.vae_scale_factor * 2)
            control.append(self._encode_control(c, dtype, generator))
        if not multi:
            control = control[0]
        if latents is None:
            latents = torch.randn(B, hl * wl, tr.config.in_channels, generator=generator, device=dev if generator is None or generator.device.type != "cpu" else "cpu",
                                  dtype=torch.float32).to(dev)
        latents = latents.to(device=dev, dtype=dtype).clone()
        ids = prepare_latent_image_ids(hl, wl, dev, dtype)
        if condition_ids is None:
            condition_ids = [ids for _ in control] if multi else ids
        text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=dev, dtype=dtype)
        cast = lambda t: t.to(device=dev, dtype=dtype)
        out = denoise_loop(tr, latents=latents, control_tokens=[cast(c) for c in control] if multi else cast(control),
                           prompt_embeds=cast(prompt_embeds), pooled_prompt_embeds=cast(pooled_prompt_embeds),
                           condition_pooled_prompt_embeds=[cast(c) for c in condition_pooled_prompt_embeds] if multi else cast(condition_pooled_prompt_embeds),
                           text_ids=text_ids, latent_image_ids=ids, condition_ids=condition_ids, num_inference_steps=num_inference_steps,
                           sigmas=sigmas, guidance_scale=guidance_scale, conditioning_scale=conditioning_scale,
                           shift=self.scheduler.config["shift"], use_dynamic_shifting=self.scheduler.config["use_dynamic_shifting"],
                           gate_uniforms=gate_uniforms, true_cfg_scale=true_cfg_scale if do_true_cfg else 1.0,
                           negative_prompt_embeds=cast(negative_prompt_embeds) if do_true_cfg else None,
                           negative_pooled_prompt_embeds=cast(negative_pooled_prompt_embeds) if do_true_cfg else None,
                           negative_gate_uniforms=negative_gate_uniforms, callback_on_step_end=callback_on_step_end,
                           callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs, pipeline=self, shift_params=self.scheduler.config)
        if output_type != "latent":
            out = self._decode(out, height, width, output_type)
        if not return_dict:
            return (out,)
        return SimpleNamespace(images=out)


def sd3_default_sigmas_unshifted(num_inference_steps: int, num_train_timesteps: int = 1000) -> List[float]:
    """The same default when the scheduler uses dynamic shifting: its __init__ then leaves the training sigmas unshifted (sigma_min = 1 / T)."""
    n = num_inference_steps
    return [1.0 + i * (1.0 / num_train_timesteps - 1.0) / max(n - 1, 1) for i in range(n)]


def control_keep(num_steps: int, start=0.0, end=1.0) -> List[float]:
    """`controlnet_keep` of the reference's SD3 pipeline (src/UniGenPipeline.py:364-370): step i keeps the control branch (1.0) unless
    i / n < control_guidance_start or (i + 1) / n > control_guidance_end (0.0); lists are taken at their first entry, as the reference does."""
    s = start[0] if isinstance(start, (list, tuple)) else start
    e = end[0] if isinstance(end, (list, tuple)) else end
    return [1.0 - float(i / num_steps < s or (i + 1) / num_steps > e) for i in range(num_steps)]


@torch.no_grad()
def sd3_denoise_loop(transformer, *, latents: torch.Tensor, control_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                     pooled_prompt_embeds: torch.Tensor, condition_pooled_prompt_embeds: torch.Tensor, num_inference_steps: int = 28,
                     guidance_scale: float = 7.0, conditioning_scale=1.0, shift: float = 3.0, sigmas: Optional[Sequence[float]] = None,
                     gate_uniforms=None, use_dynamic_shifting: bool = False, mu: Optional[float] = None, control_guidance_start=0.0,
                     control_guidance_end=1.0, callback_on_step_end=None, callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",),
                     pipeline=None, condition_types=None) -> torch.Tensor:
    """UniGenSD3Pipeline.__call__ loop (src/UniGenPipeline.py:372-433). With guidance_scale > 1 the caller passes prompt / pooled / condition
    embeds already doubled as [negative | positive] (reference :286-290); latents [B, C, H, W] are duplicated per step, the two halves of
    the prediction are combined with classifier-free guidance, then the flow-match Euler step. The timestep is passed unscaled. Per step
    `conditioning_scale * controlnet_keep[i]` (:383-389; a list-valued scale is taken at its first entry); `callback_on_step_end(pipeline, i, t,
    {name: tensor})` may replace `latents` and `prompt_embeds` (:416-427; the negative embeds it may also return are not read again by the loop)."""
    cfg_on = guidance_scale > 1.0
    if sigmas is None:
        sigmas = sd3_default_sigmas_unshifted(num_inference_steps) if use_dynamic_shifting else sd3_default_sigmas(num_inference_steps, shift)
    else:
        num_inference_steps = len(sigmas)
    sig = flow_match_sigmas(num_inference_steps, sigmas=sigmas, shift=shift, use_dynamic_shifting=use_dynamic_shifting, mu=mu)
    keep = control_keep(num_inference_steps, control_guidance_start, control_guidance_end)
    scale = conditioning_scale[0] if isinstance(conditioning_scale, (list, tuple)) else conditioning_scale
    B = latents.shape[0]
    latents = latents.contiguous()
    ctrl = torch.cat([control_latents] * 2) if cfg_on and control_latents.shape[0] == B else control_latents
    pred = torch.empty_like(latents)
    for i in range(num_inference_steps):
        x_in = torch.cat([latents] * 2) if cfg_on else latents
        t = torch.full((x_in.shape[0],), _t32(sig[i]), device=latents.device, dtype=torch.float32)
        uni = None if gate_uniforms is None else gate_uniforms[i]
        out = transformer(hidden_states=x_in, condition_hidden_states=ctrl, conditioning_scale=scale * keep[i], timestep=t,
                          encoder_hidden_states=prompt_embeds, pooled_projections=pooled_prompt_embeds,
                          condition_pooled_projections=condition_pooled_prompt_embeds, gate_uniform=uni, condition_types=condition_types)[0]
        if cfg_on:
            ops.cfg_combine(out[:B].contiguous(), out[B:].contiguous(), guidance_scale, pred)
        else:
            pred = out
        ops.euler_step(latents, pred, _step32(sig[i], sig[i + 1]))
        if callback_on_step_end is not None:
            have = dict(latents=latents, prompt_embeds=prompt_embeds, noise_pred=pred, timestep=t)
            outs = callback_on_step_end(pipeline, i, t[0], {k: have[k] for k in callback_on_step_end_tensor_inputs})
            new_latents = outs.pop("latents", latents)
            if new_latents is not latents:
                latents = new_latents.to(latents.dtype).contiguous()
            prompt_embeds = outs.pop("prompt_embeds", prompt_embeds)
    return latents


class UniGenSD3Pipeline:
    """Call-surface twin of the reference `UniGenSD3Pipeline` for the transformer side (encoders / VAE out of scope, as above)."""

    def __init__(self, transformer=None, scheduler_config: Optional[dict] = None, vae_scale_factor: int = 8, encode_prompt=None, vae=None,
                 image_processor=None):
        self.transformer = transformer
        self.vae_scale_factor = vae_scale_factor
        self.default_sample_size = 128
        sc = dict(shift=3.0, use_dynamic_shifting=False, base_image_seq_len=256, max_image_seq_len=4096, base_shift=0.5, max_shift=1.15)
        sc.update(scheduler_config or {})
        self.scheduler = SimpleNamespace(config=sc)
        # delegated stages, as in UniGenFLUXPipeline: encode_prompt(prompt=, ..., do_classifier_free_guidance=) -> (embeds, negative embeds,
        # pooled, negative pooled) like StableDiffusion3Pipeline.encode_prompt; vae = AutoencoderKL surface
        self.encode_prompt, self.vae, self.image_processor = encode_prompt, vae, image_processor

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, transformer=None, **kwargs) -> "UniGenSD3Pipeline":
        return cls(transformer=transformer)

    def to(self, device=None, dtype=None):
        if self.transformer is not None:
            self.transformer.to(device=device, dtype=dtype)
        return self

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype, do_classifier_free_guidance=False, guess_mode=False):
        """The reference's method of the same name (src/UniGenPipeline.py:107-141): as UniGenFLUXPipeline.prepare_image, then the batch doubled for
        classifier-free guidance (unless guess_mode) and a one-channel map (depth) repeated to three channels."""
        if not isinstance(image, torch.Tensor):
            if self.image_processor is None:
                raise NotImplementedError("control_image is not a tensor and no image processor is attached: set `pipe.image_processor` (.preprocess(image, height=, width=))")
            image = self.image_processor.preprocess(image, height=height, width=width)
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        image = image.repeat_interleave(repeat_by, dim=0).to(device=device, dtype=dtype)
        if do_classifier_free_guidance and not guess_mode:
            image = torch.cat([image] * 2)
        if image.shape[1] == 1:
            image = image.repeat(1, 3, 1, 1)
        return image

    @torch.no_grad()
    def __call__(self, prompt=None, condition_prompt=None, control_image=None, conditioning_scale: float = 1.0, height=None, width=None,
                 num_inference_steps: int = 28, guidance_scale: float = 7.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 condition_pooled_prompt_embeds=None, output_type: str = "latent", return_dict: bool = True, gate_uniforms=None,
                 num_images_per_prompt: int = 1, control_use_vae_shift_factor: bool = True, sigmas=None, mu: Optional[float] = None,
                 control_guidance_start=0.0, control_guidance_end=1.0, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",), **kwargs):
        tr = self.transformer
        dev = tr.device
        cast = lambda t: t.to(device=dev, dtype=tr.dtype)
        cfg_on = guidance_scale > 1.0
        if prompt is not None or condition_prompt is not None:
            if self.encode_prompt is None:
                raise NotImplementedError("prompts given as text but no text encoder is attached: set `pipe.encode_prompt` (StableDiffusion3Pipeline.encode_prompt "
                                          "signature; CLIP/T5 are outside this package), or pass the embeds")
            if prompt is not None:
                prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
                    prompt=prompt, prompt_2=None, prompt_3=None, do_classifier_free_guidance=cfg_on, device=dev, num_images_per_prompt=num_images_per_prompt)[:4]
            if condition_prompt is not None:
                condition_pooled_prompt_embeds = self.encode_prompt(prompt=condition_prompt, prompt_2=None, prompt_3=None, do_classifier_free_guidance=False,
                                                                    device=dev, num_images_per_prompt=num_images_per_prompt)[2]
        if control_image is None:
            raise ValueError("control_image is required (pixels [B, 3 or 1, H, W] with a VAE attached, or VAE latents [B, C, H/8, W/8])")
        is_latent = isinstance(control_image, torch.Tensor) and control_image.ndim == 4 and control_image.shape[1] == tr.config.in_channels
        if not is_latent:
            # :293-308: prepare_image (one image serves the batch; every image repeated per num_images_per_prompt; doubled under CFG), then the VAE
            if self.vae is None:
                raise NotImplementedError("control_image given as pixels but no VAE is attached: set `pipe.vae` or pass VAE latents [B, C, H/8, W/8]")
            nb = (prompt_embeds.shape[0] if prompt_embeds is not None else 1)            # batch_size * num_images_per_prompt (the encoder repeats per image)
            control_image = self.prepare_image(image=control_image, width=width, height=height, batch_size=nb, num_images_per_prompt=num_images_per_prompt,
                                               device=dev, dtype=getattr(self.vae, "dtype", tr.dtype), do_classifier_free_guidance=cfg_on, guess_mode=False)
            z = self.vae.encode(control_image).latent_dist.sample(generator=generator)
            control_image = (z - (self.vae.config.shift_factor if control_use_vae_shift_factor else 0.0)) * self.vae.config.scaling_factor
        if output_type != "latent" and self.vae is None:
            raise NotImplementedError("output_type other than 'latent' needs a VAE: set `pipe.vae` (AutoencoderKL surface)")
        if cfg_on:
            if negative_prompt_embeds is None or negative_pooled_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds and negative_pooled_prompt_embeds")
            prompt_embeds = torch.cat([cast(negative_prompt_embeds), cast(prompt_embeds)], 0)
            pooled_prompt_embeds = torch.cat([cast(negative_pooled_prompt_embeds), cast(pooled_prompt_embeds)], 0)
            condition_pooled_prompt_embeds = torch.cat([cast(condition_pooled_prompt_embeds)] * 2, 0)
        if control_image.ndim != 4 or control_image.shape[1] != tr.config.in_channels:
            raise ValueError("control_image must be VAE latents [B, C, H/8, W/8] (vae.encode happens upstream)")
        # a prepared pixel batch arrives doubled under CFG (prepare_image); latents the caller passes directly hold one copy per sample
        B = control_image.shape[0] // (2 if (cfg_on and not is_latent) else 1)
        if latents is None:
            latents = torch.randn((B,) + tuple(control_image.shape[1:]), generator=generator, device=dev, dtype=torch.float32)
        # :323-339: with a dynamically shifting scheduler, mu from the latent grid's token count and the scheduler's own shift parameters
        sc = self.scheduler.config
        dyn = bool(sc.get("use_dynamic_shifting"))
        if dyn and mu is None:
            ps = tr.config.patch_size
            mu = calculate_shift((latents.shape[2] // ps) * (latents.shape[3] // ps), sc["base_image_seq_len"], sc["max_image_seq_len"], sc["base_shift"], sc["max_shift"])
        out = sd3_denoise_loop(tr, latents=cast(latents).clone(), control_latents=cast(control_image), prompt_embeds=cast(prompt_embeds),
                               pooled_prompt_embeds=cast(pooled_prompt_embeds), condition_pooled_prompt_embeds=cast(condition_pooled_prompt_embeds),
                               num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, conditioning_scale=conditioning_scale,
                               shift=self.scheduler.config["shift"], gate_uniforms=gate_uniforms, sigmas=sigmas,
                               use_dynamic_shifting=dyn, mu=mu, control_guidance_start=control_guidance_start, control_guidance_end=control_guidance_end,
                               callback_on_step_end=callback_on_step_end, callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs,
                               pipeline=self, condition_types=condition_prompt)
        if output_type != "latent":
            z = (out / self.vae.config.scaling_factor) + self.vae.config.shift_factor
            out = self.vae.decode(z.to(getattr(self.vae, "dtype", z.dtype)), return_dict=False)[0]
            if self.image_processor is not None:
                out = self.image_processor.postprocess(out, output_type=output_type)
        return SimpleNamespace(images=out) if return_dict else (out,)
