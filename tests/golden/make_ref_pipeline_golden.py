"""Fixtures that ORIGINATE IN THE REFERENCE: the control-image preparation of its two pipelines, run in the build container.

    python tests/golden/make_ref_pipeline_golden.py        # needs /root/reference; writes tests/golden/ref_pipeline.safetensors

src/UniGenPipeline.py   UniGenFLUXPipeline.prepare_image :457, UniGenSD3Pipeline.prepare_image :107
compiled one definition at a time by tests/golden/ref_harness.py (torch-only namespace, pinned sha256, builtin whitelist). `self` is an attribute bag
whose `image_processor` raises when touched: every case passes tensors, the branch the reference itself takes for tensors. Cases: one image for
the whole batch, one image per prompt with num_images_per_prompt > 1, packed latents (3-D), classifier-free guidance, guess mode, a one-channel
depth map. The file holds tensors only (inputs as int64 argument vectors + images, and the reference's outputs)."""
from __future__ import annotations

import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import REF, _tripwire, bag, compile_reference_function  # noqa: E402

# (name, image shape, batch_size (= prompts x images per prompt), num_images_per_prompt, cfg, guess_mode)
FLUX_CASES = [("one_for_all", (1, 3, 16, 24), 4, 1, 0, 0), ("per_prompt_x2", (2, 3, 16, 24), 4, 2, 0, 0), ("packed_latents", (1, 6, 64), 3, 1, 0, 0),
              ("per_prompt_x1", (3, 3, 8, 8), 3, 1, 0, 0)]
SD3_CASES = [("one_for_all_cfg", (1, 3, 16, 24), 2, 1, 1, 0), ("per_prompt_x2_cfg", (2, 3, 16, 16), 4, 2, 1, 0), ("depth_one_channel", (2, 1, 16, 16), 2, 1, 1, 0),
             ("guess_mode", (1, 3, 8, 8), 2, 1, 1, 1), ("no_cfg", (2, 3, 8, 8), 2, 1, 0, 0)]


def main() -> None:
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found: the fixtures can only be regenerated in the build container")
    P = "src/UniGenPipeline.py"
    flux, l0 = compile_reference_function(P, "prepare_image", "UniGenFLUXPipeline")
    sd3, l1 = compile_reference_function(P, "prepare_image", "UniGenSD3Pipeline")
    print("compiled reference definitions at lines", l0, l1)
    me = bag(image_processor=_tripwire("image_processor"))
    g = torch.Generator().manual_seed(12443)
    fx = {}
    for name, shape, bs, nipp, cfg, guess in FLUX_CASES:
        img = torch.randn(*shape, generator=g)
        out = flux(me, img, shape[-1], shape[-2], bs, nipp, "cpu", torch.bfloat16)
        fx[f"flux.{name}.image"], fx[f"flux.{name}.args"], fx[f"flux.{name}.out"] = img, torch.tensor([bs, nipp, cfg, guess]), out.contiguous()
    for name, shape, bs, nipp, cfg, guess in SD3_CASES:
        img = torch.randn(*shape, generator=g)
        out = sd3(me, img, shape[-1], shape[-2], bs, nipp, "cpu", torch.bfloat16, do_classifier_free_guidance=bool(cfg), guess_mode=bool(guess))
        fx[f"sd3.{name}.image"], fx[f"sd3.{name}.args"], fx[f"sd3.{name}.out"] = img, torch.tensor([bs, nipp, cfg, guess]), out.contiguous()
    path = os.path.join(os.environ.get("UG_GOLDEN_OUT", HERE), "ref_pipeline.safetensors")
    save_file(fx, path, metadata={"origin": "outputs of the reference's own prepare_image methods (src/UniGenPipeline.py:107, :457), executed from /root/reference by tests/golden/make_ref_pipeline_golden.py"})
    print("wrote", path, len(fx), "tensors", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
