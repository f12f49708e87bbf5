"""SURVEY 8(f2) on the GPU: the path the reference actually takes to a runnable model - `from_pretrained(<local dir>)` (infer.py:115-119), then
`init_condition_block`, then `load_state_dict(<control checkpoint>, strict=False)` from a ZeRO directory / a `.bin` / a safetensors directory
(infer.py:124-141) or the `{module}_weights_{idx}.bin` files of src/hook.py:10-27 - followed by a HIP forward. Every wire format must give the
SAME BITS as the model whose weights were set from memory, and that forward must meet the committed `flux_tiny_single` fixture."""
import importlib
import json

import pytest
import torch
from safetensors.torch import save_file

from oracle import unigen_ref as R
from tests.test_flux_gpu import CONTROL, _to_dev
from tests.util import rel_l2, report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
CTRL_PREFIXES = ("control", "moe.", "shared_expert")


def _write_base(dirpath, cfg_d, base):
    """A diffusers-style transformer directory: config.json + two `diffusion_pytorch_model-0000x-of-00002.safetensors` shards + their index."""
    dirpath.mkdir(parents=True)
    (dirpath / "config.json").write_text(json.dumps(dict(cfg_d, _class_name="FluxTransformer2DModel", in_channels=64, guidance_embeds=False,
                                                         axes_dims_rope=[16, 56, 56])))
    keys = sorted(base)
    shards = {"diffusion_pytorch_model-00001-of-00002.safetensors": keys[::2], "diffusion_pytorch_model-00002-of-00002.safetensors": keys[1::2]}
    for fn, ks in shards.items():
        save_file({k: base[k].contiguous() for k in ks}, str(dirpath / fn))
    (dirpath / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps(
        {"metadata": {}, "weight_map": {k: fn for fn, ks in shards.items() for k in ks}}))


def test_every_wire_format_from_disk_gives_the_in_memory_forward(gpu, tmp_path):
    from tests.test_host_cpu import _write_zero_checkpoint
    from tests.test_oracle_cpu import load_golden
    from unigen_amd.checkpoint import load_control_checkpoint
    cfg_d, case, inp, g = load_golden("flux_tiny_single")
    rcfg = R.FluxConfig(condition_nums=1, **cfg_d)
    state = R.make_state(rcfg, seed=case["state_seed"], std=0.05, bias_std=0.02)          # bf16 tensors under the reference's key names
    cls = importlib.import_module("src.UniGenTransformer").UniGenFlux
    dev_inp = {k: _to_dev(v, gpu) for k, v in inp.items()}
    t = g["timestep"].to(gpu)

    def fwd(m):
        out, losses, outs = m(timestep=t, **dev_inp)
        return out.clone(), float(losses["moe_loss"]), outs["expert_counts"].clone()

    # the model set from memory (what every other GPU test does), checked against the committed fixture
    mem = cls.from_config(cfg_d, device=gpu, dtype=BF)
    mem.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
    res = mem.load_state_dict({k: v.to(gpu) for k, v in state.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    want, want_loss, want_cnt = fwd(mem)
    m = report("f2_memory_model_vs_golden", want, g["out.bf16"], err_hip_vs_fp32=rel_l2(want, g["out.fp32"]),
               err_oraclebf16_vs_fp32=rel_l2(g["out.bf16"], g["out.fp32"]))
    assert m["err_hip_vs_fp32"] <= 1.25 * m["err_oraclebf16_vs_fp32"] + 1e-3, m

    base = {k: v for k, v in state.items() if not k.startswith(CTRL_PREFIXES)}
    ctrl = {k: v.float() for k, v in state.items() if k.startswith(CTRL_PREFIXES)}            # fp32 on disk, as zero_to_fp32 leaves it
    assert base and ctrl and set(base) | set(ctrl) == set(state)
    _write_base(tmp_path / "FLUX.1-schnell" / "transformer", cfg_d, base)
    keys = sorted(ctrl)
    formats = {}
    # 1a. ZeRO directory holding the consolidated file (script/infer.sh:44-46)
    d = tmp_path / "zero_consolidated"; d.mkdir(); (d / "latest").write_text("global_step100")
    torch.save(ctrl, str(d / "pytorch_model_fp32.bin")); formats["zero_dir_consolidated"] = d
    # 1b. raw ZeRO-3 and ZeRO-2 shards under <dir>/<tag>/ (merged by unigen_amd.checkpoint.merge_zero_checkpoint)
    for stage, world in ((3, 4), (2, 2)):
        d = tmp_path / f"zero{stage}"; d.mkdir()
        _write_zero_checkpoint(d, "global_step100", ctrl, {}, {}, world, stage)
        formats[f"zero{stage}_shards"] = d
    # 2. one torch.save file
    f = tmp_path / "pytorch_model_fp32.bin"; torch.save(ctrl, str(f)); formats["single_bin"] = f
    # 3. a directory of safetensors shards
    d = tmp_path / "st"; d.mkdir()
    save_file({k: ctrl[k].contiguous() for k in keys[::2]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: ctrl[k].contiguous() for k in keys[1::2]}, str(d / "model-00002-of-00002.safetensors")); formats["safetensors_dir"] = d
    # 4. {module}_weights_{idx}.bin, one partial state dict per trainable module family (src/hook.py:10-27)
    d = tmp_path / "hook"; d.mkdir()
    fam = {}
    for k in keys:
        fam.setdefault(k.split(".")[0], {})[k] = ctrl[k]
    for i, (name, part) in enumerate(sorted(fam.items())):
        torch.save(part, str(d / f"{name}_weights_{i}.bin"))
    formats["hook_files"] = d

    for name, path in formats.items():
        m2 = cls.from_pretrained(pretrained_model_name_or_path=str(tmp_path / "FLUX.1-schnell"), subfolder="transformer", revision=None, variant=None)
        m2 = m2.to(gpu, dtype=BF)
        m2.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(CONTROL))
        res = load_control_checkpoint(m2, str(path))
        assert not [k for k in res.missing_keys if k.startswith(CTRL_PREFIXES)], (name, res.missing_keys[:4])
        unexpected = set(res.unexpected_keys) - {"alias.weight"}          # the ZeRO writer's shared-parameter alias
        assert not unexpected, (name, sorted(unexpected)[:4])
        sd2 = m2.state_dict()
        assert all(torch.equal(sd2[k].cpu(), state[k]) for k in state), name
        out, loss, cnt = fwd(m2)
        assert torch.equal(out, want), f"{name}: the forward of the model loaded from disk differs from the in-memory model"
        assert loss == want_loss and torch.equal(cnt, want_cnt), name
        report(f"f2_{name}_vs_memory", out, want)
        del m2
