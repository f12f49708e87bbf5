#!/usr/bin/env python
"""Headline benchmark: images/sec at 1024^2, FLUX-schnell geometry + canny condition, 4 denoise steps (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--batch B]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
          or plain `python bench.py --gpus N`: with no torchrun environment the script launches its own N ranks (one fresh child
          process per GPU, started BEFORE anything touches the GPU; reference script/infer.sh:49-67 `accelerate launch`)

A "step" is one pass of the hot path over one batch: the full 4-step denoise loop (4 UniGenFlux forwards + 4 Euler steps) of a
batch of B = 4 synthetic 1024x1024 samples (cfg2: N = 4096 image tokens, T = 512 text tokens, 19 double + 38 single base blocks,
9 + 19 control blocks, CoMoE with 6 experts). Inputs and weights are resident in HBM before the timed region. VAE / CLIP / T5 are
excluded. Multi-GPU is batch-parallel: every rank holds a weight replica and its own samples; the only collectives
are the barriers around the timed region and one MAX all-reduce of the elapsed time (weak scaling).

The JSON line carries `roofline` (dominant kernel = the bf16 MFMA GEMM: algorithmic FLOPs per launch / HIP-event duration of
the launches inside the timed region, against the 2.5 PFLOP/s datasheet peak and against the MFMA-only rate measured on this chip) and
`cpu_baseline` (the oracle = CPU restatement of the reference, timed on this box's host cores on ONE full-depth forward of the same
workload at B = 1 with the GPU run's own weights; an image = 4 such forwards).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)
CONTROL_PARAMS = dict(use_rope=True, use_shared_expert=True, use_consis_module=False, use_single_trans_blocks=True, single_control_dev=2,
                      single_block_control_method="overall_add", top_num=1, expert_num_each_condition=3)


def canonical_flops_per_forward(D, N, T, n_double, n_single, n_cj, n_cs, n_cond, in_ch=64, txt_dim=4096):
    """Necessary-work FLOPs of one sample-forward: 2MNK per contraction, 4 Lq Lkv D per attention."""
    J = lambda a, b: 24 * D * D * (a + b) + 4 * (a + b) ** 2 * D
    Sg = lambda l: 24 * D * D * l + 4 * l * l * D
    Jk = lambda a, b, r: 24 * D * D * a + 4 * D * D * b * r + 4 * a * (a + b) * D
    base = n_double * J(N, T) + n_single * Sg(N + T)
    ctrl = n_double * Jk(N, T, n_cj / max(n_double, 1)) + n_single * Sg(N + T)
    zero = n_double * 2 * D * D * N + n_single * 2 * D * D * (N + T)
    comoe = n_cond * (4 * D * D * N + J(N, N) + Jk(2 * N, T, 1.0))
    embeds = 2 * D * in_ch * N * (2 + n_cond) + 2 * D * txt_dim * T + 2 * D * D * T
    return base + ctrl + zero + comoe + embeds


def fixture_parity(device):
    """SURVEY 8(d) `parity`: the HIP forward on the committed golden fixture (tests/golden/flux_tiny_single.safetensors: inputs and the
    oracle's bf16 / fp32 outputs) with the fixture's seeded weights. The oracle is only the checker here: it re-creates the weights."""
    import importlib, json as _json
    from safetensors import safe_open
    from oracle import unigen_ref as R
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "flux_tiny_single.safetensors")
    with safe_open(path, "pt") as f:
        meta = f.metadata()
        g = {k: f.get_tensor(k) for k in f.keys()}
    cfg_d, case = _json.loads(meta["config"]), _json.loads(meta["case"])
    rcfg = R.FluxConfig(condition_nums=case["n_cond"], **cfg_d)
    state = R.make_state(rcfg, seed=case["state_seed"], std=0.05, bias_std=0.02)
    model = importlib.import_module("src.UniGenTransformer").UniGenFlux.from_config(cfg_d, device=device, dtype=torch.bfloat16)
    model.init_condition_block(condition_nums=1, condition_types=["canny"], control_params=dict(
        use_rope=True, use_shared_expert=True, use_single_trans_blocks=True, single_control_dev=2, single_block_control_method="overall_add",
        top_num=1, expert_num_each_condition=3))
    model.load_state_dict({k: v.to(device) for k, v in state.items()}, strict=False)
    inp = {k[3:]: v.to(device) for k, v in g.items() if k.startswith("in.")}
    with torch.no_grad():
        out = model(timestep=g["timestep"].to(device), **inp)[0].float().cpu()
    rel = lambda a, b: float((a - b.float()).norm() / b.float().norm())
    return dict(fixture="tests/golden/flux_tiny_single.safetensors", rel_l2_vs_oracle_bf16=rel(out, g["out.bf16"]),
                max_abs_vs_oracle_bf16=float((out - g["out.bf16"].float()).abs().max()), rel_l2_vs_oracle_fp32=rel(out, g["out.fp32"]),
                oracle_bf16_vs_fp32=rel(g["out.bf16"].float(), g["out.fp32"]))


def _host_info(max_threads: int):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, max_threads))
    _host_info.available = dict(affinity=avail, cpu_count=os.cpu_count() or avail)      # what the host offers, beside what was used (`cores`)
    cpu_model, mem_gb = "", 0.0
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable"):
                    mem_gb = int(ln.split()[1]) / 1e6
                    break
    except OSError:
        pass
    return cores, cpu_model, mem_gb


def _cores_note(d: dict) -> dict:
    """`cores` = the torch threads the oracle was given (min(affinity, 16): a GPU box grants one GPU's share of the host); `cores_available` = the
    affinity mask of this process and `os.cpu_count()` of the host, so the choice is visible beside the number."""
    av = getattr(_host_info, "available", None)
    if av:
        d["cores_available"] = av["affinity"]
        d["host_cpu_count"] = av["cpu_count"]
    return d


def cpu_baseline_other(model, config: str, max_threads: int = 16):
    """cfg3 / cfg5: the CPU oracle on ONE full-depth forward of the stated workload at B = 1 image, on the GPU run's own weights.
    cfg3: MultiCondtionUniGenFlux (3 conditions, E = 12), 1024^2, an image = 4 forwards. cfg5: UniGenSD3 with CFG (an image-step = one
    2-sample forward), 1024^2, T = 333, an image = 28 such forwards."""
    from oracle import unigen_ref as R
    cores, cpu_model, mem_gb = _host_info(max_threads)
    torch.set_num_threads(cores)
    st = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
    if config == "cfg3":
        cfg = R.FluxConfig(condition_nums=3)
        inp = R.make_inputs(cfg, B=1, grid=64, T=512, n_cond=3)
        t = torch.full((1,), 1.0, dtype=torch.bfloat16)
        run = lambda: R.unigen_flux_forward(st, cfg, timestep=t, dtype=torch.bfloat16, **inp)[0]
        per_image, what = 4, "MultiCondtionUniGenFlux depth+canny+openpose (E=12), 1024^2 (N=4096, T=512), full depth, B=1"
    else:
        cfg = R.SD3Config()
        inp = R.make_sd3_inputs(cfg, B=2, hw=128, T=333)
        t = torch.full((2,), 500.0)
        run = lambda: R.unigen_sd3_forward(st, cfg, timestep=t, dtype=torch.bfloat16, **inp)[0]
        per_image, what = 28, "UniGenSD3 (SD3.5-medium, 24 blocks, D=1536), 1024^2 (N=4096, T=333), one CFG forward = 2 samples"
    t0 = time.perf_counter()
    with torch.no_grad():
        out = run()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    return _cores_note(dict(value=1.0 / (per_image * dt), unit="images/s", cores=cores, kind="port", cpu=cpu_model, mode="full",
                sample=f"oracle (bf16 torch CPU restatement of the reference) timed on ONE full-depth forward of {what}, the GPU run's own random-init "
                       f"weights: {dt:.1f} s on {cores} threads; an image = {per_image} such forwards -> images/s = 1 / ({per_image} x {dt:.1f} s)",
                sample_seconds=dt))


def cpu_baseline(model, mode: str = "auto", max_threads: int = 16):
    """The CPU oracle (port of the reference) timed on this box's host cores, rank 0, N = 1 only.

    full  : ONE full-depth 1024^2 forward (B = 1, N = 4096, T = 512, 19 + 38 base blocks, 9 + 19 control blocks, full CoMoE) in the
            reference's bf16 arithmetic, on the SAME weights the GPU run used (the model's state dict copied to the host: 37 GB); an image
            is 4 such forwards + 4 Euler steps, so images/s = 1 / (4 x that time) - a timing of the stated workload's step, not a
            FLOP-scaled extrapolation (SURVEY 8(d) "time one full step x4 and state that").
    slice : the round-1 estimate (depth cut to 6 + 8 base blocks, tiled weights, scaled by canonical FLOPs) for hosts without the RAM.
    The GPU box grants one GPU's share of the host (16 cores), so at most 16 threads are used whatever the affinity mask says."""
    from oracle import unigen_ref as R
    cores, cpu_model, mem_gb = _host_info(max_threads)
    torch.set_num_threads(cores)
    if mode == "auto":
        mode = "full" if mem_gb >= 90.0 else "slice"
    if mode == "cfg1":
        # BASELINE.json configs[0], the reference's own CPU-runnable case, END TO END on both sides: 512^2 (N = 1024, T = 512), B = 1, the 4-step
        # denoise loop (4 full-depth forwards + 4 Euler steps) of the oracle on the host and of the HIP path on the GPU, same weights and inputs.
        from unigen_amd.pipeline import denoise_loop
        cfg = R.FluxConfig()
        st = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
        inp = R.make_inputs(cfg, B=1, grid=32, T=512)
        unis = [torch.rand(1024, cfg.expert_nums, generator=torch.Generator().manual_seed(i)) for i in range(4)]
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = R.denoise(st, cfg, latents=inp["hidden_states"], num_steps=4, gate_uniforms=unis,
                            **{k: v for k, v in inp.items() if k not in ("hidden_states", "gate_uniform")})
        dt = time.perf_counter() - t0
        del st
        dev = next(model.parameters()).device
        g = {k: v.to(dev) for k, v in inp.items()}
        def gpu_run():
            return denoise_loop(model, latents=g["hidden_states"].clone(), control_tokens=g["condition_hidden_states"], prompt_embeds=g["encoder_hidden_states"],
                                pooled_prompt_embeds=g["pooled_projections"], condition_pooled_prompt_embeds=g["condition_pooled_projections"],
                                text_ids=g["txt_ids"], latent_image_ids=g["img_ids"], condition_ids=g["condition_ids"], num_inference_steps=4,
                                gate_uniforms=[u.to(dev) for u in unis])
        with torch.no_grad():
            out = gpu_run(); torch.cuda.synchronize()
            t1 = time.perf_counter(); out = gpu_run(); torch.cuda.synchronize(); dg = time.perf_counter() - t1
        rel = float((out.float().cpu() - ref.float()).norm() / ref.float().norm())
        fl = 4 * canonical_flops_per_forward(3072, 1024, 512, 19, 38, 9, 19, 1)
        return _cores_note(dict(value=1.0 / dt, unit="images/s", cores=cores, kind="port", cpu=cpu_model, mode="cfg1",
                    sample=f"cfg1 END TO END (512^2, B=1, 4 steps, full depth): oracle {dt:.1f} s per image on {cores} threads ({fl / dt / 1e12:.2f} TFLOP/s); "
                           f"the HIP path runs the same job in {dg * 1e3:.0f} ms; relL2(final latents, oracle bf16) = {rel:.3e}",
                    sample_seconds=dt, sample_tflops=fl / 1e12, cpu_tflops_per_s=fl / dt / 1e12, gpu_cfg1_images_per_s=1.0 / dg, cfg1_rel_l2_vs_oracle=rel))
    B, grid, T = 1, 64, 512
    full_flops_per_image = 4 * canonical_flops_per_forward(3072, 4096, 512, 19, 38, 9, 19, 1)
    if mode == "full":
        cfg = R.FluxConfig()
        t_copy = time.perf_counter()
        st = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
        t_copy = time.perf_counter() - t_copy
        n_d, n_s = cfg.num_layers, cfg.num_single_layers
    else:
        n_d, n_s = 6, 8
        cfg = R.FluxConfig(num_layers=n_d, num_single_layers=n_s)
        st, t_copy = R.make_state(cfg, seed=0, fast=True), 0.0
    inp = R.make_inputs(cfg, B=B, grid=grid, T=T)
    t = torch.full((B,), 1.0, dtype=torch.bfloat16)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = R.unigen_flux_forward(st, cfg, timestep=t, dtype=torch.bfloat16, **inp)[0]
    dt = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    del st
    sample_flops = canonical_flops_per_forward(cfg.inner_dim, grid * grid, T, n_d, n_s, cfg.cn_joint_layers, cfg.cn_single_layers, 1)
    if mode == "full":
        img_per_s = 1.0 / (4.0 * dt)
        sample = (f"oracle (bf16 torch CPU restatement of the reference) timed on ONE FULL-DEPTH forward of the cfg2 workload at B=1: 1024^2 (N=4096, T=512), "
                  f"19 double + 38 single base blocks, 9+19 control blocks, CoMoE E=6, the GPU run's own random-init weights (state dict copied to the host in "
                  f"{t_copy:.1f} s): {sample_flops / 1e12:.1f} TFLOP in {dt:.1f} s = {sample_flops / dt / 1e12:.2f} TFLOP/s on {cores} threads; "
                  f"an image = 4 such forwards (+4 Euler steps) -> images/s = 1 / (4 x {dt:.1f} s)")
    else:
        img_per_s = (sample_flops / dt) / full_flops_per_image
        sample = (f"ESTIMATE (host has {mem_gb:.0f} GB free, the full model needs ~90): oracle on ONE forward, B=1, 1024^2, FLUX width, depth cut to {n_d} double + "
                  f"{n_s} single base blocks, tiled weights: {sample_flops / 1e12:.2f} TFLOP in {dt:.1f} s on {cores} threads, scaled by algorithmic FLOPs to "
                  f"the full 4-step image ({full_flops_per_image / 1e12:.1f} TFLOP)")
    return _cores_note(dict(value=img_per_s, unit="images/s", cores=cores, kind="port", cpu=cpu_model, mode=mode, sample=sample, sample_seconds=dt,
                sample_tflops=sample_flops / 1e12, cpu_tflops_per_s=sample_flops / dt / 1e12))


# ----------------------------------------------------------------------------------------------------------------------
# workloads: BASELINE.json configs[1..4]. Each builder returns (model, one_step, info); one_step() = one pass of the hot path over the
# per-GPU batch (the whole denoise loop of B images), inputs and weights resident in HBM.
# ----------------------------------------------------------------------------------------------------------------------

def _flux_workload(config, B, rank, dev, small):
    from unigen_amd.flux import MultiCondtionUniGenFlux, UniGenFlux
    from unigen_amd.pipeline import denoise_loop, prepare_latent_image_ids
    from unigen_amd import dist_utils as DU
    K = 3 if config == "cfg3" else 1
    cfg = dict(num_layers=2, num_single_layers=4) if small else {}
    cls = MultiCondtionUniGenFlux if K > 1 else UniGenFlux
    model = cls.from_config(cfg, device=dev, dtype=torch.bfloat16)
    model.init_condition_block(condition_nums=K, condition_types=["depth", "canny", "openpose"][:K] if K > 1 else ["canny"],
                               control_params=dict(CONTROL_PARAMS))
    model.init_synthetic_(seed=0, std=0.02)
    grid, T, steps_per_image = 64, 512, 4
    N, E = grid * grid, model._ctl.expert_nums
    g = torch.Generator(device=dev).manual_seed(DU.rank_seed(12443, rank))   # reference default seed (infer.py:61) + rank
    rn = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32)
    latents0 = rn(B, N, 64).to(torch.bfloat16)
    conds = [rn(B, N, 64).to(torch.bfloat16) for _ in range(K)]
    prompt = (0.1 * rn(B, T, 4096)).to(torch.bfloat16)
    pooled = rn(B, 768).to(torch.bfloat16)
    cpools = [rn(B, 768).to(torch.bfloat16) for _ in range(K)]
    ids = prepare_latent_image_ids(grid, grid, dev, torch.bfloat16)
    txt_ids = torch.zeros(T, 3, device=dev, dtype=torch.bfloat16)
    unis = [[torch.rand(B * N, E, generator=g, device=dev) for _ in range(K)] for _ in range(steps_per_image)]
    if K == 1:
        conds, cpools, cids, unis = conds[0], cpools[0], ids, [u[0] for u in unis]
    else:
        cids = [ids] * K

    def one_step():
        return denoise_loop(model, latents=latents0.clone(), control_tokens=conds, prompt_embeds=prompt, pooled_prompt_embeds=pooled,
                            condition_pooled_prompt_embeds=cpools, text_ids=txt_ids, latent_image_ids=ids, condition_ids=cids,
                            num_inference_steps=steps_per_image, gate_uniforms=unis)

    n_d, n_s = model.config.num_layers, model.config.num_single_layers
    fl = steps_per_image * canonical_flops_per_forward(model.inner_dim, N, T, n_d, n_s, model._ctl.cn_joint_layers, model._ctl.cn_single_layers, K)
    geom = ("N=4096 image + T=512 text tokens, FLUX-schnell geometry 19 double + 38 single blocks D=3072 H=24, 9+19 control blocks, "
            f"CoMoE E={E}, 4 denoise steps, bf16, random-init weights" + (" [--small DEBUG depth]" if small else ""))
    return model, one_step, dict(flops_per_image=fl, flops_kind="canonical (SURVEY 8(d))", geom=geom, step="one 4-step denoise loop of the per-GPU batch",
                                 attn_kernel="flash_attn_kernel<128> (ug_flash_attn_fwd)")


def _sd3_workload(B, rank, dev, small):
    from unigen_amd.sd3 import UniGenSD3
    from unigen_amd.pipeline import sd3_denoise_loop
    from unigen_amd import dist_utils as DU
    model = UniGenSD3.from_config(dict(num_layers=4, dual_attention_layers=(0, 1)) if small else {}, device=dev, dtype=torch.bfloat16)
    model.init_condition_block(condition_nums=1, condition_types=["depth"], control_params=dict(use_shared_expert=True))   # config/unigen.yaml
    model.init_synthetic_(seed=0, std=0.02)
    steps_per_image, T = 28, 333
    g = torch.Generator(device=dev).manual_seed(DU.rank_seed(12443, rank))
    rn = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32)
    lat0, cond = rn(B, 16, 128, 128).to(torch.bfloat16), rn(B, 16, 128, 128).to(torch.bfloat16)
    enc = (0.1 * rn(2 * B, T, 4096)).to(torch.bfloat16)                       # [negative | positive] (src/UniGenPipeline.py:286-290)
    pooled, cpooled = rn(2 * B, 2048).to(torch.bfloat16), rn(2 * B, 2048).to(torch.bfloat16)
    unis = [torch.rand(2 * B * 4096, model._ctl.expert_nums, generator=g, device=dev) for _ in range(steps_per_image)]

    def one_step():
        return sd3_denoise_loop(model, latents=lat0.clone(), control_latents=cond, prompt_embeds=enc, pooled_prompt_embeds=pooled,
                                condition_pooled_prompt_embeds=cpooled, num_inference_steps=steps_per_image, guidance_scale=7.0, gate_uniforms=unis)

    geom = ("SD3.5-medium geometry 24 joint blocks (13 dual-attention) D=1536 H=24 dh=64 + 24 control blocks, transformer-block experts E=6, "
            "N=4096 image + T=333 text tokens, classifier-free guidance (2 samples per image-step), 28 denoise steps, bf16, random-init weights"
            + (" [--small DEBUG depth]" if small else ""))
    return model, one_step, dict(flops_per_image=None, flops_kind="executed by the GEMM + attention launches (no canonical count for SD3 in SURVEY 8(d))",
                                 geom=geom, step="one 28-step CFG denoise loop of the per-GPU batch", attn_kernel="flash_attn_kernel<64> (ug_flash_attn_fwd)")


def _self_launch(args, argv):
    """`python bench.py --gpus N` with no torchrun environment: start N ranks ourselves. This parent has not touched the GPU
    (torch.cuda.device_count() does not initialise it) and never will; every rank is a FRESH child process, no exec of a GPU process."""
    import socket
    import subprocess
    n = args.gpus
    ndev = torch.cuda.device_count()
    print(f"[bench] self-launch: {n} ranks; torch.cuda.device_count() = {ndev}", file=sys.stderr, flush=True)
    sharing = os.environ.get("UG_DIST_BACKEND", "") == "gloo"          # rehearsal: ranks may share a GPU (no RCCL communicator)
    if not args.dry_run and ndev < n and not sharing:
        raise SystemExit(f"bench.py --gpus {n}: this node shows only {ndev} GPU(s) (torch.cuda.device_count()); nothing was measured")
    # Build the HIP library ONCE, here, before any rank exists: on a clean clone every rank's lib.load() would otherwise start the same hipcc
    # build at the same time (it is serialised by a file lock in unigen_amd.build, but N - 1 ranks would sit in it). hipcc only: this parent
    # never touches the GPU, the ranks stay fresh processes.
    from unigen_amd import build as _build
    t_b = time.perf_counter()
    lib_path = _build.build()
    print(f"[bench] self-launch: {lib_path} ready ({time.perf_counter() - t_b:.1f} s in unigen_amd.build.build())", file=sys.stderr, flush=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is not None:
                    pending.remove(p)
                    if c != 0 and rc == 0:
                        rc = c
                        for q in pending:            # one rank failed: the others would wait in a barrier forever
                            q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def _dry_run(args, world, rank):
    """Harness rehearsal without a GPU (tests/test_host_cpu.py): rendezvous, barriers, max over ranks and the JSON line, with a sleep as the step."""
    from unigen_amd import dist_utils as DU
    dev = torch.device("cpu")
    rank, world = DU.init_distributed(dev)
    for _ in range(args.warmup):
        time.sleep(0.01)
    DU.barrier(dev, world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    own = time.perf_counter() - t0                 # this rank's own loop, before the closing barrier: what the per-rank record carries
    DU.barrier(dev, world)
    elapsed = DU.max_over_ranks(time.perf_counter() - t0, dev, world)
    per_rank = [dict(rank=i, images=int(r[0]), seconds=r[1], device_index=int(r[2])) for i, r in enumerate(DU.all_gather_floats([args.steps, own, rank], dev, world))]
    if rank == 0:
        print(json.dumps(dict(metric="DRY RUN (no GPU work): harness rehearsal only", dry_run=True, value=None, unit="images/s", n_gpus=world,
                              steps=args.steps, warmup=args.warmup, ms_per_step=1000.0 * elapsed / args.steps, higher_is_better=True, scaling="weak",
                              vs_baseline=None, data="none", config=dict(workload="none"), per_rank=per_rank, dist=DU.describe(world))), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


class _PowerSampler:
    """Package power and shader clock of THIS rank's GPU during the timed region, read from the amdgpu hwmon files of the device with torch's PCI
    address (power1_input in microwatts, freq1_input in Hz; plain file reads on a side thread every 0.2 s - no subprocess, no HIP call). The
    forward runs at the package power cap (profiles/r04s_power.log); the line carries the evidence for its own run. None where sysfs is not readable."""

    def __init__(self, dev):
        self.dir, self.samples, self._stop, self._thread = None, [], None, None
        try:
            if dev.type != "cuda":
                return
            pr = torch.cuda.get_device_properties(dev)
            want = (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
            base = "/sys/class/drm"
            for card in sorted(os.listdir(base)):
                if not card.startswith("card") or "-" in card:
                    continue
                real = os.path.realpath(os.path.join(base, card, "device"))
                parts = os.path.basename(real).replace(".", ":").split(":")          # 0000:dc:00.0
                if len(parts) != 4 or (int(parts[0], 16), int(parts[1], 16), int(parts[2], 16)) != want:
                    continue
                hw = os.path.join(real, "hwmon")
                for h in sorted(os.listdir(hw)):
                    if os.path.exists(os.path.join(hw, h, "power1_input")):
                        self.dir = os.path.join(hw, h)
                        return
        except Exception:
            self.dir = None

    @staticmethod
    def _read(path):
        with open(path) as f:
            return float(f.read().strip())

    def start(self):
        if self.dir is None:
            return
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append((self._read(os.path.join(self.dir, "power1_input")) / 1e6, self._read(os.path.join(self.dir, "freq1_input")) / 1e6))
                except Exception:
                    return
                self._stop.wait(0.2)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join(timeout=2.0)
        if len(self.samples) < 2:
            return None
        w, f = sorted(x[0] for x in self.samples), sorted(x[1] for x in self.samples)
        cap = None
        try:
            cap = self._read(os.path.join(self.dir, "power1_cap")) / 1e6
        except Exception:
            pass
        return dict(watts_median=w[len(w) // 2], watts_min=w[0], watts_max=w[-1], cap_watts=cap, sclk_mhz_median=f[len(f) // 2], samples=len(w),
                    source="amdgpu hwmon power1_input / freq1_input of this rank's device, every 0.2 s inside the timed region")


# 3 + 2 of the 4 TCC counter slots: FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md, counter table); the matrix-pipe pass is the third
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))


def _under_profiler():
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def same_run_traffic(limit_s=60.0):
    """`roofline.traffic` measured by THIS run on THIS box: one child per counter pass - `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py
    --pmc-child` (the cfg2 B = 4 workload, a warm-up step + one 4-forward step of which only the second is counted, no timers / probes / CPU legs; the interpreter's ELF image goes straight after `--`) - after every timed
    region of the parent is over. Bytes beyond the XCD L2 per GEMM launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches over all gemm256 / gemm128
    launches (KiB units; gfx950's FETCH_SIZE counts wide coalesced reads at half their bytes; Infinity-Cache hits are included: an upper bound on HBM
    bytes). A pass takes 8-10 s; one that fails or overruns `limit_s` raises (the remaining passes are not started) and the caller falls back to the
    recorded profile and says so - the worst case adds `limit_s` to the run, never more."""
    import csv, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    # the program after `--` must be the interpreter itself (an ELF image), never a shim script or launcher: under --pmc the profiler's preloaded
    # library has initialised the GPU before the program starts, and any further exec from there takes the machine down
    py = os.path.realpath(sys.executable)
    with open(py, "rb") as f:
        if f.read(4) != b"\x7fELF":
            raise RuntimeError(f"{py} is not an ELF interpreter: counter passes not started")
    td = tempfile.mkdtemp(prefix="ug_bench_pmc_", dir="/tmp")
    rows, secs = {}, {}         # (kernel class, counter) -> [(start ns, value, duration ns)] in dispatch order
    try:
        for counters in PMC_PASSES:
            t0 = time.perf_counter()
            out = os.path.join(td, counters[0])
            cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "--", py, os.path.join(ROOT, "bench.py"), "--pmc-child"]
            p = subprocess.Popen(cmd, cwd=td, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                log, _ = p.communicate(timeout=limit_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)            # exactly the process group started above
                p.communicate()
                raise RuntimeError(f"PMC pass {counters} overran {limit_s:.0f} s")
            if p.returncode != 0:
                raise RuntimeError(f"PMC pass {counters} exited {p.returncode}: {log[-300:]}")
            files = [os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if not files:
                raise RuntimeError(f"PMC pass {counters}: no counter_collection.csv")
            for path in files:
                with open(path, newline="") as f:
                    for r in csv.DictReader(f):
                        name = r["Kernel_Name"]
                        classes = (("gemm_all", "gemm256") if "gemm256_kernel" in name else ("gemm_all",) if "gemm128_kernel" in name
                                   else ("attn",) if "flash_attn" in name else ())
                        for cls in classes:
                            rows.setdefault((cls, r["Counter_Name"]), []).append((float(r["Start_Timestamp"]), float(r["Counter_Value"]),
                                                                                  float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
            secs[counters[0]] = round(time.perf_counter() - t0, 1)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    # the child runs the step twice; only the SECOND step's launches count (the first packs weights, allocates workspaces and fills the caches:
    # ADVICE r5) - both steps launch the same kernels in the same order, so the later half by start time is the warm step
    acc = {}                    # (kernel class, counter) -> [sum, launches, ns]
    for key, lst in rows.items():
        lst.sort()
        if len(lst) % 2:
            raise RuntimeError(f"PMC pass saw an odd number of {key[0]} launches over two identical steps: {len(lst)}")
        warm = lst[len(lst) // 2:]
        acc[key] = [sum(v for _, v, _ in warm), len(warm), sum(d for _, _, d in warm)]
    def traffic_of(cls):
        f, w = acc.get((cls, "FETCH_SIZE")), acc.get((cls, "WRITE_SIZE"))
        if not f or not w or f[1] != w[1]:
            raise RuntimeError(f"PMC passes saw different {cls} launch counts: {f and f[1]} / {w and w[1]}")
        fetch, write = 2.0 * f[0] * 1024.0 / f[1], w[0] * 1024.0 / w[1]
        return dict(traffic=fetch + write, fetch_bytes_per_launch_corrected=fetch, write_bytes_per_launch=write, launches=f[1])

    def busy_of(cls):
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (tools/pmc_summary.py: the same arithmetic)
        b, g = acc.get((cls, "SQ_VALU_MFMA_BUSY_CYCLES")), acc.get((cls, "GRBM_GUI_ACTIVE"))
        if not b or not g or not g[0]:
            return None
        cyc = g[0] / 8.0
        return dict(mfma_busy=b[0] / (cyc * 1024.0), effective_clock_ghz=cyc / g[2], launches=g[1], avg_launch_us_profiled=g[2] / g[1] / 1e3)

    res = dict(traffic_of("gemm_all"), pass_seconds=secs)
    if busy_of("gemm256"):
        res["gemm256"] = busy_of("gemm256")
    if ("attn", "FETCH_SIZE") in acc:
        res["attn"] = dict(traffic_of("attn"), **(busy_of("attn") or {}))
    return res


def _roofline_blocks(s, elapsed, pk16, pk32, attn_kernel, traffic_file=None, live=None):
    """`roofline` (the bf16 MFMA GEMM, the dominant kernel) and `roofline_attention` from a KernelTimer summary: algorithmic FLOPs of the launches
    inside the timed region / their HIP-event durations, against the 2.5 PFLOP/s datasheet peak and the bare-MFMA rate measured in this run."""
    out = {}
    gm, at = s.get("gemm"), s.get("attn")
    ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
    traffic, traffic_note, traffic_src, pmc = None, None, None, {}
    if traffic_file is not None:
        with open(traffic_file) as f:
            tj = json.load(f)
        base = os.path.basename(traffic_file)
        traffic = tj["kernels"]["gemm_all"]["hbm_bytes_per_launch"]
        traffic_src = dict(measured_in_this_run=False, file=f"profiles/{base}",
                           collected_utc=tj.get("collected_utc", "not recorded (profile of an earlier round)"), collected=tj.get("source"))
        traffic_note = (f"NOT measured in this run - counters need their own rocprofv3 passes - read from an earlier profile of the same command on another box: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/{base}): (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                        "launch; the L2-fabric counters include Infinity-Cache hits, so this is traffic beyond the XCD L2, an upper bound on HBM bytes")
        g256 = tj["kernels"].get("gemm256", {})
        if "mfma_busy_frac" in g256:
            pmc = dict(measured_in_this_run=False, file=f"profiles/{base}", mfma_busy=g256["mfma_busy_frac"], effective_clock_ghz=g256["effective_clock_ghz"],
                       hbm_side_gbps=g256["hbm_bytes_per_launch"] / (g256["avg_launch_us_profiled"] * 1e-6) / 1e9)
    if live is not None and "traffic" in live:
        traffic = live["traffic"]
        traffic_src = dict(measured_in_this_run=True, launches=live["launches"], fetch_bytes_per_launch_corrected=live["fetch_bytes_per_launch_corrected"],
                           write_bytes_per_launch=live["write_bytes_per_launch"], pass_seconds=live["pass_seconds"],
                           collected="child processes of this run, after its timed regions: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --pmc-child")
        traffic_note = ("measured in this run on this box: (2*FETCH_SIZE + WRITE_SIZE)*1024 per GEMM launch over one WARM 4-forward step of the same workload under rocprofv3 (the child's first step - weight packing, workspace allocation, cold caches - is excluded) "
                        "(gfx950 FETCH_SIZE correction; KiB units); the L2-fabric counters include Infinity-Cache hits, so this is traffic beyond the XCD L2, an upper bound on HBM bytes")
        if "gemm256" in live:
            g = live["gemm256"]
            pmc = dict(measured_in_this_run=True, mfma_busy=g["mfma_busy"], effective_clock_ghz=g["effective_clock_ghz"], launches=g["launches"],
                       avg_launch_us_profiled=g["avg_launch_us_profiled"], collected="third child pass of this run: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
    elif live is not None and traffic_src is not None:
        traffic_src["same_run_attempt"] = live.get("error")
    # second denominator: the measured MFMA-only rate of this chip, at the clock it holds under matrix load (pk16 / pk32)
    meas = (lambda a, pk: a / pk) if pk16 else (lambda a, pk: None)
    out["roofline"] = dict(bound="mfma", kernel="gemm256_kernel / gemm128_kernel (ug_gemm_bf16)", achieved=ach, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                           frac=ach / MFMA_BF16_PEAK_TFLOPS, peak_measured=pk16 or None, frac_of_measured=meas(ach, pk16),
                           peak_measured_note=("ug_probe_mfma_bf16: bare v_mfma_f32_16x16x32_bf16 loop (the GEMM's shape), register operands with random values, "
                                               f"one wave per SIMD on every CU, HIP events; the 32x32x16 shape (attention) measures {pk32:.0f}"),
                           traffic=traffic, traffic_note=traffic_note, traffic_source=traffic_src, pmc_gemm256=pmc or None, launches=gm["launches"],
                           avg_launch_us=1000.0 * gm["ms"] / gm["launches"], avg_launch_gflop=gm["flops"] / gm["launches"] / 1e9,
                           share_of_step_time=gm["ms"] * 1e-3 / elapsed)
    if at:
        a2 = at["flops"] / (at["ms"] * 1e-3) / 1e12
        out["roofline_attention"] = dict(bound="mfma", kernel=attn_kernel, achieved=a2, peak=MFMA_BF16_PEAK_TFLOPS,
                                         unit="TFLOP/s", frac=a2 / MFMA_BF16_PEAK_TFLOPS, peak_measured=pk32 or None, frac_of_measured=meas(a2, pk32),
                                         launches=at["launches"],
                                         avg_launch_us=1000.0 * at["ms"] / at["launches"], share_of_step_time=at["ms"] * 1e-3 / elapsed)
        if live is not None and "attn" in live:           # the same three counter passes of this run, attention launches
            la = live["attn"]
            out["roofline_attention"].update(traffic=la["traffic"], pmc_attn=dict(measured_in_this_run=True, **{k: v for k, v in la.items() if k != "traffic"}))
    return out


WORKLOAD_NAMES = dict(
    cfg2=lambda B, world: f"cfg2: UniGenFlux canny single-condition, 1024x1024, batch={B}, ",
    cfg4=lambda B, world: f"cfg4: UniGenFlux canny, 1024x1024, global batch {B * world} sharded over {world} x MI355X (B={B} per GPU; 64 = 8 x 8 at N=8), ",
    cfg3=lambda B, world: f"cfg3: MultiCondtionUniGenFlux depth+canny+openpose multi-condition (per-condition CoMoE, summed), 1024x1024, batch={B}" + (f" per GPU x {world}" if world > 1 else "") + ", ",
    cfg5=lambda B, world: f"cfg5: UniGenSD3 (SD3.5-medium backbone) depth single-condition, 1024x1024, batch={B}" + (f" per GPU x {world}" if world > 1 else "") + ", ")
METRIC_NAMES = dict(cfg2="images/sec at 1024^2, FLUX-schnell+canny, 4-step", cfg4="images/sec at 1024^2, FLUX-schnell+canny, 4-step",
                    cfg3="images/sec at 1024^2, FLUX-schnell + depth+canny+openpose, 4-step", cfg5="images/sec at 1024^2, SD3.5-medium + depth, 28-step CFG")


def _pass_stats(B, per_pass):
    """min / median / max of the passes of a side block, as images/s and ms (no single samples)."""
    ps = sorted(per_pass)
    med = ps[len(ps) // 2] if len(ps) % 2 else 0.5 * (ps[len(ps) // 2 - 1] + ps[len(ps) // 2])
    return dict(ms_per_pass=[1000.0 * t for t in per_pass], images_per_s_best=B / ps[0], images_per_s_median=B / med, images_per_s_worst=B / ps[-1])


def measure_other_config(config, rank, dev, ops, pk16, pk32, steps=3, warmup=1):
    """BASELINE.json configs[2] (cfg3) / configs[4] (cfg5) at their stated batch (B = 8) inside the default N = 1 run, so the driver's own record carries
    them: the same fields as the headline block (value, ms_per_step, roofline, roofline_attention), `warmup` + `steps` passes of the hot path;
    `value` is over all timed passes, `passes` holds every pass's own time with min / median / max."""
    B = 8
    if config == "cfg5":
        model, one_step, info = _sd3_workload(B, rank, dev, False)
    else:
        model, one_step, info = _flux_workload(config, B, rank, dev, False)
    timer = ops.KernelTimer()
    per_pass = []
    elapsed, _, _ = _timed(one_step, steps, warmup, dev, 1, timer, ops, per_pass=per_pass)
    s = timer.summary()
    value = B * steps / elapsed
    fl_img = info["flops_per_image"]
    if fl_img is None:
        fl_img = sum(d["flops"] for d in s.values()) / (B * steps)
    d = dict(metric=METRIC_NAMES[config], value=value, unit="images/s", n_gpus=1, steps=steps, warmup=warmup, ms_per_step=1000.0 * elapsed / steps,
             dtype="bf16", data="synthetic", config=dict(workload=WORKLOAD_NAMES[config](B, 1) + info["geom"], baseline_config=config, per_gpu_batch=B, step=info["step"]),
             flops_per_image=fl_img, flops_per_image_kind=info["flops_kind"], e2e_mfma_frac=value * fl_img / (MFMA_BF16_PEAK_TFLOPS * 1e12))
    if pk16:
        d["e2e_frac_of_measured"] = value * fl_img / (pk16 * 1e12)
    d["passes"] = _pass_stats(B, per_pass)
    d.update(_roofline_blocks(s, elapsed, pk16, pk32, info["attn_kernel"]))
    del model, one_step
    return d


def measure_small_m(model, name, rank, dev, ops, pk16, pk32, steps=2, warmup=1):
    """The small-M regime the reference's own launch script runs (script/infer.sh:62-63: --batch_size 1; BASELINE configs[0] is 512^2, B = 1), on
    the headline run's own UniGenFlux (same weights), inside the default line: `b1_1024` = one 1024^2 image (N = 4096,
    M = 4608 joint rows), `cfg1_gpu` = configs[0]'s geometry on the GPU (512^2: N = 1024, M = 1536). 1 warm-up + 2 passes of the 4-step loop."""
    grid = dict(b1_1024=64, cfg1_gpu=32)[name]
    _, one_step, _ = _flux_workload_inputs_only(model, 1, rank, dev, grid=grid)
    per_pass = []
    elapsed, _, _ = _timed(one_step, steps, warmup, dev, 1, None, ops, per_pass=per_pass)      # no kernel timer here: its two events per launch are host work too
    value = steps / elapsed
    timer = ops.KernelTimer()
    t_elapsed, _, _ = _timed(one_step, 1, 0, dev, 1, timer, ops)                                 # one more pass under the timer for the kernels' own rates
    s = timer.summary()
    n_d, n_s = model.config.num_layers, model.config.num_single_layers
    fl_img = 4 * canonical_flops_per_forward(model.inner_dim, grid * grid, 512, n_d, n_s, model._ctl.cn_joint_layers, model._ctl.cn_single_layers, 1)
    what = {"b1_1024": "UniGenFlux canny, 1024x1024, batch=1 (the reference's script/infer.sh launch shape), N=4096 + T=512, 4 steps",
            "cfg1_gpu": "cfg1 geometry on the GPU: UniGenFlux canny, 512x512, batch=1, N=1024 + T=512, 4 steps (BASELINE configs[0] names the CPU path; this is the same job on the MI355X)"}[name]
    d = dict(metric="images/sec, FLUX-schnell+canny, 4-step, batch 1", value=value, unit="images/s", n_gpus=1, steps=steps, warmup=warmup,
             ms_per_step=1000.0 * elapsed / steps, dtype="bf16", data="synthetic", config=dict(workload=what, per_gpu_batch=1, step="one 4-step denoise loop of one image"),
             flops_per_image=fl_img, flops_per_image_kind="canonical (SURVEY 8(d))", e2e_mfma_frac=value * fl_img / (MFMA_BF16_PEAK_TFLOPS * 1e12),
             passes=_pass_stats(1, per_pass))
    if pk16:
        d["e2e_frac_of_measured"] = value * fl_img / (pk16 * 1e12)
    d.update(_roofline_blocks(s, t_elapsed, pk16, pk32, "flash_attn_kernel<128> (ug_flash_attn_fwd)"))
    d["gemm_and_attention_launches_per_image"] = int(sum(v["launches"] for v in s.values()))
    d["gemm_and_attention_kernel_ms_per_image"] = sum(v["ms"] for v in s.values())
    # The same loop captured once in a HIP graph and replayed (`--graph`): at batch 1 a forward is ~1.5 k launches of tens of
    # microseconds each, so the host's launch rate (Python + ctypes per call), not the kernels, can pace the plain loop; the replay has no host in it.
    try:
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_out = one_step()
        gp = []
        _timed(lambda: (graph.replay(), g_out)[1], steps, 1, dev, 1, None, ops, per_pass=gp)
        best = min(gp)
        d["hip_graph"] = dict(value=1.0 / best, unit="images/s", ms_per_step=1000.0 * best, e2e_mfma_frac=fl_img / best / (MFMA_BF16_PEAK_TFLOPS * 1e12),
                              passes=_pass_stats(1, gp), note="one 4-step loop captured in a HIP graph, replayed; best of the passes")
        del graph, g_out
    except Exception as e:                                   # the plain numbers above stand on their own
        d["hip_graph"] = dict(error=f"{type(e).__name__}: {e}")
    return d


def _timed(one_step, steps, warmup, dev, world, timer, ops, power=None, per_pass=None):
    """`per_pass`: a list that receives every pass's own wall time (a device synchronisation after each pass: only for the side blocks, whose
    passes last seconds or are reported per pass - the headline region is timed as ONE bracket, as the contract says)."""
    from unigen_amd import dist_utils as DU
    for _ in range(warmup):
        out = one_step()
    DU.barrier(dev, world)
    ops.set_timer(timer)
    if power is not None:
        power.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        tp = time.perf_counter()
        out = one_step()
        if per_pass is not None:
            torch.cuda.synchronize(dev)
            per_pass.append(time.perf_counter() - tp)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    own = time.perf_counter() - t0                 # this rank's own K steps; the job's time is the MAX over ranks of the barrier-bracketed region
    DU.barrier(dev, world)
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    if not torch.isfinite(out.float()).all():
        raise SystemExit("non-finite latents after denoising")
    return DU.max_over_ranks(elapsed, dev, world), out, own


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default=None,
                    help="BASELINE.json configs: cfg2 = UniGenFlux canny B=4 (the metric's configuration, default at N = 1); cfg4 = the same model at B=8 per "
                         "GPU (global 64 on 8 GPUs; default at N > 1); cfg3 = MultiCondtionUniGenFlux depth+canny+openpose B=8; cfg5 = UniGenSD3 depth B=8 (CFG, 28 steps)")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU per step; default 4 for cfg2, 8 for cfg3 / cfg4 / cfg5 (reference infer.py:173 shards samples by rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["auto", "full", "slice", "cfg1"], default="auto",
                    help="full = one full-depth 1024^2 oracle forward x 4 (needs ~90 GB host RAM); cfg1 = BASELINE configs[0] end to end (512^2, B = 1, 4 steps) on CPU and GPU")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-scaling-base", action="store_true", help="N = 1, cfg2 only: skip the extra B = 8 measurement (cfg4's per-GPU shape, the like-for-like base of the 1 -> 8 curve)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1, cfg2 only: skip the cfg3 / cfg5 blocks (other_configs: 1 warm-up + 3 passes each at B = 8) and the batch-1 blocks (small_m)")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1, cfg2 only: skip the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (then read from profiles/)")
    ap.add_argument("--pmc-child", action="store_true", help="internal: the workload of one counter pass (cfg2, B = 4, one step, nothing else)")
    ap.add_argument("--small", action="store_true", help="debug: reduced depth (NOT the headline configuration)")
    ap.add_argument("--graph", action="store_true", help="capture one step (the whole denoise loop) in a HIP graph and replay it (SURVEY 8(f) rank 1)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: rehearse the multi-process harness (spawn, rendezvous, barriers, JSON) on the CPU with gloo")
    args = ap.parse_args()
    if args.pmc_child:                 # one counter pass of same_run_traffic(): the cfg2 workload's launches and nothing else
        args.gpus, args.config, args.batch, args.steps, args.warmup = 1, "cfg2", 4, 1, 1      # warm-up + one step: the parser keeps the second step's launches
        args.no_kernel_timer = args.no_cpu_baseline = args.no_scaling_base = args.no_other_configs = args.no_pmc = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args, sys.argv[1:])                                  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch with --nproc-per-node {args.gpus}, or without a launcher")
    if args.dry_run:
        return _dry_run(args, world, rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts: RCCL needs it (already exported on the GPU boxes)
    if world > 1:                                                      # N ranks on one host: do not let every rank's torch CPU pool take all cores
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: unigen_amd has no CPU path")
    ndev = torch.cuda.device_count()
    if rank == 0:
        print(f"[bench] world = {world}; torch.cuda.device_count() = {ndev}", file=sys.stderr, flush=True)
    if ndev < world and os.environ.get("UG_DIST_BACKEND", "") != "gloo":
        raise SystemExit(f"bench.py --gpus {world}: this node shows only {ndev} GPU(s); nothing was measured")
    if os.environ.get("UG_DIST_BACKEND", "") == "gloo":
        local_rank = local_rank % max(ndev, 1)                           # rehearsal only: more ranks than GPUs share devices
    if not (0 <= local_rank < ndev):
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} has no device (torch.cuda.device_count() = {ndev})")
    print(f"[bench] rank {rank} of {world} -> cuda:{local_rank} ({torch.cuda.get_device_name(local_rank)})", file=sys.stderr, flush=True)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from unigen_amd import dist_utils as DU
    rank, world = DU.init_distributed(dev)
    from unigen_amd import ops

    config = args.config or ("cfg2" if world == 1 else "cfg4")
    if args.batch <= 0:
        args.batch = 4 if config == "cfg2" else 8
    B = args.batch
    if config == "cfg5":
        model, one_step, info = _sd3_workload(B, rank, dev, args.small)
    else:
        model, one_step, info = _flux_workload(config, B, rank, dev, args.small)

    if args.graph:
        # every workspace exists after the warm-up; capture on a side stream as torch requires, then replay on it
        if args.warmup < 1:
            raise SystemExit("--graph needs at least one warm-up step (lazy workspace allocation)")
        for _ in range(args.warmup):
            one_step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = one_step()

        def one_step():                     # noqa: F811
            graph.replay()
            return graph_out
    timer = None if (args.no_kernel_timer or args.graph) else ops.KernelTimer()
    sampler = _PowerSampler(dev) if rank == 0 else None
    elapsed, out, own_elapsed = _timed(one_step, args.steps, args.warmup, dev, world, timer, ops, power=sampler)
    power = sampler.stop() if sampler is not None else None
    # one all_gather of every rank's own record, so that a straggler (or a rank on the wrong device) shows in the one JSON line.
    # The measured MFMA-only rate of each rank's chip rides along: boxes differ by 8-12 % (DVFS), and cross-box comparisons need it.
    probe = not (args.no_kernel_timer or args.graph)          # profiling runs (--no-kernel-timer) keep the probe's launches out of the kernel statistics
    pk16 = ops.probe_mfma_peak(dev, shape=1) if probe else 0.0
    pk32 = ops.probe_mfma_peak(dev, shape=0) if probe else 0.0
    per_rank = [dict(rank=i, images=int(r[0]), seconds=r[1], images_per_s=r[0] / r[1], device_index=int(r[2]), mfma_probe_16x16x32_tflops=r[3],
                     mfma_probe_32x32x16_tflops=r[4])
                for i, r in enumerate(DU.all_gather_floats([B * args.steps, own_elapsed, local_rank, pk16, pk32], dev, world))]

    scaling_base = None
    if world == 1 and config == "cfg2" and B == 4 and not args.no_scaling_base and not args.small and not args.graph:
        # cfg4's per-GPU share (B = 8) on ONE GPU: the like-for-like base of the 1 -> 8 scaling curve (N > 1 lines run B = 8 per GPU);
        # measured after the timed region, same model
        del one_step
        _, step8, _ = _flux_workload_inputs_only(model, 8, rank, dev)
        e8, _, _ = _timed(step8, 2, 1, dev, 1, None, ops)
        scaling_base = dict(workload="cfg4's per-GPU shape on one GPU: B=8, same model", per_gpu_batch=8, steps=2, warmup=1, value=8 * 2 / e8, unit="images/s",
                            ms_per_step=1000.0 * e8 / 2, note="divide the N > 1 lines (B = 8 per GPU) by N x THIS value for a like-for-like efficiency")

    if rank == 0:
        images = B * args.steps * world
        value = images / elapsed
        names = {k: f(B, world) for k, f in WORKLOAD_NAMES.items()}
        metric = METRIC_NAMES[config]
        line = dict(metric=metric, value=value, unit="images/s", n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=1000.0 * elapsed / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype="bf16", data="synthetic",
                    config=dict(workload=names[config] + info["geom"], baseline_config=config, per_gpu_batch=B, global_batch=B * world,
                                parallelism=f"dp{world} (independent samples, RCCL barrier only)", step=info["step"]),
                    hip_graph=bool(args.graph), device_count=ndev, per_rank=per_rank,
                    dist=DU.describe(world))          # the communicator the barriers / reductions of this run went through (torch.distributed, rank 0)
        s = timer.summary() if timer is not None else {}
        fl_img = info["flops_per_image"]
        if fl_img is None and s:
            fl_img = sum(d["flops"] for d in s.values()) / (B * args.steps)
        if fl_img is not None:
            line["flops_per_image"] = fl_img
            line["flops_per_image_kind"] = info["flops_kind"]
            if config in ("cfg2", "cfg4"):
                line["flops_per_image_canonical"] = fl_img
            line["e2e_mfma_frac"] = value / world * fl_img / (MFMA_BF16_PEAK_TFLOPS * 1e12)
            # the same against the bare-MFMA rate this chip held in THIS run (16x16x32 probe, rank 0's device): the number to compare across boxes
            if probe:
                line["e2e_frac_of_measured"] = value / world * fl_img / (pk16 * 1e12)
                line["mfma_probe_tflops"] = dict(shape_16x16x32=pk16, shape_32x32x16=pk32, note="ug_probe_mfma_bf16, same run, rank 0's device")
        if scaling_base is not None:
            line["scaling_base"] = scaling_base
        if power is not None:
            line["power"] = power             # rank 0's GPU inside the timed region: the forward sits at the package power cap
        traffic_file = None
        if s:
            cands = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc.json"))      # newest round's PMC summary
            tf = os.path.join(ROOT, "profiles", cands[-1] if cands else "r01_hbm_traffic.json")
            if os.path.exists(tf) and not args.small and config == "cfg2" and B == 4:      # PMC passes of this same command (see the file's `source`, tools/pmc_summary.py)
                traffic_file = tf
            line.update(_roofline_blocks(s, elapsed, pk16, pk32, info["attn_kernel"], traffic_file))
        if world == 1 and not args.no_cpu_baseline:
            if config in ("cfg3", "cfg5"):
                line["cpu_baseline"] = cpu_baseline_other(model, config)
            else:
                line["cpu_baseline"] = cpu_baseline(model, args.cpu_baseline)
                if line["cpu_baseline"]["mode"] == "slice":           # never silently: the estimate is named in the workload too
                    line["config"]["workload"] += " [cpu_baseline: FLOP-scaled SLICE estimate, host RAM below 90 GB]"
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            if config != "cfg5":
                line["parity"] = fixture_parity(dev)
        if world == 1 and config == "cfg2" and B == 4 and not (args.small or args.graph or args.no_other_configs or args.no_kernel_timer):
            # the batch-1 regime on the same model (same weights, nothing rebuilt): a few seconds
            small = {}
            for nm in ("b1_1024", "cfg1_gpu"):
                try:
                    small[nm] = measure_small_m(model, nm, rank, dev, ops, pk16, pk32)
                except Exception as e:
                    small[nm] = dict(error=f"{type(e).__name__}: {e}")
            line["small_m"] = small
        if world == 1 and config == "cfg2" and B == 4 and not (args.small or args.graph or args.no_other_configs or args.no_kernel_timer):
            # BASELINE configs[2] and [4] inside the driver's own record: one warm-up + one pass each at their stated B = 8,
            # after everything of the headline block is final - the cfg2 numbers above are untouched by this
            import gc
            model = one_step = step8 = out = None       # every reference to the cfg2 model and its workspaces (37 GB of weights) goes first
            others = {}
            for oc in ("cfg3", "cfg5"):
                gc.collect(); torch.cuda.empty_cache()
                t_o = time.perf_counter()
                try:
                    others[oc] = measure_other_config(oc, rank, dev, ops, pk16, pk32)
                    others[oc]["wall_s_including_model_init"] = time.perf_counter() - t_o
                except Exception as e:                      # the headline line must survive: the failure is recorded, never hidden
                    others[oc] = dict(error=f"{type(e).__name__}: {e}")
            line["other_configs"] = others
        if world == 1 and config == "cfg2" and B == 4 and s and not (args.small or args.graph or args.no_pmc or _under_profiler()):
            # roofline.traffic from THIS run's own counter passes (two rocprofv3 children, after every timed region); on any failure the recorded profile stays
            import gc
            model = one_step = step8 = out = None       # the parent's 37 GB of weights go before the children build their own copy
            gc.collect(); torch.cuda.empty_cache()
            try:
                live = same_run_traffic()
            except Exception as e:
                live = dict(error=f"{type(e).__name__}: {e}")
            line.update(_roofline_blocks(s, elapsed, pk16, pk32, info["attn_kernel"], traffic_file, live))
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def _flux_workload_inputs_only(model, B, rank, dev, grid=64):
    """A second batch size / image size on an existing UniGenFlux (the B = 8 scaling base, the B = 1 blocks): same input recipe as _flux_workload."""
    from unigen_amd.pipeline import denoise_loop, prepare_latent_image_ids
    from unigen_amd import dist_utils as DU
    T = 512
    N, E = grid * grid, model._ctl.expert_nums
    g = torch.Generator(device=dev).manual_seed(DU.rank_seed(12443, rank) + 1000)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32)
    lat, cond = rn(B, N, 64).to(torch.bfloat16), rn(B, N, 64).to(torch.bfloat16)
    prompt, pooled, cpool = (0.1 * rn(B, T, 4096)).to(torch.bfloat16), rn(B, 768).to(torch.bfloat16), rn(B, 768).to(torch.bfloat16)
    ids = prepare_latent_image_ids(grid, grid, dev, torch.bfloat16)
    txt = torch.zeros(T, 3, device=dev, dtype=torch.bfloat16)
    unis = [torch.rand(B * N, E, generator=g, device=dev) for _ in range(4)]
    step = lambda: denoise_loop(model, latents=lat.clone(), control_tokens=cond, prompt_embeds=prompt, pooled_prompt_embeds=pooled,
                                condition_pooled_prompt_embeds=cpool, text_ids=txt, latent_image_ids=ids, condition_ids=ids, num_inference_steps=4,
                                gate_uniforms=unis)
    return model, step, None


if __name__ == "__main__":
    main()
