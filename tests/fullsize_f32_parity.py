"""Parity at the FULL model size (FLUX-schnell geometry: 19 + 38 base blocks, 9 + 19 control blocks, D = 3072, H = 24; cfg1's token counts: 512^2 ->
N = 1024, T = 512, B = 1), one forward:
  truth    = the CPU oracle in fp32
  hip_f32  = the HIP engine through the fp32 verification twins (same host orchestration as the product path)
  hip_bf16 = the HIP product path;  ref_bf16 = the CPU oracle in bf16
Too slow for the test suite (the fp32 oracle forward takes minutes on 16 cores); run it as a tool and keep the line under profiles/.

  python tests/fullsize_f32_parity.py [flux|multi|sd3] [GRID] [--batch B] [--hw H] [--no-ref16] [--no-truth32] [--lora] [--write-bounds --commit SHA]
--batch B      the BASELINE configs' own batch (cfg2: flux 64 --batch 4; cfg3: multi 64 --batch 8; cfg5: sd3 --hw 128 --batch 8). The MoE's capacity and
               its random token selection run over all B x N tokens at once, so this is not B independent B = 1 cases.
--no-truth32   leave the fp32 oracle out (at B = 8 it does not fit a 20-minute GPU-box call); the record then holds the three distances among
               hip_f32, hip_bf16 and the oracle's bf16 evaluation, to be read against the B = 1 record where the fp32 truth exists.
A heartbeat thread prints one line a minute: the host oracle is silent for many minutes at these sizes."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unigen_ref as R
from unigen_amd.flux import UniGenFlux
from unigen_amd.sd3 import UniGenSD3

torch.set_num_threads(int(os.environ.get("UG_ORACLE_THREADS", min(16, os.cpu_count() or 16))))     # tests/conftest.py gives the children 13 of the box's 16 cores while the suite runs beside them
dev, BF = torch.device("cuda:0"), torch.bfloat16
def _opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
BATCH, NO_TRUTH32 = _opt("--batch", 1), "--no-truth32" in sys.argv
_t_start = time.perf_counter()
def _heartbeat():
    while True:
        time.sleep(60)
        print(f"[heartbeat] {time.perf_counter() - _t_start:.0f} s", flush=True)
threading.Thread(target=_heartbeat, daemon=True).start()
SD3 = len(sys.argv) > 1 and sys.argv[1] == "sd3"        # UniGenSD3 (SD3.5-medium: 24 joint blocks, dual attention 0-12, D = 1536, dh = 64), N = 1024, T = 333
MULTI = len(sys.argv) > 1 and sys.argv[1] == "multi"    # MultiCondtionUniGenFlux, depth + canny + openpose (cfg3's model: E = 12), N = 1024, T = 512
if SD3:
    Model, CTL = UniGenSD3, dict(use_shared_expert=True, use_modulate=False)
    cfg = R.SD3Config()
    HW = _opt("--hw", 64)                                  # latent side: 64 = 512^2 (N = 1024), 128 = 1024^2 (N = 4096)
    inp = R.make_sd3_inputs(cfg, B=BATCH, hw=HW, T=333)
    t = torch.full((BATCH,), 600.0)
    oracle = R.unigen_sd3_forward
else:
    from unigen_amd.flux import MultiCondtionUniGenFlux
    Model = MultiCondtionUniGenFlux if MULTI else UniGenFlux
    CTL = dict(use_rope=True, use_shared_expert=True, use_consis_module=False, use_single_trans_blocks=True, single_control_dev=2,
               single_block_control_method="overall_add", top_num=1, expert_num_each_condition=3)
    cfg = R.FluxConfig(condition_nums=3) if MULTI else R.FluxConfig()
    GRID = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 32            # 64: the metric's own size, 1024^2 (N = 4096)
    inp = R.make_inputs(cfg, B=BATCH, grid=GRID, T=512, n_cond=3 if MULTI else 1)
    t = torch.full((BATCH,), 0.75, dtype=BF)
    oracle = R.unigen_flux_forward
UniGenFlux = Model
R_forward = oracle
m16 = Model.from_config({}, device=dev, dtype=BF)
NC, CT = (3, ["depth", "canny", "openpose"]) if MULTI else (1, ["canny"])
m16.init_condition_block(condition_nums=NC, condition_types=list(CT), control_params=dict(CTL))
m16.init_synthetic_(seed=0, std=0.02)
LORA = "--lora" in sys.argv          # round 6: per-condition rank-16 LoRA adapters on the control branch's attention projections, all live (A12 on the hot path at full size)
def attach_lora(m):
    for i, name in enumerate(CT):
        m.add_lora(["attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0"], name, 16, 32.0, prefix="control_", init_lora_weights=False, seed=40 + i)
if LORA:
    attach_lora(m16)
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
res = {}
with torch.no_grad():
    mv = lambda v, f=None: [mv(x, f) for x in v] if isinstance(v, (list, tuple)) else (v.to(dev) if f is None or not v.is_floating_point() else v.to(dev).to(f))
    out16 = m16(timestep=t.to(dev), **{k: mv(v) for k, v in inp.items()})[0].float().cpu()
    st16 = {k: v.detach().cpu() for k, v in m16.state_dict().items()}
    lora_sites = {n: [(a, lay.scaling[a]) for a in lay.live_adapters()] for n, lay in m16._lora_sites.items()} if LORA else {}
    if LORA:       # the oracle's linear() takes the same adapters (R.LORA_KEY -> R.lora_linear) from the model's own tensors
        st16[R.LORA_KEY] = {n: [(st16[f"{n}.lora_A.{a}.weight"], st16[f"{n}.lora_B.{a}.weight"], sc) for a, sc in v] for n, v in lora_sites.items()}
    sd16 = dict(m16.state_dict())
    NO_REF16 = "--no-ref16" in sys.argv          # the test suite's 1024^2 case: skip the oracle's own bf16 evaluation (65 s of host time; its ratio to the HIP bf16 error is pinned at N = 1024)
    if NO_REF16:
        ref16 = None
    else:
        t0 = time.perf_counter(); ref16 = R_forward(st16, cfg, timestep=t, dtype=BF, **inp)[0].float(); res["oracle_bf16_s"] = round(time.perf_counter() - t0, 1)
    print("bf16 done", res, flush=True)
    m32 = UniGenFlux.from_config({}, device=dev, dtype=torch.float32)
    m32.init_condition_block(condition_nums=NC, condition_types=list(CT), control_params=dict(CTL))
    if LORA:
        attach_lora(m32)
    m32.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sd16.items()})
    del m16, sd16
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    out32 = m32(timestep=t.to(dev), **{k: (mv(v) if k == "gate_uniform" else mv(v, torch.float32)) for k, v in inp.items()})[0].cpu()
    torch.cuda.synchronize(); res["hip_f32_s"] = round(time.perf_counter() - t0, 1)
    del m32
    torch.cuda.empty_cache()          # the fp32 oracle below runs for minutes on the host: the GPU memory goes back first (the suite runs beside this child)
    st32 = {k: (v.float() if v.is_floating_point() else v) for k, v in st16.items() if k != R.LORA_KEY}
    if LORA:
        st32[R.LORA_KEY] = {n: [(st32[f"{n}.lora_A.{a}.weight"], st32[f"{n}.lora_B.{a}.weight"], sc) for a, sc in v] for n, v in lora_sites.items()}
    del st16
    print("hip f32 done", res, flush=True)
    truth = None
    if not NO_TRUTH32:
        t0 = time.perf_counter(); truth = R_forward(st32, cfg, timestep=t, dtype=torch.float32, **inp)[0]; res["oracle_f32_s"] = round(time.perf_counter() - t0, 1)
if SD3:
    wl = f"UniGenSD3, SD3.5-medium depth and width, N={(HW // cfg.patch_size) ** 2}, T=333, B={BATCH}"
else:
    wl = ("MultiCondtionUniGenFlux (3 conditions, E = 12), " if MULTI else "") + f"one forward at full depth and width, {16 * GRID}^2 (N={GRID * GRID}, T=512), B={BATCH}"
if LORA:
    wl += f"; rank-16 LoRA adapters ({', '.join(CT)}) live on {len(lora_sites)} control-branch attention projections"
opt = lambda a, b: rel(a, b) if a is not None and b is not None else None
res.update(workload=wl, rel_l2_hip_f32_vs_oracle_f32=opt(out32, truth), rel_l2_hip_bf16_vs_oracle_f32=opt(out16, truth),
           rel_l2_oracle_bf16_vs_oracle_f32=opt(ref16, truth), rel_l2_hip_bf16_vs_oracle_bf16=opt(out16, ref16),
           rel_l2_hip_bf16_vs_hip_f32=rel(out16, out32), rel_l2_oracle_bf16_vs_hip_f32=opt(ref16, out32))
print("FULLSIZE_PARITY", json.dumps(res))
if "--write-bounds" in sys.argv and BATCH == 1 and res["rel_l2_oracle_bf16_vs_oracle_f32"] is not None:
    # the record tests/test_fullsize_gpu.py::test_full_model_forward_parity reads when it skips the oracle's own bf16 evaluation (--no-ref16): written
    # under gpurun_out/ (the only directory a GPU box hands back); the builder copies it into tests/golden/fullsize_bounds.json
    which = "sd3" if SD3 else ("multi" if MULTI else f"flux{GRID}")
    commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else "unknown"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"fullsize_bounds_{which}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({which: dict(rel_l2_oracle_bf16_vs_oracle_f32=res["rel_l2_oracle_bf16_vs_oracle_f32"], rel_l2_hip_bf16_vs_oracle_f32=res["rel_l2_hip_bf16_vs_oracle_f32"],
                               rel_l2_hip_f32_vs_oracle_f32=res["rel_l2_hip_f32_vs_oracle_f32"], workload=res["workload"], written_utc=time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                               commit=commit, written_by="tests/fullsize_f32_parity.py --write-bounds")}, f, indent=1)
    print("wrote", out)
