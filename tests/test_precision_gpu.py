"""GPU: the precision schedule beyond the Gaussian synthetic checkpoint - the fp16 sub-phase (three-phase schedule: plain bf16 -> plain fp16
-> split-bf16, rgn_set_f16_steps) against the reference's goldens, a stress family of trained-like checkpoints against the oracle, and the
calibration's fp32 anchor. Bound everywhere: 1e-3 abs on rot6d (BASELINE.json north_star); the defaults must keep a 3x margin."""
import numpy as np
import pytest
import torch

from tests.helpers import build_hip, fixture_inputs, y_to_device

pytestmark = pytest.mark.gpu

MARGIN = 1e-3 / 3.0


def _wrap(model, guided):
    if not guided:
        return model
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    return ClassifierFreeSampleModel(model)


@pytest.mark.parametrize("name", ["ntu_ddpm1000", "ntu_action_ddim100_cfg", "ntu_eval_ddim5", "ntu_action_eval_ddim5", "text150_ddim50_cfg", "chi3d_ddim20_cfg",
                                  "chi3d_ddpm20"])
def test_three_phase_precision_schedule_sweep(golden, name):
    """Plain bf16 -> `f16_steps` plain fp16 steps -> split-bf16 tail, on the forms that have fp16 instantiations - the one-kernel decoder stack
    (60 frames; forced on for the goldens' B = 2) and the kernel-per-stage chain of the 150-frame models (k_qkv_attn_long + k_mlp2 + k_step) -
    against the reference's own outputs: the engine's default plan (8 fp16 steps, 2 split-bf16 steps) and its neighbours keep the 3x margin;
    what other (f16_steps, tail) pairs cost is printed. Reference: diffusion/gaussian_diffusion.py:508-560, model/cmdm.py:227."""
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    S, guided = int(g["S"]), bool(g["guided"])
    shape = (int(g["B"]), cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    errs, plans = {}, {}
    for n16, tail in [(None, None), (0, None), (8, 2), (8, 1), (2, 2), (10000, 2), (10000, 1), (8, 0), (0, 2)]:
        model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision="bf16_x3tail/throughput", x3_tail=tail, f16_steps=n16,
                                     engine_options={"LAYERS_MIN_B": 1})
        fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
        out = fn(_wrap(model, guided), shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
        errs[(n16, tail)] = float(np.abs(out.cpu().numpy() - g["final"]).max())
        plans[(n16, tail)] = model._engine.precision_plan(shape[0], guided)
        model._engine.close()
    print(f"\n[three-phase schedule] {name}: " + ", ".join(f"(f16 {k[0]}, tail {k[1]}) -> plan {plans[k]}: {v:.2e}" for k, v in errs.items()))
    assert plans[(None, None)] == (min(8, S - 2), 2)                       # the default where k_layers<true> runs the plain phase
    assert plans[(0, None)][0] == 0 and plans[(0, None)][1] == (3 if S <= 10 else 5)   # no fp16 phase: the bf16 rule's tail
    for k in [(None, None), (0, None), (8, 2), (10000, 2)]:
        assert errs[k] < MARGIN, (name, k, errs[k])
    for k in [(8, 1), (2, 2), (10000, 1)]:
        assert errs[k] < 1e-3, (name, k, errs[k])
    assert errs[(None, None)] == errs[(8, 2)]                              # the default IS (8, 2)


@pytest.mark.parametrize("name", ["ntu_eval_ddim5", "ntu_ddpm50", "ntu_action_ddim100_cfg"])
def test_three_phase_schedule_on_the_kernel_per_stage_chain_at_60_frames(golden, name):
    """Below the one-kernel stack's batch threshold (64 motions) a 60-frame model runs k_qkv_attn_rs + k_mlp2 + k_step per step: the same three-phase
    plan there (the reference's default batch is 64 and a strong-scaled shard is smaller: utils/parser_util.py:95), against the reference's outputs."""
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    S, guided = int(g["S"]), bool(g["guided"])
    shape = (int(g["B"]), cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    errs = {}
    for n16 in (None, 0):
        model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision="bf16_x3tail/throughput", f16_steps=n16, engine_options={"LAYERS": 0})
        fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
        out = fn(_wrap(model, guided), shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
        errs[n16] = float(np.abs(out.cpu().numpy() - g["final"]).max())
        plan, kern = model._engine.precision_plan(shape[0], guided), model._engine.plan_query(shape[0], guided)
        assert kern["qkv_attn"]["kernel"] == "k_qkv_attn_rs" and kern["mlp"]["kernel"] == "k_mlp2" and "step_fused" in kern, kern
        assert plan == ((min(8, S - 2), 2) if n16 is None else (0, 3 if S <= 10 else 5)), plan
        model._engine.close()
    print(f"\n[three-phase schedule, kernel per stage at 60 frames] {name}: default {errs[None]:.2e}, bf16 rule {errs[0]:.2e}")
    assert errs[None] < MARGIN and errs[0] < MARGIN, errs


def test_three_phase_schedule_is_invariant_under_range_cuts_and_batch_composition():
    """A call cut into ranges (the progressive API's shape of work) crosses the bf16 -> fp16 boundary at a different point of the launch
    sequence (planes re-encoded behind an embedding instead of behind a bf16 launch), and a motion's result may not depend on its batch:
    B = 64 in one piece == the same motions in two ranges (up to the first embedding of a range: <= 2e-5) and row b == the B = 1 run."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    B, S = 64, 24
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=1)}
    model, diffusion = build_hip(cfg, sd, resp=str(S), precision="bf16_x3tail/throughput")
    shape = (B, 56, 6, 60)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, seed=7)
    assert model._engine.precision_plan(B) == (8, 2)
    plan = model._engine.plan_query(B)
    assert "steps_fused" in plan, plan
    # ranges: [23 .. 12], [11 .. 6] (inside the fp16 sub-phase: indices 9 .. 2), [5 .. 0]
    eng, dev = model._engine, torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty(shape, device=dev)
    eng.randn(x, B, 7, 0, st)
    for first, count in ((23, 12), (11, 6), (5, 6)):
        eng.sample_range("ddpm", False, 0.0, x, None, 7, 0, first, count, None, True, False, st)
    torch.cuda.synchronize()
    # (every range starts from the up-front embedding - x through bf16 operand planes, one rounding - where the uncut call hands the in-kernel
    #  embedding's image over; a cut inside the fp16 sub-phase, five steps from the end, is the case where that difference matters most)
    assert float((x - out).abs().max()) < 1e-4
    model1, diffusion1 = build_hip(cfg, sd, resp=str(S), precision="bf16_x3tail/throughput", engine_options={"LAYERS_MIN_B": 1})
    for b in (0, 31, 63):
        yb = {"cmotion": torch.from_numpy(y["cmotion"][b:b + 1]).cuda()}
        one = diffusion1.p_sample_loop(model1, (1, 56, 6, 60), clip_denoised=False, model_kwargs={"y": yb}, seed=7, sample_offset=b)
        assert float((one - out[b:b + 1]).abs().max()) < 2e-5, b
    model._engine.close()
    model1._engine.close()


def test_fp16_weight_range_is_checked_at_load():
    """fp16 has a 5-bit exponent: a checkpoint with a weight beyond 6e4 would turn into inf in the fp16 weight planes - rgn_finalize_weights refuses
    it and names the key; with the fp16 phase switched off ("BULK_F16": 0) the same checkpoint loads and samples."""
    from regennet_amd import synth
    from regennet_amd._lib import RgnError
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    sd["seqTransDecoder.layers.3.linear1.weight"] = sd["seqTransDecoder.layers.3.linear1.weight"].copy()
    sd["seqTransDecoder.layers.3.linear1.weight"][5, 7] = 7.0e4
    model, diffusion = build_hip(cfg, sd, resp="5", precision="bf16_x3tail")
    with pytest.raises(RgnError) as ei:
        model._get_engine(2)
    assert "seqTransDecoder.layers.3.linear1.weight" in str(ei.value) and "BULK_F16" in str(ei.value)
    model, diffusion = build_hip(cfg, sd, resp="5", precision="bf16_x3tail", engine_options={"BULK_F16": 0})
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, 2, seed=1)).cuda()}
    out = diffusion.p_sample_loop(model, (2, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, seed=3)
    assert torch.isfinite(out).all() and model._engine.precision_plan(2) == (0, 3)
    model._engine.close()


# ---- the stress family -----------------------------------------------------------------------------------------------------------------------
_ORACLE = {}


def _family_case(family, sched):
    """(cfg, sd, y numpy, tape, oracle result) of a family x schedule, oracle computed once per session (CPU, seconds)."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    key = (family, sched)
    if key not in _ORACLE:
        cfg = synth.get_config("ntu")
        sd = synth.make_state_dict_family(cfg, family, seed=3)
        B = 2
        resp, S = ("50", 50) if sched == "ddpm50" else ("ddim5", 5)
        y = {"cmotion": synth.make_cmotion(cfg, B, seed=71)}
        tape = synth.make_noise_tape(cfg, B, S, seed=72)
        ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", resp), tape, {k: torch.from_numpy(v) for k, v in y.items()}, mode="ddpm").numpy()
        _ORACLE[key] = (cfg, sd, y, tape, ref, resp)
    return _ORACLE[key]


@pytest.mark.parametrize("form", ["k_layers", "engine default"])
@pytest.mark.parametrize("sched", ["ddpm50", "ddim5"])
@pytest.mark.parametrize("family", ["heavy_tailed", "outlier_channels", "peaky_attention", "big_output", "small_signal"])
def test_stress_family_checkpoints(family, sched, form, capfd):
    """The 1e-3 claim on checkpoints that are NOT i.i.d. Gaussian (synth.make_state_dict_family: Student-t weights, six outlier channels with
    LayerNorm gain and shift x2 and 1/2 reader weights, q/k gain 6, poseFinal x3, all gains x0.3): HIP against the oracle on a 50-step DDPM
    schedule and the reference's 5-step evaluation schedule, on the one-kernel decoder stack (three-phase schedule) and on what the engine
    picks for B = 2 by itself.
      * x3_tail="auto" - what a checkpoint loaded through the factory path gets (utils/model_util.py:5-8 -> CMDM.load_state_dict): the switch
        point is MEASURED against an fp32 run of the first motions. It must meet 1e-3 on every family.
      * the default rule (derived on the Gaussian family; what synth.build_model / bench.py use on their own synthetic checkpoints) must meet it
        with a 3x margin on four families. On the outlier-channel family it does NOT, by two orders of magnitude - plain 16-bit steps are not
        contracted away by that denoiser (uniform split-bf16 itself sits 6e-4 from the oracle where the Gaussian family sits at 5e-5) - which is
        asserted, so that nobody makes the rule the factory path's default: there the calibration keeps the whole schedule split-bf16."""
    cfg, sd, y, tape, ref, resp = _family_case(family, sched)
    opts = {"LAYERS_MIN_B": 1} if form == "k_layers" else {}
    prec = "bf16_x3tail/throughput" if form == "k_layers" else "bf16_x3tail"
    got = {}
    for tail in (None, "auto"):
        model, diffusion = build_hip(cfg, sd, resp=resp, precision=prec, x3_tail=tail, engine_options=opts)
        out = diffusion.p_sample_loop(model, ref.shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
        got[tail] = (float(np.abs(out.cpu().numpy() - ref).max()), model._engine.precision_plan(2), model._auto_tail)
        model._engine.close()
    log = capfd.readouterr().err
    print(f"\n[stress family] {family} {sched} {form}: default rule {got[None][0]:.2e} plan {got[None][1]}; measured {got['auto'][0]:.2e} (tail {got['auto'][2]})")
    assert got["auto"][0] < 1e-3, got
    if family == "outlier_channels":
        assert got[None][0] > 1e-3, "the default rule now holds on the outlier family: re-derive this test's statement"
        assert got["auto"][2] == len(tape) - 1 and "split-bf16 arithmetic itself differs from fp32" in log, (got, log)
    else:
        assert got[None][0] < MARGIN, got
        assert got["auto"][2] <= max(got[None][1][1], 8), got            # ... and the measured switch point is the rule's (or close to it)
    if form == "k_layers":
        assert got[None][1][0] > 0                                        # the fp16 sub-phase took part


def test_parity_presupposes_a_sampler_map_that_is_not_chaotic():
    """What the stress family cannot contain: LayerNorm gains x8 on six channels WITHOUT the small reader weights training pairs them with make the
    denoiser's gain so large that the sampling loop amplifies any difference exponentially - the engine's fp32 mode (exact-product MFMA; 6e-6 ...
    1.6e-5 on every golden) and the fp32 oracle, which differ only in summation order, end O(1) apart after 50 steps. No arithmetic meets an
    element-wise bound on such a checkpoint; the 1e-3 claim is about checkpoints whose sampler map is not chaotic - every trained denoiser, or
    its samples would not be reproducible across GPU models either."""
    cfg, sd, y, tape, ref, resp = _family_case("outlier_channels_uncompensated", "ddpm50")
    model, diffusion = build_hip(cfg, sd, resp=resp, precision="f32")
    out = diffusion.p_sample_loop(model, ref.shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
    err = float(np.abs(out.cpu().numpy() - ref).max())
    print(f"\n[chaotic checkpoint] fp32-mode HIP vs fp32 oracle after 50 steps: {err:.2e} (outputs up to {float(np.abs(ref).max()):.1f})")
    assert err > 1e-2
    model._engine.close()


def test_calibration_refuses_a_checkpoint_on_which_split_bf16_itself_drifts(capfd):
    """`x3_tail="auto"` compares candidates with the uniform split-bf16 run - which is only a reference while split-bf16 itself tracks fp32. On the
    "hostile" family (Student-t weights, 24 uncompensated outlier channels x64, q/k gain 10, poseFinal x8) it does not by a wide margin: the calibration
    measures against an fp32-mode run of the same motions, keeps the whole schedule split-bf16 and says so once."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict_family(cfg, "hostile", seed=3)
    B = 2
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=71)}
    model, diffusion = build_hip(cfg, sd, resp="ddim5", precision="bf16_x3tail/throughput", x3_tail="auto", engine_options={"LAYERS_MIN_B": 1})
    out = diffusion.p_sample_loop(model, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y_to_device(y)}, seed=5)
    log = capfd.readouterr().err
    m32, d32 = build_hip(cfg, sd, resp="ddim5", precision="f32")
    o32 = d32.p_sample_loop(m32, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y_to_device(y)}, seed=5)
    mx3, dx3 = build_hip(cfg, sd, resp="ddim5", precision="bf16x3/throughput")
    ox3 = dx3.p_sample_loop(mx3, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y_to_device(y)}, seed=5)
    drift = float((ox3 - o32).abs().max())
    print(f"\n[hostile checkpoint] uniform split-bf16 vs fp32: {drift:.2e}; calibrated tail {model._auto_tail}")
    assert drift > 2.5e-4, "the hostile family no longer defeats split-bf16: make it harsher"
    assert model._auto_tail == 5 and "split-bf16 arithmetic itself differs from fp32" in log, (model._auto_tail, log)
    assert torch.equal(out, ox3)                                          # the schedule ran split-bf16 throughout
    for m in (model, m32, mx3):
        m._engine.close()
