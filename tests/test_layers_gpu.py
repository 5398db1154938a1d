"""GPU: the one-kernel decoder stack (rgn_layers.hip: one workgroup per sample carries the residual stream through all layers -
and, unguided, through whole runs of sampler steps) against the reference's goldens, the oracle and the kernel-per-stage chain it
replaces. The engine takes it for evaluations of >= 64 samples of 52 .. 64 tokens; REGENNET_LAYERS_MIN_B=1 lets test-sized batches
reach it. Tolerance: BASELINE.json north_star, 1e-3 abs on rot6d."""
import numpy as np
import pytest
import torch

from tests.helpers import build_hip, fixture_inputs, y_to_device

pytestmark = pytest.mark.gpu

# per evaluation: k_layers<false> + k_step / k_update per step; whole runs: k_layers<true> (unguided, no emb_trans_dec token)
FORMS = {"per-step": {"REGENNET_LAYERS_MIN_B": "1", "REGENNET_LAYERS_STEPS": "0"}, "multi-step": {"REGENNET_LAYERS_MIN_B": "1", "REGENNET_LAYERS_GUIDED": "2"},   # (guided: a motion per workgroup - forced: the engine's rule takes it only for 2 B > #CUs)
         "kernel-per-stage": {"REGENNET_LAYERS": "0"}}


def _engine_with(monkeypatch, env, model, B):
    """The kernel-selection switches of THIS model's engines (rgn_set_option through CMDM.engine_options: per handle, no process-wide
    environment involved); they are read when an engine is built."""
    del monkeypatch
    model.engine_options = {k[len("REGENNET_"):]: int(v) for k, v in env.items()}
    model._engine_stale = True
    model._get_engine(B)


def _wrap(model, guided):
    if not guided:
        return model
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    return ClassifierFreeSampleModel(model)


@pytest.mark.parametrize("form", ["per-step", "multi-step"])
@pytest.mark.parametrize("name", ["ntu_ddpm50", "ntu_action_ddim100_cfg", "ntu_add_etd_ddpm20", "ntu_ddpm1000"])
def test_decoder_stack_kernel_against_the_reference(golden, monkeypatch, name, form):
    """The reference's own sampling-loop outputs on identical noise: 50 / 1000-step DDPM (unguided: whole runs of steps in one
    launch in the multi-step form), guided 100-step DDIM (k_layers per evaluation over the 2B rows + the guided k_step), 61 tokens
    with the emb_trans_dec token (k_layers + the three-kernel step boundary)."""
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision="bf16_x3tail/throughput")
    B = int(g["B"])
    _engine_with(monkeypatch, FORMS[form], model, B)
    fm = _wrap(model, bool(g["guided"]))
    shape = (B, cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
    out = fn(fm, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
    err = np.abs(out.cpu().numpy() - g["final"]).max()
    print(f"\n[k_layers {form}] {name}: {err:.2e}")
    assert err < 1e-3, (name, form, err)
    model._engine.close()


@pytest.mark.parametrize("sampler,clip,guided", [("ddpm", False, False), ("ddim", True, False), ("ddim", False, True), ("ddpm", True, True)])
def test_decoder_stack_kernel_against_the_kernel_per_stage_chain(monkeypatch, sampler, clip, guided):
    """Same noise stream (on-device Philox), same sampler arithmetic: the three forms differ only by where bf16 roundings fall in
    the plain-bf16 phase and end within half the parity margin of each other behind the split-bf16 tail; the multi-step form also
    honours clip_denoised like the per-step one. Guided (per-sample scales): in the multi-step form a workgroup owns a MOTION and runs
    its conditional and unconditional evaluations back to back, the conditional x0 parked in global scratch meanwhile."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu_action" if guided else "ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    B = 9
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1)).cuda()}
    if guided:
        y["action"] = torch.from_numpy(synth.make_actions(cfg, B, seed=2)).cuda()
        y["scale"] = torch.linspace(1.5, 3.5, B).cuda()
    outs = {}
    for form, env in FORMS.items():
        model, diffusion = build_hip(cfg, sd, resp="40" if sampler == "ddpm" else "ddim40", precision="bf16_x3tail/throughput")
        _engine_with(monkeypatch, env, model, B)
        fn = diffusion.p_sample_loop if sampler == "ddpm" else diffusion.ddim_sample_loop
        outs[form] = fn(_wrap(model, guided), (B, 56, 6, 60), clip_denoised=clip, model_kwargs={"y": y}, seed=5)
        model._engine.close()
    for form in ("per-step", "multi-step"):
        assert torch.isfinite(outs[form]).all()
        dev = (outs[form] - outs["kernel-per-stage"]).abs().max().item()
        print(f"\n[k_layers {form} vs kernel per stage] {sampler} clip={clip} guided={guided}: {dev:.2e}")
        assert 0.0 < dev < 5e-4
    dev = (outs["per-step"] - outs["multi-step"]).abs().max().item()
    print(f"\n[k_layers per-step vs multi-step] {sampler} clip={clip} guided={guided}: {dev:.2e}")
    assert dev < 5e-4


def test_decoder_stack_kernel_is_independent_of_the_batch_composition(monkeypatch):
    """One workgroup per sample, nothing shared but the weights: a motion drawn alone with its global sample index (same Philox key)
    equals its row of the batch BIT FOR BIT in the plain-bf16 phase - what sharding over GPUs relies on."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    B = 70                                                    # >= 64: the default engine takes the one-kernel form
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1)).cuda()}
    model, diffusion = build_hip(cfg, sd, resp="12", precision="bf16_x3tail/throughput", x3_tail=0)
    full = diffusion.p_sample_loop(model, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, seed=11)
    model._engine.close()
    model1, diffusion1 = build_hip(cfg, sd, resp="12", precision="bf16_x3tail/throughput", x3_tail=0)
    _engine_with(monkeypatch, FORMS["multi-step"], model1, 1)
    for b in (0, 33, 69):
        yb = {k: v[b:b + 1].contiguous() for k, v in y.items()}
        one = diffusion1.p_sample_loop(model1, (1, 56, 6, 60), clip_denoised=False, model_kwargs={"y": yb}, seed=11, sample_offset=b)
        assert torch.equal(full[b:b + 1], one), (b, (full[b:b + 1] - one).abs().max().item())
    model1._engine.close()


def test_decoder_stack_kernel_quad_shared_noise_is_the_per_element_stream(monkeypatch):
    """The multi-step launch draws its Philox normals per quad of lanes (four frames of one feature per call, transposed inside the
    quad by two butterfly stages): bit-identical to the per-element draw (REGENNET_STEP_NO_QUADS=1), for DDPM and DDIM."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    B = 3
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1)).cuda()}
    for sampler, resp in (("ddpm", "8"), ("ddim", "ddim8")):
        outs = []
        for extra in ({}, {"REGENNET_STEP_NO_QUADS": "1"}):
            model, diffusion = build_hip(cfg, sd, resp=resp, precision="bf16_x3tail/throughput", x3_tail=0)
            _engine_with(monkeypatch, dict(FORMS["multi-step"], **extra), model, B)
            fn = diffusion.p_sample_loop if sampler == "ddpm" else diffusion.ddim_sample_loop
            kw = {"eta": 1.0} if sampler == "ddim" else {}
            outs.append(fn(model, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, seed=21, **kw))
            model._engine.close()
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()


def test_decoder_stack_kernel_lengths_and_partial_ranges(monkeypatch):
    """52 .. 64 tokens per sample (padding rows replicate the last token and are masked as keys) against the oracle, and a sampling
    call cut into ranges through the C-ABI (rgn_sample_range with a count below the schedule: the loop index lives on the device and
    a multi-step launch moves it by the steps it ran) against the same call in one piece: equal up to the first embedding of each
    range, which adds the fp32 condition rows where a step boundary inside a run adds their bf16 copy."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    for T in (52, 57, 64):
        cfg = synth.get_config("ntu", num_frames=T)
        sd = synth.make_state_dict(cfg, seed=0)
        B = 3
        y = {"cmotion": synth.make_cmotion(cfg, B, seed=7)}
        tape = synth.make_noise_tape(cfg, B, 12, seed=8)
        ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", "12"), tape, {k: torch.from_numpy(v) for k, v in y.items()}, mode="ddpm").numpy()
        model, diffusion = build_hip(cfg, sd, resp="12", precision="bf16_x3tail/throughput")
        _engine_with(monkeypatch, FORMS["multi-step"], model, B)
        out = diffusion.p_sample_loop(model, (B, 56, 6, T), clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
        err = float(np.abs(out.cpu().numpy() - ref).max())
        print(f"\n[k_layers multi-step vs oracle] {T} frames: {err:.2e}")
        assert err < 1e-3, (T, err)
        if T == 57:   # the same 12 steps as ranges of 4 + 5 + 3 (the last two straddle the switch to the split-bf16 tail at index 4)
            eng = model._engine
            st = torch.cuda.current_stream().cuda_stream
            td = torch.from_numpy(tape).cuda()
            x = td[0].clone()
            for first, count in ((11, 4), (7, 5), (2, 3)):
                eng.sample_range("ddpm", False, 0.0, x, td[1 + (11 - first):], 0, 0, first, count, None, True, False, st)
            torch.cuda.synchronize()
            assert (x - out).abs().max().item() < 2e-4, (x - out).abs().max().item()
        model._engine.close()


HEADLINE_ROWS = (0, 31, 32, 128, 255)   # first / last sample of XCD 0's range, first of XCD 1's, the middle, the last workgroup of the launch


@pytest.mark.parametrize("tail", [0, None])
def test_headline_launch_at_its_real_shape_rows_equal_single_sample_runs(monkeypatch, tail):
    """BASELINE configs[1] EXACTLY as bench.py issues it (gaussian_diffusion.py:675-742): NTU B = 256, the full 1000-step DDPM call, default
    engine and schedule, on-device Philox - ONE k_layers<true> launch of 256 workgroups x 995 steps (1000 with tail = 0), where the
    XCD-affine block map, the last-workgroup index hand-off and 995 iterations of the in-kernel Philox run together. Rows
    {0, 31, 32, 128, 255} against B = 1 runs of the same kernel form with the motion's global Philox key (sample_offset = b):
    bit for bit in the plain-bf16 phase (tail = 0), within 2e-5 behind the default split-bf16 tail."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    B = 256
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1)).cuda()}
    model, diffusion = synth.build_model(cfg, sd, resp="", precision="bf16_x3tail", device="cuda:0", x3_tail=tail)   # (bench.py's construction)
    assert diffusion.num_timesteps == 1000
    full = diffusion.p_sample_loop(model, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, seed=100, sample_offset=0)
    assert torch.isfinite(full).all()
    model._engine.close()
    model1, diffusion1 = build_hip(cfg, sd, resp="", precision="bf16_x3tail/throughput", x3_tail=tail)
    _engine_with(monkeypatch, FORMS["multi-step"], model1, 1)
    worst = 0.0
    for b in HEADLINE_ROWS:
        yb = {k: v[b:b + 1].contiguous() for k, v in y.items()}
        one = diffusion1.p_sample_loop(model1, (1, 56, 6, 60), clip_denoised=False, model_kwargs={"y": yb}, seed=100, sample_offset=b)
        dev = (full[b:b + 1] - one).abs().max().item()
        worst = max(worst, dev)
        if tail == 0:
            assert torch.equal(full[b:b + 1], one), (b, dev)
        else:
            assert dev <= 2e-5, (b, dev)
    print(f"\n[headline launch, B = 256 x 1000 steps, tail = {tail}] rows {HEADLINE_ROWS} vs single-sample runs: max |dev| = {worst:.1e}")
    model1._engine.close()


def test_headline_launch_shape_against_the_oracle_on_a_100_step_schedule():
    """The same launch shape (256 workgroups, default engine, default precision schedule, on-device Philox) on a 100-step DDPM schedule,
    every 32nd motion and the last (one workgroup of every XCD) against the ORACLE on the very noise the kernel drew: x_T and the 100 per-step draws
    are re-drawn through rgn_randn_step - the fused loop's own Philox stream, element for element
    (test_model_kwargs_the_fused_loop_does_not_read) - and handed to the oracle as its tape. Bound: north_star's 1e-3."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    B, S, seed = 256, 100, 77
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=71)}
    model, diffusion = synth.build_model(cfg, sd, resp=str(S), precision="bf16_x3tail", device="cuda:0")
    yd = y_to_device(y)
    shape = (B, 56, 6, 60)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": yd}, seed=seed, sample_offset=0)
    assert torch.isfinite(out).all()
    idx = np.r_[np.arange(0, B, 32), B - 1]
    eng = model._engine
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.empty(shape, device="cuda")
    tape = np.empty((S + 1, len(idx)) + shape[1:], dtype=np.float32)
    for k, loop_index in enumerate([-1] + list(range(S - 1, -1, -1))):     # draw order: x_T, then loop indices S-1 .. 0
        eng.randn_step(buf, B, seed, 0, loop_index, st)
        tape[k] = buf[idx].cpu().numpy()
    ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", str(S)), tape,
                          {k: torch.from_numpy(np.ascontiguousarray(v[idx])) for k, v in y.items()}, mode="ddpm").numpy()
    err = float(np.abs(out.cpu().numpy()[idx] - ref).max())
    print(f"\n[headline launch shape, 95 + 5 steps, B = 256, on-device Philox] a subset of the motions vs oracle: {err:.2e}")
    assert err < 1e-3, err
    model._engine.close()


def test_guided_batches_that_do_not_fill_the_chip_run_an_evaluation_per_workgroup():
    """Guided sampling on the one-kernel stack has two forms: a MOTION per workgroup (k_layers<true, true>: its two evaluations back to back inside
    whole runs of steps) and an EVALUATION per workgroup and step (k_layers<false> over the 2 B rows + the guided k_step, hipGraph replays). The first
    fills the chip only when B alone does; at 2 B <= #CUs the second runs a step in one evaluation's latency instead of two (cfg3's shape at B = 64:
    1340 vs 786 motions/s same-box), so the engine picks by batch size (rgn_host.h; LAYERS_GUIDED = 0 / 1 / 2, settable per call). Checked: the plan
    the engine reports for B = 64 and B = 256, the three-phase precision plan on both, B = 64 against the ORACLE on the kernels' own Philox draws,
    rows against single-motion runs of the same form, and the two forms against each other.
    Reference: model/cfg_sampler.py:24-31, utils/parser_util.py:95 (--batch_size 64)."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    cfg = synth.get_config("ntu_action")
    sd = synth.make_state_dict(cfg, seed=0)
    B, S, seed = 64, 20, 91
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=71), "action": synth.make_actions(cfg, B, seed=72), "scale": np.full((B,), 2.5, np.float32)}
    model, diffusion = synth.build_model(cfg, sd, resp="ddim20", precision="bf16_x3tail", device="cuda:0")
    fm = ClassifierFreeSampleModel(model)
    shape = (B, 56, 6, 60)
    out = diffusion.ddim_sample_loop(fm, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, seed=seed)
    eng = model._engine
    plan = eng.plan_query(B, guided=True)
    assert plan.get("layers", {}).get("kernel") == "k_layers<false>" and plan["step_fused"]["kernel"] == "k_step<guided>" and "steps_fused" not in plan, plan
    assert eng.precision_plan(B, True) == (8, 2)
    assert "steps_fused" in eng.plan_query(256, guided=True) if eng.max_batch >= 256 else True
    # against the oracle on the very noise the kernels drew
    idx = np.arange(0, B, 8)
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.empty(shape, device="cuda")
    tape = np.empty((S + 1, len(idx)) + shape[1:], dtype=np.float32)
    for k, loop_index in enumerate([-1] + list(range(S - 1, -1, -1))):
        eng.randn_step(buf, B, seed, 0, loop_index, st)
        tape[k] = buf[idx].cpu().numpy()
    ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", "ddim20"), tape, {k: torch.from_numpy(np.ascontiguousarray(v[idx])) for k, v in y.items()},
                          mode="ddim", guided=True).numpy()
    err = float(np.abs(out.cpu().numpy()[idx] - ref).max())
    # rows against single-motion runs of the same form (LAYERS_MIN_B = 1, an evaluation per workgroup), and the other form on the whole batch
    model1, diffusion1 = build_hip(cfg, sd, resp="ddim20", precision="bf16_x3tail/throughput", engine_options={"LAYERS_MIN_B": 1})
    model1.layers_guided = 0
    fm1 = ClassifierFreeSampleModel(model1)
    worst = 0.0
    for b in (0, 31, 63):
        yb = {k: torch.from_numpy(v[b:b + 1]).cuda() for k, v in y.items()}
        one = diffusion1.ddim_sample_loop(fm1, (1, 56, 6, 60), clip_denoised=False, model_kwargs={"y": yb}, seed=seed, sample_offset=b)
        worst = max(worst, float((one - out[b:b + 1]).abs().max()))
    model.layers_guided = 2
    other = diffusion.ddim_sample_loop(fm, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, seed=seed)
    assert "steps_fused" in eng.plan_query(B, guided=True)
    dev = float((other - out).abs().max())
    print(f"\n[guided B = 64, ddim20] evaluation per workgroup vs oracle {err:.2e}; rows vs single-motion runs {worst:.1e}; vs a motion per workgroup {dev:.2e}")
    assert err < 1e-3 and worst <= 2e-5 and dev < 5e-4, (err, worst, dev)
    model._engine.close()
    model1._engine.close()
