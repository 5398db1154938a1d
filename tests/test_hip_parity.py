"""GPU: the HIP path (through the C-ABI) against golden vectors recorded from the reference and against
the oracle on the same seeded inputs. Tolerance from BASELINE.json north_star: 1e-3 abs on rot6d
(written per test); timestep indices bit-exact."""
import numpy as np
import pytest
import torch

from tests.helpers import autoreg_inputs, build_hip, fixture_inputs, fixture_opts, y_to_device

pytestmark = pytest.mark.gpu

PRECISIONS = ["f32", "bf16x3"]
TOL = {"f32": 2e-4, "bf16x3": 1e-3}       # abs; both inside the 1e-3 contract

FWD = ["tiny_fwd", "tiny_fwd_cfg", "tiny_add_fwd", "tiny_etd_fwd", "tiny_wope_fwd", "tiny_text_fwd_cfg", "ntu_fwd",
       "ntu_action_fwd_cfg", "chi3d_fwd"]
LOOPS = ["tiny_ddpm10", "tiny_ddim10_cfg", "tiny_add_ddpm1000", "tiny_text_ddim20_cfg", "tiny_etd_ddim10_cfg",
         "tiny_wope_ddpm10", "ntu_add_etd_ddpm20", "ntu_ddpm50",
         "ntu_action_ddim100_cfg", "text150_ddim50_cfg"]


def _wrap(model, guided):
    if not guided:
        return model
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    return ClassifierFreeSampleModel(model)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", FWD)
def test_denoiser_forward(golden, name, precision):
    g = golden(name)
    cfg, sd, y, x = fixture_inputs(g, loop=False)
    model, _ = build_hip(cfg, sd, precision=precision)
    fm = _wrap(model, bool(g["guided"]))
    yd = y_to_device(y)
    xd = torch.from_numpy(x).cuda()
    for i, t in enumerate(g["ts"]):
        out = fm(xd, torch.full((x.shape[0],), int(t), dtype=torch.long, device="cuda"), y=yd)
        assert out.shape == xd.shape and out.dtype == torch.float32 and out.is_cuda
        err = np.abs(out.cpu().numpy() - g["out"][i]).max()
        assert err < TOL[precision], (name, int(t), err)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", LOOPS)
def test_sampling_loop(golden, name, precision):
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision=precision)
    # bit-exact timestep indices (respace.py:124-129)
    step = 2 if bool(g["guided"]) else 1
    assert diffusion.timestep_map[::-1] == g["model_t"][::step].tolist()
    fm = _wrap(model, bool(g["guided"]))
    shape = (int(g["B"]), cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
    out = fn(fm, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
    err = np.abs(out.cpu().numpy() - g["final"]).max()
    assert err < 1e-3, (name, err)
    if "x0" in g:   # per-step trace through the progressive API
        pfn = diffusion.p_sample_loop_progressive if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop_progressive
        for k, o in enumerate(pfn(fm, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)},
                                  noise_tape=torch.from_numpy(tape))):
            assert np.abs(o["pred_xstart"].cpu().numpy() - g["x0"][k]).max() < 1e-3
            assert np.abs(o["sample"].cpu().numpy() - g["x"][k]).max() < 1e-3


@pytest.mark.parametrize("precision", PRECISIONS)
def test_headline_1000_step_ddpm(golden, precision):
    """BASELINE configs[0]/[1] shape: NTU120-AS, 1000-step DDPM, identical noise; <= 1e-3 abs vs the reference."""
    g = golden("ntu_ddpm1000")
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    model, diffusion = build_hip(cfg, sd, precision=precision)
    assert diffusion.timestep_map == list(range(1000)) and g["model_t"].tolist() == list(range(999, -1, -1))
    shape = (2, 56, 6, 60)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)},
                                  noise_tape=torch.from_numpy(tape), use_graph=True)
    err = np.abs(out.cpu().numpy() - g["final"]).max()
    assert err < 1e-3, err


def test_graph_replay_equals_eager(golden):
    g = golden("ntu_ddpm50")
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    model, diffusion = build_hip(cfg, sd, resp="50")
    shape = (2, 56, 6, 60)
    kw = dict(clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
    a = diffusion.p_sample_loop(model, shape, use_graph=False, **kw)
    b = diffusion.p_sample_loop(model, shape, use_graph=True, **kw)
    c = diffusion.p_sample_loop(model, shape, use_graph=True, **kw)   # replays the cached graph
    assert torch.equal(a, b) and torch.equal(b, c)


@pytest.mark.parametrize("frames,etd", [(16, False), (64, False), (63, True), (64, True)])
def test_oracle_parity_sequence_length_edges(frames, etd):
    """The fused in_proj+attention kernel takes Tq <= 64 tokens: one token tile only (16), exactly full tiles (64, and
    63 + the emb_trans_dec token), and one token too many (64 + 1 -> the unfused GEMM + attention kernels)."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    cfg = synth.get_config("ntu_action", layers=2, num_frames=frames, emb_trans_dec=etd)
    sd = synth.make_state_dict(cfg, seed=9)
    B = 3
    model, diffusion = build_hip(cfg, sd, resp="ddim5", precision="bf16x3")
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=31), "action": synth.make_actions(cfg, B, seed=32)}
    tape = synth.make_noise_tape(cfg, B, 5, seed=33)
    ty = {k: torch.from_numpy(v) for k, v in y.items()}
    ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", "ddim5"), tape, ty, mode="ddim").numpy()
    out = diffusion.ddim_sample_loop(model, (B, 56, 6, frames), clip_denoised=False, model_kwargs={"y": y_to_device(y)},
                                     noise_tape=torch.from_numpy(tape))
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-3


@pytest.mark.parametrize("precision", PRECISIONS)
def test_oracle_parity_random_inputs(precision):
    """HIP vs oracle on fresh seeded inputs (not the golden ones), ragged batch (odd: the fused in_proj+attention kernel
    gets full sample pairs and a half-empty one), sequence shorter than the 64-token tile, mid-size model."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    cfg = synth.get_config("ntu_action", layers=3, num_frames=37)
    sd = synth.make_state_dict(cfg, seed=7)
    B = 5
    model, diffusion = build_hip(cfg, sd, resp="ddim20", precision=precision)
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=21), "action": synth.make_actions(cfg, B, seed=22)}
    tape = synth.make_noise_tape(cfg, B, 20, seed=23)
    ty = {k: torch.from_numpy(v) for k, v in y.items()}
    ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", "ddim20"), tape, ty, mode="ddim").numpy()
    out = diffusion.ddim_sample_loop(model, (B, 56, 6, 37), clip_denoised=False, model_kwargs={"y": y_to_device(y)},
                                     noise_tape=torch.from_numpy(tape))
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-3


def test_philox_stream_properties():
    """On-device N(0,1): moments, world-size invariance (sample_offset), determinism, x_T != per-step stream."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu", layers=1)
    sd = synth.make_state_dict(cfg, seed=0)
    model, diffusion = build_hip(cfg, sd, resp="ddim5")
    eng, dev = model._get_engine(8)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.empty(8, 56, 6, 60, device="cuda")
    eng.randn(a, 8, 1234, 0, st)
    b = torch.empty(4, 56, 6, 60, device="cuda")
    eng.randn(b, 4, 1234, 4, st)
    assert torch.equal(a[4:], b)                       # shard [4,8) of an 8-batch == 4-batch at offset 4
    c = torch.empty_like(a)
    eng.randn(c, 8, 1235, 0, st)
    assert not torch.equal(a, c)
    n = a.numel()
    assert abs(a.mean().item()) < 4 / n ** 0.5 and abs(a.var().item() - 1) < 0.02
    assert abs((a ** 3).mean().item()) < 0.05 and abs((a ** 4).mean().item() - 3) < 0.1
    # sampling with Philox: same seed -> same samples; sharded == unsharded
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, 8)).cuda()}
    s1 = diffusion.p_sample_loop(model, (8, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, seed=77)
    s2 = diffusion.p_sample_loop(model, (8, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, seed=77)
    assert torch.equal(s1, s2)
    y2 = {"cmotion": y["cmotion"][4:].contiguous()}
    s3 = diffusion.p_sample_loop(model, (4, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y2}, seed=77, sample_offset=4)
    assert torch.allclose(s1[4:], s3, atol=1e-5)
    assert torch.isfinite(s1).all()


def test_postproc_rows(golden):
    g = golden("postproc")
    from regennet_amd import synth
    cfg = synth.get_config("tiny")
    model, _ = build_hip(cfg, synth.make_state_dict(cfg, seed=0))
    eng, _ = model._get_engine(1)
    st = torch.cuda.current_stream().cuda_stream
    d6 = torch.from_numpy(g["d6"]).cuda().contiguous()
    mat = torch.empty(d6.shape[:-1] + (3, 3), device="cuda")
    eng.rot6d_to_matrix(d6, mat, d6.numel() // 6, st)
    assert np.abs(mat.cpu().numpy() - g["mats"]).max() < 1e-6
    for xk, gk in (("x", "gf"), ("x3", "gf3")):
        x = torch.from_numpy(g[xk]).cuda().contiguous()
        out = torch.empty_like(x)
        eng.gaussian_filter1d(x, out, x.numel() // x.shape[-1], x.shape[-1], 1.0, st)
        assert np.abs(out.cpu().numpy() - g[gk]).max() < 1e-6


def test_boundary_protocol_and_errors():
    """Model/diffusion object protocol of SURVEY.md §8b and error behaviour of the C-ABI."""
    from regennet_amd import _lib, synth
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from regennet_amd.utils.model_util import load_model_wo_clip
    cfg = synth.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=0)
    model, diffusion = build_hip(cfg, sd, resp="ddim5")
    assert next(model.parameters()).device.type == "cuda"
    assert (model.njoints, model.nfeats, model.data_rep, model.cond_mode) == (5, 6, "rot6d", "action")
    assert set(model.state_dict().keys()) == set(sd.keys())
    with pytest.raises(AssertionError):                       # unexpected key (model_util.py:7)
        load_model_wo_clip(model, {**{k: torch.from_numpy(v) for k, v in sd.items()}, "bogus.weight": torch.zeros(1)})
    load_model_wo_clip(model, {**{k: torch.from_numpy(v) for k, v in sd.items()}, })
    model.to("cuda:0")
    B = 2
    y = y_to_device({"cmotion": synth.make_cmotion(cfg, B), "action": synth.make_actions(cfg, B)})
    with pytest.raises(AssertionError):                       # shape mismatch
        diffusion.p_sample_loop(model, (B, 5, 6, 9), clip_denoised=False, model_kwargs={"y": y})
    with pytest.raises(NotImplementedError):
        diffusion.ddim_sample_loop(model, (B, 5, 6, 8), model_kwargs={"y": y}, dump_steps=[1])
    with pytest.raises(KeyError):                             # guided needs y['scale'] (cfg_sampler.py:31)
        diffusion.p_sample_loop(ClassifierFreeSampleModel(model), (B, 5, 6, 8), clip_denoised=False, model_kwargs={"y": y})
    dumped = diffusion.p_sample_loop(model, (B, 5, 6, 8), clip_denoised=False, model_kwargs={"y": y}, dump_steps=[0, 4], seed=3)
    assert len(dumped) == 2 and dumped[0].shape == (B, 5, 6, 8)
    clipped = diffusion.p_sample_loop(model, (B, 5, 6, 8), clip_denoised=True, model_kwargs={"y": y}, seed=3)
    assert torch.isfinite(clipped).all()
    # y is never mutated
    assert set(y.keys()) == {"cmotion", "action"}
    # raw C-ABI error codes
    eng = _lib.Engine(cfg, 2, 0, "f32")
    with pytest.raises(_lib.RgnError) as e:
        eng.load_weight("not.a.key", np.zeros((1,), np.float32))
    assert e.value.code == -2
    with pytest.raises(_lib.RgnError) as e:
        eng.load_weight("fuse_process.bias", np.zeros((3,), np.float32))
    assert e.value.code == -3
    eng.load_weight("clip_model.anything", np.zeros((3,), np.float32))   # accepted and ignored
    with pytest.raises(_lib.RgnError) as e:
        eng.finalize()
    assert e.value.code == -4 and "missing keys" in str(e.value)
    eng.close()


def test_cgenerate_cli_synthetic(tmp_path):
    """The cgenerate-style entry point end to end (synthetic checkpoint + clips, ddim5, CFG)."""
    from regennet_amd.sample import cgenerate
    out = cgenerate.main(["--synthetic", "--num_samples", "3", "--num_repetitions", "2", "--timestep_respacing", "ddim5",
                          "--use_ddim", "--guidance_param", "2.5", "--output_dir", str(tmp_path)])
    res = np.load(out, allow_pickle=True).item()
    assert res["output"].shape == (6, 56, 6, 60) and res["cmotion"].shape == (6, 56, 6, 60)
    assert np.isfinite(res["output"]).all()


def test_weight_blob_view_transfers_a_checkpoint():
    """The multi-GPU start-up path: copying rank 0's packed blob into another engine (what the RCCL broadcast does)
    makes that engine reproduce rank 0's outputs bit for bit."""
    from regennet_amd import synth
    from regennet_amd.utils import dist_util
    cfg = synth.get_config("tiny")
    ma, da = build_hip(cfg, synth.make_state_dict(cfg, seed=0), resp="ddim5", precision="bf16x3")
    mb, db = build_hip(cfg, synth.make_state_dict(cfg, seed=123), resp="ddim5", precision="bf16x3")
    B = 2
    y = y_to_device({"cmotion": synth.make_cmotion(cfg, B), "action": synth.make_actions(cfg, B)})
    kw = dict(clip_denoised=False, model_kwargs={"y": y}, seed=5)
    ea, _ = ma._get_engine(B)
    eb, _ = mb._get_engine(B)
    out_a = da.ddim_sample_loop(ma, (B, 5, 6, 8), **kw)
    out_b0 = db.ddim_sample_loop(mb, (B, 5, 6, 8), **kw)
    assert not torch.allclose(out_a, out_b0)
    (pa, na), (pb, nb) = ea.weight_blob(), eb.weight_blob()
    assert na == nb and na > 0
    va, vb = dist_util.device_view(pa, na, "cuda:0"), dist_util.device_view(pb, nb, "cuda:0")
    assert va.dtype == torch.uint8 and va.numel() == na and va.data_ptr() == pa
    vb.copy_(va)
    torch.cuda.synchronize()
    eb.set_schedule(db.timestep_map, db._engine_tables(), db._sched_token)   # per-schedule tables depend on the weights
    mb._cond_key = None                                                      # ... and so does the hoisted condition
    out_b1 = db.ddim_sample_loop(mb, (B, 5, 6, 8), **kw)
    assert torch.equal(out_a, out_b1)


def test_chain_count_does_not_change_results(golden, monkeypatch):
    """1, 2 or 4 concurrent kernel chains (REGENNET_STREAMS) are a scheduling choice only: bit-identical samples."""
    g = golden("ntu_ddpm50")
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    outs = []
    for n in ("1", "2", "4"):
        monkeypatch.setenv("REGENNET_STREAMS", n)
        model, diffusion = build_hip(cfg, sd, resp="50", precision="bf16x3")
        outs.append(diffusion.p_sample_loop(model, (2, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y_to_device(y)},
                                            noise_tape=torch.from_numpy(tape)))
        model._engine.close()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert np.abs(outs[0].cpu().numpy() - g["final"]).max() < 1e-3


def test_fused_layernorm_gemm_variant(golden, monkeypatch):
    """The opt-in row-complete GEMM with both LayerNorms in its epilogue (REGENNET_FUSED_LN=1) meets the same bound."""
    monkeypatch.setenv("REGENNET_FUSED_LN", "1")
    for name in ("ntu_ddpm50", "ntu_action_ddim100_cfg"):
        g = golden(name)
        cfg, sd, y, tape = fixture_inputs(g, loop=True)
        model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision="bf16x3")
        fm = _wrap(model, bool(g["guided"]))
        fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
        out = fn(fm, (2, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
        assert np.abs(out.cpu().numpy() - g["final"]).max() < 1e-3
        model._engine.close()


def test_big_gemm_tiles_meet_the_same_bound(golden, monkeypatch):
    """Launches of >= 7000 rows per chain use the 256x256 GEMM tile; forced here on small goldens (edge tiles included)."""
    monkeypatch.setenv("REGENNET_BIG_TILE_ROWS", "1")
    for name in ("ntu_ddpm50", "ntu_action_ddim100_cfg", "text150_ddim50_cfg"):
        g = golden(name)
        cfg, sd, y, tape = fixture_inputs(g, loop=True)
        model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision="bf16x3")
        fm = _wrap(model, bool(g["guided"]))
        fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
        out = fn(fm, (int(g["B"]), 56, 6, cfg["num_frames"]), clip_denoised=False, model_kwargs={"y": y_to_device(y)},
                 noise_tape=torch.from_numpy(tape))
        assert np.abs(out.cpu().numpy() - g["final"]).max() < 1e-3
        model._engine.close()


def test_full_size_batch_is_row_independent():
    """BASELINE configs[1] size (B=256, NTU): every sample's chain is independent, so sample b of a 256-batch must equal
    the same sample drawn alone with the same Philox key (sample_offset=b) — a size-independent property that checks
    tiling, chain splitting and row mapping at the full bench size without needing a 256-sample reference run."""
    from regennet_amd import synth
    cfg = synth.get_config("ntu")
    sd = synth.make_state_dict(cfg, seed=0)
    model, diffusion = build_hip(cfg, sd, resp="ddim5", precision="bf16x3")
    B = 256
    cm = torch.from_numpy(synth.make_cmotion(cfg, B, seed=1)).cuda()
    full = diffusion.ddim_sample_loop(model, (B, 56, 6, 60), clip_denoised=False, model_kwargs={"y": {"cmotion": cm}}, seed=9)
    assert torch.isfinite(full).all()
    for b in (0, 63, 64, 129, 255):          # first/last rows of different chains
        one = diffusion.ddim_sample_loop(model, (1, 56, 6, 60), clip_denoised=False,
                                         model_kwargs={"y": {"cmotion": cm[b:b + 1].contiguous()}}, seed=9, sample_offset=b)
        assert torch.allclose(full[b:b + 1], one, atol=2e-5), (b, (full[b:b + 1] - one).abs().max().item())
    # guided, ragged batch that does not divide into the chains evenly
    cfg2 = synth.get_config("ntu_action")
    model2, diffusion2 = build_hip(cfg2, synth.make_state_dict(cfg2, seed=0), resp="ddim5", precision="bf16x3")
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    g2 = ClassifierFreeSampleModel(model2)
    B2 = 37
    y2 = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg2, B2, seed=3)).cuda(),
          "action": torch.from_numpy(synth.make_actions(cfg2, B2, seed=4)).cuda(), "scale": torch.full((B2,), 2.5, device="cuda")}
    full2 = diffusion2.ddim_sample_loop(g2, (B2, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y2}, seed=11)
    for b in (0, 18, 36):
        yb = {k: v[b:b + 1].contiguous() for k, v in y2.items()}
        one = diffusion2.ddim_sample_loop(g2, (1, 56, 6, 60), clip_denoised=False, model_kwargs={"y": yb}, seed=11, sample_offset=b)
        assert torch.allclose(full2[b:b + 1], one, atol=2e-5), (b, (full2[b:b + 1] - one).abs().max().item())


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["tiny_opts_clip", "tiny_opts_large_linear", "tiny_opts_skip_init", "tiny_opts_skip_zero",
                                  "tiny_opts_eta"])
def test_sampler_options(golden, name, precision):
    """Options of p_sample_loop / ddim_sample_loop beyond the cgenerate defaults, against the reference:
    clip_denoised=True, FIXED_LARGE variance + linear schedule, skip_timesteps (+ init_image), DDIM eta > 0."""
    from regennet_amd import synth
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    o = fixture_opts(g)
    model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision=precision,
                                 noise_schedule=o.get("noise_schedule", "cosine"), sigma_small=o.get("sigma_small", True))
    fm = _wrap(model, bool(g["guided"]))
    shape = (int(g["B"]), cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    kw = dict(clip_denoised=o.get("clip_denoised", False), model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape),
              skip_timesteps=o.get("skip_timesteps", 0))
    if o.get("init_image"):
        kw["init_image"] = torch.from_numpy(synth.make_noise_tape(cfg, int(g["B"]), 0, seed=12)[0] * 0.5).cuda()
    if str(g["mode"]) == "ddpm":
        out = diffusion.p_sample_loop(fm, shape, **kw)
    else:
        out = diffusion.ddim_sample_loop(fm, shape, eta=o.get("eta", 0.0), **kw)
    err = np.abs(out.cpu().numpy() - g["final"]).max()
    assert err < 1e-3, (name, err)


# ---- next-3 row: auto_regressive generation (eval/a2m/stgcn_eval.py:50-67) --------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["tiny_autoreg_ddpm10", "tiny_add_autoreg_ddpm20"])
def test_auto_regressive_matches_reference(golden, name, precision):
    """All T runs as one batch (and as several calls) against the reference's frame-by-frame loop on the same noise."""
    from regennet_amd.eval import sample_auto_regressive

    g = golden(name)
    cfg, sd, y, tapes = autoreg_inputs(g)
    model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision=precision)
    B, T = int(g["B"]), int(g["T"])
    shape = (B, cfg["njoints"], cfg["nfeats"], T)
    tapes_t = [torch.from_numpy(t) for t in tapes]
    for fpc in (T, 3, 1):
        out = sample_auto_regressive(diffusion.p_sample_loop, model, shape, {"y": y_to_device(y)}, frames_per_call=fpc,
                                     noise_tapes=tapes_t)
        err = np.abs(out.cpu().numpy() - g["output"]).max()
        assert err < 1e-3, (name, fpc, err)


def test_auto_regressive_grouping_invariance_with_device_rng():
    """Philox keyed by the global (frame, sample) index: the grouping of frames into sampler calls cannot change a bit."""
    from regennet_amd import synth
    from regennet_amd.eval import sample_auto_regressive

    cfg = synth.get_config("tiny")
    sd = synth.make_state_dict(cfg, seed=0)
    model, diffusion = build_hip(cfg, sd, resp="10", precision="bf16x3")
    B, T = 3, cfg["num_frames"]
    y = y_to_device({"cmotion": synth.make_cmotion(cfg, B, seed=1), "action": synth.make_actions(cfg, B, seed=2)})
    shape = (B, cfg["njoints"], cfg["nfeats"], T)
    ref = sample_auto_regressive(diffusion.p_sample_loop, model, shape, {"y": y}, frames_per_call=T, seed=11)
    for fpc in (1, 3):
        out = sample_auto_regressive(diffusion.p_sample_loop, model, shape, {"y": y}, frames_per_call=fpc, seed=11)
        assert torch.equal(out, ref), fpc
    other = sample_auto_regressive(diffusion.p_sample_loop, model, shape, {"y": y}, frames_per_call=T, seed=12)
    assert not torch.equal(other, ref)
