"""CPU: pin the oracle (oracle/regennet_oracle.py) against vectors recorded from the reference itself."""
import numpy as np
import pytest
import torch

from oracle import regennet_oracle as orc
from regennet_amd import synth
from tests.helpers import autoreg_inputs, fixture_inputs, fixture_opts

FWD = ["tiny_fwd", "tiny_fwd_cfg", "tiny_add_fwd", "tiny_etd_fwd", "tiny_wope_fwd", "tiny_text_fwd_cfg", "ntu_fwd",
       "ntu_action_fwd_cfg", "chi3d_fwd"]
LOOPS = ["tiny_ddpm10", "tiny_ddim10_cfg", "tiny_add_ddpm1000", "tiny_text_ddim20_cfg", "tiny_etd_ddim10_cfg",
         "tiny_wope_ddpm10", "ntu_ddpm50", "ntu_add_etd_ddpm20", "chi3d_ddpm20", "chi3d_ddim20_cfg",
         "ntu_eval_ddim5", "ntu_eval_5", "ntu_action_eval_ddim5"]   # (the last three: the reference's shipped evaluation setting, README.md:134-137)


def _ty(y):
    return {k: torch.from_numpy(v) for k, v in y.items()}


@pytest.mark.parametrize("name", FWD)
def test_denoiser_forward_matches_reference(golden, name):
    g = golden(name)
    cfg, sd, y, x = fixture_inputs(g, loop=False)
    f = orc.cfg_forward if bool(g["guided"]) else orc.cmdm_forward
    with torch.no_grad():
        for i, t in enumerate(g["ts"]):
            out = f(sd, cfg, torch.from_numpy(x), torch.tensor([int(t)] * x.shape[0]), _ty(y)).numpy()
            np.testing.assert_allclose(out, g["out"][i], atol=2e-5, rtol=0)


@pytest.mark.parametrize("name", LOOPS)
def test_sampling_loop_matches_reference(golden, name):
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    sched = orc.make_schedule("cosine", str(g["resp"]))
    # bit-exact timestep indices handed to the denoiser (respace.py:124-129)
    assert np.array_equal(np.array(sched[0])[::-1], g["model_t"][:: (2 if bool(g["guided"]) else 1)])
    trace = {}
    out = orc.sample_loop(sd, cfg, sched, tape, _ty(y), mode=str(g["mode"]), guided=bool(g["guided"]), trace=trace)
    np.testing.assert_allclose(out.numpy(), g["final"], atol=1e-4, rtol=0)
    if "x0" in g:
        np.testing.assert_allclose(torch.stack(trace["x0"]).numpy(), g["x0"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(torch.stack(trace["x"]).numpy(), g["x"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("name", ["tiny_autoreg_ddpm10", "tiny_add_autoreg_ddpm20"])
def test_auto_regressive_generation_matches_reference(golden, name):
    """next-3 row: eval/a2m/stgcn_eval.py:50-67 run around the reference's own sampler (make_golden.gen_autoreg)."""
    g = golden(name)
    cfg, sd, y, tapes = autoreg_inputs(g)
    sched = orc.make_schedule("cosine", str(g["resp"]))
    out = orc.sample_auto_regressive(sd, cfg, sched, tapes, _ty(y)).numpy()
    np.testing.assert_allclose(out, g["output"], atol=1e-4, rtol=0)
    F = cfg["nfeats"]
    assert np.array_equal(out[:, :, :F], y["cmotion"])                       # actor rows = the (fully revealed) actor motion
    # the last run saw the whole actor motion: its last frame is the last column of the reactor rows
    np.testing.assert_allclose(out[:, :, F:, -1], g["last_run"][:, :, :, -1], atol=1e-4, rtol=0)


def test_schedule_tables_and_respacing_match_reference(golden):
    g = golden("schedules")
    for key in g.files:
        kind, _, tag = key.partition("__")
        if kind == "space":
            resp = tag.replace("_", ",") if tag and tag[0].isdigit() else tag
            n = 300 if resp == "10,15,20" else 1000
            assert sorted(orc.space_timesteps(n, resp or [n])) == g[key].tolist(), key
        elif kind == "map":
            sched, _, resp = tag.partition("__")
            tmap, _ = orc.make_schedule(sched, resp)
            assert tmap == g[key].tolist()
        else:
            sched, _, resp = tag.partition("__")
            _, tb = orc.make_schedule(sched, resp)
            assert np.array_equal(tb[kind], g[key]), key   # fp64 bit-exact


def test_schedule_known_answers():
    """KATs recorded in SURVEY.md §8a (a1/a2/a3)."""
    b = orc.get_named_beta_schedule("cosine", 1000)
    assert b[0] == pytest.approx(4.12842248e-05, rel=1e-8) and b[500] == pytest.approx(3.15569144e-03, rel=1e-8)
    assert b[998] == pytest.approx(7.49999393e-01, rel=1e-8) and b[999] == 0.999
    tb = orc.diffusion_tables(b)
    assert tb["alphas_cumprod"][500] == pytest.approx(4.92285172e-01, rel=1e-8)
    assert tb["alphas_cumprod"][999] == pytest.approx(2.42876691e-09, rel=1e-7)
    assert tb["posterior_variance"][0] == 0 and tb["posterior_variance"][1] == pytest.approx(2.17894961e-05, rel=1e-8)
    assert tb["posterior_log_variance_clipped"][0] == tb["posterior_log_variance_clipped"][1] == pytest.approx(-10.73408253, rel=1e-8)
    assert tb["posterior_mean_coef1"][0] == 1 and tb["posterior_mean_coef2"][0] == 0
    assert sorted(orc.space_timesteps(1000, "ddim100")) == list(range(0, 1000, 10))
    assert sorted(orc.space_timesteps(1000, "ddim5")) == [0, 200, 400, 600, 800]
    s100 = sorted(orc.space_timesteps(1000, "100"))
    assert s100[:2] == [0, 10] and s100[-3:] == [979, 989, 999]
    with pytest.raises(ValueError):
        orc.space_timesteps(1000, "250,250,500")     # respace.py:47
    with pytest.raises(ValueError):
        orc.space_timesteps(1000, "ddim999")          # respace.py:36
    with pytest.raises(NotImplementedError):
        orc.get_named_beta_schedule("sqrt", 10)      # gaussian_diffusion.py:45


def test_postproc_rows_match_reference(golden):
    g = golden("postproc")
    m = orc.rotation_6d_to_matrix(torch.from_numpy(g["d6"])).numpy()
    np.testing.assert_allclose(m, g["mats"], atol=1e-6)
    np.testing.assert_allclose(orc.gaussian_filter1d_lastaxis(g["x"]), g["gf"], atol=1e-6)
    np.testing.assert_allclose(orc.gaussian_filter1d_lastaxis(g["x3"]), g["gf3"], atol=1e-6)


OPTS = ["tiny_opts_clip", "tiny_opts_large_linear", "tiny_opts_skip_init", "tiny_opts_skip_zero", "tiny_opts_eta"]


@pytest.mark.parametrize("name", OPTS)
def test_sampler_options_match_reference(golden, name):
    """clip_denoised, FIXED_LARGE + linear schedule, skip_timesteps (+init_image), DDIM eta > 0."""
    g = golden(name)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    o = fixture_opts(g)
    sched = orc.make_schedule(o.get("noise_schedule", "cosine"), str(g["resp"]), sigma_small=o.get("sigma_small", True))
    init = synth.make_noise_tape(cfg, int(g["B"]), 0, seed=12)[0] * 0.5 if o.get("init_image") else None
    out = orc.sample_loop(sd, cfg, sched, tape, _ty(y), mode=str(g["mode"]), guided=bool(g["guided"]), eta=o.get("eta", 0.0),
                          clip_denoised=o.get("clip_denoised", False), skip_timesteps=o.get("skip_timesteps", 0),
                          init_image=None if init is None else torch.from_numpy(init))
    np.testing.assert_allclose(out.numpy(), g["final"], atol=1e-4, rtol=0)


# ---- next-4 row: ST-GCN evaluator, diversity / multimodality, accuracy ------------------------------------------------------
def _stgcn_sd(g):
    sd = synth.make_stgcn_state_dict(g["A"], num_class=26, seed=0)
    from tests.helpers import sd_digest
    assert sd_digest(sd) == str(g["sd_digest"]), "synthetic ST-GCN checkpoint drifted from the golden fixture"
    return sd


@pytest.mark.parametrize("tag", ["ntu", "chi3d", "one"])
def test_stgcn_forward_matches_reference(golden, tag):
    from oracle import stgcn_oracle as so
    g = golden("stgcn")
    feats, yhat = so.stgcn_forward(_stgcn_sd(g), g[f"x_{tag}"])
    np.testing.assert_allclose(feats.numpy(), g[f"features_{tag}"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(yhat.numpy(), g[f"yhat_{tag}"], atol=2e-5, rtol=1e-5)


def test_diversity_multimodality_accuracy_match_reference(golden):
    from oracle import stgcn_oracle as so
    g = golden("stgcn")
    div, mm = so.calculate_diversity_multimodality(g["div_act"], g["div_labels"], 26, seed=123)
    assert abs(div - float(g["diversity"])) < 1e-5 and abs(mm - float(g["multimodality"])) < 1e-5
    yh, ys = g["acc_yhat"], g["acc_y"]
    acc, conf = so.calculate_accuracy([yh[i:i + 16] for i in range(0, 64, 16)], [ys[i:i + 16] for i in range(0, 64, 16)], 26)
    assert abs(acc - float(g["accuracy"])) < 1e-7 and np.array_equal(conf.numpy(), g["confusion"])
