"""CPU: host-side mirror of the reference interface — schedule tables, respacing, factory, flags.
Bit-exact integer work (timestep sets/maps) and fp64 tables are checked against the golden vectors
recorded from the reference."""
import types

import numpy as np
import pytest
import torch

from regennet_amd.diffusion import gaussian_diffusion as gd
from regennet_amd.diffusion.respace import SpacedDiffusion, _WrappedModel, space_timesteps
from regennet_amd.utils import model_util
from regennet_amd.utils.parser_util import cgenerate_args


def _diff(sched, resp):
    return SpacedDiffusion(use_timesteps=space_timesteps(1000, resp or [1000]), betas=gd.get_named_beta_schedule(sched, 1000, 1.0),
                           model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                           loss_type=gd.LossType.MSE)


def test_tables_maps_and_spacings_bit_exact_vs_reference(golden):
    g = golden("schedules")
    cache = {}
    for key in g.files:
        kind, _, tag = key.partition("__")
        if kind == "space":
            resp = tag.replace("_", ",") if tag and tag[0].isdigit() else tag
            n = 300 if resp == "10,15,20" else 1000
            assert sorted(space_timesteps(n, resp or [n])) == g[key].tolist(), key
            continue
        sched, _, resp = tag.partition("__")
        d = cache.setdefault(tag, _diff(sched, resp))
        if kind == "map":
            assert d.timestep_map == g[key].tolist()
        else:
            assert np.array_equal(np.asarray(getattr(d, kind)), g[key]), key


def test_error_behaviour_matches_reference():
    with pytest.raises(ValueError):
        space_timesteps(1000, "250,250,500")           # respace.py:47
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")               # respace.py:36
    with pytest.raises(NotImplementedError):
        gd.get_named_beta_schedule("sqrt", 10)         # gaussian_diffusion.py:45
    with pytest.raises(AssertionError):
        gd.GaussianDiffusion(betas=np.array([0.5, 1.5]), model_mean_type=gd.ModelMeanType.START_X,
                             model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE)


def test_wrapped_model_maps_timesteps_bit_exact():
    d = _diff("cosine", "ddim100")
    seen = {}

    def model(x, ts, **kw):
        seen["ts"] = ts
        return x

    wm = d._wrap_model(model)
    assert isinstance(wm, _WrappedModel) and d._wrap_model(wm) is wm
    t = torch.tensor([99, 0, 57])
    wm(torch.zeros(3), t)
    assert seen["ts"].dtype == torch.int64 and seen["ts"].tolist() == [990, 0, 570]
    assert d._model_timesteps(t).tolist() == [990, 0, 570]


def _args(**over):
    a = cgenerate_args(["--unconstrained"])
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_factory_mirrors_reference_contract():
    data = types.SimpleNamespace(dataset=types.SimpleNamespace(num_actions=26, num_person=2))
    args = _args()
    assert args.num_person == 2
    model, diffusion = model_util.create_model_and_diffusion(args, data)
    assert args.num_person == 1                                   # side effect model_util.py:15
    assert (model.njoints, model.nfeats, model.num_frames, model.latent_dim, model.ff_size, model.num_heads) == (56, 6, 60, 512, 1024, 4)
    assert model.cond_mode == "no_cond" and model.arch == "online" and model.cm_mode == "concat"
    assert diffusion.num_timesteps == 1000 and diffusion.timestep_map == list(range(1000))
    n = sum(p.numel() for p in model.parameters())
    assert n == 26_803_024                                         # parameter count of the shipped config
    keys = set(model.state_dict().keys())
    assert "sequence_pos_encoder.pe" in keys and "embed_timestep.sequence_pos_encoder.pe" in keys and len(keys) == 158
    # action model on chi3d with guidance flags
    a2 = _args(unconstrained=False, dataset="chi3d", timestep_respacing="ddim5")
    data2 = types.SimpleNamespace(dataset=types.SimpleNamespace(num_actions=8, num_person=2))
    m2, d2 = model_util.create_model_and_diffusion(a2, data2)
    assert m2.cond_mode == "action" and m2.num_frames == 150 and m2.embed_action.action_embedding.shape == (8, 512)
    assert d2.timestep_map == [0, 200, 400, 600, 800]
    # FIXED_LARGE when sigma_small is falsy
    d3 = model_util.create_gaussian_diffusion(_args(sigma_small=False))
    assert d3.model_var_type == gd.ModelVarType.FIXED_LARGE
    with pytest.raises(NotImplementedError):
        model_util.create_model_and_diffusion(_args(arch="trans_enc"), data)


def test_checkpoint_loader_rules():
    data = types.SimpleNamespace(dataset=types.SimpleNamespace(num_actions=1, num_person=2))
    model, _ = model_util.create_model_and_diffusion(_args(layers=1, latent_dim=64), data)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model_util.load_model_wo_clip(model, sd)
    with pytest.raises(AssertionError):
        model_util.load_model_wo_clip(model, {**sd, "extra.weight": torch.zeros(1)})
    short = dict(sd)
    short.pop("fuse_process.bias")
    with pytest.raises(AssertionError):
        model_util.load_model_wo_clip(model, short)                # missing non-clip key (model_util.py:8)


def test_flag_quirks():
    assert cgenerate_args(["--cond_mask_prob", "0"]).guidance_param == 1       # parser_util.py:68-69
    assert cgenerate_args(["--sigma_small", "False"]).sigma_small is True      # type=bool quirk of the reference
    a = cgenerate_args([])
    assert (a.seed, a.batch_size, a.noise_schedule, a.latent_dim, a.layers, a.cm_mode) == (10, 64, "cosine", 512, 8, "concat")


# ---- next-3 row: batched auto_regressive generation (host logic; the sampler is faked) -------------------------------
def test_auto_regressive_batching_masks_orders_and_gathers():
    import torch

    from regennet_amd.eval import sample_auto_regressive

    B, V, C, T = 3, 2, 4, 5
    gen = torch.Generator().manual_seed(0)
    cm = torch.randn(B, V, C, T, generator=gen)
    y = {"cmotion": cm, "action": torch.arange(B).reshape(B, 1), "lengths": torch.full((B,), T), "action_text": ["a", "b", "c"]}
    calls = []

    def fake_sample_fn(model, shape, clip_denoised=True, model_kwargs=None, seed=None, sample_offset=0, noise_tape=None):
        yy = model_kwargs["y"]
        n = shape[0] // B
        calls.append((shape, sample_offset, seed))
        assert yy["cmotion"].shape == shape and yy["action"].shape == (n * B, 1) and len(yy["action_text"]) == n * B
        assert clip_denoised is False
        # sample (f, b) = masked actor motion + 1000 * (global sample index): lets the caller's gather be checked exactly
        gidx = sample_offset + torch.arange(n * B).reshape(-1, 1, 1, 1)
        assert torch.equal(yy["action"][:, 0], torch.arange(B).repeat(n))
        return yy["cmotion"] + 1000.0 * gidx

    for fpc in (1, 2, 5, None):
        calls.clear()
        out = sample_auto_regressive(fake_sample_fn, None, (B, V, C, T), {"y": y}, frames_per_call=fpc, seed=7)
        assert out.shape == (B, V, 2 * C, T)
        assert torch.equal(out[:, :, :C], cm)
        for f in range(T):
            for b in range(B):
                # frame f of run f: the actor's frame f is revealed in that run, offset identifies (f, b)
                assert torch.equal(out[b, :, C:, f], cm[b, :, :, f] + 1000.0 * (f * B + b))
        assert [c[1] for c in calls] == list(range(0, T * B, (fpc or T) * B))
        assert all(c[2] == 7 for c in calls)
    # masking: run f must not see later actor frames
    seen = []

    def spy(model, shape, clip_denoised=True, model_kwargs=None, **kw):
        seen.append(model_kwargs["y"]["cmotion"].clone())
        return torch.zeros(shape)

    out = sample_auto_regressive(spy, None, (B, V, C, T), {"y": y}, setting="other", frames_per_call=T, seed=1)
    assert out.shape == (B, V, C, T)
    m = seen[0].reshape(T, B, V, C, T)
    for f in range(T):
        assert torch.equal(m[f, :, :, :, : f + 1], cm[:, :, :, : f + 1]) and not m[f, :, :, :, f + 1:].any()


# ---- next-4 row: Frechet distance (torch, device-agnostic) vs the reference's numpy/scipy on seeded feature sets -----------
def _fid_sets(g, case):
    n1, n2, dim, shift, scale = g[f"cfg_{case}"]
    n1, n2, dim = int(n1), int(n2), int(dim)
    rng = np.random.Generator(np.random.PCG64(50 + case))
    mix = rng.standard_normal((dim, dim)) / np.sqrt(dim)
    a = (rng.standard_normal((n1, dim)) @ mix).astype(np.float32)
    b = (rng.standard_normal((n2, dim)) @ mix * scale + shift).astype(np.float32)
    return a, b


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_frechet_distance_matches_reference(golden, case):
    import torch

    from regennet_amd.eval import calculate_activation_statistics, calculate_fid

    g = golden("fid")
    a, b = _fid_sets(g, case)
    s1 = calculate_activation_statistics(torch.from_numpy(a))
    s2 = calculate_activation_statistics(torch.from_numpy(b))
    np.testing.assert_allclose(s2[0].numpy(), g[f"mu_{case}"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(torch.diagonal(s2[1]).numpy(), g[f"sigma_diag_{case}"], rtol=1e-6)
    ref = float(g[f"fid_{case}"])
    got = float(calculate_fid(s1, s2))
    assert abs(got - ref) <= 1e-6 * max(1.0, abs(ref)), (got, ref)
    assert abs(float(calculate_fid(s1, s1))) < 1e-6 and abs(float(g[f"fid_same_{case}"])) < 1e-3
