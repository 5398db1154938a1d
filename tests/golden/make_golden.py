"""Generate golden vectors by RUNNING THE REFERENCE (CPU) in the build container.

    python tests/golden/make_golden.py [--only NAME] [--skip-long]

Imports /root/reference through tests/golden/_ref_import.py (stub clip/timm/smplx), loads the
build's synthetic checkpoint (regennet_amd.synth, reference key names -> proves key/shape
compatibility), injects a recorded noise tape in the reference's own draw order and records
outputs of the reference's `CMDM.forward`, `ClassifierFreeSampleModel.forward`,
`SpacedDiffusion.p_sample_loop` / `ddim_sample_loop`, `space_timesteps` and the schedule tables.

Only DATA is written (tests/golden/*.npz): inputs are re-derivable from the recorded seeds via
regennet_amd.synth (a digest of every input is stored to detect drift). Nothing from the
reference's source travels.
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402
from regennet_amd import synth  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def sd_digest(sd):
    return digest(*[sd[k] for k in sorted(sd)])


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_schedules():
    _ref_import.install()
    from diffusion import gaussian_diffusion as gd
    from diffusion.respace import SpacedDiffusion, space_timesteps
    out = {}
    for resp in ["", "ddim100", "100", "ddim50", "50", "ddim5", "ddim20", "20", "10", "250,250,300", "10,15,20"]:
        n = 300 if resp == "10,15,20" else 1000
        s = space_timesteps(n, resp or [n])
        out["space__" + resp.replace(",", "_")] = np.array(sorted(s), dtype=np.int64)
    for sched in ["cosine", "linear"]:
        betas = gd.get_named_beta_schedule(sched, 1000, 1.0)
        for resp in ["", "ddim100", "50"]:
            d = SpacedDiffusion(use_timesteps=space_timesteps(1000, resp or [1000]), betas=betas,
                                model_mean_type=gd.ModelMeanType.START_X,
                                model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE)
            tag = f"{sched}__{resp}"
            out[f"map__{tag}"] = np.array(d.timestep_map, dtype=np.int64)
            for k in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                      "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                      "posterior_mean_coef1", "posterior_mean_coef2"]:
                out[f"{k}__{tag}"] = np.asarray(getattr(d, k), dtype=np.float64)
    save("schedules", **out)


def make_y(cfg, B, guided, scale=2.5):
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1))}
    if "action" in cfg["cond_mode"]:
        y["action"] = torch.from_numpy(synth.make_actions(cfg, B, seed=2))
    if "text" in cfg["cond_mode"]:
        y["text"] = ["synthetic"] * B
        y["text_features"] = torch.from_numpy(synth.make_text_features(cfg, B, seed=3))
    if guided:
        y["scale"] = torch.ones(B) * scale
    return y


def build(cfg, resp, sd):
    model, diffusion = _ref_import.build_reference(cfg, sd, resp)
    if "text" in cfg["cond_mode"]:
        # CLIP is out of scope: the text encoder output is an INPUT
        holder = {}
        model.encode_text = lambda raw_text: holder["feat"]
        model._feat_holder = holder
    return model, diffusion


def gen_forward(name, cfg_name, B, ts, guided=False, **over):
    cfg = synth.get_config(cfg_name, **over)
    sd = synth.make_state_dict(cfg, seed=0)
    model, _ = build(cfg, "", sd)
    y = make_y(cfg, B, guided)
    if "text" in cfg["cond_mode"]:
        model._feat_holder["feat"] = y["text_features"]
    x = torch.from_numpy(synth.make_noise_tape(cfg, B, 0, seed=11)[0])
    if guided:
        from model.cfg_sampler import ClassifierFreeSampleModel
        fmodel = ClassifierFreeSampleModel(model)
    else:
        fmodel = model
    outs = []
    with torch.no_grad():
        for t in ts:
            outs.append(fmodel(x, torch.tensor([t] * B), y=y).numpy())
    save(name, cfg_name=cfg_name, over=repr(over), B=B, ts=np.array(ts), guided=guided,
         out=np.stack(outs), sd_digest=sd_digest(sd), in_digest=digest(x.numpy(), y["cmotion"].numpy()))


def gen_loop(name, cfg_name, B, resp, mode, guided=False, keep_trace=False, opts=None, **over):
    opts = dict(opts or {})
    cfg = synth.get_config(cfg_name, **over)
    sd = synth.make_state_dict(cfg, seed=0)
    cfg_d = dict(cfg, noise_schedule=opts.get("noise_schedule", "cosine"), sigma_small=opts.get("sigma_small", True))
    model, diffusion = build(cfg_d, resp, sd)
    S = diffusion.num_timesteps
    y = make_y(cfg, B, guided)
    if "text" in cfg["cond_mode"]:
        model._feat_holder["feat"] = y["text_features"]
    tape = synth.make_noise_tape(cfg, B, S, seed=10)
    if guided:
        from model.cfg_sampler import ClassifierFreeSampleModel
        fmodel = ClassifierFreeSampleModel(model)
    else:
        fmodel = model
    shape = (B, cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    fn = diffusion.p_sample_loop_progressive if mode == "ddpm" else diffusion.ddim_sample_loop_progressive
    x0s, xs, seen_t = [], [], []
    # record the timestep indices the model is actually called with (bit-exact requirement)
    orig_forward = model.forward

    def spy(x, timesteps, y=None):
        seen_t.append(timesteps.clone().numpy())
        return orig_forward(x, timesteps, y)

    model.forward = spy
    t0 = time.time()
    skip = int(opts.get("skip_timesteps", 0))
    call_kw = dict(clip_denoised=bool(opts.get("clip_denoised", False)), model_kwargs={"y": y}, skip_timesteps=skip)
    if opts.get("init_image", False):
        call_kw["init_image"] = torch.from_numpy(synth.make_noise_tape(cfg, B, 0, seed=12)[0] * 0.5)
    if mode == "ddim" and "eta" in opts:
        call_kw["eta"] = float(opts["eta"])
    with _ref_import.NoiseTape(tape) as nt:
        for out in fn(fmodel, shape, **call_kw):
            if keep_trace:
                x0s.append(out["pred_xstart"].numpy().copy())
                xs.append(out["sample"].numpy().copy())
            final = out["sample"]
        assert nt.pos == S + 1 - skip, (nt.pos, S, skip)
    dt = time.time() - t0
    print(f"{name}: reference {mode} S={S} B={B} took {dt:.1f}s")
    kw = dict(cfg_name=cfg_name, over=repr(over), opts=repr(opts), B=B, resp=resp, mode=mode, guided=guided, S=S,
              final=final.numpy(), model_t=np.stack(seen_t)[:, 0], ref_seconds=dt,
              sd_digest=sd_digest(sd), in_digest=digest(tape[0], tape[-1], y["cmotion"].numpy()))
    if keep_trace:
        kw["x0"] = np.stack(x0s)
        kw["x"] = np.stack(xs)
    save(name, **kw)


def gen_autoreg(name, cfg_name, B, resp, **over):
    """next-3 row: the auto_regressive generation of eval/a2m/stgcn_eval.py:50-67 (setting 'cmdm').

    The frame loop below re-states those lines around the REFERENCE's own model and p_sample_loop (NewDataloader itself
    needs the SMPL-X assets for rot2xyz, which the container lacks): for every frame index the actor's frames up to it are
    revealed, a full sampler run is made with fresh noise (run f pops tape f) and only that frame of the result is kept.
    """
    cfg = synth.get_config(cfg_name, **over)
    sd = synth.make_state_dict(cfg, seed=0)
    model, diffusion = build(dict(cfg, noise_schedule="cosine", sigma_small=True), resp, sd)
    S = diffusion.num_timesteps
    y = make_y(cfg, B, False)
    T = cfg["num_frames"]
    shape = (B, cfg["njoints"], cfg["nfeats"], T)
    cmotion_bak = y["cmotion"]
    cmotion = torch.zeros_like(cmotion_bak)
    output = torch.zeros((B, cfg["njoints"], cfg["nfeats"] * 2, T))
    samples = []
    t0 = time.time()
    for f in range(T):
        cmotion[:, :, :, f] = cmotion_bak[:, :, :, f]
        y["cmotion"] = cmotion
        tape = synth.make_noise_tape(cfg, B, S, seed=100 + f)
        with _ref_import.NoiseTape(tape) as nt:
            sample = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y})
            assert nt.pos == S + 1
        tmp = torch.cat((y["cmotion"], sample), axis=2)
        output[:, :, :, f] = tmp[:, :, :, f]
        samples.append(sample.numpy().copy())
    dt = time.time() - t0
    print(f"{name}: reference auto_regressive T={T} S={S} B={B} took {dt:.1f}s")
    save(name, cfg_name=cfg_name, over=repr(over), B=B, resp=resp, S=S, T=T, guided=False, output=output.numpy(),
         last_run=samples[-1], ref_seconds=dt, sd_digest=sd_digest(sd),
         in_digest=digest(synth.make_noise_tape(cfg, B, S, seed=100)[0], cmotion_bak.numpy()))


def gen_post():
    """next-1/next-2 rows: rotation_6d_to_matrix and the cgenerate.py:142 smoothing."""
    _ref_import.install()
    from scipy.ndimage import gaussian_filter1d
    import utils.rotation_conversions as rc
    rng = np.random.Generator(np.random.PCG64(5))
    d6 = rng.standard_normal((4, 7, 6)).astype(np.float32)
    mats = rc.rotation_6d_to_matrix(torch.from_numpy(d6)).numpy()
    x = rng.standard_normal((2, 5, 6, 13)).astype(np.float32)
    gf = gaussian_filter1d(x, sigma=1, axis=-1)
    x3 = rng.standard_normal((1, 2, 6, 3)).astype(np.float32)   # T smaller than the 4-tap radius
    gf3 = gaussian_filter1d(x3, sigma=1, axis=-1)
    save("postproc", d6=d6, mats=mats, x=x, gf=gf, x3=x3, gf3=gf3)


def gen_fid():
    """next-4 row: calculate_activation_statistics (eval/a2m/stgcn/evaluate.py:48-53) + calculate_frechet_distance
    (eval/a2m/stgcn/fid.py:11-61) of the reference on seeded synthetic feature sets (only seeds and results are stored)."""
    _ref_import.install()
    from eval.a2m.stgcn.fid import calculate_fid
    out = {}
    for case, (n1, n2, dim, shift, scale) in enumerate([(400, 300, 32, 0.0, 1.0), (500, 500, 64, 0.3, 1.5), (256, 256, 16, 2.0, 0.5),
                                                        (40, 60, 48, 0.1, 1.0)]):   # last: rank-deficient covariances
        rng = np.random.Generator(np.random.PCG64(50 + case))
        mix = rng.standard_normal((dim, dim)) / np.sqrt(dim)
        a = (rng.standard_normal((n1, dim)) @ mix).astype(np.float32)
        b = (rng.standard_normal((n2, dim)) @ mix * scale + shift).astype(np.float32)
        stats = [(np.mean(x, axis=0), np.cov(x, rowvar=False)) for x in (a, b)]          # evaluate.py:50-52
        out[f"fid_{case}"] = np.float64(calculate_fid(stats[0], stats[1]))
        out[f"fid_same_{case}"] = np.float64(calculate_fid(stats[0], stats[0]))
        out[f"cfg_{case}"] = np.array([n1, n2, dim, shift, scale], dtype=np.float64)
        out[f"mu_{case}"] = stats[1][0]
        out[f"sigma_diag_{case}"] = np.diag(stats[1][1])
    save("fid", **out)


# SMPL-X kinematic tree (parent of joint i; 55 joints), the public skeleton layout the reference reads from the licensed
# SMPLX_NEUTRAL.npz (eval/a2m/recognition/models/stgcnutils/graph.py:81-88). Only used to let the reference build its graph
# here; the product takes the adjacency from the checkpoint's `A` buffer.
SMPLX_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15, 20, 25, 26, 20, 28, 29, 20, 31,
                 32, 20, 34, 35, 20, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]


def build_ref_stgcn(num_class):
    _ref_import.install()
    import tempfile
    import utils.config as rc
    d = tempfile.mkdtemp()
    kt = np.stack([np.array(SMPLX_PARENTS, dtype=np.int64) % (2 ** 32), np.arange(55, dtype=np.int64)])
    np.savez(os.path.join(d, "SMPLX_NEUTRAL.npz"), kintree_table=kt)
    rc.SMPLX_KINTREE_PATH = os.path.join(d, "SMPLX_NEUTRAL.npz")
    sys.modules.pop("eval.a2m.recognition.models.stgcnutils.graph", None)
    from eval.a2m.recognition.models.stgcn import STGCN
    return STGCN(in_channels=12, num_class=num_class, num_person=2, graph_args={"layout": "smplx", "strategy": "spatial"},
                 edge_importance_weighting=True, device="cpu")


def gen_stgcn():
    """next-4 row: the ST-GCN feature extractor / classifier (eval/a2m/recognition/models/stgcn.py:76-123) on synthetic
    weights (regennet_amd.synth.make_stgcn_state_dict, reference key names) + diversity / multimodality
    (eval/a2m/stgcn/diversity.py:6) and accuracy (accuracy.py:4) on seeded activations."""
    num_class = 26
    model = build_ref_stgcn(num_class)
    A = model.A.numpy().copy()
    sd = synth.make_stgcn_state_dict(A, num_class=num_class, seed=0)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    out = {"A": A, "sd_digest": sd_digest(sd)}
    for tag, (N, T) in {"ntu": (5, 60), "chi3d": (2, 150), "one": (1, 60)}.items():
        rng = np.random.Generator(np.random.PCG64(70 + T + N))
        x = rng.standard_normal((N, 56, 12, T)).astype(np.float32)
        with torch.no_grad():
            b = model({"output": torch.from_numpy(x)})
        out[f"x_{tag}"] = x
        out[f"features_{tag}"] = b["features"].numpy().reshape(N, -1)
        out[f"yhat_{tag}"] = b["yhat"].numpy()
    from eval.a2m.stgcn.diversity import calculate_diversity_multimodality
    from eval.a2m.stgcn.accuracy import calculate_accuracy
    rng = np.random.Generator(np.random.PCG64(90))
    act = rng.standard_normal((300, 256)).astype(np.float32)
    labels = rng.integers(0, num_class, 300).astype(np.int64)
    labels[labels == 7] = 8                                  # one class absent: its multimodality quota stays zero
    div, mm = calculate_diversity_multimodality(torch.from_numpy(act), torch.from_numpy(labels), num_class, seed=123)
    out.update(div_act=act, div_labels=labels, diversity=np.float64(div), multimodality=np.float64(mm))

    class _Cls:      # stand-in classifier: yhat supplied per batch (accuracy.py only reads classifier(batch)["yhat"])
        def __call__(self, batch):
            return {"yhat": batch["yhat_in"]}
    yh = rng.standard_normal((64, num_class)).astype(np.float32)
    ys = rng.integers(0, num_class, 64).astype(np.int64)
    loader = [{"yhat_in": torch.from_numpy(yh[i:i + 16]), "y": torch.from_numpy(ys[i:i + 16])} for i in range(0, 64, 16)]
    acc, conf = calculate_accuracy(None, loader, num_class, _Cls(), "cpu")
    out.update(acc_yhat=yh, acc_y=ys, accuracy=np.float64(acc), confusion=conf.numpy())
    save("stgcn", **out)


JOBS = {
    "schedules": gen_schedules,
    "postproc": gen_post,
    "fid": gen_fid,
    "stgcn": gen_stgcn,
    "tiny_fwd": lambda: gen_forward("tiny_fwd", "tiny", 3, [0, 1, 500, 999]),
    "tiny_fwd_cfg": lambda: gen_forward("tiny_fwd_cfg", "tiny", 3, [0, 700], guided=True),
    "tiny_add_fwd": lambda: gen_forward("tiny_add_fwd", "tiny_add", 2, [3, 999]),
    "tiny_etd_fwd": lambda: gen_forward("tiny_etd_fwd", "tiny", 2, [10, 999], emb_trans_dec=True),
    "tiny_wope_fwd": lambda: gen_forward("tiny_wope_fwd", "tiny", 2, [7, 640], wo_pos_emb=True),
    "tiny_text_fwd_cfg": lambda: gen_forward("tiny_text_fwd_cfg", "tiny_text", 3, [0, 321], guided=True),
    "tiny_ddpm10": lambda: gen_loop("tiny_ddpm10", "tiny", 2, "10", "ddpm", keep_trace=True),
    "tiny_ddim10_cfg": lambda: gen_loop("tiny_ddim10_cfg", "tiny", 2, "ddim10", "ddim", guided=True, keep_trace=True),
    "tiny_etd_ddim10_cfg": lambda: gen_loop("tiny_etd_ddim10_cfg", "tiny", 2, "ddim10", "ddim", guided=True, emb_trans_dec=True),
    "tiny_wope_ddpm10": lambda: gen_loop("tiny_wope_ddpm10", "tiny", 2, "10", "ddpm", wo_pos_emb=True),
    "ntu_add_etd_ddpm20": lambda: gen_loop("ntu_add_etd_ddpm20", "ntu", 2, "20", "ddpm", cm_mode="add", emb_trans_dec=True),
    "tiny_opts_clip": lambda: gen_loop("tiny_opts_clip", "tiny", 2, "10", "ddpm", opts=dict(clip_denoised=True)),
    "tiny_opts_large_linear": lambda: gen_loop("tiny_opts_large_linear", "tiny", 2, "20", "ddpm", opts=dict(sigma_small=False, noise_schedule="linear")),
    "tiny_opts_skip_init": lambda: gen_loop("tiny_opts_skip_init", "tiny", 2, "10", "ddpm", opts=dict(skip_timesteps=3, init_image=True)),
    "tiny_opts_skip_zero": lambda: gen_loop("tiny_opts_skip_zero", "tiny", 2, "ddim10", "ddim", opts=dict(skip_timesteps=4)),
    "tiny_opts_eta": lambda: gen_loop("tiny_opts_eta", "tiny", 2, "ddim10", "ddim", guided=True, opts=dict(eta=0.7)),
    "tiny_add_ddpm1000": lambda: gen_loop("tiny_add_ddpm1000", "tiny_add", 2, "", "ddpm"),
    "tiny_text_ddim20_cfg": lambda: gen_loop("tiny_text_ddim20_cfg", "tiny_text", 3, "ddim20", "ddim", guided=True),
    "tiny_autoreg_ddpm10": lambda: gen_autoreg("tiny_autoreg_ddpm10", "tiny", 2, "10"),
    "tiny_add_autoreg_ddpm20": lambda: gen_autoreg("tiny_add_autoreg_ddpm20", "tiny_add", 3, "20"),
    "ntu_fwd": lambda: gen_forward("ntu_fwd", "ntu", 2, [0, 500, 999]),
    "ntu_action_fwd_cfg": lambda: gen_forward("ntu_action_fwd_cfg", "ntu_action", 2, [0, 990], guided=True),
    "ntu_ddpm50": lambda: gen_loop("ntu_ddpm50", "ntu", 2, "50", "ddpm"),
    "ntu_action_ddim100_cfg": lambda: gen_loop("ntu_action_ddim100_cfg", "ntu_action", 2, "ddim100", "ddim", guided=True),
    "chi3d_fwd": lambda: gen_forward("chi3d_fwd", "chi3d", 2, [0, 999]),
    # Chi3D (T=150, 8 action classes) sampling loops: DDPM and guided DDIM
    "chi3d_ddpm20": lambda: gen_loop("chi3d_ddpm20", "chi3d", 2, "20", "ddpm"),
    "chi3d_ddim20_cfg": lambda: gen_loop("chi3d_ddim20_cfg", "chi3d", 2, "ddim20", "ddim", guided=True),
    "text150_ddim50_cfg": lambda: gen_loop("text150_ddim50_cfg", "text150", 2, "ddim50", "ddim", guided=True),
    # the reference's shipped evaluation setting (README.md:134-137: `--timestep_respacing ddim5` through p_sample_loop,
    # eval/a2m/stgcn_eval.py:61,69) and the plain 5-step spacing, 8-layer NTU model, per-step trace kept
    "ntu_eval_ddim5": lambda: gen_loop("ntu_eval_ddim5", "ntu", 2, "ddim5", "ddpm", keep_trace=True),
    "ntu_eval_5": lambda: gen_loop("ntu_eval_5", "ntu", 2, "5", "ddpm", keep_trace=True),
    "ntu_action_eval_ddim5": lambda: gen_loop("ntu_action_eval_ddim5", "ntu_action", 2, "ddim5", "ddpm"),
    # long: the headline configuration (1000-step DDPM), B=2
    "ntu_ddpm1000": lambda: gen_loop("ntu_ddpm1000", "ntu", 2, "", "ddpm"),
}
LONG = {"ntu_ddpm1000"}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-long", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    for k, fn in JOBS.items():
        if a.only and k not in a.only.split(","):
            continue
        if a.skip_long and k in LONG:
            continue
        fn()
