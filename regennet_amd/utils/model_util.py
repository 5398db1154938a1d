"""Factory — mirror of the reference's `utils/model_util.py` (create_model_and_diffusion :11-17,
get_model_args :20-72, create_gaussian_diffusion :75-117, load_model_wo_clip :5-8)."""
from ..diffusion import gaussian_diffusion as gd
from ..diffusion.respace import SpacedDiffusion, space_timesteps
from ..model.cmdm import CMDM


def load_model_wo_clip(model, state_dict):
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    assert len(unexpected_keys) == 0
    assert all([k.startswith("clip_model.") for k in missing_keys])


def create_model_and_diffusion(args, data):
    if args.setting != "cmdm":
        raise NotImplementedError("only setting='cmdm' is on the hot path (model_util.py:13)")
    model = CMDM(**get_model_args(args, data))
    args.num_person = 1  # side effect of the reference (model_util.py:15)
    diffusion = create_gaussian_diffusion(args)
    return model, diffusion


def get_model_args(args, data):
    if args.unconstrained:
        cond_mode = "no_cond"
    elif args.dataset in ["kit", "humanml"]:
        cond_mode = "text"
    else:
        cond_mode = "action"
    cond_mode = getattr(args, "cond_mode_override", None) or cond_mode   # text-conditioned a2m-shaped models (cfg5)
    ds = data.dataset
    num_actions = getattr(ds, "num_actions", 1)
    num_person = getattr(ds, "num_person", 1)
    data_rep = args.pose_rep
    njoints = {"smpl": 25, "smplx": 56}[args.body_model]
    nfeats = {"rot6d": 6, "xyz": 3}[data_rep]
    if args.dataset in ("humanml", "kit"):
        raise NotImplementedError("humanml/kit are not wired in the reference either (get_data.py:6-20)")
    num_frames = {"ntu": 60, "chi3d": 150}[args.dataset]
    num_frames = getattr(args, "num_frames_override", None) or num_frames
    return {"modeltype": "", "njoints": njoints, "nfeats": nfeats, "num_actions": num_actions,
            "num_person": num_person, "num_frames": num_frames, "translation": True, "pose_rep": "rot6d",
            "glob": True, "glob_rot": True, "latent_dim": args.latent_dim, "ff_size": 1024,
            "num_layers": args.layers, "num_heads": 4, "dropout": 0.1, "activation": "gelu",
            "data_rep": data_rep, "cond_mode": cond_mode, "cond_mask_prob": args.cond_mask_prob,
            "action_emb": "tensor", "arch": args.arch, "cm_mode": args.cm_mode, "body_model": args.body_model,
            "wo_pos_emb": args.wo_pos_emb, "emb_trans_dec": args.emb_trans_dec, "clip_version": "ViT-B/32",
            "dataset": args.dataset}


def create_gaussian_diffusion(args):
    steps = 1000                       # hard-coded by the reference (model_util.py:78); --diffusion_steps is ignored
    betas = gd.get_named_beta_schedule(args.noise_schedule, steps, 1.0)
    respacing = args.timestep_respacing or [steps]
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, respacing),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=gd.ModelVarType.FIXED_SMALL if args.sigma_small else gd.ModelVarType.FIXED_LARGE,
        loss_type=gd.LossType.MSE,
        rescale_timesteps=False,
        lambda_vel=getattr(args, "lambda_vel", 0.0), lambda_rcxyz=getattr(args, "lambda_rcxyz", 0.0),
        lambda_fc=getattr(args, "lambda_fc", 0.0), lambda_orient=getattr(args, "lambda_orient", 0.0),
        lambda_body=getattr(args, "lambda_body", 0.0), lambda_transl=getattr(args, "lambda_transl", 0.0),
        data_rep=args.pose_rep, num_person=args.num_person, body_model=args.body_model,
        vel_threshold=getattr(args, "vel_threshold", 0.01),
    )
