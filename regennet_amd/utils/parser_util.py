"""Flag subset of the reference's `utils/parser_util.py` that the sampling hot path reads:
base (:90-98), diffusion (:100-106), model (:109-139), dataset (:142-153), sampling (:190-204), generate
(:207-220). As in the reference, for sampling the `model`/`diffusion` groups are overwritten from `args.json`
beside the checkpoint (parse_and_load_from_model_wo_data :40-70) and guidance is disabled when the model was
trained without condition masking (:68-69). `type=bool` flags keep the reference's (quirky) semantics: any
non-empty string is True."""
import argparse
import json
import os


def add_base_options(p):
    g = p.add_argument_group("base")
    g.add_argument("--cuda", default=True, type=bool)
    g.add_argument("--device", default=0, type=int)
    g.add_argument("--seed", default=10, type=int)
    g.add_argument("--batch_size", default=64, type=int)
    g.add_argument("--use_ddim", action="store_true")
    g.add_argument("--timestep_respacing", default="", type=str)


def add_diffusion_options(p):
    g = p.add_argument_group("diffusion")
    g.add_argument("--noise_schedule", default="cosine", choices=["linear", "cosine"], type=str)
    g.add_argument("--diffusion_steps", default=1000, type=int)   # ignored, like the reference (model_util.py:78)
    g.add_argument("--sigma_small", default=True, type=bool)


def add_model_options(p):
    g = p.add_argument_group("model")
    g.add_argument("--setting", default="cmdm", choices=["mdm", "cmdm"], type=str)
    g.add_argument("--arch", default="online", choices=["trans_enc", "trans_dec", "gru", "mlp", "online", "offline"], type=str)
    g.add_argument("--emb_trans_dec", default=False, type=bool)
    g.add_argument("--wo_pos_emb", action="store_true")
    g.add_argument("--cm_mode", default="concat", choices=["concat", "add"], type=str)
    g.add_argument("--layers", default=8, type=int)
    g.add_argument("--latent_dim", default=512, type=int)
    g.add_argument("--cond_mask_prob", default=0.1, type=float)
    g.add_argument("--unconstrained", action="store_true")
    for name in ("lambda_rcxyz", "lambda_vel", "lambda_fc", "lambda_orient", "lambda_body", "lambda_transl"):
        g.add_argument("--" + name, default=0.0, type=float)
    g.add_argument("--vel_threshold", default=0.01, type=float)


def add_data_options(p):
    g = p.add_argument_group("dataset")
    g.add_argument("--dataset", default="ntu", choices=["ntu", "chi3d"], type=str)
    g.add_argument("--data_path", default="", type=str)
    g.add_argument("--num_person", default=2, type=int)
    g.add_argument("--pose_rep", default="rot6d", type=str)
    g.add_argument("--body_model", default="smplx", type=str)


def add_sampling_options(p):
    g = p.add_argument_group("sampling")
    g.add_argument("--model_path", default="", type=str, help="model####.pt (args.json is read from the same directory)")
    g.add_argument("--output_dir", default="", type=str)
    g.add_argument("--num_samples", default=10, type=int)
    g.add_argument("--num_repetitions", default=3, type=int)
    g.add_argument("--guidance_param", default=2.5, type=float)


def add_generate_options(p):
    g = p.add_argument_group("generate")
    g.add_argument("--motion_length", default=60, type=float)
    g.add_argument("--action_file", default="", type=str)
    g.add_argument("--action_name", default="", type=str)
    # inputs the reference takes from its (licence-restricted) h5 datasets; here an .npz or synthetic data
    g.add_argument("--cmotion_npz", default="", type=str, help="npz with 'cmotion' [N,56,6,T] (+ 'action' [N]) actor clips")
    g.add_argument("--synthetic", action="store_true", help="synthetic checkpoint + actor motions (no assets needed)")
    g.add_argument("--precision", default="bf16_x3tail", choices=["f32", "bf16x3", "bf16", "bf16_x3tail"], type=str)


def _group_keys(parser, args, title):
    for grp in parser._action_groups:
        if grp.title == title:
            return [a.dest for a in grp._group_actions]
    raise ValueError("group_name was not found.")


def cgenerate_args(argv=None):
    p = argparse.ArgumentParser()
    add_base_options(p)
    add_data_options(p)
    add_sampling_options(p)
    add_generate_options(p)
    add_model_options(p)
    add_diffusion_options(p)
    args = p.parse_args(argv)
    if args.model_path:
        args_path = os.path.join(os.path.dirname(args.model_path), "args.json")
        assert os.path.exists(args_path), "Arguments json file was not found!"
        with open(args_path) as fr:
            model_args = json.load(fr)
        for key in _group_keys(p, args, "model") + _group_keys(p, args, "diffusion"):
            if key in model_args:
                setattr(args, key, model_args[key])
            elif "cond_mode" in model_args:
                setattr(args, "unconstrained", model_args["cond_mode"] == "no_cond")
            else:
                print(f"Warning: was not able to load [{key}], using default value [{getattr(args, key)}] instead.")
    if args.cond_mask_prob == 0:
        args.guidance_param = 1
    return args
