"""`python -m regennet_amd.sample.cgenerate` — counterpart of the reference's `sample/cgenerate.py` for the
sampling hot path: build model+diffusion through the reference-shaped factory, loop `num_repetitions` x
sample_fn with the reference's own timing hook (cgenerate.py:123,136-140,169), smooth with the sigma=1
temporal Gaussian (cgenerate.py:142) — here on the device — and save `results.npy`.

The reference pulls actor clips from its h5 datasets (licence-restricted, absent here) and finishes with SMPL-X
`rot2xyz`; this CLI takes actor clips from an .npz (or synthetic ones) and stores the rot6d output ('output',
'cmotion' keys as in the reference); 'motion' (xyz) is only produced when `model.rot2xyz` has been supplied."""
import os
import time
import types

import numpy as np
import torch

from .. import synth
from ..model.cfg_sampler import ClassifierFreeSampleModel
from ..utils import dist_util
from ..utils.fixseed import fixseed
from ..utils.model_util import create_model_and_diffusion, load_model_wo_clip
from ..utils.parser_util import cgenerate_args


def _load_clips(args, cfg, n):
    if args.cmotion_npz:
        z = np.load(args.cmotion_npz)
        cm = np.asarray(z["cmotion"], dtype=np.float32)
        act = np.asarray(z["action"], dtype=np.int64).reshape(-1, 1) if "action" in z else np.zeros((len(cm), 1), np.int64)
        return cm, act
    reps = max(1, args.num_repetitions)
    return synth.make_cmotion(cfg, n * reps, seed=1), synth.make_actions(cfg, n * reps, seed=2)


def main(argv=None):
    """Single process: one GPU samples `num_samples` motions per repetition. Under torchrun
    (`python -m torch.distributed.run --nproc-per-node N -m regennet_amd.sample.cgenerate ...`): the `num_samples` of every
    repetition are sharded over the N ranks (contiguous shards, `dist_util.shard_bounds`), rank 0's checkpoint is
    broadcast once over RCCL (`dist_util.sync_model_weights`: one flat fp32 buffer), every rank samples its shard with the Philox stream keyed by the GLOBAL sample index (so the
    result does not depend on N), and rank 0 gathers and saves. No collective inside the sampling loop."""
    args = cgenerate_args(argv)
    fixseed(args.seed)
    max_frames = 150 if args.dataset == "chi3d" else 60
    n_frames = min(max_frames, int(args.motion_length))
    dev = dist_util.setup_dist()
    rank, world = dist_util.world()
    assert args.num_samples <= args.batch_size, \
        f"Please either increase batch_size({args.batch_size}) or reduce num_samples({args.num_samples})"
    args.batch_size = args.num_samples
    cfg = synth.get_config("chi3d" if args.dataset == "chi3d" else ("ntu" if args.unconstrained else "ntu_action"))
    data = types.SimpleNamespace(dataset=types.SimpleNamespace(num_actions=cfg["num_actions"], num_person=2))
    if rank == 0:
        print("Creating model and diffusion...")
    model, diffusion = create_model_and_diffusion(args, data)
    model.precision = args.precision
    if args.synthetic or not args.model_path:
        # rank 0 owns "the checkpoint"; other ranks start from different values and receive rank 0's through the start-up broadcast
        sd = {k: torch.from_numpy(v) for k, v in
              synth.make_state_dict(model.engine_config() | {"layers": model.num_layers}, seed=0 if rank == 0 else 1000 + rank).items()}
    else:
        if rank == 0:
            print(f"Loading checkpoints from [{args.model_path}]...")
        sd = torch.load(args.model_path, map_location="cpu")
    load_model_wo_clip(model, sd)
    if args.guidance_param != 1:
        model = ClassifierFreeSampleModel(model)
    model.to(dev)
    model.eval()
    dist_util.sync_model_weights(model, 0)                  # THE collective of the run's start-up (none in a single process)
    clips, actions = _load_clips(args, cfg, args.num_samples)
    assert clips.shape[1:3] == (cfg["njoints"], cfg["nfeats"]) and clips.shape[3] >= n_frames, \
        f"actor clips {clips.shape} do not cover [N, {cfg['njoints']}, {cfg['nfeats']}, {n_frames}]"
    clips = clips[..., :n_frames]                           # --motion_length shorter than the dataset length (cgenerate.py:40,126)
    B = args.batch_size
    lo, hi = dist_util.shard_bounds(B)                      # this rank's samples of every repetition
    Bl = hi - lo
    sample_fn = diffusion.p_sample_loop if not args.use_ddim else diffusion.ddim_sample_loop
    inner = model.model if isinstance(model, ClassifierFreeSampleModel) else model
    eng, _ = inner._get_engine(max(Bl, 1), n_frames)
    all_outputs, all_cmotions, time_all = [], [], 0.0
    shape = (Bl, inner.njoints, inner.nfeats, n_frames)

    def make_y(rep_i):
        idx = (np.arange(lo, hi) + rep_i * B) % len(clips)
        y = {"cmotion": torch.from_numpy(np.ascontiguousarray(clips[idx])).to(dev), "lengths": torch.full((Bl,), n_frames),
             "mask": torch.ones(Bl, 1, 1, n_frames, dtype=torch.bool)}
        if inner.cond_mode == "action":
            y["action"] = torch.from_numpy(actions[idx]).to(dev)
        if args.guidance_param != 1:
            y["scale"] = torch.ones(Bl, device=dev) * args.guidance_param
        return y

    if world > 1:
        # one precision-schedule switch point for all ranks (x3_tail="auto": measured on the loaded checkpoint), agreed HERE, where every
        # rank passes whatever its shard - a rank with no samples contributes 0 and still takes part in the all-reduce
        diffusion.agree_x3_tail(model, shape, {"y": make_y(0)} if Bl > 0 else None, sampler="ddim" if args.use_ddim else "ddpm")
    for rep_i in range(args.num_repetitions):
        if rank == 0:
            print(f"### Sampling [repetitions #{rep_i}]")
        y = make_y(rep_i)
        dist_util.synchronize()
        t_start = time.time()
        if Bl > 0:
            sample = sample_fn(model, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
                               progress=(rank == 0), dump_steps=None, noise=None, const_noise=False,
                               seed=args.seed * 1000003 + rep_i, sample_offset=rep_i * B + lo)
            smooth = torch.empty_like(sample)      # scipy.ndimage.gaussian_filter1d(sigma=1, axis=-1) on the device (cgenerate.py:142)
            eng.gaussian_filter1d(sample.contiguous(), smooth, sample.numel() // n_frames, n_frames, 1.0, dist_util.stream_handle(dev))
        else:
            smooth = torch.empty(shape, device=dev)
        dist_util.synchronize()
        t_end = time.time()
        if rep_i >= 1:
            time_all += (t_end - t_start) * 1000
        if rank == 0:
            print("Generating time consumption: %s ms" % ((t_end - t_start) * 1000))
        all_outputs.append(dist_util.all_gather_samples(smooth, B).cpu().numpy())
        all_cmotions.append(dist_util.all_gather_samples(y["cmotion"], B).cpu().numpy())
        if rank == 0:
            print(f"created {len(all_outputs) * B} samples")
    npy_path = None
    if rank == 0:
        if args.num_repetitions != 1:
            print("Average Time Consumption: %s ms" % (time_all / (args.num_repetitions - 1)))
        out_path = args.output_dir or os.path.join(os.path.dirname(args.model_path) or ".", f"samples_seed{args.seed}")
        os.makedirs(out_path, exist_ok=True)
        npy_path = os.path.join(out_path, "results.npy")
        print(f"saving results file to [{npy_path}]")
        np.save(npy_path, {"output": np.concatenate(all_outputs), "cmotion": np.concatenate(all_cmotions),
                           "lengths": np.full((len(all_outputs) * B,), n_frames), "num_samples": args.num_samples,
                           "num_repetitions": args.num_repetitions, "world_size": world})
        print(f"[Done] Results are at [{os.path.abspath(out_path)}]")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return npy_path


if __name__ == "__main__":
    main()
