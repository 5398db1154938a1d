"""End-to-end FID-delta proxy for north_star's "FID within +-0.1 of reference" (test infrastructure; `python -m tests.fid_proxy --n 4096`
for a larger N). Real data, a trained denoiser and the trained recogniser are absent (licence-restricted / not shipped), so the claim is
measured on what IS available: N action-conditioned NTU motions sampled with the reference's shipped evaluation setting
(`--timestep_respacing ddim5` through p_sample_loop, eval/a2m/stgcn_eval.py:61-81; README.md:134-137) by the HIP path under its DEFAULT
precision schedule and by the oracle (CPU restatement of the reference's loop) on the SAME noise tape, concatenated with the actor motion
as stgcn_eval.py:71 does, passed through the SAME recogniser (a synthetic ST-GCN checkpoint with the reference's keys, rgn_stgcn_forward),
then eval/a2m/stgcn/evaluate.py:48-53 statistics and eval/a2m/stgcn/fid.py:11-61:

    fid_oracle_hip = FID(oracle set, HIP set)                        - what the arithmetic difference alone is worth, in FID units
    delta_vs_gt    = |FID(gt*, HIP set) - FID(gt*, oracle set)|      - the quantity north_star bounds by 0.1, against a synthetic gt* set
    argmax_agree   = share of motions whose predicted action agrees  - evaluate.py's accuracy sees the same labels
"""
import numpy as np
import torch


def run(n_motions=1024, chunk=256, threads=32, seed=0, verbose=False, recogniser_f16=False):
    """recogniser_f16: additionally pass every set through a second recogniser engine on single fp16 operand planes (SG_F16) and report what THAT
    arithmetic is worth in FID units: FID(features of the default recogniser, features of the fp16 one) on the HIP-sampled set, and the change of
    FID(gt*, HIP set) when the whole evaluation (both sets) runs on the fp16 recogniser."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    from regennet_amd.eval import STGCN
    from regennet_amd.eval.fid import calculate_activation_statistics, calculate_fid
    from tests.conftest import golden_path
    cfg = synth.get_config("ntu_action")
    sd = synth.make_state_dict(cfg, seed=0)
    S = 5
    model, diffusion = synth.build_model(cfg, sd, resp="ddim5", precision="bf16_x3tail", device="cuda:0")
    sched = orc.make_schedule("cosine", "ddim5")
    A = np.load(golden_path("stgcn"), allow_pickle=False)["A"]
    rec_sd = synth.make_stgcn_state_dict(A, num_class=26, seed=0)
    rec = STGCN(in_channels=12, num_class=26, num_person=2, graph_args={"layout": "smplx", "strategy": "spatial"}, device="cuda:0")
    rec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rec_sd.items()}, strict=True)
    rec.to("cuda:0").eval()
    rec16 = None
    if recogniser_f16:
        rec16 = STGCN(in_channels=12, num_class=26, num_person=2, graph_args={"layout": "smplx", "strategy": "spatial"}, device="cuda:0")
        rec16.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rec_sd.items()}, strict=True)
        rec16.to("cuda:0").eval()
        rec16.engine_options["SG_F16"] = 1
    old_threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, old_threads))
    feats = {"hip": [], "oracle": [], "gt": [], "hip_f16": [], "gt_f16": []}
    preds = {"hip": [], "oracle": [], "hip_f16": []}
    worst = 0.0
    try:
        for c0 in range(0, n_motions, chunk):
            nb = min(chunk, n_motions - c0)
            cm = synth.make_cmotion(cfg, nb, seed=1000 + seed + c0)
            act = synth.make_actions(cfg, nb, seed=2000 + seed + c0)
            tape = synth.make_noise_tape(cfg, nb, S, seed=3000 + seed + c0)
            y = {"cmotion": torch.from_numpy(cm).cuda(), "action": torch.from_numpy(act).cuda()}
            hip = diffusion.p_sample_loop(model, (nb, 56, 6, 60), clip_denoised=False, model_kwargs={"y": y}, noise_tape=torch.from_numpy(tape))
            ref = orc.sample_loop(sd, cfg, sched, tape, {"cmotion": torch.from_numpy(cm), "action": torch.from_numpy(act)}, mode="ddpm")
            worst = max(worst, float((hip.cpu() - ref).abs().max()))
            gt_other = torch.from_numpy(synth.make_cmotion(cfg, nb, seed=4000 + seed + c0)).cuda()      # gt*: a second valid rot6d motion as the reactor
            for name, other in (("hip", hip), ("oracle", ref.cuda()), ("gt", gt_other)):
                batch = rec({"output": torch.cat((y["cmotion"], other), dim=2)})                       # stgcn_eval.py:71
                feats[name].append(batch["features"].reshape(nb, 256).clone())
                if name in preds:
                    preds[name].append(batch["yhat"].max(dim=1).indices.clone())
                if rec16 is not None and name != "oracle":
                    b16 = rec16({"output": torch.cat((y["cmotion"], other), dim=2)})
                    feats[name + "_f16"].append(b16["features"].reshape(nb, 256).clone())
                    if name == "hip":
                        preds["hip_f16"].append(b16["yhat"].max(dim=1).indices.clone())
            if verbose:
                print(f"[fid proxy] motions {c0 + nb}/{n_motions}, max |hip - oracle| so far {worst:.2e}", flush=True)
    finally:
        torch.set_num_threads(old_threads)
    st = {k: calculate_activation_statistics(torch.cat(v)) for k, v in feats.items() if v}
    fid_gt_hip, fid_gt_orc = float(calculate_fid(st["gt"], st["hip"])), float(calculate_fid(st["gt"], st["oracle"]))
    out = {"n": n_motions, "max_abs_motion_dev": worst, "fid_oracle_hip": float(calculate_fid(st["oracle"], st["hip"])),
           "fid_gt_hip": fid_gt_hip, "fid_gt_oracle": fid_gt_orc, "delta_vs_gt": abs(fid_gt_hip - fid_gt_orc),
           "argmax_agree": float((torch.cat(preds["hip"]) == torch.cat(preds["oracle"])).float().mean()),
           "feature_scale": float(torch.cat(feats["oracle"]).abs().mean())}
    if rec16 is not None:
        f3, f16 = torch.cat(feats["hip"]), torch.cat(feats["hip_f16"])
        fid16 = float(calculate_fid(st["gt_f16"], st["hip_f16"]))
        out["recogniser_f16"] = {"fid_default_vs_f16_features": float(calculate_fid(st["hip"], st["hip_f16"])), "fid_gt_hip_on_f16": fid16,
                                 "delta_of_fid_gt_hip": abs(fid16 - fid_gt_hip), "features_max_abs": float((f3 - f16).abs().max()),
                                 "features_abs_max": float(f3.abs().max()), "features_rms_rel": float(((f3 - f16).pow(2).mean() / f3.pow(2).mean()).sqrt()),
                                 "argmax_agree": float((torch.cat(preds["hip"]) == torch.cat(preds["hip_f16"])).float().mean())}
    model._engine.close()
    return out


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--recogniser-f16", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.n, threads=a.threads, verbose=True, recogniser_f16=a.recogniser_f16)))
