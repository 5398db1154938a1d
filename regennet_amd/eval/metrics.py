"""Metrics of the a2m evaluation harness on device tensors — mirror of the reference's `eval/a2m/stgcn/{accuracy,diversity,
evaluate}.py`. The index draws use NumPy's global generator exactly like the reference (same seed -> same pairs); the
distances are computed on the activations' device in one batched pass instead of 200 + 20 * num_labels `torch.dist` calls."""
import numpy as np
import torch

from .fid import calculate_fid
from .stgcn import STGCN


def calculate_accuracy(model, motion_loader, num_labels, classifier, device):
    """accuracy.py:4-14. Returns (accuracy, confusion [num_labels, num_labels] int64)."""
    confusion = torch.zeros(num_labels, num_labels, dtype=torch.long)
    with torch.no_grad():
        for batch in motion_loader:
            pred = classifier(batch)["yhat"].max(dim=1).indices.cpu()
            y = torch.as_tensor(batch["y"]).cpu().long()
            confusion.index_put_((y, pred), torch.ones_like(y), accumulate=True)
    return (torch.trace(confusion) / torch.sum(confusion)).item(), confusion


def calculate_diversity_multimodality(activations, labels, num_labels, seed=None, unconstrained=False):
    """diversity.py:6-71 (multimodality is computed for every cond_mode, as there)."""
    diversity_times, multimodality_times = 200, 20
    labels_np = torch.as_tensor(labels).long().cpu().numpy()
    num_motions = activations.shape[0]
    if seed is not None:
        np.random.seed(seed)
    first = np.random.randint(0, num_motions, diversity_times)
    second = np.random.randint(0, num_motions, diversity_times)
    pairs_a, pairs_b = [], []
    quotas = np.zeros(num_labels)
    quotas[np.unique(labels_np)] = multimodality_times       # a label absent from the batch keeps a zero quota
    while np.any(quotas > 0):
        a = np.random.randint(0, num_motions)
        la = labels_np[a]
        if not quotas[la]:
            continue
        b = np.random.randint(0, num_motions)
        while la != labels_np[b]:
            b = np.random.randint(0, num_motions)
        quotas[la] -= 1
        pairs_a.append(a)
        pairs_b.append(b)
    act = activations.float()
    dev = act.device

    def dist_sum(i, j):
        i, j = torch.as_tensor(np.asarray(i), device=dev), torch.as_tensor(np.asarray(j), device=dev)
        return (act[i] - act[j]).norm(dim=1).sum()

    diversity = dist_sum(first, second) / diversity_times
    multimodality = dist_sum(pairs_a, pairs_b) / (multimodality_times * num_labels)
    return diversity.item(), multimodality.item()


class Evaluation:
    """evaluate.py:9-125: STGCN features -> accuracy / FID / diversity / multimodality per loader."""

    def __init__(self, dataname, body_model, parameters, device, seed=None):
        layout = "smplx" if body_model == "smplx" else "smpl"
        model = STGCN(in_channels=parameters["nfeats"], num_class=parameters["num_classes"], num_person=parameters["num_person"],
                      graph_args={"layout": layout, "strategy": "spatial"}, edge_importance_weighting=True, device=device)
        state_dict = parameters.get("state_dict")
        if state_dict is None:
            state_dict = torch.load(parameters["model_path"], map_location="cpu")
        model.load_state_dict(state_dict)
        model = model.to(device)
        model.eval()
        if parameters.get("recogniser_f16"):        # not in the reference: the recogniser's opt-in fp16 arithmetic (STGCN.engine_options["SG_F16"])
            model.engine_options["SG_F16"] = 1
        self.num_classes, self.model = parameters["num_classes"], model
        self.dataname, self.device, self.seed = dataname, device, seed

    def compute_features(self, model, motionloader):
        activations, labels = [], []
        with torch.no_grad():
            for batch in motionloader:
                activations.append(self.model(batch)["features"].reshape(-1, 256))
                labels.append(torch.as_tensor(batch["y"]).to(activations[-1].device))
        return torch.cat(activations, dim=0), torch.cat(labels, dim=0)

    def compute_features_and_accuracy(self, motionloader):
        """evaluate.py:41-52 and accuracy.py:4-14 from one forward per batch: (activations [N, 256], labels [N], accuracy)."""
        activations, labels = [], []
        confusion = torch.zeros(self.num_classes, self.num_classes, dtype=torch.long)
        with torch.no_grad():
            for batch in motionloader:
                out = self.model(batch)
                activations.append(out["features"].reshape(-1, 256))
                y = torch.as_tensor(batch["y"])
                labels.append(y.to(activations[-1].device))
                confusion.index_put_((y.cpu().long(), out["yhat"].max(dim=1).indices.cpu()), torch.ones_like(y.cpu().long()), accumulate=True)
        return torch.cat(activations, dim=0), torch.cat(labels, dim=0), (torch.trace(confusion) / torch.sum(confusion)).item()

    @staticmethod
    def calculate_activation_statistics(activations):
        from .fid import calculate_activation_statistics
        return calculate_activation_statistics(activations)

    def evaluate(self, model, loaders, setting):
        metrics_all = {}
        for sets in ["train", "test"]:
            computedfeats, metrics = {}, {}
            for key, loader_sets in loaders.items():
                loader = loader_sets[sets]
                # ONE pass of the recogniser per loader: the reference runs it twice over the same batches - once for yhat (accuracy.py:4-14), once for
                # the features (evaluate.py:41-52) - and both come out of the same forward
                feats, labels, metrics[f"accuracy_{key}"] = self.compute_features_and_accuracy(loader)
                computedfeats[key] = {"feats": feats, "labels": labels, "stats": self.calculate_activation_statistics(feats)}
                ret = calculate_diversity_multimodality(feats, labels, self.num_classes, seed=self.seed,
                                                        unconstrained=(getattr(model, "cond_mode", None) == "no_cond"))
                metrics[f"diversity_{key}"], metrics[f"multimodality_{key}"] = ret
            gtstats = computedfeats["gt"]["stats"]
            for key in computedfeats:
                metrics[f"fid_{key}"] = float(calculate_fid(gtstats, computedfeats[key]["stats"]))
            metrics_all[sets] = metrics
        return {f"{key}_{sets}": metrics_all[sets][key] for sets in ["train", "test"] for key in metrics_all[sets]}

    def evaluate_acc(self, model, loaders, setting):
        """evaluate.py:127-162 (the `acc_only` runs of stgcn_eval.py:200): recognition accuracy per loader and split only."""
        out = {}
        for sets in ["train", "test"]:
            for key, loader_sets in loaders.items():
                out[f"accuracy_{key}_{sets}"], _ = calculate_accuracy(model, loader_sets[sets], self.num_classes, self.model, self.device)
        return out
