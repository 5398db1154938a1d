"""ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.

CPU restatement (PyTorch CPU functional ops, fp32) of the evaluation harness next to the sampling hot path
(SURVEY.md §8f next-4): the ST-GCN feature extractor / classifier, diversity + multimodality, accuracy. Only tests/ may
import this module. Pinned against outputs of the reference itself on a synthetic checkpoint (tests/golden/stgcn.npz,
written by tests/golden/make_golden.py::gen_stgcn).

Reference lines followed: eval/a2m/recognition/models/stgcn.py:76-123 (STGCN.forward), :145-228 (st_gcn block),
stgcnutils/tgcn.py:61-71 (graph convolution), eval/a2m/stgcn/diversity.py:6-71, eval/a2m/stgcn/accuracy.py:4-14.
"""
import numpy as np
import torch
import torch.nn.functional as F

# (in, out, temporal stride) of the ten st_gcn blocks (stgcn.py:51-62); block 0 has no residual branch (:52)
BLOCKS = [(None, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1), (128, 256, 2),
          (256, 256, 1), (256, 256, 1)]


def _t(sd, k):
    return torch.from_numpy(np.asarray(sd[k]))


def _bn(sd, prefix, x):
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"), _t(sd, prefix + ".weight"),
                        _t(sd, prefix + ".bias"), training=False, eps=1e-5)


def stgcn_forward(sd, output, num_person=2):
    """output [N, V, C * num_person, T] (batch['output']) -> (features [N, 256], yhat [N, num_class])."""
    x = torch.as_tensor(output, dtype=torch.float32)
    N, V, C2, T = x.shape
    M = num_person
    C = C2 // M
    x = x.reshape(N, V, M, C, T).permute(0, 3, 4, 1, 2).contiguous()            # stgcn.py:83-88: N, C, T, V, M
    x = x.permute(0, 4, 3, 1, 2).contiguous().view(N, M * V * C, T)              # :92-95
    x = _bn(sd, "data_bn", x)                                                    # :98
    x = x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)   # :99-101
    A0 = _t(sd, "A")
    K = A0.shape[0]
    for i, (_, co, stride) in enumerate(BLOCKS):
        p = f"st_gcn_networks.{i}."
        A = A0 * _t(sd, f"edge_importance.{i}")                                  # :105
        if i == 0:
            res = 0
        elif (p + "residual.0.weight") in sd:
            res = _bn(sd, p + "residual.1", F.conv2d(x, _t(sd, p + "residual.0.weight"), _t(sd, p + "residual.0.bias"), stride=(stride, 1)))
        else:
            res = x
        y = F.conv2d(x, _t(sd, p + "gcn.conv.weight"), _t(sd, p + "gcn.conv.bias"))   # tgcn.py:64
        n, kc, t, v = y.shape
        y = torch.einsum("nkctv,kvw->nctw", y.view(n, K, kc // K, t, v), A)           # tgcn.py:66-68
        y = F.relu(_bn(sd, p + "tcn.0", y))
        y = F.conv2d(y, _t(sd, p + "tcn.2.weight"), _t(sd, p + "tcn.2.bias"), stride=(stride, 1), padding=(4, 0))
        y = _bn(sd, p + "tcn.3", y)
        x = F.relu(y + res)                                                      # stgcn.py:224-228
    x = F.avg_pool2d(x, x.shape[2:])                                             # :113
    x = x.view(N, M, -1, 1, 1).mean(dim=1)                                       # :114
    feats = x.reshape(N, -1)
    yhat = F.conv2d(x, _t(sd, "fcn.weight"), _t(sd, "fcn.bias")).view(N, -1)     # :120-121
    return feats, yhat


def calculate_diversity_multimodality(activations, labels, num_labels, seed=None):
    """diversity.py:6-71 (the branch that runs: multimodality is computed for every cond_mode)."""
    act = torch.as_tensor(activations)
    labels = torch.as_tensor(labels).long()
    num = act.shape[0]
    if seed is not None:
        np.random.seed(seed)
    first = np.random.randint(0, num, 200)
    second = np.random.randint(0, num, 200)
    diversity = sum(torch.dist(act[a], act[b]) for a, b in zip(first, second)) / 200
    multimodality = 0
    quotas = np.zeros(num_labels)
    quotas[labels.unique()] = 20
    while np.any(quotas > 0):
        a = np.random.randint(0, num)
        la = labels[a]
        if not quotas[la]:
            continue
        b = np.random.randint(0, num)
        while la != labels[b]:
            b = np.random.randint(0, num)
        quotas[la] -= 1
        multimodality += torch.dist(act[a], act[b])
    multimodality /= 20 * num_labels
    return float(diversity), float(multimodality)


def calculate_accuracy(yhat_batches, y_batches, num_labels):
    """accuracy.py:4-14 given the classifier outputs per batch."""
    confusion = torch.zeros(num_labels, num_labels, dtype=torch.long)
    for yh, y in zip(yhat_batches, y_batches):
        pred = torch.as_tensor(yh).max(dim=1).indices
        for label, p in zip(torch.as_tensor(y), pred):
            confusion[label][p] += 1
    return float(torch.trace(confusion) / torch.sum(confusion)), confusion
