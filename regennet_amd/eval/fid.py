"""Frechet distance between feature sets, on whatever device the features live on.

Reference: eval/a2m/stgcn/evaluate.py:48-53 (`calculate_activation_statistics`: mean + `np.cov(rowvar=False)`) and
eval/a2m/stgcn/fid.py:11-61 (`calculate_frechet_distance`, scipy `sqrtm` of the covariance product on the host).
The reference pulls every activation to the CPU and calls scipy; here the statistics and the distance stay on the GPU
in fp64 (plain torch linear algebra - this row is glue around the sampler, not a hot path, so no kernel of its own).

    d^2 = |mu1 - mu2|^2 + Tr(C1) + Tr(C2) - 2 Tr((C1 C2)^(1/2))

C1 C2 is similar to the symmetric PSD matrix C1^(1/2) C2 C1^(1/2), so Tr((C1 C2)^(1/2)) is the sum of the square roots
of that matrix's eigenvalues: two `eigh` calls, no general (complex) matrix square root, no imaginary residue to discard.
The ST-GCN feature extractor (`regennet_amd/eval/stgcn.py` over the HIP kernels of `rgn_stgcn.hip`) and the diversity / multimodality /
accuracy metrics (`regennet_amd/eval/metrics.py`) of the same harness are their own modules.
"""
import torch as th


def calculate_activation_statistics(activations):
    """(mu [D], sigma [D, D]) in fp64 on the activations' device; sigma is the unbiased covariance (np.cov default)."""
    x = activations.to(th.float64)
    mu = x.mean(dim=0)
    xc = x - mu
    sigma = xc.T @ xc / (x.shape[0] - 1)
    return mu, sigma


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2):
    mu1, mu2 = th.atleast_1d(mu1).to(th.float64), th.atleast_1d(mu2).to(th.float64)
    sigma1, sigma2 = th.atleast_2d(sigma1).to(th.float64), th.atleast_2d(sigma2).to(th.float64)
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"
    diff = mu1 - mu2
    lam, u = th.linalg.eigh((sigma1 + sigma1.T) * 0.5)
    root1 = (u * lam.clamp_min(0).sqrt()) @ u.T                      # C1^(1/2)
    inner = root1 @ sigma2 @ root1
    ev = th.linalg.eigvalsh((inner + inner.T) * 0.5).clamp_min(0)
    return diff.dot(diff) + th.trace(sigma1) + th.trace(sigma2) - 2 * ev.sqrt().sum()


def calculate_fid(statistics_1, statistics_2):
    """fid.py:6-8."""
    return calculate_frechet_distance(statistics_1[0], statistics_1[1], statistics_2[0], statistics_2[1])
