"""ST-GCN action recogniser of the evaluation harness — host-side mirror of the reference's
`eval/a2m/recognition/models/stgcn.py` (STGCN :12-139, st_gcn :145-228) for inference.

Like `regennet_amd.model.cmdm.CMDM`, the module is a checkpoint container with the reference's parameter / buffer names
(`load_state_dict` of a reference checkpoint works unchanged, evaluate.py:24-25); `forward` hands `batch['output']` to
libregennet_hip.so (`rgn_stgcn_forward`), where the whole network runs as HIP kernels with every BatchNorm folded.
There is no eager fallback. The skeleton graph is taken from the checkpoint's `A` buffer: the reference rebuilds it from
the licensed SMPL-X kinematic tree (stgcnutils/graph.py:81-88) and then stores it in the state_dict anyway (stgcn.py:42).
"""
import torch
import torch.nn as nn

from .. import _lib

_BLOCKS = [(None, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1), (128, 256, 2),
           (256, 256, 1), (256, 256, 1)]      # stgcn.py:51-62


class _Gcn(nn.Module):                         # ConvTemporalGraphical's parameter names (stgcnutils/tgcn.py:51-59)
    def __init__(self, ci, co, K):
        super().__init__()
        self.conv = nn.Conv2d(ci, co * K, kernel_size=(1, 1))


class _Block(nn.Module):                       # st_gcn's parameter names (stgcn.py:183-219)
    def __init__(self, ci, co, K, stride, residual):
        super().__init__()
        self.gcn = _Gcn(ci, co, K)
        self.tcn = nn.Sequential(nn.BatchNorm2d(co), nn.ReLU(inplace=True), nn.Conv2d(co, co, (9, 1), (stride, 1), (4, 0)),
                                 nn.BatchNorm2d(co), nn.Dropout(0.0, inplace=True))
        if residual and not (ci == co and stride == 1):
            self.residual = nn.Sequential(nn.Conv2d(ci, co, kernel_size=1, stride=(stride, 1)), nn.BatchNorm2d(co))


class STGCN(nn.Module):
    def __init__(self, in_channels, num_class, num_person, graph_args=None, edge_importance_weighting=True, device=None,
                 num_nodes=None, spatial_kernel_size=3, **kwargs):
        super().__init__()
        if not edge_importance_weighting:
            raise NotImplementedError("the evaluator is always built with edge_importance_weighting=True (evaluate.py:19)")
        layout = (graph_args or {}).get("layout", "smplx")
        V = num_nodes or {"smplx": 56, "smpl": 25}[layout]            # stgcnutils/graph.py:63,82
        K = spatial_kernel_size                                        # 'spatial' strategy with max_hop 1: 3 partitions
        self.device, self.in_channels, self.num_class, self.num_person = device, in_channels, num_class, num_person
        self.losses = ["accuracy", "cross_entropy", "mixed"]
        self.criterion = nn.CrossEntropyLoss(reduction="mean")
        self.register_buffer("A", torch.zeros(K, V, V))
        self.data_bn = nn.BatchNorm1d(in_channels * V)
        c0 = in_channels // num_person
        self.st_gcn_networks = nn.ModuleList(
            [_Block(c0 if ci is None else ci, co, K, s, residual=i > 0) for i, (ci, co, s) in enumerate(_BLOCKS)])
        self.edge_importance = nn.ParameterList([nn.Parameter(torch.ones(K, V, V)) for _ in _BLOCKS])
        self.fcn = nn.Conv2d(256, num_class, kernel_size=1)
        self._engine, self._stale, self._applied_options = None, True, {}
        self.engine_options = {}                  # kernel-selection switches for this model's engine (rgn_stgcn_set_option; tools and tests)
        for p in self.parameters():
            p.requires_grad_(False)

    def load_state_dict(self, state_dict, strict=True):
        self._stale = True
        return super().load_state_dict(state_dict, strict=strict)

    def _apply(self, fn, *a, **k):
        self._stale = True
        return super()._apply(fn, *a, **k)

    def _get_engine(self, N, T):
        dev = self.A.device
        if dev.type != "cuda":
            raise RuntimeError("regennet_amd STGCN runs on an AMD GPU only: call model.to(device) first (no CPU fallback)")
        V = self.A.shape[1]
        eng = self._engine
        if eng is None or self._stale or N > eng.max_batch or eng.shape != (V, self.in_channels, T):
            if eng is not None:
                torch.cuda.synchronize(dev)
                eng.close()
            eng = _lib.StgcnEngine(self.in_channels, self.num_class, self.num_person, V, T, max(N, eng.max_batch if eng and eng.shape == (V, self.in_channels, T) else 0),
                                   dev.index or 0, options=self.engine_options)
            for k, v in self.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    continue
                eng.load_weight(k, v.detach().float().cpu().numpy())
            eng.finalize()
            self._engine, self._stale, self._applied_options = eng, False, dict(self.engine_options)
        elif self.engine_options != self._applied_options:
            # the recogniser's switches may be set at any time before a forward (rgn_stgcn_set_option): an edit of the dict reaches the live engine
            torch.cuda.synchronize(dev)
            for k, v in self.engine_options.items():
                if self._applied_options.get(k) != v:
                    eng.set_option(k, v)
            for k in set(self._applied_options) - set(self.engine_options):
                self._stale = True                                  # (a switch taken away: only a rebuild restores the environment's default)
            self._applied_options = dict(self.engine_options)
            if self._stale:
                return self._get_engine(N, T)
        return eng, dev

    def forward(self, batch):
        """batch['output'] [N, V, C * num_person, T] -> adds batch['features'] [N, 256] (squeezed like the reference,
        stgcn.py:117) and batch['yhat'] [N, num_class]."""
        x = batch["output"]
        N, V, C, T = x.shape
        assert (V, C) == (self.A.shape[1], self.in_channels), f"batch['output'] {tuple(x.shape)} does not match the model"
        eng, dev = self._get_engine(N, T)
        xc = x.to(device=dev, dtype=torch.float32).contiguous()
        feats = torch.empty(N, 256, device=dev)
        yhat = torch.empty(N, self.num_class, device=dev)
        eng.forward(N, xc, feats, yhat, torch.cuda.current_stream(dev).cuda_stream)
        batch["features"] = feats.squeeze()
        batch["yhat"] = yhat
        return batch

    def compute_accuracy(self, batch):
        """stgcn.py:125-132."""
        confusion = torch.zeros(self.num_class, self.num_class, dtype=int)
        yhat = batch["yhat"].max(dim=1).indices
        for label, pred in zip(batch["y"], yhat):
            confusion[label][pred] += 1
        return torch.trace(confusion) / torch.sum(confusion)
