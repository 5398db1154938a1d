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
        self._stale_person = {}
        self.engine_options = {}                  # switches of this model's engine (rgn_stgcn_set_option): kernel forms (tools and tests) and "SG_F16": 1 - the
                                                  # recogniser on single fp16 operand planes (one MFMA per product; features 4e-4 of the largest instead of 5e-6)
        for p in self.parameters():
            p.requires_grad_(False)

    def load_state_dict(self, state_dict, strict=True):
        self._stale, self._stale_person = True, {}
        return super().load_state_dict(state_dict, strict=strict)

    def _apply(self, fn, *a, **k):
        self._stale, self._stale_person = True, {}
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

    # ---- per-person evaluation (not in the reference: an evaluation loop that shows the recogniser the SAME actor clip again and again - every
    #      repetition / seed of eval/a2m/stgcn/evaluate.py re-samples the reactor for the same actors - can keep the actor's half) -------------------
    def person_features(self, x, person=0):
        """Pooled features [N, 256] of ONE person's motion x [N, V, C, T] (slot `person` of batch['output']'s channel axis): stgcn.py:96-113 for that
        person alone. In eval mode the persons of a clip never meet before the final mean (data_bn is a per-channel affine, every st_gcn block runs
        on the (N M) sequences independently, stgcn.py:99-114), so `features_from_persons` reproduces `forward` from per-person results."""
        N, V, C, T = x.shape
        M = self.num_person
        assert (V, C * M) == (self.A.shape[1], self.in_channels) and 0 <= person < M, f"x {tuple(x.shape)} is not one person of this model's input"
        dev = self.A.device
        if dev.type != "cuda":
            raise RuntimeError("regennet_amd STGCN runs on an AMD GPU only: call model.to(device) first (no CPU fallback)")
        engs = self.__dict__.setdefault("_person_engines", {})
        popts = self.__dict__.setdefault("_person_options", {})
        eng = engs.get(person)
        if eng is not None and popts.get(person) != self.engine_options:        # an edit of the switches reaches the live per-person engines too
            torch.cuda.synchronize(dev)
            if set(popts.get(person, {})) - set(self.engine_options):
                self._stale_person[person] = True                               # (a switch taken away: only a rebuild restores the environment's default)
            else:
                for k, v in self.engine_options.items():
                    if popts[person].get(k) != v:
                        eng.set_option(k, v)
                popts[person] = dict(self.engine_options)
        if eng is None or self._stale_person.get(person, True) or N > eng.max_batch or eng.shape != (V, C, T):
            if eng is not None:
                torch.cuda.synchronize(dev)
                eng.close()
            eng = _lib.StgcnEngine(C, self.num_class, 1, V, T, max(N, eng.max_batch if eng and eng.shape == (V, C, T) else 0), dev.index or 0, options=self.engine_options)
            lo, hi = person * V * C, (person + 1) * V * C               # BatchNorm1d channel (m V + v) C + c (stgcn.py:96-101): person m's slice
            for k, v in self.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    continue
                if k.startswith("data_bn."):
                    v = v[lo:hi]
                eng.load_weight(k, v.detach().float().cpu().numpy())
            eng.finalize()
            engs[person] = eng
            popts[person] = dict(self.engine_options)
            self._stale_person[person] = False
        xc = x.to(device=dev, dtype=torch.float32).contiguous()
        feats = torch.empty(N, 256, device=dev)
        eng.forward(N, xc, feats, None, torch.cuda.current_stream(dev).cuda_stream)
        return feats

    def features_from_persons(self, persons):
        """`persons`: one entry per person slot, each either that person's motion [N, V, C, T] (evaluated now) or its pooled features [N, 256] from an
        earlier `person_features` call (reused). Returns {'features', 'yhat'} as `forward` does for the concatenated clip: the mean over the persons
        (stgcn.py:114) and the 256 -> classes layer (stgcn.py:117-119)."""
        assert len(persons) == self.num_person
        fs = [p if p.dim() == 2 else self.person_features(p, m) for m, p in enumerate(persons)]
        f = torch.stack(fs).mean(dim=0)
        w = self.fcn.weight.reshape(self.num_class, 256).to(f.device)
        return {"features": f.squeeze(), "yhat": f @ w.t() + self.fcn.bias.to(f.device)}

    def compute_accuracy(self, batch):
        """stgcn.py:125-132."""
        confusion = torch.zeros(self.num_class, self.num_class, dtype=int)
        yhat = batch["yhat"].max(dim=1).indices
        for label, pred in zip(batch["y"], yhat):
            confusion[label][pred] += 1
        return torch.trace(confusion) / torch.sum(confusion)
