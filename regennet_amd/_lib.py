"""ctypes binding of libregennet_hip.so (C-ABI: include/regennet_hip.h).

The product path has NO CPU fallback: if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"`) importing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libregennet_hip.so")

RGN_OK = 0
RGN_ERR_UNSUPPORTED = -7
ERR_NAMES = {-1: "INVALID_ARG", -2: "BAD_KEY", -3: "BAD_SHAPE", -4: "MISSING_KEY", -5: "STATE", -6: "HIP", -7: "UNSUPPORTED", -8: "INTERNAL"}
CM = {"add": 0, "concat": 1}
COND = {"no_cond": 0, "action": 1, "text": 2}
PREC = {"f32": 0, "bf16x3": 1, "bf16": 2, "bf16_x3tail": 3}
DEFAULT_PRECISION = "bf16_x3tail"
SAMPLER = {"ddpm": 0, "ddim": 1}
FLAG_UNCOND, FLAG_GUIDED = 1, 2


class RgnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"regennet_hip error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class RgnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "njoints", "nfeats", "num_frames", "latent_dim", "ff_size", "num_heads", "num_layers", "cm_mode",
        "cond_mode", "num_actions", "clip_dim", "emb_trans_dec", "wo_pos_emb", "max_batch", "precision", "device")]


class RgnSchedule(C.Structure):
    _fields_ = [("S", C.c_int32), ("timestep_map", C.POINTER(C.c_int64))] + [
        (n, C.POINTER(C.c_double)) for n in (
            "posterior_mean_coef1", "posterior_mean_coef2", "model_log_variance", "sqrt_recip_alphas_cumprod",
            "sqrt_recipm1_alphas_cumprod", "alphas_cumprod", "alphas_cumprod_prev")]


# every symbol include/regennet_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
SYMBOLS = {
    "rgn_create": (C.c_int, [C.POINTER(RgnConfig), C.POINTER(_vp)]),
    "rgn_destroy": (C.c_int, [_vp]),
    "rgn_last_error": (C.c_char_p, [_vp]),
    "rgn_load_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32]),
    "rgn_finalize_weights": (C.c_int, [_vp]),
    "rgn_weight_blob": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    "rgn_set_schedule": (C.c_int, [_vp, C.POINTER(RgnSchedule)]),
    "rgn_set_condition": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rgn_denoise": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "rgn_sample_range": (C.c_int, [_vp, _i32, _i32, _f32, _vp, _vp, _u64, _u64, _i32, _i32, _vp, _i32, _i32, _vp]),
    "rgn_set_x3_tail": (C.c_int, [_vp, _i32]),
    "rgn_set_f16_steps": (C.c_int, [_vp, _i32]),
    "rgn_precision_plan": (C.c_int, [_vp, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "rgn_set_small_batch_rows": (C.c_int, [_vp, _i32]),
    "rgn_set_const_noise": (C.c_int, [_vp, _i32]),
    "rgn_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "rgn_set_layers_min_b": (C.c_int, [_vp, _i32]),
    "rgn_plan_query": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rgn_randn": (C.c_int, [_vp, _vp, _i32, _u64, _u64, _vp]),
    "rgn_randn_step": (C.c_int, [_vp, _vp, _i32, _u64, _u64, _i32, _vp]),
    "rgn_rot6d_to_matrix": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "rgn_gaussian_filter1d": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "rgn_profile_enable": (C.c_int, [_vp, _i32]),
    "rgn_profile_query": (C.c_int, [_vp, _i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "rgn_profile_bracket_overhead": (C.c_int, [_vp, C.POINTER(C.c_double)]),
}

class RgnStgcnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "num_class", "num_person", "num_nodes", "num_frames", "max_batch", "device")]


SYMBOLS.update({
    "rgn_stgcn_create": (C.c_int, [C.POINTER(RgnStgcnConfig), C.POINTER(_vp)]),
    "rgn_stgcn_destroy": (C.c_int, [_vp]),
    "rgn_stgcn_last_error": (C.c_char_p, [_vp]),
    "rgn_stgcn_load_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32]),
    "rgn_stgcn_finalize": (C.c_int, [_vp]),
    "rgn_stgcn_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "rgn_stgcn_forward": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
})

_lib = None


def default_x3_tail(S, layers=8, etd=False):
    """The engine's default split-bf16 tail where the plain phase has NO fp16 sub-phase (rgn_plan.cpp default_tail(): kernel-per-stage forms,
    150-frame models, batches below 64, "BULK_F16": 0): how many of the last loop indices run split-bf16. Host-side mirror for tests; what an
    engine will actually do for a batch on its bound schedule is Engine.precision_plan()."""
    if etd:
        return S
    if layers >= 8:
        if S <= 10:
            return min(S, 3)
        return min(S, max(5, -(-S // 200)))
    return min(S, -(-max(8, -(-S // 100)) * 8 // max(1, layers)))


def load():
    """Load the shared library (once) and type every entry point. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` from the repo root.")
    # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7). It must be in the process BEFORE
    # our library so that both bind to ONE HIP runtime (streams/events/pointers are shared with torch).
    import torch  # noqa: F401
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """Owns one rgn_handle. Thin, typed wrappers; tensors are torch CUDA(HIP) tensors, fp32/int64 contiguous."""

    def __init__(self, cfg, max_batch, device_index, precision="f32", options=None):
        """options: {switch: int} for rgn_set_option - the REGENNET_<KEY> kernel-selection switches for THIS engine only."""
        self.lib = load()
        self.cfg = dict(cfg)
        self.max_batch = int(max_batch)
        self.precision = precision
        rc = RgnConfig(
            njoints=cfg["njoints"], nfeats=cfg["nfeats"], num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim"],
            ff_size=cfg["ff_size"], num_heads=cfg["num_heads"], num_layers=cfg["layers"], cm_mode=CM[cfg["cm_mode"]],
            cond_mode=COND[cfg["cond_mode"]], num_actions=int(cfg.get("num_actions", 1)), clip_dim=int(cfg.get("clip_dim", 512)),
            emb_trans_dec=int(bool(cfg.get("emb_trans_dec", False))), wo_pos_emb=int(bool(cfg.get("wo_pos_emb", False))),
            max_batch=self.max_batch, precision=PREC[precision], device=int(device_index))
        h = C.c_void_p()
        code = self.lib.rgn_create(C.byref(rc), C.byref(h))
        if code != RGN_OK:
            raise RgnError(code, (self.lib.rgn_last_error(None) or b"").decode())
        self.h = h
        self.schedule_id = None
        self.cond_key = None
        for k, v in (options or {}).items():
            self._ck(self.lib.rgn_set_option(self.h, str(k).encode(), int(v)))

    def _ck(self, code):
        if code != RGN_OK:
            raise RgnError(code, (self.lib.rgn_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rgn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def load_weight(self, key, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self._ck(self.lib.rgn_load_weight(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    def finalize(self):
        self._ck(self.lib.rgn_finalize_weights(self.h))

    def weight_blob(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.lib.rgn_weight_blob(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- schedule / condition ----------------------------------------------------------------------
    def set_schedule(self, timestep_map, tables, sched_id=None):
        keep = []

        def dptr(name):
            a = np.ascontiguousarray(tables[name], dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))

        tm = np.ascontiguousarray(timestep_map, dtype=np.int64)
        s = RgnSchedule(S=len(tm), timestep_map=tm.ctypes.data_as(C.POINTER(C.c_int64)),
                        posterior_mean_coef1=dptr("posterior_mean_coef1"), posterior_mean_coef2=dptr("posterior_mean_coef2"),
                        model_log_variance=dptr("model_log_variance"),
                        sqrt_recip_alphas_cumprod=dptr("sqrt_recip_alphas_cumprod"),
                        sqrt_recipm1_alphas_cumprod=dptr("sqrt_recipm1_alphas_cumprod"),
                        alphas_cumprod=dptr("alphas_cumprod"), alphas_cumprod_prev=dptr("alphas_cumprod_prev"))
        self._ck(self.lib.rgn_set_schedule(self.h, C.byref(s)))
        self.schedule_id = sched_id

    def set_condition(self, B, cmotion, action, text_feat, scale, stream):
        self._ck(self.lib.rgn_set_condition(self.h, int(B), _ptr(cmotion), _ptr(action), _ptr(text_feat), _ptr(scale),
                                            C.c_void_p(stream)))

    # ---- compute -----------------------------------------------------------------------------------
    def denoise(self, x, t, flags, out, stream):
        self._ck(self.lib.rgn_denoise(self.h, _ptr(x), _ptr(t), int(flags), _ptr(out), C.c_void_p(stream)))

    def sample_range(self, sampler, guided, eta, x, noise, seed, sample_offset, first_index, count, x0_out, use_graph,
                     clip_denoised, stream):
        self._ck(self.lib.rgn_sample_range(self.h, SAMPLER[sampler], int(bool(guided)), float(eta), _ptr(x), _ptr(noise),
                                           int(seed) & (2 ** 64 - 1), int(sample_offset), int(first_index), int(count),
                                           _ptr(x0_out), int(bool(use_graph)), int(bool(clip_denoised)), C.c_void_p(stream)))

    def set_x3_tail(self, tail_steps):
        """Precision schedule ('bf16_x3tail'): split-bf16 for the last `tail_steps` loop indices (-1: default)."""
        self._ck(self.lib.rgn_set_x3_tail(self.h, int(tail_steps)))

    def set_f16_steps(self, steps):
        """Precision schedule: the `steps` plain loop indices in front of the split-bf16 tail run on fp16 MFMA operands (-1: default 8; 0: none)."""
        self._ck(self.lib.rgn_set_f16_steps(self.h, int(steps)))

    def precision_plan(self, B, guided=False):
        """(f16_steps, x3_tail) rgn_sample_range will use for B motions on the bound schedule: loop indices < x3_tail split-bf16, the next f16_steps
        plain fp16, the rest plain bf16."""
        n16, tail = _i32(), _i32()
        self._ck(self.lib.rgn_precision_plan(self.h, int(B), int(bool(guided)), C.byref(n16), C.byref(tail)))
        return n16.value, tail.value

    def set_const_noise(self, on):
        """p_sample's const_noise (gaussian_diffusion.py:544-547) for the following sample_range calls."""
        self._ck(self.lib.rgn_set_const_noise(self.h, int(bool(on))))

    def set_small_batch_rows(self, rows):
        """Evaluations of at most `rows` token rows run the small-batch (column-split) kernels; -1: default, 0: off."""
        self._ck(self.lib.rgn_set_small_batch_rows(self.h, int(rows)))

    def set_option(self, key, value):
        """rgn_set_option on the live handle: before finalize any switch; afterwards only the dispatch rules that may change between calls
        ("LAYERS_GUIDED": 0 an evaluation per workgroup | 1 by batch size | 2 a motion per workgroup | -1 the engine's default)."""
        self._ck(self.lib.rgn_set_option(self.h, str(key).encode(), int(value)))

    def set_layers_min_b(self, samples):
        """Evaluations of at least `samples` samples (<= 64 tokens) run the one-kernel decoder stack (k_layers); -1: default (64)."""
        self._ck(self.lib.rgn_set_layers_min_b(self.h, int(samples)))

    def plan_query(self, B, guided=False, split_phase=False):
        """The engine's plan for one denoiser evaluation of B motions: {class: dict(kernel, launches_per_eval, flops, l2_bytes)} for the
        classes that take part (rgn_plan_query: filled by the dispatching code itself)."""
        out = {}
        for i in range(32):
            name, kern, n, fl, l2 = C.c_char_p(), C.c_char_p(), C.c_double(), C.c_double(), C.c_double()
            code = self.lib.rgn_plan_query(self.h, int(B), int(bool(guided)), int(bool(split_phase)), i, C.byref(name), C.byref(kern),
                                           C.byref(n), C.byref(fl), C.byref(l2))
            if code != RGN_OK:
                break
            if kern.value:
                out[name.value.decode()] = {"kernel": kern.value.decode(), "launches_per_eval": n.value, "flops": fl.value, "l2_bytes": l2.value}
        return out

    def randn(self, x, B, seed, sample_offset, stream):
        self._ck(self.lib.rgn_randn(self.h, _ptr(x), int(B), int(seed) & (2 ** 64 - 1), int(sample_offset), C.c_void_p(stream)))

    def randn_step(self, x, B, seed, sample_offset, loop_index, stream):
        """The fused loop's noise of loop index `loop_index` (-1: x_T) into x [B,njoints,nfeats,T]."""
        self._ck(self.lib.rgn_randn_step(self.h, _ptr(x), int(B), int(seed) & (2 ** 64 - 1), int(sample_offset), int(loop_index), C.c_void_p(stream)))

    def rot6d_to_matrix(self, d6, mat, n, stream):
        self._ck(self.lib.rgn_rot6d_to_matrix(self.h, _ptr(d6), _ptr(mat), int(n), C.c_void_p(stream)))

    def gaussian_filter1d(self, x, out, rows, T, sigma, stream):
        self._ck(self.lib.rgn_gaussian_filter1d(self.h, _ptr(x), _ptr(out), int(rows), int(T), float(sigma), C.c_void_p(stream)))

    def profile_enable(self, on):
        self._ck(self.lib.rgn_profile_enable(self.h, int(bool(on))))

    def profile_bracket_overhead_ms(self):
        ms = C.c_double()
        self._ck(self.lib.rgn_profile_bracket_overhead(self.h, C.byref(ms)))
        return ms.value

    def profile_query(self):
        out = {}
        for i in range(16):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
            code = self.lib.rgn_profile_query(self.h, i, C.byref(name), C.byref(ms), C.byref(n))
            if code != RGN_OK:
                break
            out[name.value.decode()] = (ms.value, n.value)
        return out


class StgcnEngine:
    """Owns one rgn_stgcn_handle (the ST-GCN evaluator, include/regennet_hip.h)."""

    def __init__(self, in_channels, num_class, num_person, num_nodes, num_frames, max_batch, device_index, options=None):
        self.lib = load()
        self.shape = (int(num_nodes), int(in_channels), int(num_frames))
        self.num_class, self.max_batch = int(num_class), int(max_batch)
        cfg = RgnStgcnConfig(in_channels=in_channels, num_class=num_class, num_person=num_person, num_nodes=num_nodes,
                             num_frames=num_frames, max_batch=max_batch, device=int(device_index))
        h = C.c_void_p()
        code = self.lib.rgn_stgcn_create(C.byref(cfg), C.byref(h))
        if code != RGN_OK:
            raise RgnError(code, (self.lib.rgn_stgcn_last_error(None) or b"").decode())
        self.h = h
        for k, v in (options or {}).items():         # kernel-selection switches of this handle (rgn_stgcn_set_option)
            self.set_option(k, v)

    def _ck(self, code):
        if code != RGN_OK:
            raise RgnError(code, (self.lib.rgn_stgcn_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rgn_stgcn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weight(self, key, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        self._ck(self.lib.rgn_stgcn_load_weight(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    def finalize(self):
        self._ck(self.lib.rgn_stgcn_finalize(self.h))

    def set_option(self, key, value):
        self._ck(self.lib.rgn_stgcn_set_option(self.h, key.encode(), int(value)))

    def forward(self, N, output, features, yhat, stream):
        self._ck(self.lib.rgn_stgcn_forward(self.h, int(N), _ptr(output), _ptr(features), _ptr(yhat), C.c_void_p(stream)))
