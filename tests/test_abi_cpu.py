"""CPU: the C-ABI library loads, exports every symbol include/regennet_hip.h declares, and fails loudly
(no silent CPU fallback) when no GPU is present. No compute calls."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "regennet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgn_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from regennet_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    assert sorted(_lib.SYMBOLS) == declared, (sorted(_lib.SYMBOLS), declared)
    for name in declared:
        assert getattr(lib, name) is not None
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        getattr(raw, name)


def test_the_c_abi_is_the_only_dynamic_symbol_set():
    """Built with -fvisibility=hidden + a linker version script (regennet_amd/csrc/exports.map): `nm -D --defined-only` lists exactly
    the entry points include/regennet_hip.h declares - no rgn::launch_*, no __device_stub__*, no weak libstdc++ instantiations."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    from regennet_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    rows = [ln.split() for ln in out.splitlines() if ln.strip()]
    exported = sorted(r[-1] for r in rows)
    assert exported == _declared_symbols(), sorted(set(exported) ^ set(_declared_symbols()))
    assert all(r[-2] == "T" for r in rows), [r for r in rows if r[-2] != "T"]


def test_direct_to_lds_loads_find_their_m0_in_the_isa():
    """rgn_layers.hip issues direct-to-LDS loads both through the compiler's builtin and from inline asm (the step boundary's L2 prefetch); m0 - their LDS
    base - is a register the compiler manages and does not preserve across an asm that names it as a clobber. tools/check_m0.py compiles the source to
    ISA and checks every such load for an m0 write inside its own basic block, the asm's save / restore pairing, and a warning-free compile."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_m0.py"), "rgn_layers.hip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) direct-to-LDS loads checked, 0 problem\(s\)", r.stdout)
    assert m and int(m.group(1)) >= 60, r.stdout                          # (8 - 16 per instantiation of k_layers)


def test_null_and_bad_arguments_return_status_codes():
    from regennet_amd import _lib
    lib = _lib.load()
    assert lib.rgn_create(None, None) == -1
    assert b"null" in lib.rgn_last_error(None)
    assert lib.rgn_destroy(None) == -1
    assert lib.rgn_finalize_weights(None) == -1
    assert lib.rgn_set_layers_min_b(None, 1) == -1
    assert lib.rgn_set_f16_steps(None, 1) == -1
    assert lib.rgn_precision_plan(None, 1, 0, None, None) == -1
    assert lib.rgn_plan_query(None, 1, 0, 0, 0, None, None, None, None, None) == -1
    cfg = _lib.RgnConfig(njoints=56, nfeats=6, num_frames=60, latent_dim=500, ff_size=1024, num_heads=4, num_layers=8,
                         cm_mode=1, cond_mode=0, num_actions=1, clip_dim=512, emb_trans_dec=0, wo_pos_emb=0, max_batch=1,
                         precision=0, device=0)
    h = ctypes.c_void_p()
    assert lib.rgn_create(ctypes.byref(cfg), ctypes.byref(h)) == -7      # latent_dim not 64*2^k
    cfg.latent_dim = 512
    cfg.cm_mode = 9
    assert lib.rgn_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_fallback():
    from regennet_amd import _lib, synth
    with pytest.raises(_lib.RgnError) as e:
        _lib.Engine(synth.get_config("tiny"), 1, 0, "f32")
    assert e.value.code == -6
    from tests.helpers import build_hip
    cfg = synth.get_config("tiny")
    model, diffusion = build_hip(cfg, synth.make_state_dict(cfg), device="cpu")
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, 1))}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 5, 6, 8), torch.zeros(1, dtype=torch.long), y=y)
    with pytest.raises(TypeError):
        diffusion.p_sample_loop(lambda *a, **k: None, (1, 5, 6, 8), model_kwargs={"y": y})
