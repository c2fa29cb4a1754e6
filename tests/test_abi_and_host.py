"""CPU-side checks: the C-ABI library builds/loads and exports exactly what include/adaface_hip.h declares,
the ctypes struct mirrors the C struct, and the host-side weight re-layout is what the kernels expect.
No compute calls are made (no GPU here)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "adaface_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(af_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from adaface_dev_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_library_exports_every_declared_symbol(lib):
    declared = header_functions()
    assert len(declared) >= 18
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (af_[a-z0-9_]+)", out)))
    assert exported == declared, (set(declared) ^ set(exported))
    assert sorted(lib.EXPORTS) == declared
    L = lib.lib()
    for name in declared:
        getattr(L, name)
    assert L.af_version() >= 100


def test_bad_arguments_return_codes_without_a_gpu(lib):
    """Argument validation happens before any launch, so it is observable on CPU."""
    L = lib.lib()
    assert L.af_gemm(None, None) == lib.AF_E_BADARG
    assert b"null descriptor" in L.af_last_error()
    d = lib.GemmDesc()
    assert L.af_gemm(ctypes.byref(d), None) == lib.AF_E_BADARG
    assert L.af_layernorm(None, None, None, None, 4, 320, 1e-5, None) == lib.AF_E_BADARG
    assert L.af_groupnorm_ws_floats(8) == 8 * 128 * 32 * 2        # [B][<=128 partial blocks][32 groups][sum, sumsq]


def test_gemm_desc_matches_c_struct(lib, tmp_path):
    """sizeof/offsetof of af_gemm_desc as compiled by gcc == the ctypes mirror."""
    fields = [f[0] for f in lib.GemmDesc._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"adaface_hip.h\"\nint main(){printf(\"%zu\\n\", sizeof(af_gemm_desc));\n"
    for f in fields:
        prog += f'printf("%zu\\n", offsetof(af_gemm_desc, {f}));\n'
    prog += "return 0;}\n"
    c = tmp_path / "t.c"
    c.write_text(prog)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(lib.GemmDesc)
    assert vals[1:] == [getattr(lib.GemmDesc, f).offset for f in fields]


def test_weight_packing_layouts():
    from adaface_dev_amd import ops
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)          # [Cout=2, Cin=3, 3, 3]
    pw = ops.pack_conv3x3(w, None, "cpu", cin_pad=8)
    assert pw.wt.shape == (128, 128) and pw.K == 72 and pw.cin == 8 and pw.taps == 9
    k = (1 * 3 + 2) * 8 + 1                                                             # (ky=1, kx=2, cin=1)
    assert float(pw.wt[1, k]) == float(w[1, 1, 1, 2])
    assert float(pw.wt[:, 72:].abs().max()) == 0 and float(pw.wt[2:].abs().max()) == 0  # zero padding
    wg, bg = torch.arange(64 * 4, dtype=torch.float32).reshape(64, 4), torch.arange(64, dtype=torch.float32)
    wi, bi = ops.interleave_geglu(wg, bg)
    assert torch.equal(wi[:16], wg[:16]) and torch.equal(wi[16:32], wg[32:48]) and torch.equal(wi[32:48], wg[16:32])
    assert torch.equal(bi[16:32], bg[32:48])
    kb = ops.make_keybias(torch.tensor([[1, 0, 1]]), 3)
    assert kb.shape == (1, 64) and kb[0, 1] == -torch.finfo(torch.float32).max and kb[0, 0] == 0


def test_module_tree_matches_reference_names():
    """SD-1.5 U-Net: 686 tensors / 859,520,964 params (SURVEY.md section 4 known answers); layers refuse to run on CPU."""
    from adaface_dev_amd import SD15_UNET_CONFIG
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel, unet_param_shapes
    from oracle.unet_oracle import unet_topology
    shapes = dict(unet_param_shapes(SD15_UNET_CONFIG))
    assert len(shapes) == 686 and sum(int(np.prod(s)) for s in shapes.values()) == 859520964
    assert shapes["input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert shapes["output_blocks.11.0.skip_connection.weight"] == (320, 640, 1, 1)
    ins, mid, outs = unet_topology(SD15_UNET_CONFIG)
    ca = [i for i, b in enumerate(ins + [mid] + outs) if any(k == "attn" for k, _ in b)]
    assert ca == UNetModel.ALL_CA_LAYER_INDICES                                        # openaimodel.py:721
    from adaface_dev_amd.ldm.modules.diffusionmodules.util import linear
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        linear(8, 8)(torch.zeros(2, 8))


def test_ddim_host_schedule():
    from adaface_dev_amd.ldm.models.diffusion.ddim import DDIMSampler
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd import TINY_UNET_CONFIG
    g = np.load(os.path.join(ROOT, "tests", "golden", "schedule.npz"))
    ld = LatentDiffusion(TINY_UNET_CONFIG)
    assert np.array_equal(ld.alphas_cumprod.numpy(), g["alphas_cumprod"].astype(np.float32))
    s = DDIMSampler(ld)
    s.make_schedule(50, verbose=False)
    assert np.array_equal(s.ddim_timesteps, g["ddim_timesteps"])
    assert np.array_equal(s.ddim_alphas, g["ddim_alphas"]) and np.array_equal(s.ddim_alphas_prev, g["ddim_alphas_prev"])
    sc = s.guide_scales(50, (4.0, 1.0))
    assert sc[0] == 4.0 and abs(sc[-1] - 1.0) < 1e-9 and len(sc) == 50


def test_halo_kernel_lds_swizzle_is_conflict_free_under_the_real_ds_read_b128_lane_groups():
    """The halo-resident 3x3 kernel XORs the 16-byte chunk index of a halo pixel with pixel & 7 (af_gemm3.hip).  Under the lane groups a ds_read_b128 is
    really served in (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, ...) that swizzle must cost no extra LDS cycle at ANY fragment start (every tap shifts
    the start by one pixel); the swizzle of rounds 3 - 5, derived for groups of 16 consecutive lanes, costs one per group at three of four starts
    (SQ_LDS_BANK_CONFLICT 0.24 of the kernel's LDS cycles, profiles/r05ac, 0.03 after the change, r05ad)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("lds_swizzle_check", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "lds_swizzle_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.score(lambda hp: hp & 7) == 0
    assert m.score(lambda hp: (hp >> 1) & 7) == 384
    assert m.score(lambda hp: 0) > 384
