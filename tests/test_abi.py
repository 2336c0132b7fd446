"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/cdcmdr.h declares,
and the ctypes structures of _lib.py have the sizes and field offsets the C compiler gives the header's structs.
No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cdcmdr.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cdc_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from cdcmdr_amd import _lib, build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    assert lib.cdc_abi_version() == 1
    names = declared_functions()
    assert len(names) >= 30
    raw = C.CDLL(path)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/cdcmdr.h but not exported by libcdcmdr.so"
    # and the binding knows every declared function (nothing callable only from C)
    missing = sorted(set(names) - set(_lib.exported_symbols()))
    assert not missing, f"declared but not bound in _lib.py: {missing}"
    extra = sorted(set(_lib.exported_symbols()) - set(names))
    assert not extra, f"bound in _lib.py but not declared in the header: {extra}"


STRUCTS = {
    "cdc_adam_hp": "AdamHP", "cdc_lin_group": "LinGroup", "cdc_lin_fwd_args": "LinFwdArgs", "cdc_bwdx_seg": "BwdxSeg",
    "cdc_bwdx_out": "BwdxOut", "cdc_lin_bwdx_args": "LinBwdxArgs", "cdc_bwdw_group": "BwdwGroup",
    "cdc_lin_bwdw_args": "LinBwdwArgs", "cdc_pool_fwd_args": "PoolFwdArgs", "cdc_pool_bwd_args": "PoolBwdArgs",
    "cdc_bn_seg": "BnSeg", "cdc_bn_fwd_args": "BnFwdArgs", "cdc_bn_bseg": "BnBSeg", "cdc_bn_bwd_args": "BnBwdArgs",
    "cdc_rowdot_group": "RowdotGroup", "cdc_rowdot_fwd_args": "RowdotFwdArgs", "cdc_rowdot_bgroup": "RowdotBGroup",
    "cdc_rowdot_bwd_args": "RowdotBwdArgs", "cdc_adam_tensor": "AdamTensor", "cdc_adam_args": "AdamArgs",
    "cdc_star_fuse_args": "StarFuseArgs", "cdc_transpose_args": "TransposeArgs", "cdc_add_n_args": "AddNArgs",
    "cdc_g2_out": "G2Out", "cdc_g2_seg": "G2Seg", "cdc_g2_args": "G2Args", "cdc_wshadow_args": "WShadowArgs",
    "cdc_shadow_args": "ShadowArgs", "cdc_head_tower": "HeadTower", "cdc_head_args": "HeadArgs",
    "cdc_mid_gate1": "MidGate1", "cdc_mid_expert2": "MidExpert2", "cdc_mid_gate2": "MidGate2", "cdc_cgc_mid_fwd_args": "CgcMidFwdArgs",
    "cdc_tower_layer": "TowerLayer", "cdc_tower_desc": "TowerDesc", "cdc_tower_args": "TowerArgs",
    "cdc_mid_bgate1": "MidBGate1", "cdc_mid_bexpert2": "MidBExpert2", "cdc_mid_bgate2": "MidBGate2", "cdc_cgc_mid_bwd_args": "CgcMidBwdArgs",
}


def test_ctypes_layouts_match_the_header():
    """sizeof and the offset of the LAST field of every struct, as gcc lays the header out, vs ctypes."""
    from cdcmdr_amd import _lib
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, pyname in STRUCTS.items():
        cls = getattr(_lib, pyname)
        last = cls._fields_[-1][0]
        lines.append(f'  printf("{cname} %zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));')
    lines += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "abi.c"), os.path.join(td, "abi")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-std=c11", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        cname, size, off = line.split()
        cls = getattr(_lib, STRUCTS[cname])
        last = cls._fields_[-1][0]
        assert C.sizeof(cls) == int(size), f"{cname}: C sizeof {size} != ctypes {C.sizeof(cls)}"
        assert getattr(cls, last).offset == int(off), f"{cname}.{last}: C offset {off} != ctypes {getattr(cls, last).offset}"
        assert C.sizeof(cls) <= 4096, f"{cname} travels as a kernel argument and must stay under 4 KB"


def test_limits_match_the_header():
    from cdcmdr_amd import _lib
    src = open(HEADER).read()
    for macro, val in [("CDC_MAX_GROUPS", _lib.MAX_GROUPS), ("CDC_MAX_TENSORS", _lib.MAX_TENSORS), ("CDC_MAX_GATES", _lib.MAX_GATES),
                       ("CDC_MAX_SEL", _lib.MAX_SEL), ("CDC_MAX_BN_SEGS", _lib.MAX_BN_SEGS), ("CDC_SORT_MAX_B", _lib.SORT_MAX_B), ("CDC_SORT_MAX_ROWS", _lib.SORT_MAX_ROWS),
                       ("CDC_BN_ROWS_PER_BLOCK", _lib.BN_ROWS_PER_BLOCK), ("CDC_ROWDOT_PARTS", _lib.ROWDOT_PARTS),
                       ("CDC_G2_MAX_OUT", _lib.G2_MAX_OUT), ("CDC_G2_MAX_SEG", _lib.G2_MAX_SEG), ("CDC_HEAD_MAX_TOWERS", _lib.HEAD_MAX_TOWERS),
                       ("CDC_TOWER_MAX", _lib.TOWER_MAX), ("CDC_TOWER_ROWS", _lib.TOWER_ROWS),
                       ]:
        m = re.search(rf"#define\s+{macro}\s+(\d+)", src)
        assert m and int(m.group(1)) == val, macro


def test_product_path_refuses_to_run_without_a_gpu():
    """No CPU fallback: a forward on a CPU-resident model raises instead of silently computing something else."""
    import torch
    from cdcmdr_amd import _lib
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.optim import FusedAdam
    m = PLE([5, 6, 7], 4, 3, 1, 1, ((8,), (4,)), (4,), dropout=0.0)
    with pytest.raises(_lib.HipExtensionError):
        m(torch.zeros(2, 3, dtype=torch.int32))
    with pytest.raises(_lib.HipExtensionError):
        FusedAdam(m)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "causal-domain-clustering-for-multi-domain-recommendation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                hit = re.search(r"(^|\n)\s*(import|from)\s+oracle|cdc_oracle|libadam_elem_ref|oracle/_build", text)
                assert not hit, f"{f} uses the oracle ({hit.group(0).strip()!r}): the product path must not"


def test_mirror_state_dict_keys_equal_the_reference_goldens():
    """state_dict keys (names AND shapes) of every mirrored model == the reference's, from the golden fixtures."""
    import types
    import numpy as np
    from cdcmdr_amd.model.ple import PLE
    from cdcmdr_amd.model.mmoe import MMoE
    from cdcmdr_amd.model.dcn import DCN
    from cdcmdr_amd.model.dcnv2 import DCNv2
    from cdcmdr_amd.model.star import STAR
    FD = [7, 100, 3, 50, 11, 29]
    FD13 = [11, 50, 7, 100, 3, 29, 64, 5, 17, 200, 9, 31, 13]
    cases = {
        "g2_ple3": PLE(FD, 4, 3, 2, 2, ((32, 16), (8,)), (8, 4), dropout=0.0),
        "g2_mmoe8": MMoE(FD, 4, 3, 8, (32, 16, 8), (8, 4), dropout=0.0),
        "g2_dcn13": DCN(FD13, 4, 3, (32, 16, 8), dropout=0.0),
        "g2_dcnv2_mix": DCNv2(FD13, 4, 3, (32, 16, 8), dropout=0.0, low_rank=8, num_experts=4),
        "g2_star30_all": STAR(FD, 4, 30, (16, 8), dropout=0.0),
    }
    gold = os.path.join(ROOT, "tests", "golden")
    for name, model in cases.items():
        d = np.load(os.path.join(gold, name + ".npz"))
        want = {k[3:]: d[k].shape for k in d.files if k.startswith("sd/")}
        got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert got == want, f"{name}: state_dict differs from the reference"
    # the regularisation registry covers the same tensors as the reference's filters (F9: MLP BatchNorm gammas included)
    from oracle import cdc_oracle as O
    m = cases["g2_ple3"]
    names = {id(p): n for n, p in m.named_parameters()}
    reg = sorted(names[id(p)] for p, _, l2 in m.regularized_parameters() if l2 > 0)
    assert reg == sorted(O.reg_names(list(m.state_dict()), "ple"))
    assert "towers.0.layers.1.weight" in reg           # a BatchNorm gamma, regularised by the reference's name filter


def test_every_model_of_the_reference_registry_constructs():
    """run.py:15-26 imports twelve model classes; all of them are mirrored (no placeholder left)."""
    import types
    from cdcmdr_amd.model import adasparse, adl, autoint, dfm, hinet, pepnet
    fd = [5, 6, 7]
    off = types.SimpleNamespace(use_atten=False, use_dcn=False)
    assert dfm.DeepFM(fd, 4, (8,)).model_name == "deepfm"
    assert autoint.AutoInt(fd, 4, 8, 1, 2, True, (8,)).model_name == "autoint"
    assert hinet.HiNet(fd, 4, n_tower=2, sei_dims=(8, 4), tower_dims=(4,), domain_idx=1, config=off).model_name == "hinet"
    assert adasparse.AdaSparse(fd, 4, (8,), domain_idx=1, config=off).model_name == "adasparse"
    assert pepnet.PEPNet(fd, 4, 2, (8,), 4, 1, True, 0.0, off).model_name == "pepnet"
    assert adl.ADL(fd, 4, 2, (8,), domain_idx=1, device="cpu", config=off).model_name == "adl"


def test_compat_package_serves_the_reference_import_names():
    """`from model.ple import PLE` (run.py:15-26) resolves to the mirror when <repo>/compat is on sys.path."""
    code = ("import sys; sys.path[:0] = [%r, %r]; "
            "from model.dfm import DeepFM; from model.dcn import DCN; from model.dcnv2 import DCNv2; from model.autoint import AutoInt; "
            "from model.ple import PLE; from model.mmoe import MMoE; from model.pepnet import PEPNet; from model.star import STAR; "
            "from model.cdc import CDC; from model.adl import ADL; from model.hinet import HiNet; from model.adasparse import AdaSparse; "
            "from model.layer import BaseModel, FeaturesEmbedding, FeaturesLinear, MultiLayerPerceptron, DNN, CrossNetwork, CrossNetV2, CrossNetMix; "
            "import cdcmdr_amd.model.ple as m; assert PLE is m.PLE; print('ok')") % (os.path.join(ROOT, "compat"), ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_entry_points_reject_bad_arguments_without_touching_the_device():
    """Every entry point validates its arguments before any launch: a negative return code and a message from
    cdc_last_error(), no GPU needed (and none used: this runs in the CPU-only container)."""
    from cdcmdr_amd import _lib
    lib = _lib.load()
    rc = lib.cdc_embed_gather_fwd(None, None, None, None, None, None, 4, 3, 8, 100, None)
    assert rc < 0 and b"null pointer" in lib.cdc_last_error()
    one = C.c_void_p(16)                                    # a non-null address that is never dereferenced on the host
    rc = lib.cdc_embed_gather_fwd(one, one, one, one, None, None, 4, 0, 8, 100, None)
    assert rc < 0 and b"bad sizes" in lib.cdc_last_error()
    rc = lib.cdc_embed_sort_dedupe(one, one, one, one, one, None, 40000, 3, None)
    assert rc < 0 and b"exceeds" in lib.cdc_last_error()
    rc = lib.cdc_embed_sort_dedupe(one, one, one, one, one, None, 20000, 3, None)
    assert rc < 0 and b"scratch" in lib.cdc_last_error()
    a = _lib.LinFwdArgs()
    a.n_groups = 0
    assert lib.cdc_glinear_fwd(C.byref(a), 0, None) < 0
    a.n_groups = 1
    a.drop_p = 1.5
    assert lib.cdc_glinear_fwd(C.byref(a), 0, None) < 0 and b"dropout" in lib.cdc_last_error()
    assert lib.cdc_bce_fwd_bwd(None, 1, None, None, None, None, None, 1, 4, 1, 1.0, None) < 0
    assert lib.cdc_eval_metrics(one, one, None, 0, 10, 3, one, one, None, one, 1 << 20, None) < 0      # n_domain > 1 needs the domain column
    assert lib.cdc_shard_bucket(one, one, one, one, one, 64, 3, 99, 8, None) < 0                       # more ranks than supported
    # a replay SLICE (rows chosen on the device) beyond the 2^31 work items one launch counts is refused (CDC_E_TOOBIG), not wrapped;
    # a whole-table flush of that size goes out as row windows (tests/test_gpu_configs.py)
    hp = _lib.AdamHP()
    hp.step_scalars, hp.n_scalars = 16, 4
    rc = lib.cdc_embed_lazy_flush(one, one, one, one, 1 << 40, 4, hp, one, 0, 2, 0, 0, None)
    assert rc == -2 and b"exceeds the 2^31 work items" in lib.cdc_last_error()
    assert lib.cdc_glinear_bwd_w_pair_reduce(None, None, None, None) == -1 and b"null argument" in lib.cdc_last_error()
    # the rows + dense-parameter update in one launch needs the dense descriptor table
    rc = lib.cdc_embed_segsum_lazy_update_dense(one, one, one, one, one, one, one, one, one, hp, one, 8, 2, 4, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"descriptor table" in lib.cdc_last_error()
    # the fused PLE level boundary's LDS budget, as plan.CGCMid.match asks it: 3 domains fit, 6 and 7 domains (2 + 2 experts) do not
    assert lib.cdc_cgc_mid_fits(8, 4, 8, 3) == 1 and lib.cdc_cgc_mid_fits(12, 6, 12, 5) == 1
    assert lib.cdc_cgc_mid_fits(14, 7, 14, 6) == 0 and lib.cdc_cgc_mid_fits(16, 8, 16, 7) == 0
    assert lib.cdc_cgc_mid_fits(0, 1, 1, 1) == -1 and lib.cdc_cgc_mid_fits(17, 1, 1, 1) == -1
    with pytest.raises(RuntimeError):
        _lib.check(-1, "probe")


def test_fused_tower_launch_refuses_bad_arguments_without_a_launch():
    from cdcmdr_amd import _lib as L
    lib = L.load()
    a = L.TowerArgs()
    assert lib.cdc_tower_workspace_bytes(C.byref(a)) == -1
    a.n_tower, a.H0, a.H1, a.H2, a.M = 3, 64, 64, 32, 4096
    assert lib.cdc_tower_workspace_bytes(C.byref(a)) > 4096
    assert lib.cdc_tower_fwd(C.byref(a), None) == -1                     # no workspace / buffers
    a.H0 = 96
    assert lib.cdc_tower_fwd(C.byref(a), None) == -1 and b"instantiated" in lib.cdc_last_error()
    a.H0, a.M = 64, 1
    assert lib.cdc_tower_bwd(C.byref(a), None) == -1 and b"two rows" in lib.cdc_last_error()
    a.M = 128 * 100
    assert lib.cdc_tower_fwd(C.byref(a), None) == -2 and b"resident" in lib.cdc_last_error()
    assert lib.cdc_tower_step(C.byref(a), None) == -2 and b"tower_step" in lib.cdc_last_error()
    a.M = 1
    assert lib.cdc_tower_step(C.byref(a), None) == -1 and b"two rows" in lib.cdc_last_error()
