"""CPU, only where /root/reference exists: the libtorch-facing declarations of include/lfs_gsplat_torch.hpp against the TEXT of the reference's
gsplat/Ops.h and fastgs/optimizer/include/{adam_api.h, adam.h} - function by function, the return type and the ordered list of parameter
types and names must be the same (comments, whitespace and the `OptT` / `at::optional<at::Tensor>` spelling aside). The reference header itself
needs glm to compile; this is the check that the restated declarations did not drift from it."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "gsplat", "Ops.h")), reason="reference tree not present")


def _strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return text


def _declarations(text):
    """{function name: (return type, [(type, name), ...])} of the free-function declarations in `text`"""
    text = _strip(text)
    out = {}
    for m in re.finditer(r"([\w:<>,\s&\*]+?)\b(\w+)\s*\(([^;{}()]*(?:\([^()]*\)[^;{}()]*)*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ret = re.sub(r"\s+", " ", ret).strip()
        ret = re.sub(r"^(.*\b(inline|static|extern|LFS_API)\b)", "", ret).strip()
        if not ret or ret.endswith(("return", "namespace", "using")) or name in ("defined", "static_assert"):
            continue
        plist = []
        for p in _split_params(params):
            p = re.sub(r"=\s*[^,]+$", "", p).strip()   # default values
            if not p:
                continue
            t, n = p.rsplit(None, 1) if not p.endswith(("&", "*")) else (p, "")
            if n.startswith(("&", "*")):
                t, n = t + n[0], n[1:]
            plist.append((_norm_type(t), n))
        out[name] = (_norm_type(ret), plist)
    return out


def _split_params(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "<(":
            depth += 1
        elif ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def _norm_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = t.replace("OptT", "at::optional<at::Tensor>").replace("gsplat::", "")
    t = re.sub(r"\s*([<>,&\*])\s*", r"\1", t)
    return t


OURS = _declarations(open(os.path.join(ROOT, "include", "lfs_gsplat_torch.hpp")).read())


@pytest.mark.parametrize("ref_file,names", [
    ("gsplat/Ops.h", ["spherical_harmonics_fwd", "spherical_harmonics_bwd", "intersect_tile", "intersect_offset", "quats_to_rotmats", "relocation", "add_noise",
                      "projection_ut_3dgs_fused", "rasterize_to_pixels_from_world_3dgs_fwd", "rasterize_to_pixels_from_world_3dgs_bwd"]),
    ("fastgs/optimizer/include/adam_api.h", ["adam_step_wrapper"]),
    ("fastgs/optimizer/include/adam.h", ["adam_step"]),
])
def test_declarations_match_the_reference_header_text(ref_file, names):
    ref = _declarations(open(os.path.join(REF, ref_file)).read())
    for name in names:
        assert name in ref, (name, sorted(ref))
        assert name in OURS, name
        r_ret, r_par = ref[name]
        o_ret, o_par = OURS[name]
        assert o_ret == r_ret, (name, o_ret, r_ret)
        assert [t for t, _ in o_par] == [t for t, _ in r_par], (name, o_par, r_par)
        # parameter names too - except adam_api.h's last two, which the reference NAMES bias_correction1 / bias_correction2_sqrt but USES as
        # reciprocals (adam_api.cu:17-18, fused_adam.cpp:78-79): ours say what they are
        if name != "adam_step_wrapper":
            assert [n for _, n in o_par] == [n for _, n in r_par], (name, o_par, r_par)
