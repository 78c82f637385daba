"""The MEX gateways (mex/*.cpp) and the MATLAB shadows (matlab/*.m) have never seen a MATLAB: the image has none
(SURVEY 0 / 8b).  What can be checked here is checked: the gateways parse and type-check as C++17 against the
library's own header and a hand-written stand-in for mex.h that declares exactly the mx* / mex* functions they use
(tests/stubs/mex.h), and every shadow keeps the call signature of the reference file it shadows.  A syntax check of
~600 lines that were never compiled -- not parity evidence."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_mex_gateways_are_well_formed_cxx():
    srcs = sorted(glob.glob(os.path.join(ROOT, "mex", "*.cpp")))
    assert len(srcs) >= 8, srcs
    for src in srcs:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-unused-function",
                            "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + os.path.join(ROOT, "include"), src],
                           capture_output=True, text=True)
        assert r.returncode == 0, f"{os.path.basename(src)}:\n{r.stderr}"
        assert "warning" not in r.stderr, f"{os.path.basename(src)}:\n{r.stderr}"
    # every symbol the gateways call exists in the C ABI header
    hdr = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    used = set()
    for src in srcs + [os.path.join(ROOT, "mex", "vo_mex_common.h")]:
        used |= set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", open(src).read()))
    used = {s for s in used if not s.endswith("_mex")} - {"vo_mex_ctx", "vo_mex_check", "vo_mex_cleanup", "vo_opt"}   # (gateway names in usage texts)
    missing = [s for s in sorted(used) if not re.search(r"\b%s\s*\(" % s, hdr)]
    assert not missing, f"gateways call functions include/vo_hip.h does not declare: {missing}"
    # and the stub declares nothing the gateways do not use (it stays a minimal header)
    stub = open(os.path.join(ROOT, "tests", "stubs", "mex.h")).read()
    text = "".join(open(s).read() for s in srcs + [os.path.join(ROOT, "mex", "vo_mex_common.h")])
    declared = set(re.findall(r"\b((?:mx|mex)[A-Z][A-Za-z0-9]*)\s*\(", stub)) - {"mexFunction"}
    unused = [f for f in sorted(declared) if not re.search(r"\b%s\s*\(" % f, text)]
    assert not unused, f"tests/stubs/mex.h declares functions no gateway uses: {unused}"


def _signature(path, func):
    """(outputs, inputs) of `function [outs] = func(ins)` in a .m file"""
    text = re.sub(r"\.\.\.[^\n]*\n", " ", open(path).read())      # MATLAB line continuations
    for line in text.splitlines():
        m = re.match(r"\s*function\s+(?:\[?([^\]=]*?)\]?\s*=\s*)?%s\s*\(([^)]*)\)" % re.escape(func), line)
        if m:
            outs = [o.strip() for o in re.split(r"[,\s]+", (m.group(1) or "").strip()) if o.strip()]
            ins = [i.strip() for i in m.group(2).split(",") if i.strip()]
            return outs, ins
    raise AssertionError(f"no `function ... {func}(...)` line in {path}")


# (shadow file, function, reference file, reference line, outputs, positional inputs, has a trailing name-value struct)
SIGNATURES = [
    ("KLTTracker.m", "KLTTracker", "modules/KLTTracker.m", 9, ["obj"], [], True),
    ("KLTTracker.m", "track", "modules/KLTTracker.m", 22, ["points", "validity", "scores"], ["obj", "img1", "img2", "img1_pts"], False),
    ("selectCandidateKeypoints.m", "selectCandidateKeypoints", "modules/selectCandidateKeypoints.m", 1, ["new_kps_loc"], ["img", "curr_kps"], True),
    ("candidateTriangulation.m", "candidateTriangulation", "modules/candidateTriangulation.m", 1, ["curr_state", "tracked_keypoints"],
     ["prev_img", "prev_state", "curr_img", "curr_pose", "camera_params", "tracker"], True),
    ("calculateAngleDeg.m", "calculateAngleDeg", "utils/calculateAngleDeg.m", 1, ["angle"], ["landmark", "cam_pose0", "cam_pose1"], False),
    ("isInFrontOfCamera.m", "isInFrontOfCamera", "modules/isInFrontOfCamera.m", 1, ["valid_idx"], ["camMat", "points3d"], False),
]


@pytest.mark.parametrize("shadow,func,ref,line,outs,ins,named", SIGNATURES)
def test_matlab_shadows_keep_the_reference_signatures(shadow, func, ref, line, outs, ins, named):
    """Number and order of outputs and of positional inputs, and whether a name-value struct follows (the object's and
    the struct's own names are local to a function and may differ).  The expected values restate the reference's
    signature lines (file:line above); where /root/reference is present they are read from there as well."""
    o, i = _signature(os.path.join(ROOT, "matlab", shadow), func)
    pos = i[:-1] if named else i
    assert len(o) == len(outs) and (func == "KLTTracker" or o == outs), (o, outs)
    assert len(pos) == len(ins) and pos[1 if func == "track" else 0:] == ins[1 if func == "track" else 0:], (pos, ins)
    assert len(i) == len(ins) + (1 if named else 0)
    ref_path = os.path.join("/root/reference", ref)
    if os.path.exists(ref_path):                      # (absent on the GPU box; this test is a CPU test)
        ro, ri = _signature(ref_path, func)
        with open(ref_path) as f:
            assert re.search(r"function.*\b%s\s*\(" % func, f.readlines()[line - 1]), f"{ref}:{line} is not the signature line"
        assert len(ro) == len(o) and len(ri) == len(i)
        assert ri[:len(ins)][1 if func == "track" else 0:] == ins[1 if func == "track" else 0:] and (func == "KLTTracker" or ro == outs)


def test_toolbox_shadows_take_what_the_reference_passes():
    """triangulate and estimateWorldCameraPose are toolbox names: the reference holds no signature line for them, only
    call sites (modules/VisualOdometry.m:97-101 and :285-289).  The shadows take as many positional arguments and
    return as many outputs as those calls use."""
    o, i = _signature(os.path.join(ROOT, "matlab", "triangulate.m"), "triangulate")
    assert len(o) == 3 and len(i) == 4, (o, i)
    o, i = _signature(os.path.join(ROOT, "matlab", "estimateWorldCameraPose.m"), "estimateWorldCameraPose")
    assert len(o) == 4 and len(i) == 4, (o, i)          # three positional arguments + the name-value struct
    ref_path = "/root/reference/modules/VisualOdometry.m"
    if os.path.exists(ref_path):
        text = re.sub(r"\.\.\.[^\n]*\n", " ", open(ref_path).read())
        m = re.search(r"\[([^\]]*)\]\s*=\s*triangulate\(([^;]*)\);", text)
        assert m and len(m.group(1).split(",")) == 3
        args = [a for a in (x.strip() for x in m.group(2).split(",")) if a]
        # (i,:) index expressions hold commas of their own: count top-level arguments
        depth = 0; n = 1
        for ch in m.group(2):
            depth += ch == "("; depth -= ch == ")"
            n += ch == "," and depth == 0
        assert n - (1 if m.group(2).strip().startswith(",") else 0) == 4, m.group(2)
        m = re.search(r"\[([^\]]*)\]\s*=\s*estimateWorldCameraPose\(", text)
        assert m and len(m.group(1).split(",")) == 4


def test_name_value_options_of_the_shadows_are_the_references():
    """the option names in the `arguments` blocks (modules/KLTTracker.m:10-15, modules/selectCandidateKeypoints.m:10-17,
    modules/candidateTriangulation.m:3-15)"""
    want = {"KLTTracker.m": ["NumPyramidLevels", "MaxBidirectionalError", "BlockSize", "MaxIterations"],
            "selectCandidateKeypoints.m": ["MinDistance", "MinQuality", "FilterSize", "CandidatesToKeep"]}
    for f, names in want.items():
        text = open(os.path.join(ROOT, "matlab", f)).read()
        got = re.findall(r"^\s*\w+\.(\w+)\s.*=", text, flags=re.M)
        assert [n for n in got if n in names] == names, (f, got)
        ref_path = os.path.join("/root/reference/modules", f)
        if os.path.exists(ref_path):
            ref_names = re.findall(r"^\s*\w+\.(\w+)\s.*=", open(ref_path).read(), flags=re.M)
            assert [n for n in ref_names if n in names] == names, (f, ref_names)


def test_every_gateway_a_shadow_calls_exists_and_takes_that_many_arguments():
    """matlab/*.m call `vo_*_mex(...)`: the gateway's source is in mex/, and the number of arguments of the call is the
    one the gateway's `nrhs` check accepts (a gateway says `nrhs != N` or `nrhs < A || nrhs > B` in its first lines)."""
    calls = []
    for f in sorted(glob.glob(os.path.join(ROOT, "matlab", "*.m"))):
        text = re.sub(r"\.\.\.[^\n]*\n", " ", open(f).read())
        text = "\n".join(l.split("%")[0] for l in text.splitlines())          # comments
        for m in re.finditer(r"\b(vo_[a-z0-9_]+_mex)\s*\(", text):
            depth, i, n = 1, m.end(), 1
            while depth and i < len(text):
                ch = text[i]
                depth += ch in "([{"; depth -= ch in ")]}"
                n += ch == "," and depth == 1
                i += 1
            calls.append((os.path.basename(f), m.group(1), n))
    assert len(calls) >= 6, calls
    for shadow, gw, nargs in calls:
        src = os.path.join(ROOT, "mex", gw + ".cpp")
        assert os.path.exists(src), f"{shadow} calls {gw}, which mex/ does not hold"
        code = open(src).read()
        m = re.search(r"nrhs\s*!=\s*(\d+)", code)
        if m:
            assert nargs == int(m.group(1)), f"{shadow}: {gw} called with {nargs} arguments, the gateway takes {m.group(1)}"
            continue
        m = re.search(r"nrhs\s*<\s*(\d+)\s*\|\|\s*nrhs\s*>\s*(\d+)", code)
        if m:
            assert int(m.group(1)) <= nargs <= int(m.group(2)), f"{shadow}: {gw} called with {nargs} arguments, the gateway takes {m.group(1)}..{m.group(2)}"
