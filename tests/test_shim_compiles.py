"""The drop-in boundary, proven by the compiler and the linker (CPU only).

integration/rgbdslam_shim.cpp is the file a maintainer builds into the reference tree in place of Kernel.cu / ICP.cu /
Primitive.cu / RG.cu.  Here it is compiled against the reference's OWN four operator headers (copied next to a stand-in
cudaTypes.cuh that supplies the CUDA / OpenCV / Windows type names: integration/stub/) and linked with librgbdslam_hip.so
together with a generated translation unit that takes the address of every operator function the reference's host code
calls -- found by scanning Frame.cpp / Graph.cpp / RG.cpp / RG.h / Primitive.* / GP-demo.cpp for the names the headers
declare.  An undefined or mis-typed symbol fails the link.  Needs /root/reference (absent on the GPU box: skipped there).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DemoStep1/DemoStep1"
HEADERS = ["Kernel.cuh", "ICP.cuh", "Primitive.cuh", "RG.cuh"]
HOST_SOURCES = ["Frame.cpp", "Frame.h", "Graph.cpp", "Graph.h", "RG.cpp", "RG.h", "Primitive.cpp", "Primitive.h", "GP-demo.cpp"]
LIB = os.path.join(ROOT, "rgb-d-slam_amd", "librgbdslam_hip.so")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _declared_functions():
    names = {}
    for h in HEADERS:
        text = _strip_comments(open(os.path.join(REF, h), errors="replace").read())
        for mm in re.finditer(r"^[ \t]*[\w:<> \*]+?[ \*]+(\w+)\s*\(", text, flags=re.M):
            names.setdefault(mm.group(1), h)
    return names


def _called_by_host_code(names):
    called = set()
    for f in HOST_SOURCES:
        text = _strip_comments(open(os.path.join(REF, f), errors="replace").read())
        for name in names:
            if re.search(r"(?<![\w:.>])%s\s*\(" % re.escape(name), text):
                called.add(name)
    return called


def test_shim_defines_every_operator_the_reference_host_code_calls(tmp_path):
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    for h in HEADERS:                       # the reference's declarations, next to the stand-in cudaTypes.cuh
        shutil.copy(os.path.join(REF, h), tmp_path / h)
    shutil.copy(os.path.join(ROOT, "integration", "stub", "cudaTypes.cuh"), tmp_path / "cudaTypes.cuh")
    declared = _declared_functions()
    called = _called_by_host_code(declared)
    # what SURVEY 8(b) lists must be among them
    for must in ["SetCalibrationMatrix", "SetPoseMatrix", "VertexMapKinect", "gpu_cpy_char3", "ComputeNormal", "gpu_cpy",
                 "AllocBuffers", "FreeBuffers", "EstimateSystemGaussNewton", "TransformVertexMap", "QuadTrimFrame", "ReadFrame_cu",
                 "ProjectMask_cu", "Segment_cu", "MergeDepthOff", "MergeDepthKinect", "VertexMapGL", "gpu_add_zbuffOff",
                 "gpu_add_zbuffKinect", "InitMask_cu", "Update_cu", "Affect", "AffectChar", "AffectShort", "AllocPosePrimMatrix",
                 "AllocBufPrim", "FreeBufPrim", "FreePosePrimMatrix", "VotePlan_cu", "AffectPlan_cu", "Overlap_cu", "MapData2VBO_cu",
                 "gpu_cpyN", "gpu_Discont", "gpu_Sum", "gpu_Thresh"]:
        assert must in called, must
    also = ["VertexFromBump", "QuadTrim", "SetPosePrimMatrix"]   # declared, no D1 caller, but part of the ABI (D2 calls them)
    overloaded = {"ImDilate"}
    names = sorted((called | set(also)) - overloaded)
    users = ['#include "Kernel.cuh"', '#include "ICP.cuh"', '#include "Primitive.cuh"', '#include "RG.cuh"',
             "void *rgbdslam_shim_users[] = {"]
    users += ["    (void *)&%s," % nm for nm in names]
    users += ["};", ""]
    (tmp_path / "users.cpp").write_text("\n".join(users))
    inc = ["-I", str(tmp_path), "-I", os.path.join(ROOT, "include")]
    shim = os.path.join(ROOT, "integration", "rgbdslam_shim.cpp")
    common = ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-fPIC"] + inc
    subprocess.check_call(common + ["-fsyntax-only", shim])
    out = str(tmp_path / "libshimcheck.so")
    r = subprocess.run(common + ["-shared", "-Wl,--no-undefined", "-o", out, shim, str(tmp_path / "users.cpp"),
                                 "-L", os.path.dirname(LIB), "-lrgbdslam_hip", "-Wl,-rpath," + os.path.dirname(LIB)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # and nothing else is missing: every declared function without a definition in the shim has NO caller in D1's host code
    syms = subprocess.check_output(["nm", "-C", "--defined-only", out], text=True)
    undefined_but_called = [nm for nm in called if not re.search(r"\b%s\(" % re.escape(nm), syms)]
    assert undefined_but_called == [], undefined_but_called
