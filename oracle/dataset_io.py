"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain Python / numpy, of the reference's dataset listing and trajectory output
(SURVEY.md 8f, row N3).  Parity unpinned: the reference holds no fixtures for these functions and cannot
be built here; the restatement is pinned by hand-derived known answers in tests/test_dataset_io.py.

  load_time_stamps      D1/Utilities.cpp:171-276   (rgb.txt x depth.txt, 20 ms window)
  load_associated       D1/Utilities.cpp:278-304
  frame_files           D1/RG.cpp:105-139
  matrix2quaternion     D1/Utilities.cpp:11-68
  trajectory_text       D1/Utilities.cpp:500-546
"""
import math
import os
import re

import numpy as np

f32 = np.float32


def _lines(filename, skip):
    """Lines the reference's getline/eof loop parses: only '\\n'-terminated ones, after `skip` header lines."""
    with open(filename, "rb") as f:
        data = f.read().decode("latin-1")
    parts = data.split("\n")[:-1]          # the unterminated tail (possibly empty) is never parsed
    return parts[skip:]


_SCAN = re.compile(r"\s*([+-]?\d+)\.([+-]?\d+)\s+(\S+)")


def _parse(lines):
    """sscanf("%d.%d %s"): fraction read as an integer / 1e6; a failed conversion keeps the previous values."""
    t, names = [], []
    a = b = 0
    s = ""
    for l in lines:
        m = _SCAN.match(l)
        if m:
            a, b, s = int(m.group(1)), int(m.group(2)), m.group(3)
        t.append(float(a) + float(b) / 1000000.0)
        names.append(s)
    return t, names


def load_time_stamps(path):
    """-> (timestamp_depth, timestamp_color) name lists.  Restates the iterator/index walk literally."""
    try:
        dt, dn = _parse(_lines(os.path.join(path, "depth.txt"), 3))
        ct, cn = _parse(_lines(os.path.join(path, "rgb.txt"), 3))
    except OSError:
        return [], []
    out_d, out_c = [], []
    max_diff = 0.02
    it = 0          # position of the reference's vector iterator
    idx = -1        # indx_color
    while it < len(ct):      # the reference compares != end(); it > end() is undefined there, a stop here
        idx += 1
        val = ct[idx]
        min_val = max_diff
        position = 0
        for k, val2 in enumerate(dt):
            if val2 > val + max_diff:
                break
            d = abs(val - val2)
            if d < min_val and d < max_diff:
                min_val = d
                position = k
        if min_val == max_diff:
            it += 1
            continue
        out_d.append(dn[position])
        out_c.append(cn[idx])
        del ct[idx], cn[idx], dt[position], dn[position]
        it = idx + 1         # it = erase(begin + idx) -> idx ; then the loop's it++
        idx -= 1
    return out_d, out_c


def load_associated(path):
    pat = re.compile(r"\s*[+-]?\d+\.[+-]?\d+\s+(\S+)\s+[+-]?\d+\.[+-]?\d+\s+(\S+)")
    out_d, out_c = [], []
    s1 = s2 = ""
    try:
        lines = _lines(os.path.join(path, "associated.txt"), 0)
    except OSError:
        return [], []
    for l in lines:
        m = pat.match(l)
        if m:
            s1, s2 = m.group(1), m.group(2)
        out_c.append(s1)
        out_d.append(s2)
    return out_d, out_c


def frame_files(path, count, real=False, rgbd_rep=True, ts_depth=None, ts_color=None):
    if not real:
        if rgbd_rep:
            return path + "//" + ts_depth[count], path + "//" + ts_color[count]
        k = count + 1
        assert k < 10000
        return "%s//depth//%06d.png" % (path, k), "%s//color//%06d.png" % (path, k)
    return "%s//Depth_%d.tiff" % (path, count), "%s//RGB_%d.tiff" % (path, count)


def matrix2quaternion(R):
    """R[r][c] float32 -> (w, x, y, z) float32, rounding as the C++ expressions do."""
    R = np.asarray(R, f32)
    r00, r11, r22 = R[0, 0], R[1, 1], R[2, 2]

    def branch(a, b, c):
        r = f32(np.sqrt(f32(f32(f32(f32(1) + a) - b) - c)))
        s = f32(0.5 / float(r))
        return r, s

    if r00 > r11 and r00 > r22:
        r, s = branch(r00, r11, r22)
        return np.array([f32(R[2, 1] - R[1, 2]) * s, f32(0.5 * float(r)), f32(R[0, 1] + R[1, 0]) * s, f32(R[2, 0] + R[0, 2]) * s], f32)
    if r11 > r00 and r11 > r22:
        r, s = branch(r11, r00, r22)
        return np.array([f32(R[2, 0] - R[0, 2]) * s, f32(R[0, 1] + R[1, 0]) * s, f32(0.5 * float(r)), f32(R[2, 1] + R[1, 2]) * s], f32)
    if r22 > r00 and r22 > r11:
        r, s = branch(r22, r00, r11)
        return np.array([f32(R[1, 0] - R[0, 1]) * s, f32(R[0, 2] + R[2, 0]) * s, f32(R[2, 1] + R[1, 2]) * s, f32(0.5 * float(r))], f32)
    qw = f32(math.sqrt(1.0 + float(r00) + float(r11) + float(r22)) / 2.0)
    d = 4.0 * float(qw)
    return np.array([qw, f32(float(f32(R[2, 1] - R[1, 2])) / d), f32(float(f32(R[0, 2] - R[2, 0])) / d),
                     f32(float(f32(R[1, 0] - R[0, 1])) / d)], f32)


def _g(x):
    """operator<<(ostream, float) with default flags: %g, 6 significant digits."""
    return "%g" % float(f32(x))


def trajectory_text(Rs, ts, names):
    out = ["# Etimated camera trajectory\n"]
    for R, t, name in zip(Rs, ts, names):
        t = np.asarray(t, f32)
        if float(t[0]) == 100.0:
            continue
        q = matrix2quaternion(R)
        out.append("%s %s %s %s %s %s %s %s\n" % (name[6:23], _g(t[0]), _g(t[1]), _g(t[2]), _g(q[1]), _g(q[2]), _g(q[3]), _g(q[0])))
    return "".join(out)
