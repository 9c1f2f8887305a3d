"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the reference's on-disk model format; parity unpinned (the
reference ships no dump files), pinned by hand-derived known answers in tests/test_model_io.py.

  clean            D1/Primitive.cpp:4-134
  pose_text        D1/Graph.cpp:107-119
  graph_text       D1/Graph.cpp:124-160
  plane_images     D1/Graph.cpp:166-231      (the three 16-bit x 3 images of a plane, memory order B,G,R)
  frame_images     D1/Frame.cpp:191-227
  planes_from_images / pose_from_text   D2/Model3D.cpp:1084-1166, D2/KeyFrame.h:84-95
"""
import numpy as np

f32 = np.float32
RES_PLANE = 0.004


def _g(x):
    return "%g" % float(f32(x))


def clean(bump, rgb, mask, shift):
    """-> (bump, rgb, mask, shift) cropped to the bounding box of mask > 10, or None when fewer than 10 such cells."""
    bump, rgb, mask = bump.copy(), rgb.copy(), mask.copy()
    obs = mask > 10
    rgb[~obs] = 0
    bump[~obs] = 0
    if obs.sum() < 10:
        return None
    ii, jj = np.nonzero(obs)
    i0, i1, j0, j1 = ii.min(), ii.max(), jj.min(), jj.max()
    keep = mask >= 11
    out_mask = np.where(keep, mask, 0)[i0:i1 + 1, j0:j1 + 1]
    new_shift = (f32(float(shift[0]) + float(f32(i0)) * RES_PLANE), f32(float(shift[1]) + float(f32(j0)) * RES_PLANE))
    return bump[i0:i1 + 1, j0:j1 + 1], rgb[i0:i1 + 1, j0:j1 + 1], out_mask, new_shift


def pose_text(pose16):
    return " ".join(_g(v) for v in pose16) + "\n"


def graph_text(nodes):
    """nodes: list of dicts(nmle, dist, e1, e2, shift, scale, size, anchor) or None for a plane without data."""
    out = ["%d\n" % len(nodes)]
    for nd in nodes:
        if nd is None:
            out.append(" ".join(["0"] * 16) + "\n")
            continue
        vals = [_g(v) for v in list(nd["nmle"]) + [nd["dist"]] + list(nd["e1"]) + list(nd["e2"]) + list(nd["shift"]) + list(nd["scale"])]
        vals += ["%d" % nd["size"][0], "%d" % nd["size"][1]] + [_g(v) for v in nd["anchor"]]
        out.append(" ".join(vals) + "\n")
    return "".join(out)


def plane_images(bump, rgb, mask):
    """Images as cv::imwrite receives them: rows flipped, channel [2] <- component 0."""
    ib = bump[::-1, :, ::-1].astype(np.uint16)
    im = np.repeat((200 * mask.astype(np.int64))[::-1, :, None], 3, axis=2).astype(np.uint16)
    ic = (200 * rgb.astype(np.int64))[::-1, :, ::-1].astype(np.uint16)
    return ib, im, ic


def frame_images(rgbd):
    d = np.trunc(rgbd[:, :, 3].astype(np.float64) * 100.0 * 5000.0).astype(np.int64).astype(np.uint16)[::-1]
    c = (200 * np.trunc(rgbd[:, :, :3].astype(np.float64) * 255.0).astype(np.int64).astype(np.uint16).astype(np.int64)).astype(np.uint16)
    return d, c[::-1, :, ::-1]


def planes_from_images(ib, im, ic):
    bump = ib[::-1, :, ::-1].copy()
    rgb = (ic[::-1, :, ::-1] // 200).astype(np.uint8)
    mask = (im[::-1, :, 0] // 200).astype(np.uint16)
    return bump, rgb, mask


def pose_from_text(line):
    return np.array([f32(x) for x in line.split()[:16]], f32)
