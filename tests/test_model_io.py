"""SURVEY 8f N2 -- the reference's on-disk model format: writer (Primitive::Clean, Graph::save,
PredictedFrame::save) and reader (Model3D::Update, KeyFrame).  CPU tests of the host library against
oracle/model_io.py (restatement, parity unpinned) and hand-derived known answers."""
import importlib
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model_io as ora  # noqa: E402

pkg = importlib.import_module("rgb-d-slam_amd")


def tiff_decode_reference(path):
    """Independent minimal TIFF reader (little-endian, strips, LZW + horizontal predictor, 16-bit) written from
    the TIFF 6.0 text, used to pin the container of the 16-bit x 3 images no other codec in this image reads."""
    b = open(path, "rb").read()
    assert b[:4] == b"II*\x00"
    ifd = struct.unpack_from("<I", b, 4)[0]
    tags = {}
    for e in range(struct.unpack_from("<H", b, ifd)[0]):
        tag, typ, cnt = struct.unpack_from("<HHI", b, ifd + 2 + 12 * e)
        sz = {1: 1, 3: 2, 4: 4}[typ]
        off = ifd + 2 + 12 * e + 8 if sz * cnt <= 4 else struct.unpack_from("<I", b, ifd + 2 + 12 * e + 8)[0]
        tags[tag] = [struct.unpack_from("<" + {1: "B", 3: "H", 4: "I"}[typ], b, off + sz * k)[0] for k in range(cnt)]
    W, H, spp, rps = tags[256][0], tags[257][0], tags[277][0], tags[278][0]
    assert tags[258] == [16] * spp and tags[259] == [5] and tags[317] == [2] and tags[284] == [1]
    rows = []
    for off, cnt, s in zip(tags[273], tags[279], range(len(tags[273]))):
        data, out = b[off:off + cnt], bytearray()
        table, nbits, acc, have, old = None, 9, 0, 0, None
        pos = 0
        while True:
            while have < nbits:
                acc = (acc << 8) | data[pos]
                pos += 1
                have += 8
            code = (acc >> (have - nbits)) & ((1 << nbits) - 1)
            have -= nbits
            if code == 257:
                break
            if code == 256:
                table, nbits, old = [bytes([k]) for k in range(256)] + [b"", b""], 9, None
                continue
            if old is None:
                entry = table[code]
            else:
                entry = table[code] if code < len(table) else old + old[:1]
                table.append(old + entry[:1])
            out += entry
            old = entry
            if len(table) == (1 << nbits) - 1 and nbits < 12:
                nbits += 1
        nrows = min(rps, H - s * rps)
        a = np.frombuffer(bytes(out[:nrows * W * spp * 2]), "<u2").reshape(nrows, W, spp).astype(np.uint32)
        rows.append((np.cumsum(a, axis=1) & 0xFFFF).astype(np.uint16))
    img = np.concatenate(rows, 0)
    return img[:, :, ::-1] if spp == 3 else img[:, :, 0]       # RGB in the file -> B,G,R in memory


def make_plane(rng, n, m, box, k):
    i0, i1, j0, j1 = box
    mask = np.zeros((n, m), np.uint8)
    mask[i0:i1 + 1, j0:j1 + 1] = rng.choice(np.array([0, 5, 10, 11, 12, 40, 255], np.uint8), (i1 - i0 + 1, j1 - j0 + 1))
    mask[i0, j0 + 1] = 30
    mask[i1, j0] = 30
    mask[i0 + 1, j1] = 30                                         # pin the bounding box
    return {"nmle": np.float32([0.1 * k, 0.95, -0.2]), "dist": np.float32(-1.25 - k), "e1": np.float32([1, 0, 0.5]),
            "e2": np.float32([0, 0.2, 1]), "shift": np.float32([-2.048 + k, 1.0e-3]), "scale": np.float32([500, 500]), "size": (n, m),
            "anchor": np.float32([0.5, -0.25, 3.0 + k]), "bump": rng.integers(1, 65536, (n, m, 3)).astype(np.uint16),
            "rgb": rng.integers(1, 256, (n, m, 3)).astype(np.uint8), "mask": mask}


def test_clean_known_answer(tmp_path):
    """6x7 plane, hand-derived: observed cells (mask > 10) span rows 1..3 x columns 2..5 -> 3x4 crop, shift moves by
    (1, 2) cells of 4 mm; cells with mask <= 10 inside the box come out as zeros; mask 10 outside does not widen it."""
    mask = np.zeros((6, 7), np.uint8)
    mask[1, 2:6] = (11, 12, 10, 13)
    mask[2, 2:6] = (14, 0, 15, 16)
    mask[3, 2:6] = (17, 18, 19, 255)
    mask[5, 6] = 10
    bump = np.arange(6 * 7 * 3, dtype=np.uint16).reshape(6, 7, 3) + 1000
    rgb = np.arange(6 * 7 * 3, dtype=np.uint8).reshape(6, 7, 3) + 1
    plane = {"nmle": np.float32([0, 0, 1]), "dist": np.float32(2), "e1": np.float32([1, 0, 0]), "e2": np.float32([0, 1, 0]),
             "shift": np.float32([0.5, -0.25]), "scale": np.float32([500, 500]), "size": (6, 7), "anchor": np.float32([0, 0, 2]),
             "bump": bump, "rgb": rgb, "mask": mask}
    d = str(tmp_path)
    written = pkg.save_graph(d, 0, np.eye(4, dtype=np.float32).reshape(-1), [plane])
    assert written[0]["size"] == (3, 4)
    assert np.array_equal(written[0]["shift"], np.float32([0.5 + 1 * 0.004, -0.25 + 2 * 0.004]))
    m_exp = np.array([[11, 12, 0, 13], [14, 0, 15, 16], [17, 18, 19, 255]], np.uint16)
    b_exp = bump[1:4, 2:6].copy()
    b_exp[m_exp == 0] = 0
    pkg.save_frame(d, 1, np.zeros((2, 2, 4), np.float32))
    q = pkg.load_model(d, 0)["planes"][0]
    assert np.array_equal(q["mask"], m_exp) and np.array_equal(q["bump"], b_exp)
    assert np.array_equal(q["rgb"], np.where(m_exp[:, :, None] == 0, 0, rgb[1:4, 2:6]))
    # one observed cell fewer (9 < 10): the plane is dropped and written as a 0 x 0 line
    mask[3, 5] = 10
    written = pkg.save_graph(d, 0, np.eye(4, dtype=np.float32).reshape(-1), [plane])
    assert written[0]["size"] == (0, 0)
    assert open(d + "/Graph-eq0.txt").read() == "1\n" + " ".join(["0"] * 16) + "\n"


def test_clean_and_graph_files_match_oracle(tmp_path):
    rng = np.random.default_rng(21)
    planes = [make_plane(rng, 40, 56, (5, 30, 8, 50), 0), make_plane(rng, 33, 21, (0, 32, 0, 20), 1)]
    tiny = make_plane(rng, 16, 16, (2, 4, 2, 4), 2)
    tiny["mask"][:] = 0
    tiny["mask"][3, 3:8] = 50                                     # 5 observed cells < 10: the plane is dropped
    planes.append(tiny)
    planes.append(dict(planes[0], bump=None))                     # a Primitive that never got data
    pose = np.float32([0.9, 0.1, -0.2, 0, -0.1, 0.95, 0.05, 0, 0.2, -0.04, 0.97, 0, 1.5, -2.25e-3, 1234.5678, 1])
    d = str(tmp_path)
    written = pkg.save_graph(d, 7, pose, planes)
    assert sorted(os.listdir(d)) == sorted(["Pose8.txt", "Graph-eq7.txt", "Dummy7.txt"] + ["%s7-%d.tif" % (s, k) for s in ("Bump", "Mask", "RGB")
                                                                                         for k in (0, 1)])
    assert open(d + "/Pose8.txt").read() == ora.pose_text(pose)
    assert open(d + "/Pose8.txt").read() == "0.9 0.1 -0.2 0 -0.1 0.95 0.05 0 0.2 -0.04 0.97 0 1.5 -0.00225 1234.57 1\n"
    nodes = []
    for k, p in enumerate(planes):
        c = ora.clean(p["bump"], p["rgb"], p["mask"], p["shift"]) if p["bump"] is not None else None
        if c is None:
            nodes.append(None)
            assert written[k]["size"] == (0, 0)
            continue
        cb, cr, cm, sh = c
        nodes.append(dict(p, shift=sh, size=cm.shape))
        assert written[k]["size"] == cm.shape and np.array_equal(written[k]["shift"], np.float32(sh))
        ib, im, ic = ora.plane_images(cb, cr, cm)
        assert np.array_equal(pkg.read_image("%s/Bump7-%d.tif" % (d, k)), ib)
        assert np.array_equal(pkg.read_image("%s/Mask7-%d.tif" % (d, k)), im)
        assert np.array_equal(pkg.read_image("%s/RGB7-%d.tif" % (d, k)), ic)
        assert np.array_equal(tiff_decode_reference("%s/Bump7-%d.tif" % (d, k)), ib)     # container pinned independently
    assert open(d + "/Graph-eq7.txt").read() == ora.graph_text(nodes)
    lines = open(d + "/Graph-eq7.txt").read().split("\n")
    assert lines[0] == "4" and lines[3] == lines[4] == "0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
    # crop of plane 0: rows 5..30, columns 8..50; shift moves by 5 and 8 cells of 4 mm
    assert lines[1] == "0 0.95 -0.2 -1.25 1 0 0.5 0 0.2 1 -2.028 0.033 500 500 26 43 0.5 -0.25 3"

    # reader: dropped planes are skipped (Model3D.cpp:1107), the others keep their file index
    pkg.save_frame(d, 8, np.zeros((4, 4, 4), np.float32))
    m = pkg.load_model(d, 7)
    assert [q["size"] for q in m["planes"]] == [(26, 43), written[1]["size"]]


def test_round_trip_with_key_frame(tmp_path):
    rng = np.random.default_rng(22)
    d = str(tmp_path)
    assert pkg.load_model(d, 3) is None                           # no Dummy3.txt yet
    planes = [make_plane(rng, 64, 48, (3, 60, 2, 40), 0), make_plane(rng, 20, 30, (1, 18, 4, 25), 1)]
    rgbd = rng.random((24, 32, 4)).astype(np.float32)
    rgbd[:, :, 3] *= 0.1                                          # depth / 100, up to 10 m
    rgbd[0, 0] = (1.0, 0.0, 0.5, 0.0)
    pose = np.float32(rng.normal(size=16))
    pkg.save_frame(d, 4, rgbd)                                    # SaveThread order: key frame under indx + 1, then the graph
    written = pkg.save_graph(d, 3, pose, planes)
    dimg, cimg = ora.frame_images(rgbd)
    assert np.array_equal(pkg.read_image(d + "/Depth4.tiff"), dimg)
    assert np.array_equal(pkg.read_image(d + "/RGB4.tiff"), cimg)
    assert dimg[-1, 0] == 0 and list(cimg[-1, 0]) == [200 * 127, 0, 200 * 255]   # row flip, B,G,R order, trunc(0.5*255) = 127
    m = pkg.load_model(d, 3)
    assert m is not None and len(m["planes"]) == 2
    assert np.array_equal(m["pose"], ora.pose_from_text(ora.pose_text(pose)))
    assert np.abs(m["pose"] - pose).max() <= 1e-5 * np.abs(pose).max()      # text carries 6 significant digits
    assert np.array_equal(m["depth"], dimg) and np.array_equal(m["rgb"], cimg)
    for k, p in enumerate(planes):
        cb, cr, cm, sh = ora.clean(p["bump"], p["rgb"], p["mask"], p["shift"])
        ob, orgb, omask = ora.planes_from_images(*ora.plane_images(cb, cr, cm))
        q = m["planes"][k]
        assert q["size"] == cm.shape == written[k]["size"]
        assert np.array_equal(q["bump"], ob) and np.array_equal(q["bump"], cb)
        assert np.array_equal(q["rgb"], orgb) and np.array_equal(q["rgb"], cr)
        assert np.array_equal(q["mask"], omask) and np.array_equal(q["mask"], cm)
        for key in ("nmle", "e1", "e2", "scale", "anchor"):
            assert np.allclose(q[key], p[key], rtol=1e-5)
        assert np.allclose(q["shift"], sh, rtol=1e-5, atol=1e-6)


def test_indx_minus_one_writes_only_the_pose(tmp_path):
    d = str(tmp_path)
    pkg.save_graph(d, -1, np.arange(16, dtype=np.float32), [])
    assert os.listdir(d) == ["Pose0.txt"]
    assert open(d + "/Pose0.txt").read() == " ".join(str(k) for k in range(16)) + "\n"
