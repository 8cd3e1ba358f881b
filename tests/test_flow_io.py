"""Host-side flow file formats (flow_supervisor_amd/raft_utils/frame_utils.py vs pytorch/raft_utils/frame_utils.py).
The reference module cannot be imported in the build container (it imports cv2 at module level), so the .flo layout is
checked against the Middlebury format itself: a hand-assembled byte string."""
import struct

import numpy as np

from flow_supervisor_amd.raft_utils import frame_utils as FU


def test_flo_bytes_and_round_trip(tmp_path):
    flow = np.array([[[1.5, -2.0], [0.25, 4.0], [3.0, 0.0]], [[-1.0, 1.0], [8.5, -0.5], [2.0, 2.0]]], np.float32)   # H=2, W=3
    p = tmp_path / "a.flo"
    FU.writeFlow(str(p), flow)
    want = struct.pack("<f", 202021.25) + struct.pack("<ii", 3, 2) + flow.astype("<f4").tobytes()
    assert p.read_bytes() == want
    assert p.read_bytes()[:4] == b"PIEH"
    back = FU.readFlow(str(p))
    assert back.shape == (2, 3, 2) and np.array_equal(back, flow)
    # planes given separately (writeFlow(filename, u, v))
    q = tmp_path / "b.flo"
    FU.writeFlow(str(q), flow[:, :, 0], flow[:, :, 1])
    assert q.read_bytes() == want
    assert np.array_equal(FU.read_gen(str(p)), flow)


def test_flo_bad_tag_and_empty(tmp_path, capsys):
    p = tmp_path / "bad.flo"
    p.write_bytes(struct.pack("<f", 1.0) + struct.pack("<ii", 1, 1) + b"\0" * 8)
    assert FU.readFlow(str(p)) is None
    assert "Magic number incorrect" in capsys.readouterr().out
    assert FU.read_gen(str(tmp_path / "x.unknown")) == []


def test_pfm_round_trip(tmp_path):
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    p = tmp_path / "d.pfm"
    with open(p, "wb") as f:
        f.write(b"Pf\n4 3\n-1.0\n")
        f.write(np.flipud(img).astype("<f4").tobytes())
    assert np.array_equal(FU.readPFM(str(p)), img)
    assert np.array_equal(FU.read_gen(str(p)), img)
