"""Map persistence round trip in the reference's PLY layout (gaussian_model.py:467-520,537-640)."""
import numpy as np
import torch

from monogs_amd.ply_io import attribute_names, load_ply, read_vertex_table, save_ply


def _map(P, iso, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(xyz=torch.randn(P, 3, generator=g), f_dc=torch.rand(P, 3, generator=g),
                opacity=torch.randn(P, 1, generator=g), scaling=torch.randn(P, 1 if iso else 3, generator=g),
                rotation=torch.randn(P, 4, generator=g))


def test_round_trip_and_header(tmp_path):
    for iso in (True, False):
        m = _map(257, iso)
        path = str(tmp_path / f"map_{iso}" / "point_cloud.ply")
        save_ply(path, m["xyz"], m["f_dc"].reshape(-1, 3, 1), m["opacity"], m["scaling"], m["rotation"])
        head = open(path, "rb").read(4096).split(b"end_header\n")[0].decode().split("\n")
        assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
        names = [ln.split()[2] for ln in head if ln.startswith("property")]
        assert names == attribute_names(3, 1 if iso else 3) and all(ln.split()[1] == "float" for ln in head if ln.startswith("property"))
        back = load_ply(path)
        for k in m:
            assert torch.equal(back[k], m[k]), k
        assert np.all(read_vertex_table(path)["nx"] == 0)


def test_reads_ascii_and_extra_elements(tmp_path):
    path = tmp_path / "a.ply"
    path.write_text("ply\nformat ascii 1.0\ncomment hand written\nelement vertex 2\n"
                    + "".join(f"property float {n}\n" for n in attribute_names(3, 1))
                    + "element face 0\nproperty list uchar int vertex_indices\nend_header\n"
                    + " ".join(str(0.5 * i) for i in range(15)) + "\n" + " ".join(str(-1.0 * i) for i in range(15)) + "\n")
    m = load_ply(str(path))
    assert m["xyz"].tolist() == [[0.0, 0.5, 1.0], [-0.0, -1.0, -2.0]]
    assert m["rotation"].shape == (2, 4) and m["scaling"].shape == (2, 1) and float(m["opacity"][1, 0]) == -9.0


def test_empty_map(tmp_path):
    m = _map(0, True)
    path = str(tmp_path / "e.ply")
    save_ply(path, m["xyz"], m["f_dc"], m["opacity"], m["scaling"], m["rotation"])
    back = load_ply(path)
    assert back["xyz"].shape == (0, 3) and back["rotation"].shape == (0, 4)
