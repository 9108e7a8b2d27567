"""Map persistence: the Gaussian map as a PLY file.

On-disk format of the reference's ``GaussianModel.save_ply`` / ``load_ply``
(/root/reference/gaussian_splatting/scene/gaussian_model.py:467-520,537-640): one ``vertex`` element of float32
properties ``x y z nx ny nz f_dc_0.. opacity scale_0.. rot_0..3`` (normals are zeros; this fork keeps raw RGB
in f_dc and writes no f_rest), binary little-endian as plyfile writes it.  plyfile is not installed here, so the
file is written and parsed directly with numpy; a file written by the reference loads here and vice versa.
Values are the RAW parameters (pre-activation opacity / scaling, un-normalised rotation), as in the reference.
Parity unpinned: the reference holds no PLY fixture.
"""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np
import torch


def attribute_names(n_dc: int, n_scale: int, n_rot: int = 4) -> List[str]:
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += ["opacity"]
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(path: str, xyz: torch.Tensor, f_dc: torch.Tensor, opacity: torch.Tensor, scaling: torch.Tensor,
             rotation: torch.Tensor) -> None:
    """xyz [P,3], f_dc [P,3] (or [P,C,1]/[P,1,C] as the reference stores it), opacity [P,1], scaling [P,1|3], rotation [P,4]."""
    def cpu(t):
        t = t.detach().to("cpu", torch.float32)
        return t.reshape(t.shape[0], int(np.prod(t.shape[1:]))).numpy()

    xyz_, dc, op, sc, rot = cpu(xyz), cpu(f_dc), cpu(opacity), cpu(scaling), cpu(rotation)
    P = xyz_.shape[0]
    names = attribute_names(dc.shape[1], sc.shape[1], rot.shape[1])
    table = np.concatenate([xyz_, np.zeros_like(xyz_), dc, op, sc, rot], axis=1).astype("<f4")
    assert table.shape == (P, len(names))
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}


def read_vertex_table(path: str) -> Dict[str, np.ndarray]:
    """Every scalar property of the ``vertex`` element (binary little/big endian or ascii)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
                elif props:
                    pass          # elements after the vertex table are not needed
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2)
            return {name: rows[:, i].astype(t) for i, (name, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, order + t) for name, t in props])
        raw = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        return {name: np.ascontiguousarray(raw[name]) for name, _ in props}


def load_ply(path: str, device="cpu") -> Dict[str, torch.Tensor]:
    """{'xyz' [P,3], 'f_dc' [P,C], 'opacity' [P,1], 'scaling' [P,S], 'rotation' [P,4]} as float32 tensors."""
    v = read_vertex_table(path)

    def stack(prefix):
        keys = sorted((k for k in v if k.startswith(prefix)), key=lambda k: int(k.split("_")[-1]))
        return np.stack([v[k] for k in keys], axis=1).astype(np.float32)

    out = {"xyz": np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32), "f_dc": stack("f_dc_"),
           "opacity": v["opacity"].astype(np.float32)[:, None], "scaling": stack("scale_"), "rotation": stack("rot_")}
    return {k: torch.from_numpy(a).to(device) for k, a in out.items()}
