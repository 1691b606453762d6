"""PLY interchange of the collected Gaussians: the binary little-endian layout of the
INRIA 3DGS / nerfstudio `ns-export gaussian-splat` files that the reference's viewer export panel produces
(/root/reference/mtgs/custom_viewer/export_panel.py:193) and that web viewers read:

    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3 (K-1) - 1)  opacity  scale_0..2  rot_0..3          (all float32)

with the RAW parameters MTGS stores (vanilla_gaussian_splatting.py:174-213): log-scales, logit opacities, wxyz quaternions
(not normalised), SH coefficients; f_rest is CHANNEL-major (all red coefficients, then green, then blue), i.e.
features_rest[N, K-1, 3] transposed.  Host code (numpy); the tensors it returns feed mtgs_amd.nodes / rasterization.
Note (mtgs_scene_graph.py:83-86): a model trained in antialiased mode does not look the same in a classic-mode viewer."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Union

import numpy as np
import torch
from torch import Tensor


def _fields(n_rest: int):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(n_rest)]
    return names + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]


def write_ply(path: Union[str, Path], params: Dict[str, Tensor]) -> int:
    """params: means[N,3], scales[N,3] (log), quats[N,4] (wxyz), opacities[N,1] or [N] (logit), features_dc[N,3],
    features_rest[N,K-1,3] (optional).  Rows with a non-finite value are dropped, as nerfstudio's exporter does.
    Returns the number of Gaussians written."""
    g = lambda k: params[k].detach().to(torch.float32).cpu().numpy()
    means, scales, quats = g("means"), g("scales"), g("quats")
    opac = g("opacities").reshape(-1, 1)
    dc = g("features_dc").reshape(means.shape[0], 3)
    rest = g("features_rest") if params.get("features_rest") is not None else np.zeros((means.shape[0], 0, 3), np.float32)
    N = means.shape[0]
    assert scales.shape == (N, 3) and quats.shape == (N, 4) and opac.shape == (N, 1) and rest.shape[0] == N and rest.shape[2] == 3
    rest_cm = np.ascontiguousarray(rest.transpose(0, 2, 1)).reshape(N, rest.shape[1] * 3)   # channel-major
    table = np.concatenate([means, np.zeros((N, 3), np.float32), dc, rest_cm, opac, scales, quats], axis=1).astype("<f4")
    table = table[np.isfinite(table).all(axis=1)]
    names = _fields(rest_cm.shape[1])
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {table.shape[0]}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())
    return int(table.shape[0])


def read_ply(path: Union[str, Path], device="cpu") -> Dict[str, Tensor]:
    """The inverse of write_ply for binary little-endian float32 vertex tables (any property order; the fields above are
    looked up by name, unknown ones are ignored)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        names, n, fmt = [], None, None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no end_header")
            tok = line.decode("ascii").split()
            if tok[:1] == ["format"]:
                fmt = tok[1]
            elif tok[:2] == ["element", "vertex"]:
                n = int(tok[2])
            elif tok[:1] == ["element"]:
                raise ValueError(f"{path}: only a vertex element is supported")
            elif tok[:1] == ["property"]:
                if tok[1] != "float":
                    raise ValueError(f"{path}: property {tok[-1]} is {tok[1]}, expected float")
                names.append(tok[2])
            elif tok[:1] == ["end_header"]:
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError(f"{path}: expected `format binary_little_endian` and a vertex element")
        data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    col = {k: i for i, k in enumerate(names)}
    pick = lambda keys: torch.from_numpy(np.ascontiguousarray(data[:, [col[k] for k in keys]])).to(device)
    n_rest = sum(1 for k in names if k.startswith("f_rest_"))
    assert n_rest % 3 == 0, "f_rest_* must hold 3 channels"
    out = {"means": pick(["x", "y", "z"]), "features_dc": pick(["f_dc_0", "f_dc_1", "f_dc_2"]), "opacities": pick(["opacity"]),
           "scales": pick(["scale_0", "scale_1", "scale_2"]), "quats": pick(["rot_0", "rot_1", "rot_2", "rot_3"])}
    rest = pick([f"f_rest_{i}" for i in range(n_rest)])
    out["features_rest"] = rest.reshape(n, 3, n_rest // 3).transpose(1, 2).contiguous()
    return out
