"""PLY checkpoint format of the Gaussian set: same four elements, property names and order as the reference writes with
plyfile (R/scene/gaussian_model_dpsr_dynamic_anchor.py:238-289 save_ply, :296-362 load_ply):

    element vertex P            x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3       (float32)
    element density_thres 1     density_thres
    element gaussian_center 1   gaussian_center_x gaussian_center_y gaussian_center_z
    element gaussian_scale 1    gaussian_scale

binary_little_endian 1.0 (plyfile's default for PlyData.write).  f_dc / f_rest are stored channel-major, i.e. the
(P, coeffs, 3) parameter transposed to (P, 3, coeffs) and flattened.  plyfile is not a dependency: the format is a text
header followed by packed little-endian records, read / written here with numpy structured arrays (scalar properties only,
which is all this format uses).  Reader also accepts ascii PLY and float64 / integer scalar properties.
"""
import os

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}
_NP_TO_PLY = {"f4": "float", "f8": "double", "i4": "int", "u1": "uchar", "i2": "short", "u2": "ushort", "u4": "uint", "i1": "char"}


def write_ply(path, elements):
    """elements: list of (name, structured ndarray).  binary_little_endian, plyfile-style header."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    head = ["ply", "format binary_little_endian 1.0"]
    for name, arr in elements:
        head.append(f"element {name} {len(arr)}")
        for field in arr.dtype.names:
            head.append(f"property {_NP_TO_PLY[arr.dtype[field].str[1:]]} {field}")
    head.append("end_header")
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        for _, arr in elements:
            fh.write(np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder("<"))).tobytes())


def read_ply(path):
    """-> dict name -> structured ndarray (in file order: use list(result) for positional access like plydata.elements[i])."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    lines = data[:end].decode("ascii").splitlines()
    if lines[0].strip() != "ply":
        raise ValueError(f"{path}: not a PLY file")
    fmt, elems = None, []
    for ln in lines[1:]:
        tok = ln.split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elems.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError(f"{path}: list properties are not part of this checkpoint format")
            elems[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
    out, off = {}, end
    if fmt == "ascii":
        toks = data[end:].split()
        pos = 0
        for name, n, props in elems:
            arr = np.empty(n, dtype=[(p, t) for p, t in props])
            for i in range(n):
                for p, t in props:
                    arr[p][i] = np.array(toks[pos].decode(), dtype=t)
                    pos += 1
            out[name] = arr
        return out
    bo = "<" if fmt == "binary_little_endian" else ">"
    for name, n, props in elems:
        dt = np.dtype([(p, bo + t) for p, t in props])
        out[name] = np.frombuffer(data, dtype=dt, count=n, offset=off).copy()
        off += n * dt.itemsize
    return out


def attribute_names(n_dc, n_rest, n_scale=3, n_rot=4):
    """construct_list_of_attributes (:238-251)."""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
            + ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)])


def save_gaussians(path, xyz, normal, f_dc, f_rest, opacity, scaling, rotation, density_thres=0.0, gaussian_center=(0.0, 0.0, 0.0),
                   gaussian_scale=1.0):
    """All arrays numpy float32: f_dc (P, 1, 3), f_rest (P, 15, 3) as the model keeps them."""
    P = xyz.shape[0]
    dc = np.transpose(f_dc, (0, 2, 1)).reshape(P, -1)
    rest = np.transpose(f_rest, (0, 2, 1)).reshape(P, -1)
    names = attribute_names(dc.shape[1], rest.shape[1], scaling.shape[1], rotation.shape[1])
    table = np.concatenate((xyz, normal, dc, rest, opacity.reshape(P, 1), scaling, rotation), axis=1).astype(np.float32)
    vertex = np.empty(P, dtype=[(n, "<f4") for n in names])
    for i, n in enumerate(names):
        vertex[n] = table[:, i]
    one = lambda fields, vals: np.array([tuple(np.float32(v) for v in vals)], dtype=[(f, "<f4") for f in fields])
    write_ply(path, [("vertex", vertex), ("density_thres", one(["density_thres"], [density_thres])),
                     ("gaussian_center", one(["gaussian_center_x", "gaussian_center_y", "gaussian_center_z"], gaussian_center)),
                     ("gaussian_scale", one(["gaussian_scale"], [gaussian_scale]))])


def load_gaussians(path, max_sh_degree=3):
    """-> dict of float32 arrays in the model's layout (+ density_thres, gaussian_center, gaussian_scale).  A plain
    3D-GS PLY (vertex element only) loads too; the three extra elements then default like a fresh model."""
    ply = read_ply(path)
    v = ply["vertex"] if "vertex" in ply else ply[list(ply)[0]]
    col = lambda n: np.asarray(v[n], np.float32)
    P = len(v)
    xyz = np.stack((col("x"), col("y"), col("z")), 1)
    normal = np.stack((col("nx"), col("ny"), col("nz")), 1) if "nx" in v.dtype.names else np.zeros((P, 3), np.float32)
    f_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), 1).reshape(P, 3, 1)
    rest_names = sorted((n for n in v.dtype.names if n.startswith("f_rest_")), key=lambda n: int(n.split("_")[-1]))
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties do not match sh degree {max_sh_degree}")
    f_rest = np.stack([col(n) for n in rest_names], 1).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1)
    scale_names = sorted((n for n in v.dtype.names if n.startswith("scale_")), key=lambda n: int(n.split("_")[-1]))
    rot_names = sorted((n for n in v.dtype.names if n.startswith("rot")), key=lambda n: int(n.split("_")[-1]))
    out = dict(xyz=xyz, normal=normal, features_dc=np.ascontiguousarray(np.transpose(f_dc, (0, 2, 1))),
               features_rest=np.ascontiguousarray(np.transpose(f_rest, (0, 2, 1))), opacity=col("opacity").reshape(P, 1),
               scaling=np.stack([col(n) for n in scale_names], 1), rotation=np.stack([col(n) for n in rot_names], 1),
               density_thres=np.float32(0.0), gaussian_center=np.zeros(3, np.float32), gaussian_scale=np.float32(1.0))
    if "density_thres" in ply:
        out["density_thres"] = np.float32(ply["density_thres"]["density_thres"][0])
    if "gaussian_center" in ply:
        c = ply["gaussian_center"]
        out["gaussian_center"] = np.array([c["gaussian_center_x"][0], c["gaussian_center_y"][0], c["gaussian_center_z"][0]], np.float32)
    if "gaussian_scale" in ply:
        out["gaussian_scale"] = np.float32(ply["gaussian_scale"]["gaussian_scale"][0])
    return out
