"""Minimal stand-in for the third-party `plyfile` package, covering exactly what the reference uses:

  scene/gaussian_model.py:239-256  save_ply : PlyElement.describe(structured_array, 'vertex'); PlyData([el]).write(path)
  scene/gaussian_model.py:263-314  load_ply : PlyData.read(path); plydata.elements[0]["x"]; plydata.elements[0].properties[i].name
  scene/dataset_readers.py fetchPly / storePly: plydata['vertex'] with x y z nx ny nz (f4) and red green blue (u1)

so that `point_cloud.ply` files written by the reference (binary_little_endian, one `vertex` element, scalar
properties: x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*, all float) load here and vice versa.  Scalar
properties only -- list properties (faces) are outside the reference's use and raise.  Pure numpy; one read()/tofile()
per element, no per-vertex Python loop."""
from __future__ import annotations

import io
from typing import List, Sequence

import numpy as np

__all__ = ["PlyData", "PlyElement", "PlyProperty", "PlyParseError"]

# PLY type names <-> numpy type codes (both the classic and the sized spellings are accepted on read)
_PLY_TO_NP = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}
_FORMATS = {"ascii": None, "binary_little_endian": "<", "binary_big_endian": ">"}


class PlyParseError(Exception):
    pass


class PlyProperty:
    def __init__(self, name: str, val_dtype: str):
        self.name = name
        self.val_dtype = val_dtype          # numpy code without byte order, e.g. 'f4'

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.val_dtype!r})"

    def __str__(self):
        return f"property {_NP_TO_PLY[self.val_dtype]} {self.name}"


class PlyElement:
    def __init__(self, name: str, properties: Sequence[PlyProperty], data: np.ndarray):
        self.name = name
        self.properties = tuple(properties)
        self.data = data

    @property
    def count(self) -> int:
        return int(self.data.shape[0])

    @staticmethod
    def describe(data: np.ndarray, name: str) -> "PlyElement":
        """Element from a 1-D numpy structured array (one scalar field per property)."""
        if not isinstance(data, np.ndarray) or data.dtype.names is None or data.ndim != 1:
            raise TypeError("PlyElement.describe expects a one-dimensional numpy structured array")
        props = []
        for field in data.dtype.names:
            dt = data.dtype.fields[field][0]
            if dt.shape != () or dt.kind not in "iuf" or dt.str[1:] not in _NP_TO_PLY:
                raise ValueError(f"field {field!r}: only scalar int / uint / float properties are supported")
            props.append(PlyProperty(field, dt.str[1:]))
        return PlyElement(name, props, data)

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return self.count

    def dtype(self, byte_order: str = "<") -> np.dtype:
        return np.dtype([(p.name, byte_order + p.val_dtype) for p in self.properties])

    def header(self) -> str:
        return "\n".join([f"element {self.name} {self.count}"] + [str(p) for p in self.properties])


class PlyData:
    def __init__(self, elements: Sequence[PlyElement] = (), text: bool = False, byte_order: str = "<",
                 comments: Sequence[str] = ()):
        self.elements: List[PlyElement] = list(elements)
        self.text = bool(text)
        self.byte_order = byte_order
        self.comments = list(comments)

    def __getitem__(self, name: str) -> PlyElement:
        for el in self.elements:
            if el.name == name:
                return el
        raise KeyError(name)

    def __contains__(self, name: str) -> bool:
        return any(el.name == name for el in self.elements)

    # ---- writing -------------------------------------------------------------------------------------------------
    def header(self) -> str:
        fmt = "ascii" if self.text else ("binary_little_endian" if self.byte_order == "<" else "binary_big_endian")
        lines = ["ply", f"format {fmt} 1.0"] + [f"comment {c}" for c in self.comments]
        lines += [el.header() for el in self.elements] + ["end_header"]
        return "\n".join(lines) + "\n"

    def write(self, stream) -> None:
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            f.write(self.header().encode("ascii"))
            for el in self.elements:
                if self.text:
                    buf = io.StringIO()
                    np.savetxt(buf, np.column_stack([el.data[p.name] for p in el.properties]) if el.count else np.zeros((0, 1)),
                               fmt=["%d" if p.val_dtype[0] in "iu" else "%.9g" for p in el.properties] or "%g")
                    f.write(buf.getvalue().encode("ascii"))
                else:
                    f.write(np.ascontiguousarray(el.data.astype(el.dtype(self.byte_order), copy=False)).tobytes())
        finally:
            if own:
                f.close()

    # ---- reading -------------------------------------------------------------------------------------------------
    @staticmethod
    def read(stream) -> "PlyData":
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            return PlyData._parse(f)
        finally:
            if own:
                f.close()

    @staticmethod
    def _parse(f) -> "PlyData":
        if f.readline().strip() != b"ply":
            raise PlyParseError("not a PLY file (missing 'ply' magic)")
        fmt = None
        comments: List[str] = []
        specs = []                      # [name, count, [PlyProperty]]
        while True:
            raw = f.readline()
            if not raw:
                raise PlyParseError("unexpected end of file inside the header")
            tok = raw.decode("ascii", "replace").strip().split()
            if not tok:
                continue
            if tok[0] == "format":
                if len(tok) != 3 or tok[1] not in _FORMATS:
                    raise PlyParseError(f"unsupported format line: {' '.join(tok)}")
                fmt = tok[1]
            elif tok[0] in ("comment", "obj_info"):
                comments.append(" ".join(tok[1:]))
            elif tok[0] == "element":
                specs.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if not specs:
                    raise PlyParseError("property before any element")
                if tok[1] == "list":
                    raise PlyParseError("list properties are not supported (the reference's files have none)")
                if tok[1] not in _PLY_TO_NP:
                    raise PlyParseError(f"unknown property type {tok[1]!r}")
                specs[-1][2].append(PlyProperty(tok[2], _PLY_TO_NP[tok[1]]))
            elif tok[0] == "end_header":
                break
            else:
                raise PlyParseError(f"unknown header keyword {tok[0]!r}")
        if fmt is None:
            raise PlyParseError("missing format line")
        order = _FORMATS[fmt]
        elements = []
        for name, count, props in specs:
            if order is None:           # ascii: `count` whitespace-separated rows
                native = np.dtype([(p.name, "=" + p.val_dtype) for p in props])
                data = np.empty(count, dtype=native)
                rows = []
                while len(rows) < count:
                    line = f.readline()
                    if not line:
                        raise PlyParseError(f"element {name}: expected {count} rows, file ended after {len(rows)}")
                    if line.strip():
                        rows.append(line.split())
                cols = list(zip(*rows)) if rows else [[] for _ in props]
                if rows and len(cols) != len(props):
                    raise PlyParseError(f"element {name}: rows have {len(cols)} columns, header declares {len(props)}")
                for p, col in zip(props, cols):
                    data[p.name] = np.asarray(col, dtype=np.float64).astype(p.val_dtype) if p.val_dtype[0] == "f" else \
                        np.asarray([int(v) for v in col], dtype=p.val_dtype)
            else:
                dt = np.dtype([(p.name, order + p.val_dtype) for p in props])
                buf = f.read(dt.itemsize * count)
                if len(buf) != dt.itemsize * count:
                    raise PlyParseError(f"element {name}: expected {dt.itemsize * count} bytes, got {len(buf)}")
                data = np.frombuffer(buf, dtype=dt, count=count).astype(dt.newbyteorder("="), copy=True)
            elements.append(PlyElement(name, props, data))
        return PlyData(elements, text=order is None, byte_order=order or "<", comments=comments)
