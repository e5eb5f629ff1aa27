"""Minimal stand-in for the ``plyfile`` package (absent from this image; no network) -- TEST FIXTURE, not product
code.  Implements exactly the API the reference uses:

  PlyData.read(path); plydata.elements[0]["x"]; plydata.elements[0].properties[i].name; plydata['vertex'];
  'red' in vertices; vertices['x']; PlyElement.describe(structured_ndarray, 'vertex'); PlyData([el]).write(path)
  (scene/gaussian_model.py:19,291-323,502-508; scene/dataset_readers.py:22,135-148,163-178).

PLY files with scalar properties only, ``binary_little_endian`` (written and read) and ``ascii`` (read)."""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}
_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
          "f8": "double"}


class PlyProperty:
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = tuple(PlyProperty(n, data.dtype[n].str[1:]) for n in data.dtype.names)

    @staticmethod
    def describe(data, name):
        if data.dtype.names is None:
            raise ValueError("PlyElement.describe needs a structured array")
        return PlyElement(name, np.ascontiguousarray(data))

    @property
    def count(self):
        return self.data.shape[0]

    def __getitem__(self, key):
        return self.data[key]

    def __contains__(self, key):
        return key in self.data.dtype.names

    def __len__(self):
        return self.data.shape[0]


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="<"):
        self.elements = list(elements)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name):
        return any(e.name == name for e in self.elements)

    @staticmethod
    def read(stream):
        f = open(stream, "rb") if isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__") else stream
        try:
            if f.readline().strip() != b"ply":
                raise ValueError("not a PLY file")
            fmt, elems = None, []
            while True:
                line = f.readline()
                if not line:
                    raise ValueError("PLY header without end_header")
                tok = line.decode("ascii").split()
                if not tok or tok[0] in ("comment", "obj_info"):
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    elems.append((tok[1], int(tok[2]), []))
                elif tok[0] == "property":
                    if tok[1] == "list":
                        raise ValueError("list properties are not supported by this shim")
                    elems[-1][2].append((tok[2], _TYPES[tok[1]]))
                elif tok[0] == "end_header":
                    break
            out = []
            for name, count, props in elems:
                if fmt == "ascii":
                    dt = np.dtype([(n, t) for n, t in props])
                    rows = [f.readline().split() for _ in range(count)]
                    data = np.zeros(count, dtype=dt)
                    for j, (n, t) in enumerate(props):
                        data[n] = np.array([r[j] for r in rows], dtype=np.float64).astype(t) if count else []
                else:
                    order = "<" if fmt == "binary_little_endian" else ">"
                    dt = np.dtype([(n, order + t) for n, t in props])
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count).copy()
                out.append(PlyElement(name, data))
            return PlyData(out)
        finally:
            if f is not stream:
                f.close()

    def write(self, stream):
        f = open(stream, "wb") if isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__") else stream
        try:
            head = ["ply", "format binary_little_endian 1.0"]
            for e in self.elements:
                head.append(f"element {e.name} {e.count}")
                for p in e.properties:
                    head.append(f"property {_NAMES[p.dtype]} {p.name}")
            head.append("end_header")
            f.write(("\n".join(head) + "\n").encode("ascii"))
            for e in self.elements:
                dt = np.dtype([(n, "<" + e.data.dtype[n].str[1:]) for n in e.data.dtype.names])
                f.write(e.data.astype(dt).tobytes())
        finally:
            if f is not stream:
                f.close()
