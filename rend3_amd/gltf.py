"""Minimal glTF 2.0 / GLB reader (row N1 of SURVEY.md section 8f, first slice): container parsing and accessor decoding
into numpy arrays -- enough to feed meshes, skins and factor-only PBR materials of binary glTF files into the
Renderer-shaped API.  Follows what the reference reads through the `gltf` crate in
rend3-gltf/src/lib.rs:607-678 (load_meshes: positions, normals, tangents, uv0/1, colours, joints, weights, indices) and
examples/src/static_gltf/mod.rs:5-41.  Textures / images / KTX2 / DDS (row N2) are not handled.
"""
import json
import os
import struct

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class Gltf:
    def __init__(self, path):
        self.path = path
        data = open(path, "rb").read()
        self.buffers = []
        if data[:4] == b"glTF":
            version, length = struct.unpack_from("<II", data, 4)
            assert version == 2, "only glTF 2.0"
            off, chunks = 12, []
            while off < length:
                clen, ctype = struct.unpack_from("<II", data, off)
                chunks.append((ctype, data[off + 8: off + 8 + clen]))
                off += 8 + clen
            assert chunks and chunks[0][0] == 0x4E4F534A, "first GLB chunk must be JSON"
            self.json = json.loads(chunks[0][1].decode("utf-8"))
            glb_bin = next((c for t, c in chunks[1:] if t == 0x004E4942), None)
        else:
            self.json = json.loads(data.decode("utf-8"))
            glb_bin = None
        for b in self.json.get("buffers", []):
            if "uri" not in b:
                assert glb_bin is not None, "buffer without uri needs a GLB BIN chunk"
                self.buffers.append(glb_bin)
            elif b["uri"].startswith("data:"):
                import base64
                self.buffers.append(base64.b64decode(b["uri"].split(",", 1)[1]))
            else:
                self.buffers.append(open(os.path.join(os.path.dirname(path), b["uri"]), "rb").read())

    def accessor(self, index):
        a = self.json["accessors"][index]
        assert "sparse" not in a, "sparse accessors are not supported"
        dt = np.dtype(_COMPONENT[a["componentType"]])
        width = _WIDTH[a["type"]]
        count = a["count"]
        if "bufferView" not in a:
            return np.zeros((count, width), dtype=dt)
        bv = self.json["bufferViews"][a["bufferView"]]
        buf = self.buffers[bv["buffer"]]
        base = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        elem = dt.itemsize * width
        stride = bv.get("byteStride", 0) or elem
        if stride == elem:
            arr = np.frombuffer(buf, dtype=dt, count=count * width, offset=base).reshape(count, width)
        else:
            raw = np.frombuffer(buf, dtype=np.uint8, count=(count - 1) * stride + elem, offset=base)
            idx = (np.arange(count)[:, None] * stride + np.arange(elem)[None, :]).reshape(-1)
            arr = raw[idx].view(dt).reshape(count, width)
        if a.get("normalized") and dt != np.float32:
            info = np.iinfo(dt)
            arr = np.maximum(arr.astype(np.float32) / np.float32(info.max), -1.0 if info.min < 0 else 0.0)
        return arr.copy()

    def primitive(self, mesh=0, primitive=0):
        """Attribute arrays of one primitive, as the reference's loaders read them (u32 indices, f32 attributes)."""
        p = self.json["meshes"][mesh]["primitives"][primitive]
        assert p.get("mode", 4) == 4, "only triangle lists"
        at = p["attributes"]
        out = {"positions": self.accessor(at["POSITION"]).astype(np.float32)}
        if "NORMAL" in at:
            out["normals"] = self.accessor(at["NORMAL"]).astype(np.float32)
        if "TANGENT" in at:
            out["tangents"] = self.accessor(at["TANGENT"]).astype(np.float32)[:, :3]  # Vec4::truncate
        if "TEXCOORD_0" in at:
            out["uv0"] = self.accessor(at["TEXCOORD_0"]).astype(np.float32)
        if "JOINTS_0" in at:
            out["joints"] = self.accessor(at["JOINTS_0"]).astype(np.uint16)
            out["weights"] = self.accessor(at["WEIGHTS_0"]).astype(np.float32)
        if "indices" in p:
            out["indices"] = self.accessor(p["indices"]).astype(np.uint32).reshape(-1)
        else:
            out["indices"] = np.arange(len(out["positions"]), dtype=np.uint32)
        out["material"] = p.get("material")
        return out

    def base_color_factor(self, material):
        if material is None:
            return (1.0, 1.0, 1.0, 1.0)
        pbr = self.json["materials"][material].get("pbrMetallicRoughness", {})
        return tuple(pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0]))
