#!/usr/bin/env python3
"""Packs the assets of the reference's animation example (examples/src/animation/resources: scene.gltf + scene.bin +
textures/, cube_3.gltf; CC-BY, see animation-LICENSE.txt) into two self-contained GLB fixtures next to this script:
animation-character.glb and animation-cube.glb, plus the example's screenshot as animation-screenshot.png.
Run in the build container (reads /root/reference); the outputs are committed, the reference is not needed at test time.

  python tests/golden/make_animation_fixture.py
"""
import base64
import json
import os
import shutil
import struct

SRC = "/root/reference/examples/src/animation"
HERE = os.path.dirname(os.path.abspath(__file__))


def pack(gltf_name, out_name):
    root = os.path.join(SRC, "resources")
    doc = json.load(open(os.path.join(root, gltf_name)))
    blobs = []
    for b in doc.get("buffers", []):
        uri = b.pop("uri")
        blobs.append(base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(root, uri), "rb").read())
    # one BIN chunk: the original buffers back to back, then the image files
    offsets, cur = [], 0
    for b in blobs:
        offsets.append(cur)
        cur += (len(b) + 3) & ~3
    body = bytearray(cur)
    for off, b in zip(offsets, blobs):
        body[off:off + len(b)] = b
    for bv in doc.get("bufferViews", []):
        bv["byteOffset"] = bv.get("byteOffset", 0) + offsets[bv["buffer"]]
        bv["buffer"] = 0
    for im in doc.get("images", []):
        if "uri" in im:
            uri = im.pop("uri")
            data = base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(root, uri), "rb").read()
            off = len(body)
            body += data + bytes((-len(data)) % 4)
            doc.setdefault("bufferViews", []).append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
            im["bufferView"] = len(doc["bufferViews"]) - 1
            im["mimeType"] = "image/jpeg" if uri.lower().endswith((".jpg", ".jpeg")) else "image/png"
    doc["buffers"] = [{"byteLength": len(body)}]
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    out = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(body))
    out += struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(body), b"BIN\0") + bytes(body)
    open(os.path.join(HERE, out_name), "wb").write(out)
    print(out_name, len(out), "bytes")


if __name__ == "__main__":
    pack("scene.gltf", "animation-character.glb")
    pack("cube_3.gltf", "animation-cube.glb")
    shutil.copyfile(os.path.join(SRC, "screenshot.png"), os.path.join(HERE, "animation-screenshot.png"))
    shutil.copyfile(os.path.join(SRC, "resources", "LICENSE"), os.path.join(HERE, "animation-LICENSE.txt"))
