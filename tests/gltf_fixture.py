"""Synthetic glTF 2.0 asset for the importer tests (no external assets are available offline).

write_gltf(dir, glb=False) writes `zoo.gltf` (+ `zoo.bin`, PNG images as data URIs) or `zoo.glb` (everything in the
BIN chunk) and returns the path.  The document exercises: nested TRS nodes and a `matrix` node, u16 / u32 / u8
indices, an interleaved (byteStride) vertex buffer, normalised u8 / u16 texture coordinates, tangents, base-colour /
metallic-roughness / normal / emissive textures, alphaMode MASK and BLEND, KHR_materials_transmission + ior,
the mirror shortcut, emissive x10, KHR_lights_punctual (directional, point, spot), a perspective camera on the
first-child chain and a decoy camera that find_camera's first-child-only descent must not reach."""
import base64
import io
import json
import math
import struct

import numpy as np


def _png(arr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG")
    return buf.getvalue()


def _quad(size=1.0):
    pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32) * np.float32(size)
    nrm = np.tile(np.array([0, 1, 0], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    tan = np.tile(np.array([1, 0, 0, 1], np.float32), (4, 1))
    idx = np.array([0, 2, 1, 0, 3, 2], np.uint32)
    return pos, nrm, uv, tan, idx


def _sphere(nt=8, nph=16):
    pos, idx = [], []
    for i in range(nt + 1):
        th = math.pi * i / nt
        for j in range(nph + 1):
            ph = 2 * math.pi * j / nph
            pos.append([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
    for i in range(nt):
        for j in range(nph):
            a, b = i * (nph + 1) + j, (i + 1) * (nph + 1) + j
            if i > 0:
                idx += [a, a + 1, b]
            if i < nt - 1:
                idx += [a + 1, b + 1, b]
    pos = np.array(pos, np.float32)
    return pos, pos.copy(), np.array(idx, np.uint32)


class _Builder:
    def __init__(self):
        self.bin = bytearray()
        self.views, self.accessors, self.images_png = [], [], []

    def view(self, data, stride=None, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        self.bin += data
        self.views.append(v)
        return len(self.views) - 1

    def accessor(self, arr, kind, view=None, offset=0, normalized=False, minmax=False):
        arr = np.ascontiguousarray(arr)
        ct = {np.dtype(np.float32): 5126, np.dtype(np.uint32): 5125, np.dtype(np.uint16): 5123, np.dtype(np.uint8): 5121}[arr.dtype]
        if view is None:
            view = self.view(arr.tobytes())
        a = {"bufferView": view, "componentType": ct, "count": int(arr.shape[0]), "type": kind}
        if offset:
            a["byteOffset"] = offset
        if normalized:
            a["normalized"] = True
        if minmax:
            a["min"], a["max"] = arr.min(axis=0).tolist(), arr.max(axis=0).tolist()
        self.accessors.append(a)
        return len(self.accessors) - 1


def build_document(glb):
    rng = np.random.default_rng(11)
    b = _Builder()
    # --- images ---------------------------------------------------------------------------------
    base_rgba = rng.integers(30, 255, (16, 16, 4), dtype=np.uint8)
    base_rgba[..., 3] = np.where((np.add.outer(np.arange(16) // 4, np.arange(16) // 4) % 2) == 0, 0, 255)  # alpha checker 0 / 255
    base_rgb = rng.integers(30, 255, (8, 12, 3), dtype=np.uint8)  # non power of two
    mr = rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)
    nm = np.zeros((8, 8, 3), np.uint8)
    nm[..., 0] = rng.integers(100, 156, (8, 8)); nm[..., 1] = rng.integers(100, 156, (8, 8)); nm[..., 2] = 250
    emis = np.zeros((8, 8, 3), np.uint8)
    emis[:, 4:, :] = 255  # left half black: triangles whose probes all land there get no light
    pngs = [_png(base_rgba), _png(base_rgb), _png(mr), _png(nm), _png(emis)]
    # --- geometry -------------------------------------------------------------------------------
    qp, qn, quv, qt, qi = _quad()
    meshes = []
    # 0: floor, separate accessors, u16 indices, tangents
    prim0 = {"attributes": {"POSITION": b.accessor(qp * np.float32(4), "VEC3", minmax=True), "NORMAL": b.accessor(qn, "VEC3"),
                            "TEXCOORD_0": b.accessor(quv * np.float32(3), "VEC2"), "TANGENT": b.accessor(qt, "VEC4")},
             "indices": b.accessor(qi.astype(np.uint16), "SCALAR"), "material": 0}
    meshes.append({"primitives": [prim0]})
    # 1: glass sphere, u32 indices, no uvs
    sp, sn, si = _sphere()
    prim1 = {"attributes": {"POSITION": b.accessor(sp, "VEC3", minmax=True), "NORMAL": b.accessor(sn, "VEC3")},
             "indices": b.accessor(si, "SCALAR"), "material": 1}
    meshes.append({"primitives": [prim1]})
    # 2: alpha-masked card, interleaved pos/normal/uv (stride 32), u8 indices
    inter = np.concatenate([qp, qn, quv], axis=1).astype(np.float32)
    v = b.view(inter.tobytes(), stride=32)
    prim2 = {"attributes": {"POSITION": b.accessor(qp, "VEC3", view=v, offset=0, minmax=True), "NORMAL": b.accessor(qn, "VEC3", view=v, offset=12),
                            "TEXCOORD_0": b.accessor(quv, "VEC2", view=v, offset=24)},
             "indices": b.accessor(qi.astype(np.uint8), "SCALAR"), "material": 2}
    # 3: emissive quad (two primitives in one mesh: constant emission, textured emission), normalised u16 / u8 uvs
    prim3a = {"attributes": {"POSITION": b.accessor(qp * np.float32(0.5), "VEC3", minmax=True), "NORMAL": b.accessor(qn, "VEC3")},
              "indices": b.accessor(qi.astype(np.uint16), "SCALAR"), "material": 3}
    uv16 = (quv * 65535).astype(np.uint16)
    prim3b = {"attributes": {"POSITION": b.accessor(qp * np.float32(0.5) + np.array([1.5, 0, 0], np.float32), "VEC3", minmax=True), "NORMAL": b.accessor(qn, "VEC3"),
                             "TEXCOORD_0": b.accessor(uv16, "VEC2", normalized=True)},
              "indices": b.accessor(qi.astype(np.uint16), "SCALAR"), "material": 4}
    meshes.append({"primitives": [prim2]})
    meshes.append({"primitives": [prim3a, prim3b]})
    # 4: mirror and blended quads, uv as normalised u8; one primitive without a material
    uv8 = (quv * 255).astype(np.uint8)
    uv8p = np.zeros((4, 4), np.uint8); uv8p[:, :2] = uv8  # 4-byte aligned rows (stride 4)
    v8 = b.view(uv8p.tobytes(), stride=4)
    prim4a = {"attributes": {"POSITION": b.accessor(qp, "VEC3", minmax=True), "NORMAL": b.accessor(qn, "VEC3"), "TEXCOORD_0": b.accessor(uv8, "VEC2", view=v8, normalized=True)},
              "indices": b.accessor(qi.astype(np.uint16), "SCALAR"), "material": 5}
    prim4b = {"attributes": {"POSITION": b.accessor(qp + np.array([2.5, 0, 0], np.float32), "VEC3", minmax=True), "NORMAL": b.accessor(qn, "VEC3")},
              "indices": b.accessor(qi.astype(np.uint16), "SCALAR"), "material": 6}
    prim4c = {"attributes": {"POSITION": b.accessor(qp + np.array([-2.5, 0, 0], np.float32), "VEC3", minmax=True)},
              "indices": b.accessor(qi.astype(np.uint16), "SCALAR")}
    meshes.append({"primitives": [prim4a, prim4b, prim4c]})
    # --- images / textures ----------------------------------------------------------------------
    images = []
    for png in pngs:
        if glb:
            images.append({"bufferView": b.view(png), "mimeType": "image/png"})
        else:
            images.append({"uri": "data:image/png;base64," + base64.b64encode(png).decode("ascii")})
    samplers = [{"wrapS": 10497, "wrapT": 10497}, {"wrapS": 33071, "wrapT": 33071}, {"wrapS": 33648, "wrapT": 33648}]
    textures = [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}, {"source": 2, "sampler": 2}, {"source": 3}, {"source": 4, "sampler": 0}]
    materials = [
        {"name": "floor", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.9, 1.0, 1.0], "baseColorTexture": {"index": 1}, "metallicFactor": 0.7, "roughnessFactor": 0.9,
                                                   "metallicRoughnessTexture": {"index": 2}}, "normalTexture": {"index": 3, "scale": 0.5}},
        {"name": "glass", "pbrMetallicRoughness": {"metallicFactor": 0.0}, "extensions": {"KHR_materials_transmission": {"transmissionFactor": 1.0}, "KHR_materials_ior": {"ior": 1.45}}},
        {"name": "card", "alphaMode": "MASK", "alphaCutoff": 0.5, "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0, "roughnessFactor": 0.6}},
        {"name": "lamp", "emissiveFactor": [0.5, 0.9, 0.1], "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.2, 0.2, 1.0], "metallicFactor": 0.0}},
        {"name": "lamp_tex", "emissiveFactor": [0.8, 0.0, 0.0], "emissiveTexture": {"index": 4}, "pbrMetallicRoughness": {"metallicFactor": 0.0}},
        {"name": "mirror", "pbrMetallicRoughness": {"metallicFactor": 1.0, "roughnessFactor": 0.0}},
        {"name": "blend", "alphaMode": "BLEND", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.4, 0.2, 0.4]}},
    ]
    # --- nodes ----------------------------------------------------------------------------------
    s45 = math.sin(math.radians(15)); c45 = math.cos(math.radians(15))
    nodes = [
        {"name": "root", "children": [1, 4, 5, 6, 7, 8, 9, 10]},                                                                   # 0
        {"name": "rig", "translation": [0.0, 2.2, 6.0], "rotation": [math.sin(math.radians(-7.5)), 0.0, 0.0, math.cos(math.radians(-7.5))], "children": [2, 3]},  # 1: pitched down 15 degrees
        {"name": "cam", "camera": 0},                                                                                             # 2 (first child of rig)
        {"name": "decoy", "camera": 1, "translation": [5, 5, 5]},                                                                 # 3 (never reached)
        {"name": "floor", "mesh": 0},                                                                                            # 4
        {"name": "ball", "mesh": 1, "translation": [-1.2, 0.8, 0.0], "scale": [0.8, 0.8, 0.8]},                                    # 5
        {"name": "card", "mesh": 2, "matrix": [0.7, 0, 0, 0, 0, 0, 0.7, 0, 0, -0.7, 0, 0, 1.0, 0.9, 0.5, 1]},                      # 6 (column-major: quad stood up)
        {"name": "lamps", "mesh": 3, "translation": [-0.5, 3.0, 0.0], "rotation": [1.0, 0.0, 0.0, 0.0]},                           # 7 (flipped: faces down)
        {"name": "shelf", "mesh": 4, "translation": [0.0, 0.01, -2.5], "rotation": [0.0, s45, 0.0, c45], "scale": [0.6, 1.0, 0.6]},  # 8
        {"name": "sun", "rotation": [math.sin(math.radians(-30)), 0.0, 0.0, math.cos(math.radians(-30))], "extensions": {"KHR_lights_punctual": {"light": 0}}},  # 9
        {"name": "bulbs", "translation": [2.0, 2.5, 1.0], "extensions": {"KHR_lights_punctual": {"light": 1}}, "children": [11]},   # 10
        {"name": "spot", "translation": [-4.0, 0.0, 0.0], "extensions": {"KHR_lights_punctual": {"light": 2}}},                    # 11
    ]
    doc = {
        "asset": {"version": "2.0", "generator": "tests/gltf_fixture.py"},
        "extensionsUsed": ["KHR_lights_punctual", "KHR_materials_transmission", "KHR_materials_ior"],
        "extensions": {"KHR_lights_punctual": {"lights": [
            {"type": "directional", "intensity": 2.0, "color": [0.9, 0.5, 0.1]},
            {"type": "point", "intensity": 30.0, "color": [1.0, 0.2, 0.2]},
            {"type": "spot", "intensity": 12.0, "color": [0.5, 1.0, 1.0], "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.6}}]}},
        "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes, "meshes": meshes, "materials": materials,
        "textures": textures, "images": images, "samplers": samplers,
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "znear": 0.05, "zfar": 500.0, "aspectRatio": 1.5}},
                    {"type": "perspective", "perspective": {"yfov": 1.2, "znear": 0.1}}],
        "accessors": b.accessors, "bufferViews": b.views,
    }
    return doc, bytes(b.bin)


def write_gltf(directory, glb=False):
    import os
    doc, blob = build_document(glb)
    if glb:
        doc["buffers"] = [{"byteLength": len(blob)}]
        js = json.dumps(doc).encode("utf-8")
        js += b" " * ((4 - len(js) % 4) % 4)
        blob += b"\0" * ((4 - len(blob) % 4) % 4)
        path = os.path.join(directory, "zoo.glb")
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
            f.write(struct.pack("<II", len(blob), 0x004E4942)); f.write(blob)
        return path
    doc["buffers"] = [{"uri": "zoo.bin", "byteLength": len(blob)}]
    with open(os.path.join(directory, "zoo.bin"), "wb") as f:
        f.write(blob)
    path = os.path.join(directory, "zoo.gltf")
    with open(path, "w") as f:
        json.dump(doc, f)
    return path
