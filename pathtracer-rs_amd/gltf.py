"""glTF 2.0 importer (host side): restates
  common/importer/gltf.rs:3-117           (camera search, default camera, node transforms)
  pathtracer/importer/gltf.rs:20-584      (materials, meshes, emissive area lights, punctual lights, default env light)
on top of the RenderScene builders of scene.py, so the result feeds the HIP library and the oracle alike.

What the reference gets from third-party crates is restated from their published behaviour (parity unpinned,
DESIGN.md section 6): `gltf 1.1.0` (document, accessors, buffers, KHR_lights_punctual, KHR_materials_transmission,
KHR_materials_ior; node matrices are used as given where the crate would decompose and the importer recompose
T*R*S), `image 0.23.14` (PNG/JPEG decoding -> here PIL), nalgebra-glm `quat_to_mat4`.

Quirks kept on purpose:
  * find_camera only ever descends into the FIRST child of a node (`return` inside the loop, gltf.rs:36-44);
  * emissive strength and punctual light colour use the RED factor for all three channels, emissive x10 (391-401,461-465);
  * spot lights become point lights (479-485); MirroredRepeat wraps as Repeat (30-36);
  * alphaMode BLEND with alpha < 1 becomes glass with ior 1.33 (236-256); metallic 1 / roughness 0 becomes a mirror (259-261);
  * alphaCutoff is ignored: MASK materials reject only where the alpha texel is exactly 0 (shape.rs:227-244);
  * normal-map images are read as a tightly packed RGB stream even when the file has an alpha channel (203-204).
"""
import base64
import json
import math
import os
import struct

import numpy as np

from . import abi
from . import textures as tx
from .scene import Camera, RenderScene, _quat_from_rotation_matrix

F = np.float32
DEFAULT_Z_NEAR, DEFAULT_Z_FAR = 0.01, 10000.0  # common/mod.rs:17-18

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class GltfError(ValueError):
    pass


# ---- container ------------------------------------------------------------------------------------
def _load_document(path):
    """-> (json dict, [buffer bytes]).  .glb: 12-byte header + JSON chunk + optional BIN chunk."""
    raw = open(path, "rb").read()
    base = os.path.dirname(os.path.abspath(path))
    glb_bin = None
    if raw[:4] == b"glTF":
        magic, version, length = struct.unpack_from("<4sII", raw, 0)
        if version != 2:
            raise GltfError("unsupported GLB version %d" % version)
        off, doc = 12, None
        while off + 8 <= min(length, len(raw)):
            clen, ctype = struct.unpack_from("<II", raw, off)
            chunk = raw[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                doc = json.loads(chunk.decode("utf-8"))
            elif ctype == 0x004E4942 and glb_bin is None:
                glb_bin = chunk
            off += 8 + clen + ((4 - clen % 4) % 4)
        if doc is None:
            raise GltfError("GLB without a JSON chunk")
    else:
        doc = json.loads(raw.decode("utf-8"))
    buffers = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if glb_bin is None:
                raise GltfError("buffer %d has no uri and the file has no BIN chunk" % i)
            data = glb_bin
        elif uri.startswith("data:"):
            data = base64.b64decode(uri.split(",", 1)[1])
        else:
            data = open(os.path.join(base, uri), "rb").read()
        if len(data) < b.get("byteLength", 0):
            raise GltfError("buffer %d shorter than its byteLength" % i)
        buffers.append(data)
    return doc, buffers, base


def _accessor(doc, buffers, index):
    """Accessor -> ndarray (count, ncomp) in its stored component type (byteStride honoured)."""
    a = doc["accessors"][index]
    if "sparse" in a:
        raise GltfError("sparse accessors are not supported")
    dt = np.dtype(_COMPONENT[a["componentType"]]).newbyteorder("<")
    nc = _NCOMP[a["type"]]
    count = a["count"]
    if "bufferView" not in a:
        return np.zeros((count, nc), dtype=dt)
    bv = doc["bufferViews"][a["bufferView"]]
    data = buffers[bv["buffer"]]
    start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    elem = dt.itemsize * nc
    stride = bv.get("byteStride", 0) or elem
    if count and start + stride * (count - 1) + elem > len(data):
        raise GltfError("accessor %d reaches past its buffer" % index)
    if stride == elem:
        arr = np.frombuffer(data, dtype=dt, count=count * nc, offset=start).reshape(count, nc)
    else:
        arr = np.ndarray(shape=(count, nc), dtype=dt, buffer=data, offset=start, strides=(stride, dt.itemsize))
    return np.array(arr)


def _normalized_to_f32(arr):
    """gltf `into_f32()` for texture coordinates / colours: unsigned normalised integers -> [0,1]."""
    if arr.dtype == np.float32:
        return arr.astype(np.float32)
    if arr.dtype == np.uint8:
        return (arr.astype(np.float32) / F(255.0)).astype(np.float32)
    if arr.dtype == np.uint16:
        return (arr.astype(np.float32) / F(65535.0)).astype(np.float32)
    raise GltfError("unsupported normalised component type %s" % arr.dtype)


def _decode_image(doc, buffers, base, index):
    """gltf::image::Data: (rows, cols, channels) uint8, channels 3 (R8G8B8) or 4 (R8G8B8A8); other formats -> None."""
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise GltfError("decoding glTF images needs PIL") from e
    import io
    im = doc["images"][index]
    if "uri" in im:
        uri = im["uri"]
        data = base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(base, uri), "rb").read()
    else:
        bv = doc["bufferViews"][im["bufferView"]]
        o = bv.get("byteOffset", 0)
        data = buffers[bv["buffer"]][o:o + bv["byteLength"]]
    img = Image.open(io.BytesIO(data))
    if img.mode in ("RGB", "RGBA"):
        return np.array(img, dtype=np.uint8)
    if img.mode == "P":
        return np.array(img.convert("RGBA" if "transparency" in img.info else "RGB"), dtype=np.uint8)
    return None  # R8 / R8G8 / 16-bit: "unsupported image format" in the reference (gltf.rs:91-97)


# ---- transforms -----------------------------------------------------------------------------------
def _matmul4(a, b):
    """4x4 product in binary32, inner index ascending (nalgebra's small-matrix gemm order)."""
    r = np.zeros((4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            acc = a[i, 0] * b[0, j]
            for k in range(1, 4):
                acc = acc + a[i, k] * b[k, j]
            r[i, j] = acc
    return r


def _quat_to_mat4(q):
    """glm::quat_to_mat4 = UnitQuaternion::to_homogeneous (q = i, j, k, w)."""
    i, j, k, w = (F(v) for v in q)
    ww, ii, jj, kk = w * w, i * i, j * j, k * k
    ij, wk, wj, ik, jk, wi = i * j * F(2), w * k * F(2), w * j * F(2), i * k * F(2), j * k * F(2), w * i * F(2)
    m = np.eye(4, dtype=np.float32)
    m[0, :3] = [ww + ii - jj - kk, ij - wk, wj + ik]
    m[1, :3] = [wk + ij, ww - ii + jj - kk, jk - wi]
    m[2, :3] = [ik - wj, wi + jk, ww - ii - jj + kk]
    return m


def node_transform(node):
    """trans_from_gltf (common/importer/gltf.rs:87-96): T * R * S; a `matrix` node is used as given."""
    if "matrix" in node:
        return np.array(node["matrix"], dtype=np.float32).reshape(4, 4).T.copy()  # glTF stores column-major
    t = np.eye(4, dtype=np.float32)
    t[:3, 3] = np.array(node.get("translation", [0, 0, 0]), dtype=np.float32)
    r = _quat_to_mat4(node.get("rotation", [0, 0, 0, 1]))
    s = np.eye(4, dtype=np.float32)
    sc = np.array(node.get("scale", [1, 1, 1]), dtype=np.float32)
    s[0, 0], s[1, 1], s[2, 2] = sc
    return _matmul4(_matmul4(t, r), s)


def _xf_points(m, p):
    """Projective3 * Point3 per vertex: row dot in x, y, z, w order, divided by w when the last row is not affine."""
    p = np.asarray(p, dtype=np.float32)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((m[r, 0] * p[:, 0] + m[r, 1] * p[:, 1]) + m[r, 2] * p[:, 2]) + m[r, 3]
    if not (m[3, 0] == 0 and m[3, 1] == 0 and m[3, 2] == 0 and m[3, 3] == 1):
        w = ((m[3, 0] * p[:, 0] + m[3, 1] * p[:, 1]) + m[3, 2] * p[:, 2]) + m[3, 3]
        out = out / w[:, None]
    return out.astype(np.float32)


def _xf_vectors(m, v):
    v = np.asarray(v, dtype=np.float32)
    out = np.empty_like(v)
    for r in range(3):
        out[:, r] = (m[r, 0] * v[:, 0] + m[r, 1] * v[:, 1]) + m[r, 2] * v[:, 2]
    return out.astype(np.float32)


# ---- materials ------------------------------------------------------------------------------------
def _wrap_mode(doc, tex_info):
    """wrap_mode_from_gtlf (gltf.rs:30-36); the sampler's wrapS must equal wrapT (assert at 47, 112, 196, 312)."""
    t = doc["textures"][tex_info["index"]]
    s = doc.get("samplers", [])[t["sampler"]] if "sampler" in t else {}
    ws, wt = s.get("wrapS", 10497), s.get("wrapT", 10497)
    if ws != wt:
        raise GltfError("sampler with wrapS != wrapT (the reference asserts)")
    return abi.WRAP_CLAMP if ws == 33071 else abi.WRAP_REPEAT


class _Importer:
    def __init__(self, path):
        self.doc, self.buffers, self.base = _load_document(path)
        self.scene = RenderScene()
        self._images = {}
        self.materials = []  # RenderScene material id per glTF material (+1: default first)
        self.deferred_lights = []  # `preprocess_lights`: directional lights join the list after the traversal (gltf.rs:567-575)

    def image(self, tex_info):
        src = self.doc["textures"][tex_info["index"]]["source"]
        if src not in self._images:
            self._images[src] = _decode_image(self.doc, self.buffers, self.base, src)
        return self._images[src]

    def color_texture(self, tex_info, factor):
        """color_texture_from_gltf (gltf.rs:38-98): RGB (alpha dropped), gamma-decoded, scaled by `factor`."""
        img = self.image(tex_info)
        if img is None:
            return None
        return tx.spectrum_texture(self.scene, img[..., :3], scale=factor, wrap=_wrap_mode(self.doc, tex_info), gamma=True)

    def material(self, m):
        """material_from_gltf (gltf.rs:170-296)."""
        s = self.scene
        pbr = m.get("pbrMetallicRoughness", {})
        bcf = [float(v) for v in pbr.get("baseColorFactor", [1, 1, 1, 1])]
        color_factor = tx.inverse_gamma_correct(np.array(bcf[:3], dtype=np.float32))  # Spectrum::from_slice_4(.., true)
        color_tex = None
        if "baseColorTexture" in pbr:
            color_tex = self.color_texture(pbr["baseColorTexture"], color_factor)
        if color_tex is None:
            color_tex = s.const_rgb(color_factor)
        normal_tex = None
        if "normalTexture" in m:
            info = m["normalTexture"]
            img = self.image(info)
            if img is None:
                raise GltfError("normal texture with an unsupported image format (the reference unwraps)")
            rows, cols = img.shape[:2]
            rgb = img.reshape(-1)[:rows * cols * 3].reshape(rows, cols, 3)  # RgbImage::from_raw on the raw stream
            sc = float(info.get("scale", 1.0))
            normal_tex = tx.normal_map_texture(s, rgb, scale=(sc, sc), wrap=_wrap_mode(self.doc, info))

        def with_normal(mat):
            return s.add_material(abi.MAT_NORMAL, [normal_tex], inner=mat) if normal_tex is not None else mat
        ext = m.get("extensions", {})
        transmission = float(ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0)) if "KHR_materials_transmission" in ext else 0.0
        ior = float(ext.get("KHR_materials_ior", {}).get("ior", 1.5)) if "KHR_materials_ior" in ext else 1.5
        if transmission == 1.0:  # total transparency, pure glass
            return with_normal(s.add_material(abi.MAT_GLASS, [s.const_rgb([1, 1, 1]), s.const_rgb([1, 1, 1]), s.const_f(ior)]))
        alpha = F(bcf[3])
        if m.get("alphaMode", "OPAQUE") == "BLEND" and alpha < 1.0:
            kt = (np.ones(3, dtype=np.float32) - alpha * color_factor).astype(np.float32)
            return with_normal(s.add_material(abi.MAT_GLASS, [s.const_rgb([1, 1, 1]), s.const_rgb(kt), s.const_f(1.33)]))
        metallic, roughness = float(pbr.get("metallicFactor", 1.0)), float(pbr.get("roughnessFactor", 1.0))
        if metallic == 1.0 and roughness == 0.0:
            return s.add_material(abi.MAT_MIRROR)
        metallic_tex, roughness_tex = s.const_f(metallic), s.const_f(roughness)
        if "metallicRoughnessTexture" in pbr:  # metallic = B, roughness = G (gltf.rs:100-168)
            info = pbr["metallicRoughnessTexture"]
            img = self.image(info)
            if img is not None:
                wrap = _wrap_mode(self.doc, info)
                metallic_tex = tx.float_texture(s, img[..., 2], scale=metallic, wrap=wrap)
                roughness_tex = tx.float_texture(s, img[..., 1], scale=roughness, wrap=wrap)
        return with_normal(s.add_material(abi.MAT_DISNEY, [color_tex, metallic_tex, s.const_f(ior), roughness_tex]))

    # ---- meshes / lights ---------------------------------------------------------------------------
    def primitive(self, prim, xf):
        """shapes_from_gltf_prim + the emissive part of populate_scene (gltf.rs:298-445)."""
        s, doc = self.scene, self.doc
        if prim.get("mode", 4) != 4:
            raise GltfError("only triangle-list primitives are supported (the reference unwraps read_indices on them)")
        if "indices" not in prim:
            raise GltfError("primitive without indices (the reference unwraps)")
        mat = doc["materials"][prim["material"]] if "material" in prim else {}
        alpha_tex = -1
        bct = mat.get("pbrMetallicRoughness", {}).get("baseColorTexture")
        if bct is not None and mat.get("alphaMode", "OPAQUE") == "MASK":
            img = self.image(bct)
            if img is None or img.shape[2] != 4:
                raise GltfError("alpha-mask material whose base colour image is not RGBA8 (the reference asserts)")
            alpha_tex = tx.float_texture(s, img[..., 3], scale=1.0, wrap=_wrap_mode(doc, bct))
        attr = prim["attributes"]
        idx = _accessor(doc, self.buffers, prim["indices"]).astype(np.uint32).reshape(-1)
        idx = idx[:(len(idx) // 3) * 3].reshape(-1, 3)  # chunks_exact(3)
        pos = _xf_points(xf, _accessor(doc, self.buffers, attr["POSITION"]).astype(np.float32))
        normal = _xf_vectors(xf, _accessor(doc, self.buffers, attr["NORMAL"]).astype(np.float32)) if "NORMAL" in attr else None
        tangent = _xf_vectors(xf, _accessor(doc, self.buffers, attr["TANGENT"]).astype(np.float32)[:, :3]) if "TANGENT" in attr else None
        uv = _normalized_to_f32(_accessor(doc, self.buffers, attr["TEXCOORD_0"])) if "TEXCOORD_0" in attr else None
        material = self.materials[prim["material"] + 1] if "material" in prim else self.materials[0]
        mi = s.add_mesh(pos, idx, material, normal=normal, uv=uv, tangent=tangent, alpha_mask_tex=alpha_tex)
        ef = [float(v) for v in mat.get("emissiveFactor", [0, 0, 0])]
        e = F(10.0) * F(ef[0])  # EMISSIVE_SCALING_FACTOR, red factor for all channels
        if e == 0:
            return
        ke, ke_img = s.const_rgb([e, e, e]), None
        if "emissiveTexture" in mat:
            t = self.color_texture(mat["emissiveTexture"], [e, e, e])
            if t is not None:
                ke, ke_img = t, s.textures[t]["levels"][0]
        for t_i in range(len(idx)):
            if ke_img is not None and not self._has_emission(ke_img, s.textures[ke]["wrap"], uv, idx[t_i]):
                continue
            s.lights.append(dict(kind=abi.LIGHT_AREA, mesh=mi, tri=t_i, ke_tex=ke))

    @staticmethod
    def _has_emission(level0, wrap, uv, tri):
        """populate_scene's 10x10 probe (gltf.rs:413-427): ke at Triangle::sample((x/10, y/10)), level-0 bilinear."""
        uvs = np.array([[0, 0], [1, 0], [1, 1]], dtype=np.float32) if uv is None else uv[tri]
        rows, cols = level0.shape[:2]
        for x in range(10):
            for y in range(10):
                u0, u1 = F(x) * F(0.1), F(y) * F(0.1)
                su0 = np.sqrt(u0)
                b0, b1 = F(1) - su0, u1 * su0
                b2 = (F(1) - b0) - b1
                st = b0 * uvs[0] + b1 * uvs[1] + b2 * uvs[2]
                sx, ty = st[0] * F(cols) - F(0.5), st[1] * F(rows) - F(0.5)
                s0, t0 = int(np.floor(sx)), int(np.floor(ty))
                for ds in (0, 1):
                    for dt in (0, 1):
                        si, ti = s0 + ds, t0 + dt
                        if wrap == abi.WRAP_REPEAT:
                            si, ti = si % cols, ti % rows
                        elif wrap == abi.WRAP_CLAMP:
                            si, ti = min(max(si, 0), cols - 1), min(max(ti, 0), rows - 1)
                        elif not (0 <= si < cols and 0 <= ti < rows):
                            continue
                        if np.any(level0[ti, si] != 0):
                            return True
        return False

    def light(self, light, xf):
        """populate_scene, KHR_lights_punctual part (gltf.rs:460-487)."""
        c = F(light.get("intensity", 1.0)) * F(light.get("color", [1, 1, 1])[0])
        col = [c, c, c]
        if light["type"] == "directional":
            w = _xf_vectors(xf, np.array([[0, 0, -1]], dtype=np.float32))[0]
            n_before = len(self.scene.lights)
            self.scene.add_directional_light(w, col)  # normalises
            self.deferred_lights.append(self.scene.lights.pop(n_before))
        else:  # point, and spot treated as point
            p = _xf_points(xf, np.zeros((1, 3), dtype=np.float32))[0]
            self.scene.add_point_light(p, col)

    def populate(self, parent, node_index):
        node = self.doc["nodes"][node_index]
        xf = _matmul4(parent, node_transform(node))
        if "mesh" in node:
            for prim in self.doc["meshes"][node["mesh"]]["primitives"]:
                self.primitive(prim, xf)
        lref = node.get("extensions", {}).get("KHR_lights_punctual", {}).get("light")
        if lref is not None:
            self.light(self.doc["extensions"]["KHR_lights_punctual"]["lights"][lref], xf)
        for child in node.get("children", []):
            self.populate(xf, child)

    # ---- camera ------------------------------------------------------------------------------------
    def find_camera(self, parent, node_index, resolution):
        """find_camera (common/importer/gltf.rs:3-46) incl. its first-child-only descent."""
        node = self.doc["nodes"][node_index]
        xf = _matmul4(parent, node_transform(node))
        if "camera" in node:
            cam = self.doc["cameras"][node["camera"]]
            if cam.get("type") == "perspective":
                p = cam["perspective"]
                rot = xf[:3, :3].copy()
                for c in range(3):  # try_convert to an isometry: unit columns expected; normalise against rounding
                    col = rot[:, c]
                    rot[:, c] = col / np.sqrt((col[0] * col[0] + col[1] * col[1]) + col[2] * col[2])
                res = (F(resolution[0]), F(resolution[1]))
                return Camera(_quat_from_rotation_matrix(rot), xf[:3, 3], res[0] / res[1], F(p["yfov"]), F(p["znear"]),
                              F(p.get("zfar", DEFAULT_Z_FAR)), resolution)
        for child in node.get("children", []):
            return self.find_camera(xf, child, resolution)
        return None


def default_camera(world_bound, resolution):
    """get_default_camera (common/importer/gltf.rs:68-85): eye at the bound's max corner looking at the origin,
    yfov = pi/2 * (height / width)."""
    lo, hi = world_bound
    eye = [float(v) for v in hi]  # binary64 scalars, explicit order (the C++ host does the same)
    en = math.sqrt(eye[0] * eye[0] + eye[1] * eye[1] + eye[2] * eye[2])
    f = [-eye[0] / en, -eye[1] / en, -eye[2] / en]
    up = [0.0, 1.0, 0.0]
    sv = [f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]]
    sn = math.sqrt(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2])
    sv = [x / sn for x in sv]
    u = [sv[1] * f[2] - sv[2] * f[1], sv[2] * f[0] - sv[0] * f[2], sv[0] * f[1] - sv[1] * f[0]]
    rot = np.array([[sv[k], u[k], -f[k]] for k in range(3)], dtype=np.float32)
    eye = np.array(eye, dtype=np.float64)
    res = (F(resolution[0]), F(resolution[1]))
    return Camera(_quat_from_rotation_matrix(rot), eye.astype(np.float32), res[0] / res[1], F(math.pi / 2) * (res[1] / res[0]),
                  DEFAULT_Z_NEAR, DEFAULT_Z_FAR, resolution)


def default_env_light_to_world():
    """UnitQuaternion::from_euler_angles(-pi/2, 0, 0): the env map is z-up, the scene y-up (gltf.rs:553-562)."""
    c, s = F(math.cos(-math.pi / 2)), F(math.sin(-math.pi / 2))
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def import_gltf(path, resolution, default_lights=False, env_map=None):
    """from_gltf (common/importer/gltf.rs:98-117) -> (Camera, RenderScene).

    default_lights: add the environment light of `--default_lights`.  The reference reads
    data/abandoned_tank_farm_04_1k.hdr from its source tree; here `env_map` is either an (rows, cols, 3)
    float32 array or the path of a Radiance .hdr file (required when default_lights is set)."""
    imp = _Importer(path)
    doc, s = imp.doc, imp.scene
    imp.materials.append(s.add_material(abi.MAT_MATTE, [s.const_rgb([1.0, 1.0, 1.0])]))  # default_material
    for m in doc.get("materials", []):
        imp.materials.append(imp.material(m))
    ident = np.eye(4, dtype=np.float32)
    for sc in doc.get("scenes", []):
        for n in sc.get("nodes", []):
            imp.populate(ident, n)
    if not s.meshes:
        raise GltfError("glTF file without triangle meshes")
    s.lights.extend(imp.deferred_lights)
    if default_lights:
        if env_map is None:
            raise GltfError("default_lights needs env_map (array or .hdr path): the reference's bundled HDR is not shipped")
        img = tx.read_rgbe(env_map) if isinstance(env_map, (str, os.PathLike)) else np.asarray(env_map, dtype=np.float32)
        tx.add_infinite_light(s, img, light_to_world=default_env_light_to_world())
    camera = None
    for sc in doc.get("scenes", []):
        for n in sc.get("nodes", []):
            camera = imp.find_camera(ident, n, resolution)
            if camera is not None:
                break
        if camera is not None:
            break
    if camera is None:
        camera = default_camera(s.world_bound(), resolution)
    return camera, s
