"""Host-side mirror of the reference's scene / camera construction for the render() hot path.

Mirrors (reference file:line, relative to /root/reference):
  * common/importer/mod.rs:6-25            import(path, resolution)           -> import_scene
  * common/importer/mitsuba.rs:20-79       gen_rectangle / gen_cube (genmesh 0.6.2 Plane/Cube)
  * common/importer/mitsuba.rs:685-710     get_camera
  * common/mod.rs:33-62                    Camera::new
  * pathtracer/importer/mitsuba.rs:84-428  material_from_bsdf / parse_shape / RenderScene::from_mitsuba
  * common/film.rs:132-185                 Film (resolution, sample bounds)
Only the Mitsuba subset needed by data/cornell-box.xml and simple test scenes is parsed
(rectangle, cube, twosided/diffuse/conductor/roughconductor/dielectric/plastic bsdfs with rgb
parameters, area emitters, perspective sensor).  Everything here is host-side set-up that the
reference also does once, outside PathIntegrator::render; the results cross the C ABI as flat
arrays (include/ptrs.h).  All arithmetic is binary32 in the order nalgebra 0.32.2 performs it.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import abi

F = np.float32


def _mat4(values):
    return np.array(values, dtype=np.float32).reshape(4, 4)


def transform_point(m, p):
    """nalgebra Transform * Point for a matrix whose last row is (0,0,0,1)."""
    x, y, z = F(p[0]), F(p[1]), F(p[2])
    return np.array([((m[r, 0] * x + m[r, 1] * y) + m[r, 2] * z) + m[r, 3] for r in range(3)], dtype=np.float32)


def transform_vector(m, v):
    x, y, z = F(v[0]), F(v[1]), F(v[2])
    return np.array([(m[r, 0] * x + m[r, 1] * y) + m[r, 2] * z for r in range(3)], dtype=np.float32)


# ---- genmesh 0.6.2 generators (restated from memory: UNVERIFIED) -------------------
def gen_rectangle():
    """Plane::new(): 4 shared vertices (+-1,+-1,0), normal +z, one quad (0,1,3,2) triangulated as
    (0,1,3),(0,3,2)  (common/importer/mitsuba.rs:20-38)."""
    pos = np.array([[-1, -1, 0], [1, -1, 0], [-1, 1, 0], [1, 1, 0]], dtype=np.float32)
    normal = np.tile(np.array([0, 0, 1], dtype=np.float32), (4, 1))
    indices = np.array([[0, 1, 3], [0, 3, 2]], dtype=np.uint32)
    return pos, normal, indices


_CUBE_FACES = [
    ((1, 0, 0), (0b110, 0b111, 0b101, 0b100)),
    ((-1, 0, 0), (0b000, 0b001, 0b011, 0b010)),
    ((0, 1, 0), (0b011, 0b111, 0b110, 0b010)),
    ((0, -1, 0), (0b100, 0b101, 0b001, 0b000)),
    ((0, 0, 1), (0b101, 0b111, 0b011, 0b001)),
    ((0, 0, -1), (0b000, 0b010, 0b110, 0b100)),
]


def gen_cube():
    """Cube::new(): 6 faces x 4 vertices on +-1 with per-face normals, quads (4f,4f+1,4f+2,4f+3)
    triangulated as (x,y,z),(x,z,w)  (common/importer/mitsuba.rs:40-58)."""
    pos, normal, indices = [], [], []
    for f, (n, quad) in enumerate(_CUBE_FACES):
        for vid in quad:
            pos.append([1.0 if vid & 4 else -1.0, 1.0 if vid & 2 else -1.0, 1.0 if vid & 1 else -1.0])
            normal.append(n)
        b = 4 * f
        indices += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    return np.array(pos, dtype=np.float32), np.array(normal, dtype=np.float32), np.array(indices, dtype=np.uint32)


# ---- camera ---------------------------------------------------------------------------------------
def _quat_from_rotation_matrix(r):
    """nalgebra UnitQuaternion::from_rotation_matrix; returns (i, j, k, w)."""
    tr = (r[0, 0] + r[1, 1]) + r[2, 2]
    q = F(0.25)
    if tr > 0:
        denom = np.sqrt(tr + F(1)) * F(2)
        w, i, j, k = q * denom, (r[2, 1] - r[1, 2]) / denom, (r[0, 2] - r[2, 0]) / denom, (r[1, 0] - r[0, 1]) / denom
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        denom = np.sqrt(((F(1) + r[0, 0]) - r[1, 1]) - r[2, 2]) * F(2)
        w, i, j, k = (r[2, 1] - r[1, 2]) / denom, q * denom, (r[0, 1] + r[1, 0]) / denom, (r[0, 2] + r[2, 0]) / denom
    elif r[1, 1] > r[2, 2]:
        denom = np.sqrt(((F(1) + r[1, 1]) - r[0, 0]) - r[2, 2]) * F(2)
        w, i, j, k = (r[0, 2] - r[2, 0]) / denom, (r[0, 1] + r[1, 0]) / denom, q * denom, (r[1, 2] + r[2, 1]) / denom
    else:
        denom = np.sqrt(((F(1) + r[2, 2]) - r[0, 0]) - r[1, 1]) * F(2)
        w, i, j, k = (r[1, 0] - r[0, 1]) / denom, (r[0, 2] + r[2, 0]) / denom, (r[1, 2] + r[2, 1]) / denom, q * denom
    return np.array([i, j, k, w], dtype=np.float32)


def _rotation_axis_angle(axisangle):
    """nalgebra Rotation3::new(axisangle) = from_axis_angle(normalize(axisangle), |axisangle|)."""
    a = np.array(axisangle, dtype=np.float32)
    angle = np.sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2])
    if angle == 0:
        return np.eye(3, dtype=np.float32)
    ux, uy, uz = a / angle
    s, c = F(math.sin(float(angle))), F(math.cos(float(angle)))
    omc = F(1) - c
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    return np.array([
        [sqx + (F(1) - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s],
        [ux * uy * omc + uz * s, sqy + (F(1) - sqy) * c, uy * uz * omc - ux * s],
        [ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (F(1) - sqz) * c]], dtype=np.float32)


class Film:
    """common/film.rs:132-185 -- resolution, Gaussian(alpha=2, r=2) filter, accumulators."""

    FILTER_RADIUS = 2.0

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.pixels = np.zeros((self.height, self.width), dtype=abi.FILM_DTYPE)

    def clear(self):  # film.rs:164-172
        self.pixels[...] = 0

    def get_sample_bounds(self):  # film.rs:174-185 -> (min_x, min_y, max_x, max_y)
        r = self.FILTER_RADIUS
        return (math.floor(0.5 - r), math.floor(0.5 - r), math.ceil(self.width - 0.5 + r), math.ceil(self.height - 0.5 + r))

    def to_rgb(self):  # film.rs:253-271 (to_channel_updates): rgb / weight
        w = self.pixels["weight"][..., None]
        with np.errstate(divide="ignore", invalid="ignore"):
            return self.pixels["rgb"] * (np.float32(1.0) / w)


class Camera:
    """common/mod.rs:20-62.  cam_to_world is an Isometry3 (unit quaternion i,j,k,w + translation)."""

    def __init__(self, rot_quat, trans, aspect, fovy, znear, zfar, resolution):
        w, h = F(resolution[0]), F(resolution[1])
        self.rot = np.array(rot_quat, dtype=np.float32)
        self.trans = np.array(trans, dtype=np.float32)
        # Perspective3::new(aspect, fovy, znear, zfar)
        aspect, fovy, znear, zfar = F(aspect), F(fovy), F(znear), F(zfar)
        self.m11 = F(1) / F(math.tan(float(fovy / F(2))))
        self.m00 = self.m11 / aspect
        self.m22 = (zfar + znear) / (znear - zfar)
        self.m23 = zfar * znear * F(2) / (znear - zfar)
        # screen_to_raster = S(W,H,1) * S(1/2,-1/2,1) * T(1,-1,0); raster_to_screen = inverse
        sx, sy = w * F(0.5), h * F(-0.5)
        self.screen_to_raster = _mat4([sx, 0, 0, sx * F(1), 0, sy, 0, sy * F(-1), 0, 0, 1, 0, 0, 0, 0, 1])
        ax, by = F(1) / sx, F(1) / sy
        self.raster_to_screen = _mat4([ax, 0, 0, F(-1), 0, by, 0, F(1), 0, 0, 1, 0, 0, 0, 0, 1])
        # raster_to_camera = cam_to_screen.to_projective().inverse() * raster_to_screen, applied to
        # raster (1,0,0), (0,1,0) and the origin with the homogeneous divide (common/mod.rs:44-48)
        n = self.m22 / self.m23
        inv00, inv11 = F(1) / self.m00, F(1) / self.m11
        p0 = np.array([(inv00 * F(-1)) / n, (inv11 * F(1)) / n, F(-1) / n], dtype=np.float32)
        px = np.array([(inv00 * ax + inv00 * F(-1)) / n, (inv11 * F(1)) / n, F(-1) / n], dtype=np.float32)
        py = np.array([(inv00 * F(-1)) / n, (inv11 * by + inv11 * F(1)) / n, F(-1) / n], dtype=np.float32)
        self.dx_camera = px - p0
        self.dy_camera = py - p0
        self.film = Film(int(resolution[0]), int(resolution[1]))

    def to_abi(self):
        c = abi.PtrsCamera()
        c.rot[:] = [float(x) for x in self.rot]
        c.trans[:] = [float(x) for x in self.trans]
        c.m00, c.m11, c.m22, c.m23 = float(self.m00), float(self.m11), float(self.m22), float(self.m23)
        c.raster_to_screen[:] = [float(x) for x in self.raster_to_screen.reshape(16)]
        c.dx_camera[:] = [float(x) for x in self.dx_camera]
        c.dy_camera[:] = [float(x) for x in self.dy_camera]
        return c


def camera_from_matrix(cam_to_world4, fov_deg, film_w, film_h, resolution):
    """get_camera, common/importer/mitsuba.rs:685-710 (Q30)."""
    fov = F(fov_deg) * F(math.pi / 180.0)  # f32::to_radians
    rot_y = _rotation_axis_angle([0.0, -math.pi, 0.0])
    m = _mat4(cam_to_world4)
    r4 = np.eye(4, dtype=np.float32)
    r4[:3, :3] = rot_y
    mm = np.zeros((4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            acc = F(0)
            for k in range(4):
                acc = acc + m[i, k] * r4[k, j]
            mm[i, j] = acc
    # try_convert::<Projective3, Similarity3>: normalise the columns, mean scale forced to 1
    rot = mm[:3, :3].copy()
    for col in range(3):
        c = rot[:, col]
        nrm = np.sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2])
        rot[:, col] = c / nrm
    q = _quat_from_rotation_matrix(rot)
    res = (F(resolution[0]), F(resolution[1]))
    return Camera(q, mm[:3, 3], res[0] / res[1], fov * (F(film_h) / F(film_w)), 0.01, 10000.0, resolution)


def look_at_camera(eye, target, up, fovy_deg, resolution, znear=0.01, zfar=10000.0):
    """Convenience for synthetic scenes (not a reference path): right-handed look-at, -z forward."""
    eye, target, up = (np.array(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    rot = np.stack([s, u, -f], axis=1).astype(np.float32)
    res = (F(resolution[0]), F(resolution[1]))
    return Camera(_quat_from_rotation_matrix(rot), eye.astype(np.float32), res[0] / res[1], F(fovy_deg) * F(math.pi / 180.0), znear, zfar, resolution)


# ---- RenderScene ----------------------------------------------------------------------------------
class RenderScene:
    """pathtracer/mod.rs:84-107: meshes, materials, textures, lights (+ derived flat description)."""

    def __init__(self):
        self.meshes, self.materials, self.textures, self.lights = [], [], [], []
        self._holder = None

    # builders -------------------------------------------------------------------------------------
    def add_texture(self, **kw):
        self.textures.append(kw)
        return len(self.textures) - 1

    def const_rgb(self, rgb):
        return self.add_texture(kind=abi.TEX_CONSTANT, channels=3, value=np.array(rgb, dtype=np.float32))

    def const_f(self, v):
        return self.add_texture(kind=abi.TEX_CONSTANT, channels=1, value=float(v))

    def add_material(self, kind, tex=(), flags=0, inner=-1):
        self.materials.append(dict(kind=kind, tex=list(tex), flags=flags, inner=inner))
        return len(self.materials) - 1

    def add_mesh(self, pos, indices, material, normal=None, uv=None, tangent=None, emission_rgb=None, alpha_mask_tex=-1):
        """One TriangleMesh; `emission_rgb` creates one DiffuseAreaLight per triangle in triangle
        order (pathtracer/importer/mitsuba.rs:306-331)."""
        self.meshes.append(dict(pos=pos, indices=indices, material=material, normal=normal, uv=uv, tangent=tangent, alpha_mask_tex=alpha_mask_tex))
        mi = len(self.meshes) - 1
        if emission_rgb is not None:
            ke = self.const_rgb(emission_rgb)
            for t in range(len(indices)):
                self.lights.append(dict(kind=abi.LIGHT_AREA, mesh=mi, tri=t, ke_tex=ke))
        return mi

    def add_point_light(self, p, intensity):
        self.lights.append(dict(kind=abi.LIGHT_POINT, v=p, c=intensity))

    def add_directional_light(self, w_light, radiance):
        w = np.array(w_light, dtype=np.float32)
        w = w / np.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])
        self.lights.append(dict(kind=abi.LIGHT_DIRECTIONAL, v=w, c=radiance))

    def world_bound(self):
        lo = np.min([np.asarray(m["pos"], dtype=np.float32).reshape(-1, 3).min(axis=0) for m in self.meshes], axis=0)
        hi = np.max([np.asarray(m["pos"], dtype=np.float32).reshape(-1, 3).max(axis=0) for m in self.meshes], axis=0)
        return lo, hi

    def preprocess_lights(self):
        """Light::preprocess (light.rs:209-211,480-482) with Bounds3::bounding_sphere (bounds.rs:126-134)."""
        lo, hi = self.world_bound()
        center = (lo + hi) * F(0.5)
        d = center - hi
        radius = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        for l in self.lights:
            if l["kind"] in (abi.LIGHT_DIRECTIONAL, abi.LIGHT_INFINITE):
                l["world_center"], l["world_radius"] = center, float(radius)

    def num_triangles(self):
        return sum(len(m["indices"]) for m in self.meshes)

    def desc(self, bvh=None):
        self.preprocess_lights()
        self._holder = abi.SceneDescHolder(self.meshes, self.materials, self.textures, self.lights, bvh)
        return self._holder.desc


def _parse_matrix(el):
    return _mat4([float(v) for v in el.get("value").split()])


def _bsdf_to_material(scene, el):
    """material_from_bsdf, pathtracer/importer/mitsuba.rs:84-181 (rgb-parameter subset)."""
    kind = el.get("type")
    rgbs = {c.get("name"): [float(v) for v in c.get("value").replace(",", " ").split()] for c in el.findall("rgb")}
    floats = {c.get("name"): float(c.get("value")) for c in el.findall("float")}
    floats = {"".join("_" + ch.lower() if ch.isupper() else ch for ch in k): v for k, v in floats.items()}  # heck SnakeCase
    if kind == "twosided":
        return _bsdf_to_material(scene, el.find("bsdf"))
    if kind == "diffuse":
        return scene.add_material(abi.MAT_MATTE, [scene.const_rgb(rgbs.get("reflectance", [1, 1, 1]))])
    if kind in ("conductor", "roughconductor"):
        mat = el.find("string[@name='material']")
        if kind == "conductor" and mat is not None:
            if mat.get("value") == "none":
                return scene.add_material(abi.MAT_MIRROR)
            raise ValueError("other material values not supported yet!")
        r = rgbs.get("specularReflectance", rgbs.get("specular_reflectance", [1, 1, 1]))
        alpha = 0.001 if kind == "conductor" else floats["alpha"]
        return scene.add_material(abi.MAT_METAL, [scene.const_rgb(rgbs["eta"]), scene.const_rgb(rgbs["k"]), scene.const_rgb(r), scene.const_f(alpha), -1, -1], flags=0)
    if kind == "dielectric":
        return scene.add_material(abi.MAT_GLASS, [scene.const_rgb([1, 1, 1]), scene.const_rgb([1, 1, 1]), scene.const_f(floats["int_ior"])])
    if kind in ("plastic", "roughplastic"):
        e = F(floats["int_ior"])
        r0 = ((e - F(1)) * (e - F(1))) / ((e + F(1)) * (e + F(1)))
        a = 0.001 if kind == "plastic" else floats["alpha"]
        kd = rgbs.get("diffuseReflectance", rgbs.get("diffuse_reflectance", [1, 1, 1]))
        return scene.add_material(abi.MAT_SUBSTRATE, [scene.const_rgb(kd), scene.const_rgb([r0] * 3), scene.const_f(a), scene.const_f(a)], flags=0)
    raise ValueError("unsupported bsdf type " + str(kind))


def import_scene(path, resolution, default_lights=False, env_map=None):
    """common/importer/mod.rs:6-25: dispatch on the extension; returns (Camera, RenderScene).
    .gltf / .glb -> gltf.import_gltf (default_lights = the CLI's --default_lights); .xml -> the Mitsuba subset."""
    ext = os.path.splitext(str(path))[1].lower()
    if ext in (".gltf", ".glb"):
        from .gltf import import_gltf
        return import_gltf(path, resolution, default_lights=default_lights, env_map=env_map)
    if ext != ".xml":
        raise ValueError("unsupported format!")
    root = ET.parse(path).getroot()
    sensor = root.find("sensor")
    fov = float(sensor.find("float[@name='fov']").get("value"))
    film = sensor.find("film")
    fw = int(film.find("integer[@name='width']").get("value"))
    fh = int(film.find("integer[@name='height']").get("value"))
    cam = camera_from_matrix(_parse_matrix(sensor.find("transform/matrix")).reshape(16), fov, fw, fh, resolution)
    scene = RenderScene()
    named = {}
    for b in root.findall("bsdf"):
        named[b.get("id")] = _bsdf_to_material(scene, b)
    for sh in root.findall("shape"):
        kind = sh.get("type")
        if kind == "rectangle":
            pos, normal, indices = gen_rectangle()
        elif kind == "cube":
            pos, normal, indices = gen_cube()
        else:
            raise ValueError("unsupported shape type " + kind)
        m = _parse_matrix(sh.find("transform/matrix")) if sh.find("transform/matrix") is not None else np.eye(4, dtype=np.float32)
        wpos = np.array([transform_point(m, p) for p in pos], dtype=np.float32)
        wnrm = np.array([transform_vector(m, n) for n in normal], dtype=np.float32)  # Q15: forward matrix, no renormalise
        ref = sh.find("ref")
        mat = named[ref.get("id")] if ref is not None else _bsdf_to_material(scene, sh.find("bsdf"))
        em = sh.find("emitter")
        emission = None
        if em is not None and em.get("type") == "area":
            emission = [float(v) for v in em.find("rgb").get("value").replace(",", " ").split()]
        scene.add_mesh(wpos, indices, mat, normal=wnrm, emission_rgb=emission)
    return cam, scene
