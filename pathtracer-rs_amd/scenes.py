"""Procedural scenes (seeded, no assets): parity-test scenes and stand-ins for the glTF configs of
BASELINE.json whose assets (Sponza, Classroom) are not available offline.  Geometry only uses the
records of include/ptrs.h, so the same description feeds the HIP library and the oracle."""
import math
import os

import numpy as np

from . import abi
from .scene import RenderScene, gen_cube, gen_rectangle, look_at_camera, transform_point, transform_vector


def _trs(scale, translate, rot_y_deg=0.0):
    c, s = math.cos(math.radians(rot_y_deg)), math.sin(math.radians(rot_y_deg))
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32) * np.array(scale, dtype=np.float32)[None, :]
    m[:3, 3] = translate
    return m


def _add(scene, gen, m, material, uv=None, emission=None):
    pos, normal, idx = gen
    wpos = np.array([transform_point(m, p) for p in pos], dtype=np.float32)
    wn = np.array([transform_vector(m, n) for n in normal], dtype=np.float32)
    return scene.add_mesh(wpos, idx, material, normal=wn, uv=uv, emission_rgb=emission)


def uv_sphere(n_theta=12, n_phi=24):
    """Latitude/longitude sphere with vertex normals and uvs (radius 1, centre 0)."""
    pos, uv = [], []
    for i in range(n_theta + 1):
        th = math.pi * i / n_theta
        for j in range(n_phi + 1):
            ph = 2.0 * math.pi * j / n_phi
            pos.append([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
            uv.append([j / n_phi, i / n_theta])
    idx = []
    for i in range(n_theta):
        for j in range(n_phi):
            a, b = i * (n_phi + 1) + j, (i + 1) * (n_phi + 1) + j
            if i > 0:
                idx.append([a, a + 1, b])
            if i < n_theta - 1:
                idx.append([a + 1, b + 1, b])
    pos = np.array(pos, dtype=np.float32)
    return pos, pos.copy(), np.array(idx, dtype=np.uint32), np.array(uv, dtype=np.float32)


def material_zoo(resolution=(96, 64), with_delta_lights=True):
    """A closed room with one object per material kind, checker textures, an area light, a point
    light and a directional light.  Exercises every BxDF / light code path of the hot path except
    image textures and the environment light."""
    s = RenderScene()
    white = s.add_material(abi.MAT_MATTE, [s.const_rgb([0.7, 0.7, 0.7])])
    checker = s.add_texture(kind=abi.TEX_CHECKER, channels=3, value=np.array([0.8, 0.2, 0.2], np.float32), value2=np.array([0.2, 0.2, 0.8], np.float32), su=4.0, sv=4.0, du=0.1, dv=0.2)
    floor = s.add_material(abi.MAT_MATTE, [checker])
    mirror = s.add_material(abi.MAT_MIRROR)
    glass = s.add_material(abi.MAT_GLASS, [s.const_rgb([1, 1, 1]), s.const_rgb([0.9, 0.95, 1.0]), s.const_f(1.5)])
    null_glass = s.add_material(abi.MAT_GLASS, [s.const_rgb([0, 0, 0]), s.const_rgb([0, 0, 0]), s.const_f(1.3)])
    metal = s.add_material(abi.MAT_METAL, [s.const_rgb([0.2, 0.92, 1.1]), s.const_rgb([3.9, 2.45, 2.14]), s.const_rgb([1, 1, 1]), s.const_f(0.15), -1, -1], flags=0)
    metal_aniso = s.add_material(abi.MAT_METAL, [s.const_rgb([0.14, 0.37, 1.44]), s.const_rgb([3.98, 2.38, 1.6]), s.const_rgb([0.9, 0.9, 0.9]), -1, s.const_f(0.3), s.const_f(0.05)], flags=1)
    rough_tex = s.add_texture(kind=abi.TEX_CHECKER, channels=1, value=0.2, value2=0.6, su=3.0, sv=3.0)
    disney = s.add_material(abi.MAT_DISNEY, [s.const_rgb([0.8, 0.5, 0.2]), s.const_f(0.3), s.const_f(1.5), rough_tex])
    disney_metal = s.add_material(abi.MAT_DISNEY, [checker, s.const_f(1.0), s.const_f(1.5), s.const_f(0.25)])
    substrate = s.add_material(abi.MAT_SUBSTRATE, [s.const_rgb([0.1, 0.5, 0.2]), s.const_rgb([0.04, 0.04, 0.04]), s.const_f(0.1), s.const_f(0.2)], flags=0)
    rect, cube = gen_rectangle(), gen_cube()
    uv_rect = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=np.float32)
    rx = lambda deg: np.array([[1, 0, 0, 0], [0, math.cos(math.radians(deg)), -math.sin(math.radians(deg)), 0], [0, math.sin(math.radians(deg)), math.cos(math.radians(deg)), 0], [0, 0, 0, 1]], dtype=np.float32)
    rz = lambda deg: np.array([[math.cos(math.radians(deg)), -math.sin(math.radians(deg)), 0, 0], [math.sin(math.radians(deg)), math.cos(math.radians(deg)), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    T = lambda x, y, z: _trs([1, 1, 1], [x, y, z])
    S = lambda x, y, z: _trs([x, y, z], [0, 0, 0])
    _add(s, rect, T(0, 0, 0) @ rx(-90) @ S(3, 3, 1), floor, uv=uv_rect)          # floor  (normal +y)
    _add(s, rect, T(0, 3, 0) @ rx(90) @ S(3, 3, 1), white)                        # ceiling
    _add(s, rect, T(0, 1.5, -3) @ S(3, 1.5, 1), white)                            # back wall (normal +z)
    _add(s, rect, T(-3, 1.5, 0) @ (rz(0) @ np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]], np.float32)) @ S(3, 1.5, 1), white)
    _add(s, rect, T(3, 1.5, 0) @ np.array([[0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float32) @ S(3, 1.5, 1), white)
    _add(s, rect, T(0, 2.98, 0) @ rx(90) @ S(0.6, 0.6, 1), white, emission=[12.0, 11.0, 9.0])
    sp = uv_sphere()
    def add_sphere(center, r, mat):
        pos, nrm, idx, uv = sp
        s.add_mesh((pos * np.float32(r) + np.array(center, np.float32)).astype(np.float32), idx, mat, normal=nrm, uv=uv)
    add_sphere([-1.9, 0.5, -1.2], 0.5, mirror)
    add_sphere([-0.7, 0.5, -1.4], 0.5, glass)
    add_sphere([0.5, 0.5, -1.2], 0.5, metal)
    add_sphere([1.8, 0.5, -1.5], 0.5, disney)
    _add(s, cube, _trs([0.35, 0.35, 0.35], [-1.5, 0.35, 0.6], 25), metal_aniso)
    _add(s, cube, _trs([0.35, 0.35, 0.35], [-0.3, 0.35, 0.8], -15), disney_metal)
    _add(s, cube, _trs([0.35, 0.35, 0.35], [0.9, 0.35, 0.6], 40), substrate)
    _add(s, rect, T(1.9, 0.8, 0.4) @ S(0.5, 0.8, 1), null_glass)                   # Q7/Q17: surface without BSDF
    if with_delta_lights:
        s.add_point_light([2.0, 2.2, 1.5], [6.0, 6.0, 5.0])
        s.add_directional_light([0.3, 1.0, 0.5], [0.4, 0.4, 0.5])
    cam = look_at_camera([0.0, 1.6, 5.5], [0.0, 1.0, 0.0], [0, 1, 0], 38.0, resolution)
    return cam, s


def triangle_soup(n_tris=20000, seed=1, resolution=(64, 64), extent=4.0, size=0.25):
    """Seeded random triangles (matte, one emissive quad): a BVH stress scene for traversal parity
    and for roofline measurements on a tree that does not fit in L1/LDS."""
    rng = np.random.default_rng(seed)
    s = RenderScene()
    mats = [s.add_material(abi.MAT_MATTE, [s.const_rgb(rng.uniform(0.2, 0.9, 3))]) for _ in range(8)]
    c = rng.uniform(-extent, extent, (n_tris, 1, 3)).astype(np.float32)
    pos = (c + rng.normal(0, size, (n_tris, 3, 3)).astype(np.float32)).reshape(-1, 3).astype(np.float32)
    idx = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    per = n_tris // len(mats)
    for k, m in enumerate(mats):
        a, b = k * per, (n_tris if k == len(mats) - 1 else (k + 1) * per)
        sub = idx[a:b] - 3 * a
        s.add_mesh(pos[3 * a:3 * b], sub, m)
    rect = gen_rectangle()
    m = _trs([1.5, 1.5, 1.0], [0, 0, 0])
    up = np.array([[1, 0, 0, 0], [0, 0, -1, extent + 1.0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)  # quad facing down
    _add(s, rect, up @ m, mats[0], emission=[20.0, 20.0, 20.0])
    cam = look_at_camera([0.0, 0.5, 3.0 * extent], [0, 0, 0], [0, 1, 0], 40.0, resolution)
    return cam, s


def synthetic_env_map(rows=32, cols=64, seed=3):
    """Seeded HDR environment (sky gradient, ground, a small very bright sun): stands in for
    data/abandoned_tank_farm_04_1k.hdr where the asset must not be read."""
    rng = np.random.default_rng(seed)
    v = (np.arange(rows, dtype=np.float32) + 0.5) / rows
    u = (np.arange(cols, dtype=np.float32) + 0.5) / cols
    sky = np.stack([0.3 + 0.5 * (1 - v), 0.5 + 0.4 * (1 - v), 0.9 + 0.0 * v], axis=-1)[:, None, :] * np.ones((1, cols, 1), np.float32)
    ground = np.array([0.25, 0.2, 0.15], np.float32)
    img = np.where((v < 0.55)[:, None, None], sky, ground).astype(np.float32)
    img *= (1.0 + 0.2 * rng.uniform(-1, 1, (rows, cols, 1))).astype(np.float32)
    su, sv = int(0.3 * cols), int(0.25 * rows)
    img[sv:sv + 2, su:su + 2] = np.array([900.0, 800.0, 600.0], np.float32)
    return img.astype(np.float32)


def textured_env(resolution=(96, 64), env=None):
    """Open scene lit by an environment map, with image textures (non-power-of-two: exercises the
    Lanczos resample), a float roughness texture, a normal-mapped material and glass."""
    from . import textures as tx
    rng = np.random.default_rng(21)
    s = RenderScene()
    kd_img = (rng.uniform(40, 230, (12, 20, 3))).astype(np.uint8)
    kd_tex = tx.spectrum_texture(s, kd_img, uvmap=(3.0, 3.0, 0.0, 0.0))
    floor = s.add_material(abi.MAT_MATTE, [kd_tex])
    rough_tex = tx.float_texture(s, rng.integers(20, 200, (16, 16)).astype(np.uint8), scale=0.8, uvmap=(2.0, 2.0, 0.0, 0.0))
    disney = s.add_material(abi.MAT_DISNEY, [s.const_rgb([0.9, 0.6, 0.3]), s.const_f(0.6), s.const_f(1.5), rough_tex])
    nm = np.zeros((16, 16, 3), np.uint8)
    ang = rng.uniform(0, 2 * math.pi, (16, 16))
    tilt = rng.uniform(0.0, 0.5, (16, 16))
    nm[..., 0] = np.clip(127.5 * (1 + tilt * np.cos(ang)), 0, 255)
    nm[..., 1] = np.clip(127.5 * (1 + tilt * np.sin(ang)), 0, 255)
    nm[..., 2] = np.clip(127.5 * (1 + np.sqrt(1 - tilt ** 2)), 0, 255)
    nm_tex = tx.normal_map_texture(s, nm, uvmap=(4.0, 2.0, 0.0, 0.0))
    bumpy_inner = s.add_material(abi.MAT_MATTE, [s.const_rgb([0.7, 0.7, 0.75])])
    bumpy = s.add_material(abi.MAT_NORMAL, [nm_tex], inner=bumpy_inner)
    glass = s.add_material(abi.MAT_GLASS, [s.const_rgb([1, 1, 1]), s.const_rgb([1, 1, 1]), s.const_f(1.5)])
    rect = gen_rectangle()
    uv_rect = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=np.float32)
    rx = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], np.float32)  # +z -> +y
    _add(s, rect, rx @ _trs([4, 4, 1], [0, 0, 0]), floor, uv=uv_rect)
    pos, nrm, idx, uv = uv_sphere(10, 20)
    for center, r, mat in ([-1.3, 0.6, 0.0], 0.6, disney), ([0.0, 0.6, 0.3], 0.6, bumpy), ([1.3, 0.6, 0.0], 0.6, glass):
        s.add_mesh((pos * np.float32(r) + np.array(center, np.float32)).astype(np.float32), idx, mat, normal=nrm, uv=uv)
    # alpha-masked cards (shape.rs:227-244,471-521): a checker mask (0/1) on a matte quad and on an emissive quad
    mask = s.add_texture(kind=abi.TEX_CHECKER, channels=1, value=0.0, value2=1.0, su=6.0, sv=4.0, du=0.05, dv=0.1)
    card = s.add_material(abi.MAT_MATTE, [s.const_rgb([0.2, 0.7, 0.3])])
    m1 = _trs([0.9, 0.6, 1.0], [-0.6, 0.9, 1.4], 20)
    pos_c = np.array([transform_point(m1, p) for p in rect[0]], dtype=np.float32)
    nrm_c = np.array([transform_vector(m1, n) for n in rect[1]], dtype=np.float32)
    s.add_mesh(pos_c, rect[2], card, normal=nrm_c, uv=uv_rect, alpha_mask_tex=mask)
    m2 = np.array([[1, 0, 0, 0], [0, 0, -1, 2.6], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32) @ _trs([0.7, 0.7, 1.0], [0.8, 0.0, 0.0])
    pos_l = np.array([transform_point(m2, p) for p in rect[0]], dtype=np.float32)
    nrm_l = np.array([transform_vector(m2, n) for n in rect[1]], dtype=np.float32)
    s.add_mesh(pos_l, rect[2], card, normal=nrm_l, uv=uv_rect, emission_rgb=[6.0, 5.0, 4.0], alpha_mask_tex=mask)
    # Mitsuba import's environment orientation (pathtracer/importer/mitsuba.rs:365-372): Euler(-pi/2,-pi/2,0) * scale(1,1,-1)
    cr, sr = math.cos(-math.pi / 2), math.sin(-math.pi / 2)
    rxm = np.array([[1, 0, 0, 0], [0, cr, -sr, 0], [0, sr, cr, 0], [0, 0, 0, 1]], np.float64)
    rym = np.array([[cr, 0, sr, 0], [0, 1, 0, 0], [-sr, 0, cr, 0], [0, 0, 0, 1]], np.float64)
    l2w = (rym @ rxm @ np.diag([1.0, 1.0, -1.0, 1.0])).astype(np.float32)
    tx.add_infinite_light(s, synthetic_env_map() if env is None else env, light_to_world=l2w)
    cam = look_at_camera([0.0, 1.4, 4.5], [0.0, 0.5, 0.0], [0, 1, 0], 40.0, resolution)
    return cam, s


# ---- Sponza-class stand-in (BASELINE configs[2] / [4]: the glTF asset is not available offline) ------
def _grid(nu, nv):
    """(nu+1)x(nv+1) lattice of uv in [0,1]^2 and its 2*nu*nv triangles."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float32), np.linspace(0, 1, nv + 1, dtype=np.float32), indexing="xy")
    uv = np.stack([u.reshape(-1), v.reshape(-1)], axis=1).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    a = (j * (nu + 1) + i).reshape(-1)
    idx = np.concatenate([np.stack([a, a + 1, a + nu + 2], axis=1), np.stack([a, a + nu + 2, a + nu + 1], axis=1)], axis=0).astype(np.uint32)
    return uv, idx


def _surface(scene, fn, nu, nv, material, uv_scale=(1.0, 1.0)):
    """Adds the parametric surface fn(u, v) -> (pos, normal) tessellated nu x nv."""
    uv, idx = _grid(nu, nv)
    pos, nrm = fn(uv[:, 0].astype(np.float64), uv[:, 1].astype(np.float64))
    scene.add_mesh(pos.astype(np.float32), idx, material, normal=nrm.astype(np.float32), uv=(uv * np.array(uv_scale, np.float32)).astype(np.float32))
    return len(idx)


def colonnade(resolution=(1280, 720), detail=1.0, seed=1, tex_size=1024):
    """Procedural Sponza-class hall: tessellated floor / ceiling / walls, two rows of columns with
    capitals and arches; Disney materials (metallic = 1, roughness 0.2 / 0.5) with a seeded image
    base-colour texture; 1 directional + 4 point lights.  detail = 1 gives about 262k triangles
    (BASELINE configs[2]: 1280x720, 64 spp, depth 15)."""
    from . import textures as tx
    rng = np.random.default_rng(seed)
    s = RenderScene()
    # seeded base-colour image: low-frequency colour noise under a checker
    t = np.linspace(0, 1, tex_size, endpoint=False)
    xx, yy = np.meshgrid(t, t)
    base = np.stack([0.55 + 0.35 * np.sin(2 * np.pi * (3 * xx + rng.uniform())), 0.5 + 0.3 * np.sin(2 * np.pi * (5 * yy + rng.uniform())),
                     0.45 + 0.3 * np.sin(2 * np.pi * (2 * (xx + yy) + rng.uniform()))], axis=-1)
    checker = (((np.floor(xx * 16) + np.floor(yy * 16)) % 2) * 0.25 + 0.75)[..., None]
    img8 = np.clip(255 * base * checker + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)
    col_tex = tx.spectrum_texture(s, img8)
    mats = [s.add_material(abi.MAT_DISNEY, [col_tex, s.const_f(1.0), s.const_f(1.5), s.const_f(r)]) for r in (0.2, 0.5)]
    stone = s.add_material(abi.MAT_DISNEY, [col_tex, s.const_f(0.0), s.const_f(1.5), s.const_f(0.6)])
    LX, LY, LZ = 20.0, 10.0, 7.0
    d = max(float(detail), 0.05) * 1.23  # detail = 1 -> ~262k triangles
    n = lambda k: max(2, int(round(k * math.sqrt(d))))
    tris = 0

    def plane(origin, eu, ev, normal):
        o, eu, ev, nn = (np.array(v, np.float64) for v in (origin, eu, ev, normal))
        return lambda u, v: (o + u[:, None] * eu + v[:, None] * ev, np.tile(nn, (len(u), 1)))
    tris += _surface(s, plane([-LX, 0, -LZ], [2 * LX, 0, 0], [0, 0, 2 * LZ], [0, 1, 0]), n(160), n(64), stone, (8, 3))
    tris += _surface(s, plane([-LX, LY, LZ], [2 * LX, 0, 0], [0, 0, -2 * LZ], [0, -1, 0]), n(128), n(48), mats[1], (8, 3))
    tris += _surface(s, plane([-LX, 0, -LZ], [0, 0, 2 * LZ], [0, LY, 0], [1, 0, 0]), n(64), n(48), stone, (3, 2))
    tris += _surface(s, plane([LX, 0, LZ], [0, 0, -2 * LZ], [0, LY, 0], [-1, 0, 0]), n(64), n(48), stone, (3, 2))
    tris += _surface(s, plane([LX, 0, -LZ], [-2 * LX, 0, 0], [0, LY, 0], [0, 0, 1]), n(128), n(48), stone, (8, 2))
    ncol = 12
    xs = np.linspace(-LX + 2.5, LX - 2.5, ncol)
    for row, z in enumerate((-3.2, 3.2)):
        for ci, x in enumerate(xs):
            r, h = 0.55, 6.0

            def shaft(u, v, x=x, z=z):
                th = 2 * np.pi * u
                rr = r * (1.0 + 0.06 * np.cos(16 * th)) * (1.0 - 0.08 * v)
                p = np.stack([x + rr * np.cos(th), h * v, z + rr * np.sin(th)], axis=1)
                nn = np.stack([np.cos(th), 0.08 * np.ones_like(th), np.sin(th)], axis=1)
                return p, nn / np.linalg.norm(nn, axis=1, keepdims=True)
            tris += _surface(s, shaft, n(44), n(44), mats[(ci + row) % 2], (2, 4))

            def capital(u, v, x=x, z=z):
                th, ph = 2 * np.pi * u, np.pi * v
                nn = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], axis=1)
                return np.array([x, h + 0.35, z]) + nn * np.array([0.85, 0.4, 0.85]), nn
            tris += _surface(s, capital, n(30), n(30), mats[(ci + row + 1) % 2], (2, 1))
        for ci in range(ncol - 1):
            xa, xb = xs[ci], xs[ci + 1]

            def arch(u, v, xa=xa, xb=xb, z=z):
                a, th = np.pi * u, 2 * np.pi * v
                R, rr = 0.5 * (xb - xa), 0.28
                c = np.stack([0.5 * (xa + xb) - R * np.cos(a), 6.75 + 0.9 * R * np.sin(a), np.full_like(a, z)], axis=1)
                radial = np.stack([-np.cos(a), 0.9 * np.sin(a), np.zeros_like(a)], axis=1)
                radial /= np.linalg.norm(radial, axis=1, keepdims=True)
                nn = radial * np.cos(th)[:, None] + np.array([0, 0, 1.0]) * np.sin(th)[:, None]
                return c + rr * nn, nn
            tris += _surface(s, arch, n(30), n(15), stone, (4, 1))
    s.add_directional_light([0.25, 1.0, 0.35], [2.2, 2.0, 1.8])
    for p in ([-12, 7.5, 0], [-4, 7.5, 0], [4, 7.5, 0], [12, 7.5, 0]):
        s.add_point_light(p, [55.0, 50.0, 42.0])
    cam = look_at_camera([-17.0, 3.2, 0.6], [6.0, 3.6, -0.4], [0, 1, 0], 60.0, resolution)
    return cam, s


# ---- Classroom-class stand-in (BASELINE configs[3]: glass + HDR environment light) -------------------
TANK_FARM_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "abandoned_tank_farm_04_1k.hdr")


def synthetic_env_map_1k(seed=5):
    """512x1024 seeded HDR sky (same size class as data/abandoned_tank_farm_04_1k.hdr; used where that file is absent)."""
    rows, cols = 512, 1024
    small = synthetic_env_map(64, 128, seed)
    img = np.repeat(np.repeat(small, rows // 64, axis=0), cols // 128, axis=1)
    v = (np.arange(rows, dtype=np.float32) + 0.5) / rows
    u = (np.arange(cols, dtype=np.float32) + 0.5) / cols
    img = img * (1.0 + 0.15 * np.sin(40 * np.pi * u)[None, :, None] * np.cos(24 * np.pi * v)[:, None, None]).astype(np.float32)
    return np.maximum(img, 0.0).astype(np.float32)


def classroom(resolution=(1920, 1080), detail=1.0, seed=2, tex_size=512, env=None):
    """Procedural classroom-class room: tessellated floor / ceiling / three walls, a window wall made of piers
    with solid glass panes (KHR_materials_transmission = 1, ior 1.5 -> the Glass arm of material/mod.rs:200-256)
    between them, rows of desks with rounded chairs; Disney dielectric (metallic 0) materials with a seeded image
    base colour everywhere else; lit only by the environment map through the windows, oriented like the glTF
    importer's default light (pathtracer/importer/gltf.rs:553-562: Euler(-pi/2, 0, 0)).  detail = 1 gives about
    600k triangles (BASELINE configs[3]: 1920x1080, 128 spp, depth 15)."""
    from . import textures as tx
    rng = np.random.default_rng(seed)
    s = RenderScene()
    t = np.linspace(0, 1, tex_size, endpoint=False)
    xx, yy = np.meshgrid(t, t)
    base = np.stack([0.6 + 0.3 * np.sin(2 * np.pi * (2 * xx + rng.uniform())), 0.55 + 0.3 * np.sin(2 * np.pi * (3 * yy + rng.uniform())),
                     0.5 + 0.3 * np.sin(2 * np.pi * ((xx - yy) + rng.uniform()))], axis=-1)
    planks = (((np.floor(yy * 24)) % 2) * 0.2 + 0.8)[..., None]
    img8 = np.clip(255 * base * planks + rng.normal(0, 5, base.shape), 0, 255).astype(np.uint8)
    col_tex = tx.spectrum_texture(s, img8)
    wall = s.add_material(abi.MAT_DISNEY, [col_tex, s.const_f(0.0), s.const_f(1.5), s.const_f(0.8)])
    wood = s.add_material(abi.MAT_DISNEY, [col_tex, s.const_f(0.0), s.const_f(1.5), s.const_f(0.45)])
    plastic = s.add_material(abi.MAT_DISNEY, [s.const_rgb([0.2, 0.35, 0.7]), s.const_f(0.0), s.const_f(1.5), s.const_f(0.3)])
    glass = s.add_material(abi.MAT_GLASS, [s.const_rgb([1, 1, 1]), s.const_rgb([1, 1, 1]), s.const_f(1.5)])
    LX, LY, LZ = 9.0, 3.6, 6.0
    d = max(float(detail), 0.02) * 1.9  # detail = 1 -> ~600k triangles
    n = lambda k: max(1, int(round(k * math.sqrt(d))))
    tris = 0

    def plane(origin, eu, ev, normal):
        o, eu, ev, nn = (np.array(v, np.float64) for v in (origin, eu, ev, normal))
        return lambda u, v: (o + u[:, None] * eu + v[:, None] * ev, np.tile(nn, (len(u), 1)))

    def box(lo, hi, material, nu, nv, nw, uv_scale=(1.0, 1.0)):
        lo, hi = np.array(lo, np.float64), np.array(hi, np.float64)
        e = hi - lo
        k = 0
        k += _surface(s, plane([lo[0], lo[1], hi[2]], [e[0], 0, 0], [0, e[1], 0], [0, 0, 1]), nu, nv, material, uv_scale)
        k += _surface(s, plane([hi[0], lo[1], lo[2]], [-e[0], 0, 0], [0, e[1], 0], [0, 0, -1]), nu, nv, material, uv_scale)
        k += _surface(s, plane([hi[0], lo[1], hi[2]], [0, 0, -e[2]], [0, e[1], 0], [1, 0, 0]), nw, nv, material, uv_scale)
        k += _surface(s, plane([lo[0], lo[1], lo[2]], [0, 0, e[2]], [0, e[1], 0], [-1, 0, 0]), nw, nv, material, uv_scale)
        k += _surface(s, plane([lo[0], hi[1], hi[2]], [e[0], 0, 0], [0, 0, -e[2]], [0, 1, 0]), nu, nw, material, uv_scale)
        k += _surface(s, plane([lo[0], lo[1], lo[2]], [e[0], 0, 0], [0, 0, e[2]], [0, -1, 0]), nu, nw, material, uv_scale)
        return k
    tris += _surface(s, plane([-LX, 0, -LZ], [2 * LX, 0, 0], [0, 0, 2 * LZ], [0, 1, 0]), n(220), n(150), wood, (6, 4))
    tris += _surface(s, plane([-LX, LY, LZ], [2 * LX, 0, 0], [0, 0, -2 * LZ], [0, -1, 0]), n(160), n(110), wall, (4, 3))
    tris += _surface(s, plane([-LX, 0, -LZ], [0, 0, 2 * LZ], [0, LY, 0], [1, 0, 0]), n(120), n(40), wall, (3, 1))
    tris += _surface(s, plane([LX, 0, LZ], [0, 0, -2 * LZ], [0, LY, 0], [-1, 0, 0]), n(120), n(40), wall, (3, 1))
    tris += _surface(s, plane([LX, 0, -LZ], [-2 * LX, 0, 0], [0, LY, 0], [0, 0, 1]), n(180), n(40), wall, (4, 1))
    # window wall at z = +LZ: sill, lintel, piers, and glass slabs in the openings
    nwin = 6
    tris += box([-LX, 0.0, LZ - 0.15], [LX, 0.9, LZ + 0.15], wall, n(120), n(8), n(2))
    tris += box([-LX, 3.0, LZ - 0.15], [LX, LY, LZ + 0.15], wall, n(120), n(6), n(2))
    edges = np.linspace(-LX, LX, nwin + 1)
    for i in range(nwin + 1):
        xc = edges[i]
        tris += box([max(xc - 0.25, -LX), 0.9, LZ - 0.15], [min(xc + 0.25, LX), 3.0, LZ + 0.15], wall, n(4), n(20), n(2))
    for i in range(nwin):
        tris += box([edges[i] + 0.25, 0.9, LZ - 0.02], [edges[i + 1] - 0.25, 3.0, LZ + 0.02], glass, n(6), n(6), 1)
    # desks and chairs
    rows_, cols_ = 5, 6
    for r_ in range(rows_):
        for c_ in range(cols_):
            x = -LX + 2.0 + c_ * (2 * LX - 4.0) / (cols_ - 1)
            z = -LZ + 2.2 + r_ * (2 * LZ - 4.6) / (rows_ - 1)
            tris += box([x - 0.6, 0.72, z - 0.35], [x + 0.6, 0.76, z + 0.35], wood, n(22), n(2), n(14), (1, 1))
            for lx, lz in ((-0.55, -0.3), (0.55, -0.3), (-0.55, 0.3), (0.55, 0.3)):
                tris += box([x + lx - 0.025, 0.0, z + lz - 0.025], [x + lx + 0.025, 0.72, z + lz + 0.025], plastic, n(2), n(12), n(2))

            def seat(u, v, x=x, z=z):
                th, ph = 2 * np.pi * u, np.pi * v
                nn = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], axis=1)
                rad = np.array([0.24, 0.05, 0.24])
                g = nn / rad
                return np.array([x, 0.45, z - 0.7]) + nn * rad, g / np.linalg.norm(g, axis=1, keepdims=True)
            tris += _surface(s, seat, n(36), n(18), plastic)

            def back(u, v, x=x, z=z):
                th = (u - 0.5) * 1.2
                p = np.stack([x + 0.3 * np.sin(th), 0.55 + 0.4 * v, z - 0.95 + 0.08 * (1 - np.cos(th))], axis=1)
                nn = np.stack([-np.sin(th) * 0.3, np.zeros_like(th), np.ones_like(th)], axis=1)
                return p, nn / np.linalg.norm(nn, axis=1, keepdims=True)
            tris += _surface(s, back, n(30), n(20), plastic)
    # a glass sphere on the teacher's desk and the desk itself
    tris += box([-1.2, 0.0, -LZ + 0.6], [1.2, 0.8, -LZ + 1.4], wood, n(40), n(20), n(16))
    pos, nrm, idx, uv = uv_sphere(max(8, n(60)), max(16, n(120)))
    s.add_mesh((pos * np.float32(0.22) + np.array([0.5, 1.02, -LZ + 1.0], np.float32)).astype(np.float32), idx, glass, normal=nrm, uv=uv)
    tris += len(idx)
    cr, sr = math.cos(-math.pi / 2), math.sin(-math.pi / 2)
    l2w = np.array([[1, 0, 0, 0], [0, cr, -sr, 0], [0, sr, cr, 0], [0, 0, 0, 1]], np.float32)
    if env is None:  # the map BASELINE configs[3] names, shipped as a data fixture; the seeded sky only where it is absent
        env = tx.read_rgbe(TANK_FARM_HDR) if os.path.exists(TANK_FARM_HDR) else synthetic_env_map_1k()
    tx.add_infinite_light(s, env, light_to_world=l2w)
    cam = look_at_camera([-7.5, 1.7, -4.6], [3.0, 1.2, 3.5], [0, 1, 0], 60.0, resolution)
    return cam, s
