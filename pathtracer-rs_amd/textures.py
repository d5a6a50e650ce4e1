"""Host-side construction of image textures and the environment light, mirroring what the reference
does once at import time (outside PathIntegrator::render):

  MIPMap::new            src/pathtracer/texture.rs:279-405 (Lanczos resample to powers of two,
                         box-filtered pyramid through texel() with the wrap mode)
  ImageTexture::new      texture.rs:97-177 (u8 -> f32 / Spectrum / normal-map decoding)
  read_hdr_image_to_mat  src/pathtracer/light.rs:331-346 (image 0.23 HdrDecoder: c * 2^(e-136))
  InfiniteAreaLight::new light.rs:348-398 (2x-supersampled sin-weighted luminance distribution)
  Distribution1D/2D::new src/pathtracer/sampling.rs:134-157,191-209

The results cross the C ABI as PtrsTexture.level_data / PtrsLight.dist_* (include/ptrs.h).  All
arithmetic is binary32; element order follows the reference loops.
"""
import math

import numpy as np

from . import abi

F = np.float32


def _round_up_pow2(v):
    return 1 << max(0, (int(v) - 1).bit_length())


def _lanczos(x, tau=2.0):
    x = abs(F(x))
    if x < F(1e-5):
        return F(1.0)
    if x > F(1.0):
        return F(0.0)
    x = x * F(math.pi)
    s = F(math.sin(float(x * F(tau)))) / (x * F(tau))
    return s * (F(math.sin(float(x))) / x)


def _resample_weights(old_res, new_res):
    """texture.rs:213-236: 4-tap Lanczos weights per new texel."""
    out = []
    for i in range(new_res):
        center = (F(i) + F(0.5)) * F(old_res) / F(new_res)
        first = np.floor((center - F(2.0)) + F(0.5))
        w = [_lanczos((first + F(j) + F(0.5) - center) / F(2.0)) for j in range(4)]
        inv = F(1.0) / (((w[0] + w[1]) + w[2]) + w[3])
        out.append((max(int(first), 0), [x * inv for x in w]))  # `first_texel as usize` saturates at 0
    return out


def _wrap_index(i, n, wrap):
    if wrap == abi.WRAP_REPEAT:
        return i % n
    if wrap == abi.WRAP_CLAMP:
        return min(max(i, 0), n - 1)
    return i


def build_mipmap(image, wrap=abi.WRAP_REPEAT):
    """image: (rows, cols, channels) float32.  Returns the pyramid as a list of arrays, level 0 first."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    if img.ndim == 2:
        img = img[..., None]
    rows, cols, ch = img.shape
    if (cols & (cols - 1)) or (rows & (rows - 1)):
        pc, pr = _round_up_pow2(cols), _round_up_pow2(rows)
        res = np.zeros((pr, pc, ch), dtype=np.float32)
        sw = _resample_weights(cols, pc)
        for t in range(rows):
            for s in range(pc):
                first, w = sw[s]
                for j in range(4):
                    o = _wrap_index(first + j, cols, wrap)
                    if 0 < o < cols:  # sic: texel 0 is skipped (texture.rs:321)
                        res[t, s] += img[t, o] * w[j]
        tw = _resample_weights(rows, pr)
        for s in range(pc):
            work = np.zeros((pr, ch), dtype=np.float32)
            for t in range(pr):
                first, w = tw[t]
                for j in range(4):
                    o = _wrap_index(first + j, rows, wrap)
                    if o < rows:
                        work[t] += res[o, s] * w[j]
            res[:, s] = work
        img, rows, cols = res, pr, pc
    levels = [img]
    n_levels = 1 + (max(rows, cols).bit_length() - 1)
    for _ in range(1, n_levels):
        prev = levels[-1]
        pr, pc = prev.shape[0], prev.shape[1]
        tr, sr = max(1, pr // 2), max(1, pc // 2)

        def tex(s_idx, t_idx):
            if wrap == abi.WRAP_REPEAT:
                return prev[np.mod(t_idx, pr)][:, np.mod(s_idx, pc)]
            if wrap == abi.WRAP_CLAMP:
                return prev[np.clip(t_idx, 0, pr - 1)][:, np.clip(s_idx, 0, pc - 1)]
            out = np.zeros((len(t_idx), len(s_idx), ch), dtype=np.float32)
            tv, sv = t_idx < pr, s_idx < pc
            out[np.ix_(tv, sv)] = prev[t_idx[tv]][:, s_idx[sv]]
            return out

        s2, t2 = 2 * np.arange(sr), 2 * np.arange(tr)
        nxt = (((tex(s2, t2) + tex(s2 + 1, t2)) + tex(s2, t2 + 1)) + tex(s2 + 1, t2 + 1)) * F(0.25)
        levels.append(np.ascontiguousarray(nxt, dtype=np.float32))
    return levels


def inverse_gamma_correct(v):  # common/math.rs:141-147
    v = np.asarray(v, dtype=np.float32)
    lo = v * F(1.0) / F(12.92)
    hi = np.power(((v + F(0.055)) * F(1.0) / F(1.055)).astype(np.float64), 2.4).astype(np.float32)
    return np.where(v <= F(0.04045), lo, hi).astype(np.float32)


def spectrum_texture(scene, rgb8, scale=(1.0, 1.0, 1.0), wrap=abi.WRAP_REPEAT, uvmap=(1.0, 1.0, 0.0, 0.0), gamma=True):
    """ImageTexture::<Spectrum>::new (texture.rs:123-147) from an (rows, cols, 3) uint8 image."""
    v = np.asarray(rgb8, dtype=np.float32) / F(255.0)
    if gamma:
        v = inverse_gamma_correct(v)
    v = (np.array(scale, dtype=np.float32) * v).astype(np.float32)
    su, sv, du, dv = uvmap
    return scene.add_texture(kind=abi.TEX_IMAGE, channels=3, levels=build_mipmap(v, wrap), wrap=wrap, su=su, sv=sv, du=du, dv=dv)


def float_texture(scene, gray8, scale=1.0, wrap=abi.WRAP_REPEAT, uvmap=(1.0, 1.0, 0.0, 0.0)):
    """ImageTexture::<f32>::new (texture.rs:97-121)."""
    v = (F(scale) * (np.asarray(gray8, dtype=np.float32) / F(255.0))).astype(np.float32)
    su, sv, du, dv = uvmap
    return scene.add_texture(kind=abi.TEX_IMAGE, channels=1, levels=build_mipmap(v, wrap), wrap=wrap, su=su, sv=sv, du=du, dv=dv)


def normal_map_texture(scene, rgb8, scale=(1.0, 1.0), wrap=abi.WRAP_REPEAT, uvmap=(1.0, 1.0, 0.0, 0.0)):
    """ImageTexture::<Vector3>::new (texture.rs:149-177): (p/127.5 - 1), x and y scaled."""
    p = np.asarray(rgb8, dtype=np.float32)
    v = (p / F(127.5) - F(1.0)).astype(np.float32)
    v[..., 0] *= F(scale[0])
    v[..., 1] *= F(scale[1])
    su, sv, du, dv = uvmap
    return scene.add_texture(kind=abi.TEX_IMAGE, channels=3, levels=build_mipmap(v, wrap), wrap=wrap, su=su, sv=sv, du=du, dv=dv)


def read_rgbe(path):
    """Radiance .hdr (RGBE, new-style RLE or flat) -> (rows, cols, 3) float32, value = c * 2^(e-136)."""
    data = open(path, "rb").read()
    pos = 0
    if not data.startswith(b"#?"):
        raise ValueError("not a Radiance file")
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line == b"":
            break
    end = data.index(b"\n", pos)
    res = data[pos:end].split()
    pos = end + 1
    if res[0] != b"-Y" or res[2] != b"+X":
        raise ValueError("unsupported orientation " + repr(res))
    rows, cols = int(res[1]), int(res[3])
    out = np.zeros((rows, cols, 4), dtype=np.uint8)
    buf = np.frombuffer(data, dtype=np.uint8)
    for y in range(rows):
        if cols < 8 or cols > 0x7FFF or buf[pos] != 2 or buf[pos + 1] != 2 or (buf[pos + 2] & 0x80):
            out[y] = buf[pos:pos + 4 * cols].reshape(cols, 4)
            pos += 4 * cols
            continue
        if (int(buf[pos + 2]) << 8 | int(buf[pos + 3])) != cols:
            raise ValueError("scanline width mismatch")
        pos += 4
        for c in range(4):
            x = 0
            while x < cols:
                n = int(buf[pos])
                pos += 1
                if n > 128:
                    n -= 128
                    out[y, x:x + n, c] = buf[pos]
                    pos += 1
                else:
                    out[y, x:x + n, c] = buf[pos:pos + n]
                    pos += n
                x += n
    e = out[..., 3].astype(np.int32)
    scale = np.where(e == 0, F(0.0), np.exp2((e - 136).astype(np.float32))).astype(np.float32)
    return (out[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


def _distribution_1d(f):
    """Distribution1D::new (sampling.rs:134-157) for every row of f at once."""
    f = np.asarray(f, dtype=np.float32)
    n = f.shape[-1]
    cdf = np.zeros(f.shape[:-1] + (n + 1,), dtype=np.float32)
    cdf[..., 1:] = np.cumsum(f / F(n), axis=-1, dtype=np.float32)
    func_int = cdf[..., n].copy()
    uniform = (np.arange(1, n + 1, dtype=np.float32) / F(n)).astype(np.float32)
    zero = func_int == 0
    with np.errstate(divide="ignore", invalid="ignore"):
        normed = cdf[..., 1:] / func_int[..., None]
    cdf[..., 1:] = np.where(zero[..., None], uniform, normed)
    return cdf, func_int


def _bilinear_level0(img, up, vp):
    """MIPMap::triangle(0, st) with Repeat wrap (texture.rs:413-428) on a grid of st."""
    rows, cols = img.shape[:2]
    s = up * F(cols) - F(0.5)
    t = vp * F(rows) - F(0.5)
    s0, t0 = np.floor(s), np.floor(t)
    ds, dt = (s - s0).astype(np.float32), (t - t0).astype(np.float32)
    s0, t0 = s0.astype(np.int64), t0.astype(np.int64)

    def tx(si, ti):
        return img[np.mod(ti, rows)[:, None], np.mod(si, cols)[None, :]]

    one = F(1.0)
    a = tx(s0, t0) * (one - ds)[None, :, None] * (one - dt)[:, None, None]
    b = tx(s0, t0 + 1) * (one - ds)[None, :, None] * dt[:, None, None]
    c = tx(s0 + 1, t0) * ds[None, :, None] * (one - dt)[:, None, None]
    d = tx(s0 + 1, t0 + 1) * ds[None, :, None] * dt[:, None, None]
    return (((a + b) + c) + d).astype(np.float32)


def add_infinite_light(scene, radiance_map, light_to_world=None, scale=(1.0, 1.0, 1.0)):
    """InfiniteAreaLight::new (light.rs:348-398).  radiance_map: (rows, cols, 3) float32, power-of-two
    sized (the bundled 1024x512 map is); light_to_world: 4x4 (default identity)."""
    texels = (np.asarray(radiance_map, dtype=np.float32) * np.array(scale, dtype=np.float32)).astype(np.float32)
    rows, cols = texels.shape[:2]
    levels = build_mipmap(texels, abi.WRAP_REPEAT)
    lmap = scene.add_texture(kind=abi.TEX_IMAGE, channels=3, levels=levels, wrap=abi.WRAP_REPEAT)
    width, height = 2 * levels[0].shape[1], 2 * levels[0].shape[0]
    f_width = F(0.5) / F(min(width, height))
    level = F(len(levels)) - F(1.0) + F(math.log2(max(float(f_width), 1e-8)))
    if not level < 0:
        raise NotImplementedError("environment maps whose distribution lookup is not at level 0")
    vp = ((np.arange(height, dtype=np.float32) + F(0.5)) / F(height)).astype(np.float32)
    up = ((np.arange(width, dtype=np.float32) + F(0.5)) / F(width)).astype(np.float32)
    rgb = _bilinear_level0(levels[0], up, vp)
    lum = ((rgb[..., 0] * F(0.212671) + rgb[..., 1] * F(0.715160)) + rgb[..., 2] * F(0.072169)).astype(np.float32)
    sin_theta = np.sin((F(math.pi) * vp).astype(np.float64)).astype(np.float32)
    func = (sin_theta[:, None] * lum).astype(np.float32)
    cdf, func_int = _distribution_1d(func)
    mcdf, mint = _distribution_1d(func_int[None, :])
    l2w = np.eye(4, dtype=np.float32) if light_to_world is None else np.asarray(light_to_world, dtype=np.float32)
    w2l = np.linalg.inv(l2w.astype(np.float64)).astype(np.float32)
    scene.lights.append(dict(kind=abi.LIGHT_INFINITE, lmap_tex=lmap, light_to_world=l2w, world_to_light=w2l,
                             dist=dict(nu=width, nv=height, func=func, cdf=cdf, func_int=func_int, marg_cdf=mcdf[0], marg_func_int=float(mint[0]))))
    return len(scene.lights) - 1
