"""ctypes mirror of include/ptrs.h (the C ABI of the drop-in boundary).

Field order and types must match the header exactly; tests/test_abi.py checks the struct sizes
against the values compiled into the shared library (ptrs_abi_sizeof).
"""
import ctypes as C

import numpy as np

PTRS_OK = 0
TEX_CONSTANT, TEX_CHECKER, TEX_IMAGE = 0, 1, 2
WRAP_REPEAT, WRAP_BLACK, WRAP_CLAMP = 0, 1, 2
MAT_MATTE, MAT_METAL, MAT_MIRROR, MAT_GLASS, MAT_DISNEY, MAT_SUBSTRATE, MAT_NORMAL = range(7)
LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_AREA, LIGHT_INFINITE = range(4)
FLAG_COUNTERS, FLAG_TIMING, FLAG_FILM_ZERO = 1, 2, 4
SAMPLER_SOBOL, SAMPLER_STRATIFIED = 0, 1

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


class PtrsTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("channels", C.c_int32), ("value", C.c_float * 3), ("value2", C.c_float * 3),
                ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float), ("wrap", C.c_int32),
                ("n_levels", C.c_int32), ("level_data", C.POINTER(f32p)), ("level_cols", i32p), ("level_rows", i32p)]


class PtrsMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tex", C.c_int32 * 6), ("flags", C.c_int32), ("inner", C.c_int32)]


class PtrsMesh(C.Structure):
    _fields_ = [("n_verts", C.c_uint32), ("n_tris", C.c_uint32), ("pos", f32p), ("normal", f32p), ("tangent", f32p),
                ("uv", f32p), ("indices", u32p), ("material", C.c_int32), ("alpha_mask_tex", C.c_int32),
                ("reverse_orientation", C.c_int32), ("transform_swaps_handedness", C.c_int32)]


class PtrsLight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("v", C.c_float * 3), ("c", C.c_float * 3), ("mesh", C.c_uint32), ("tri", C.c_uint32),
                ("ke_tex", C.c_int32), ("world_center", C.c_float * 3), ("world_radius", C.c_float), ("lmap_tex", C.c_int32),
                ("light_to_world", C.c_float * 16), ("world_to_light", C.c_float * 16), ("dist_nu", C.c_int32),
                ("dist_nv", C.c_int32), ("dist_func", f32p), ("dist_cdf", f32p), ("dist_func_int", f32p),
                ("marg_cdf", f32p), ("marg_func_int", C.c_float)]


class PtrsBvhNode(C.Structure):
    _fields_ = [("p_min", C.c_float * 3), ("p_max", C.c_float * 3), ("offset", C.c_uint32), ("num_prims", C.c_uint16),
                ("axis", C.c_uint8), ("pad", C.c_uint8)]


class PtrsSceneDesc(C.Structure):
    _fields_ = [("n_meshes", C.c_uint32), ("meshes", C.POINTER(PtrsMesh)), ("n_materials", C.c_uint32),
                ("materials", C.POINTER(PtrsMaterial)), ("n_textures", C.c_uint32), ("textures", C.POINTER(PtrsTexture)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(PtrsLight)), ("n_bvh_nodes", C.c_uint32),
                ("bvh_nodes", C.POINTER(PtrsBvhNode)), ("bvh_prims", u32p)]


class PtrsCamera(C.Structure):
    _fields_ = [("rot", C.c_float * 4), ("trans", C.c_float * 3), ("m00", C.c_float), ("m11", C.c_float), ("m22", C.c_float),
                ("m23", C.c_float), ("raster_to_screen", C.c_float * 16), ("dx_camera", C.c_float * 3), ("dy_camera", C.c_float * 3)]


class PtrsRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("rr_threshold", C.c_float), ("rr_start_depth", C.c_int32), ("rr_enable", C.c_int32), ("row_begin", C.c_int32),
                ("row_end", C.c_int32), ("device", C.c_int32), ("paths_per_pass", C.c_uint32), ("flags", C.c_uint32), ("sampler", C.c_int32),
                ("n_sampled_dimensions", C.c_int32)]


class PtrsStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_extension", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_mis", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("passes", C.c_uint64), ("kernel_launches", C.c_uint64),
                ("trace_launches", C.c_uint64), ("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double),
                ("ms_film", C.c_double), ("bvh_nodes", C.c_uint64), ("bvh_max_depth", C.c_uint64), ("device_bytes", C.c_uint64),
                ("ms_extend", C.c_double), ("ms_connect", C.c_double), ("ms_shade_kernels", C.c_double), ("ms_aux", C.c_double),
                ("extend_launches", C.c_uint64), ("connect_launches", C.c_uint64), ("shade_launches", C.c_uint64), ("aux_launches", C.c_uint64),
                ("film_launches", C.c_uint64), ("error_flags", C.c_uint64), ("node_steps_x64", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_steps_x64", C.c_uint64), ("debug", C.c_uint64 * 12), ("queue_segments", C.c_uint64), ("grid_wgs", C.c_uint64 * 4),
                ("resident_wgs_per_cu", C.c_uint64 * 4), ("lanes", C.c_uint64), ("grid_pct", C.c_uint64), ("ms_enqueue", C.c_double),
                ("ms_tail", C.c_double), ("tail_launches", C.c_uint64), ("tail_round", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    @property
    def rays(self):
        return self.rays_extension + self.rays_shadow + self.rays_mis


class PtrsHit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float)]


HIT_DTYPE = np.dtype([("prim", "<i4"), ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4")])
FILM_DTYPE = np.dtype([("rgb", "<f4", 3), ("weight", "<f4")])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else typ()


class SceneDescHolder:
    """Builds a PtrsSceneDesc from plain Python scene records and keeps every buffer alive."""

    def __init__(self, meshes, materials, textures, lights, bvh=None):
        self._keep = []
        k = self._keep
        tex = (PtrsTexture * max(len(textures), 1))()
        for i, t in enumerate(textures):
            x = tex[i]
            x.kind, x.channels = t.get("kind", TEX_CONSTANT), t.get("channels", 3)
            v = list(np.broadcast_to(_f32(t.get("value", 0.0)), (3,)))
            v2 = list(np.broadcast_to(_f32(t.get("value2", 0.0)), (3,)))
            x.value[:] = v
            x.value2[:] = v2
            x.su, x.sv, x.du, x.dv = t.get("su", 1.0), t.get("sv", 1.0), t.get("du", 0.0), t.get("dv", 0.0)
            x.wrap = t.get("wrap", WRAP_REPEAT)
            levels = t.get("levels")
            if levels:
                arrs = [_f32(l) for l in levels]
                k.append(arrs)
                ptrs = (f32p * len(arrs))(*[a.ctypes.data_as(f32p) for a in arrs])
                cols = np.array([a.shape[1] for a in arrs], dtype=np.int32)
                rows = np.array([a.shape[0] for a in arrs], dtype=np.int32)
                k += [ptrs, cols, rows]
                x.n_levels = len(arrs)
                x.level_data = C.cast(ptrs, C.POINTER(f32p))
                x.level_cols, x.level_rows = _ptr(cols, i32p), _ptr(rows, i32p)
        mats = (PtrsMaterial * max(len(materials), 1))()
        for i, m in enumerate(materials):
            mats[i].kind = m["kind"]
            tx = list(m.get("tex", [])) + [-1] * 6
            mats[i].tex[:] = tx[:6]
            mats[i].flags = m.get("flags", 0)
            mats[i].inner = m.get("inner", -1)
        ms = (PtrsMesh * max(len(meshes), 1))()
        for i, m in enumerate(meshes):
            pos = _f32(m["pos"]).reshape(-1, 3)
            idx = np.ascontiguousarray(m["indices"], dtype=np.uint32).reshape(-1, 3)
            nrm = _f32(m["normal"]).reshape(-1, 3) if m.get("normal") is not None else None
            tan = _f32(m["tangent"]).reshape(-1, 3) if m.get("tangent") is not None else None
            uv = _f32(m["uv"]).reshape(-1, 2) if m.get("uv") is not None else None
            k += [pos, idx, nrm, tan, uv]
            x = ms[i]
            x.n_verts, x.n_tris = pos.shape[0], idx.shape[0]
            x.pos, x.indices = _ptr(pos, f32p), _ptr(idx, u32p)
            x.normal, x.tangent, x.uv = _ptr(nrm, f32p), _ptr(tan, f32p), _ptr(uv, f32p)
            x.material = m["material"]
            x.alpha_mask_tex = m.get("alpha_mask_tex", -1)
            x.reverse_orientation = x.transform_swaps_handedness = 0
        ls = (PtrsLight * max(len(lights), 1))()
        for i, l in enumerate(lights):
            x = ls[i]
            x.kind = l["kind"]
            x.v[:] = list(_f32(l.get("v", [0, 0, 0])))
            x.c[:] = list(_f32(l.get("c", [0, 0, 0])))
            x.mesh, x.tri, x.ke_tex = l.get("mesh", 0), l.get("tri", 0), l.get("ke_tex", -1)
            x.world_center[:] = list(_f32(l.get("world_center", [0, 0, 0])))
            x.world_radius = l.get("world_radius", 0.0)
            x.lmap_tex = l.get("lmap_tex", -1)
            if l["kind"] == LIGHT_INFINITE:
                x.light_to_world[:] = list(_f32(l["light_to_world"]).reshape(16))
                x.world_to_light[:] = list(_f32(l["world_to_light"]).reshape(16))
                d = l["dist"]
                arrs = [_f32(d["func"]), _f32(d["cdf"]), _f32(d["func_int"]), _f32(d["marg_cdf"])]
                k.append(arrs)
                x.dist_nu, x.dist_nv = d["nu"], d["nv"]
                x.dist_func, x.dist_cdf, x.dist_func_int, x.marg_cdf = [a.ctypes.data_as(f32p) for a in arrs]
                x.marg_func_int = d["marg_func_int"]
        self.desc = PtrsSceneDesc()
        d = self.desc
        d.n_meshes, d.meshes = len(meshes), ms
        d.n_materials, d.materials = len(materials), mats
        d.n_textures, d.textures = len(textures), tex
        d.n_lights, d.lights = len(lights), ls
        if bvh is not None:
            nodes, prims = bvh
            nodes = np.ascontiguousarray(nodes)
            prims = np.ascontiguousarray(prims, dtype=np.uint32)
            k += [nodes, prims]
            d.n_bvh_nodes = nodes.shape[0]
            d.bvh_nodes = C.cast(nodes.ctypes.data, C.POINTER(PtrsBvhNode))
            d.bvh_prims = _ptr(prims, u32p)
        k += [tex, mats, ms, ls]
