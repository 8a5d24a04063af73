"""ctypes mirror of include/rtcore.h and include/rthost.h.

The product path is librtcore.so (hand-written HIP kernels behind the C-ABI).  There is no CPU
fallback: if the library is missing, `load_core()` raises.  Nothing in this package imports, links
or calls anything under oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_OK = 0
RT_HIT_FRONT, RT_HIT_BACK, RT_HIT_FRONT_AND_BACK = 1, 2, 3
RT_PRIM_NONE, RT_PRIM_SPHERE, RT_PRIM_PLANE, RT_PRIM_MESH = 0, 1, 2, 3
RT_MTL_BLINN, RT_MTL_WHITE, RT_MTL_PHONG, RT_MTL_MICROFACET = 1, 2, 3, 4
RT_EXT_MATERIALS, RT_EXT_TEX_FILTER = 1, 2   # rt_render_params.extensions
RT_SAMPLER_PHILOX, RT_SAMPLER_PCG_HALTON = 0, 1
RT_LIGHT_AMBIENT, RT_LIGHT_DIRECT, RT_LIGHT_POINT = 1, 2, 3
RT_K_NAMES = ["raygen", "trace_closest", "shade", "shadow", "accum", "photon_trace", "photon_gather", "denoise", "trace_mesh",
              "shadow_mesh", "photon_build"]


class rt_mat34(C.Structure):
    _fields_ = [("m", C.c_float * 12)]


class rt_node(C.Structure):
    _fields_ = [("parent", C.c_int32), ("prim_kind", C.c_int32), ("mesh_id", C.c_int32), ("material_id", C.c_int32),
                ("tm", rt_mat34), ("itm", rt_mat34)]


class rt_bvh_node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("a", C.c_uint32), ("bmax", C.c_float * 3), ("b", C.c_uint32)]


class rt_mesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("uvw", C.POINTER(C.c_float)),
                ("pos_idx", C.POINTER(C.c_uint32)), ("nrm_idx", C.POINTER(C.c_uint32)), ("uvw_idx", C.POINTER(C.c_uint32)),
                ("n_positions", C.c_uint32), ("n_normals", C.c_uint32), ("n_uvw", C.c_uint32), ("n_faces", C.c_uint32),
                ("bvh", C.POINTER(rt_bvh_node)), ("bvh_elements", C.POINTER(C.c_uint32)),
                ("n_bvh_nodes", C.c_uint32), ("n_bvh_elements", C.c_uint32),
                ("bound_min", C.c_float * 3), ("bound_max", C.c_float * 3)]


class rt_tex_color(C.Structure):
    _fields_ = [("value", C.c_float * 3), ("map", C.c_int32)]


class rt_tex_float(C.Structure):
    _fields_ = [("value", C.c_float), ("map", C.c_int32)]


class rt_texmap(C.Structure):
    _fields_ = [("kind", C.c_int32), ("image", C.c_int32), ("itm", rt_mat34), ("color", rt_tex_color * 2)]


class rt_image(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("diffuse", rt_tex_color), ("specular", rt_tex_color), ("emission", rt_tex_color),
                ("reflection", rt_tex_color), ("refraction", rt_tex_color), ("glossiness", rt_tex_float),
                ("absorption", C.c_float * 3), ("ior", C.c_float)]


class rt_light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("intensity", C.c_float * 3), ("position", C.c_float * 3),
                ("direction", C.c_float * 3), ("size", C.c_float), ("attenuation", C.c_float)]


class rt_camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float),
                ("focaldist", C.c_float), ("dof", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("srgb", C.c_int32)]


class rt_scene_desc(C.Structure):
    _fields_ = [("nodes", C.POINTER(rt_node)), ("n_nodes", C.c_uint32),
                ("meshes", C.POINTER(rt_mesh)), ("n_meshes", C.c_uint32),
                ("materials", C.POINTER(rt_material)), ("n_materials", C.c_uint32),
                ("texmaps", C.POINTER(rt_texmap)), ("n_texmaps", C.c_uint32),
                ("images", C.POINTER(rt_image)), ("n_images", C.c_uint32),
                ("lights", C.POINTER(rt_light)), ("n_lights", C.c_uint32),
                ("background", rt_tex_color), ("environment", rt_tex_color), ("camera", rt_camera)]


class rt_ray(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("dir", C.c_float * 3)]


class rt_hit(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("z", C.c_float), ("N", C.c_float * 3), ("uvw", C.c_float * 3),
                ("node", C.c_int32), ("face", C.c_int32), ("light", C.c_int32), ("front", C.c_int32)]


class rt_render_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sampler", C.c_int32),
                ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("spp_min", C.c_int32), ("spp_max", C.c_int32), ("sample_begin", C.c_int32), ("sample_count", C.c_int32),
                ("adaptive_threshold", C.c_float), ("pixel_jitter", C.c_int32), ("max_bounces", C.c_int32),
                ("mc_bounces", C.c_int32), ("mc_samples", C.c_int32), ("shadow_min", C.c_int32), ("shadow_max", C.c_int32),
                ("light_size_scale", C.c_float), ("dedup_hard_shadows", C.c_int32), ("photons", C.c_int32),
                ("n_photons", C.c_int32), ("photon_radius", C.c_float), ("photon_k", C.c_int32),
                ("ellipticity_global", C.c_float), ("ellipticity_caustic", C.c_float), ("skip_zero_weight", C.c_int32),
                ("tile_begin", C.c_int32), ("tile_count", C.c_int32), ("tile_stride", C.c_int32),
                ("photon_emit_limit", C.c_int32), ("photon_max_bounces", C.c_int32), ("frame_mem_mb", C.c_int32), ("extensions", C.c_uint32)]


class rt_framebuffer(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("radiance", C.c_void_p), ("sample_count", C.c_void_p),
                ("hit_node", C.c_void_p), ("hit_face", C.c_void_p), ("on_device", C.c_int32), ("reserved", C.c_int32),
                ("depth", C.c_void_p), ("normal", C.c_void_p), ("albedo", C.c_void_p)]


class rt_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_primary", "rays_secondary", "rays_shadow", "rays_photon", "inst_tests", "sphere_tests",
                 "plane_tests", "node_visits", "tri_tests", "shades", "tex_evals", "photon_visits", "gathers", "paths",
                 "closest_inst_tests", "closest_node_visits", "closest_tri_tests", "mesh_inst_tests", "mesh_items",
                 "rays_shadow_proven")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


RT_K_COUNT = len(RT_K_NAMES)


class rt_profile(C.Structure):
    _fields_ = [("ms", C.c_double * RT_K_COUNT), ("launches", C.c_uint64 * RT_K_COUNT), ("items", C.c_uint64 * RT_K_COUNT)]


class rt_photon(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("power", C.c_float), ("dir", C.c_float * 3), ("color", C.c_uint8 * 3),
                ("plane", C.c_uint8)]


def default_params():
    """Reference defaults (raytracer.h:17-23, renderer.h:172-174); same values rt_render_params_default fills."""
    p = rt_render_params()
    p.sampler = RT_SAMPLER_PHILOX
    p.seed_lo, p.seed_hi = 1, 0
    p.spp_min, p.spp_max = 32, 128
    p.adaptive_threshold = 0.01
    p.pixel_jitter = 1
    p.max_bounces = 3
    p.mc_bounces, p.mc_samples = 0, 1
    p.shadow_min, p.shadow_max = 16, 128
    p.light_size_scale = 1.0
    p.dedup_hard_shadows = 1
    p.photons = 1
    p.n_photons = 100000
    p.photon_radius = 3.0
    p.photon_k = 128
    p.ellipticity_global, p.ellipticity_caustic = 1.0, 0.25
    p.skip_zero_weight = 1
    return p


_host = None
_core = None


def load_host():
    """librthost.so: re-authored CPU scene loader / PNG IO (not accelerated, not the hot path)."""
    global _host
    if _host is not None:
        return _host
    path = os.path.join(LIB_DIR, "librthost.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C raytracer_amd/csrc`")
    lib = C.CDLL(path)
    lib.rth_scene_load_file.restype = C.c_void_p
    lib.rth_scene_load_file.argtypes = [C.c_char_p, C.c_char_p]
    lib.rth_scene_load_string.restype = C.c_void_p
    lib.rth_scene_load_string.argtypes = [C.c_char_p, C.c_char_p]
    lib.rth_scene_free.argtypes = [C.c_void_p]
    lib.rth_last_error.restype = C.c_char_p
    lib.rth_scene_desc.restype = C.POINTER(rt_scene_desc)
    lib.rth_scene_desc.argtypes = [C.c_void_p]
    lib.rth_scene_node_name.restype = C.c_char_p
    lib.rth_scene_node_name.argtypes = [C.c_void_p, C.c_int]
    lib.rth_scene_warnings.argtypes = [C.c_void_p]
    lib.rth_scene_generate_stress.restype = C.c_void_p
    lib.rth_scene_generate_stress.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rth_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.rth_read_image_rgb.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rth_free.argtypes = [C.c_void_p]
    lib.rth_set_host_bvh.argtypes = [C.c_int]
    lib.rth_set_host_bvh.restype = None
    lib.rth_scene_load_file_ex.restype = C.c_void_p
    lib.rth_scene_load_file_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    lib.rth_scene_load_string_ex.restype = C.c_void_p
    lib.rth_scene_load_string_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    lib.rth_scene_generate_stress_ex.restype = C.c_void_p
    lib.rth_scene_generate_stress_ex.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
    _host = lib
    return lib


CORE_SYMBOLS = [
    "rt_context_create", "rt_context_destroy", "rt_last_error", "rt_scene_commit", "rt_trace_closest", "rt_trace_any",
    "rt_render", "rt_render_begin", "rt_render_poll", "rt_render_wait", "rt_render_accumulate", "rt_resolve",
    "rt_photons_build", "rt_photons_download", "rt_photons_set", "rt_photon_gather", "rt_denoise", "rt_counters_enable",
    "rt_counters_reset", "rt_counters_get", "rt_profile_enable", "rt_profile_reset", "rt_profile_get",
    "rt_render_params_default", "rt_set_option", "rt_debug_alloc_count", "rt_denoise_noalloc", "rt_photons_download_tree", "rt_reduce_accumulators",
    "rt_scene_mesh_bvh", "rt_debug_philox", "rt_context_create_multi", "rt_alloc_pinned", "rt_free_pinned",
]
HOST_SYMBOLS = [
    "rth_scene_load_file", "rth_scene_load_string", "rth_scene_free", "rth_last_error", "rth_scene_desc",
    "rth_scene_node_name", "rth_scene_warnings", "rth_scene_generate_stress", "rth_write_png", "rth_read_image_rgb",
    "rth_free", "rth_set_host_bvh", "rth_scene_load_file_ex", "rth_scene_load_string_ex", "rth_scene_generate_stress_ex",
]
RTH_LOAD_NO_BVH = 1


def load_core():
    """librtcore.so: the HIP renderer core.  Raises if absent — there is no fallback path."""
    global _core
    if _core is not None:
        return _core
    # RTCORE_LIB: developer knob to A/B another HIP build of the same sources (kernel experiments); never a CPU library
    path = os.environ.get("RTCORE_LIB") or os.path.join(LIB_DIR, "librtcore.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP extension must be built (no CPU fallback exists)")
    _preload_torch_hip_runtime()
    _core = bind_core(C.CDLL(path))
    return _core


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as
    /opt/rocm's); if librtcore.so pulled in /opt/rocm's copy first, a later `import torch` would bring a second
    runtime and fail with "No HIP GPUs are available".  So when torch is installed, its copy is loaded first and
    librtcore.so binds to it (torch itself is not imported here)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(libdir, name)
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def bind_core(lib):
    """Declares the C-ABI signatures of include/rtcore.h on a loaded library object."""
    vp = C.c_void_p
    lib.rt_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_context_destroy.argtypes = [vp]
    lib.rt_context_create_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.rt_alloc_pinned.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.rt_free_pinned.argtypes = [vp]
    lib.rt_free_pinned.restype = None
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_scene_commit.argtypes = [vp, C.POINTER(rt_scene_desc)]
    lib.rt_trace_closest.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    lib.rt_trace_any.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.rt_render.argtypes = [vp, C.POINTER(rt_render_params), C.POINTER(rt_framebuffer)]
    lib.rt_render_begin.argtypes = [vp, C.POINTER(rt_render_params), C.POINTER(rt_framebuffer)]
    lib.rt_render_poll.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    lib.rt_render_wait.argtypes = [vp]
    lib.rt_render_accumulate.argtypes = [vp, C.POINTER(rt_render_params), vp]
    lib.rt_resolve.argtypes = [vp, C.POINTER(rt_render_params), vp, C.POINTER(rt_framebuffer)]
    lib.rt_photons_build.argtypes = [vp, C.POINTER(rt_render_params)]
    lib.rt_photons_download.argtypes = [vp, C.c_int, vp, C.c_int32, C.POINTER(C.c_int32)]
    lib.rt_photons_download_tree.argtypes = [vp, C.c_int, vp, C.c_int32, C.POINTER(C.c_int32)]
    lib.rt_scene_mesh_bvh.argtypes = [vp, C.c_int, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rt_photons_set.argtypes = [vp, C.c_int, vp, C.c_int32]
    lib.rt_photon_gather.argtypes = [vp, C.c_int, vp, vp, C.c_size_t, C.c_float, C.c_float, C.c_int, vp]
    lib.rt_denoise.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    lib.rt_counters_enable.argtypes = [vp, C.c_int]
    lib.rt_counters_reset.argtypes = [vp]
    lib.rt_counters_get.argtypes = [vp, C.POINTER(rt_counters)]
    lib.rt_profile_enable.argtypes = [vp, C.c_int]
    lib.rt_profile_reset.argtypes = [vp]
    lib.rt_profile_get.argtypes = [vp, C.POINTER(rt_profile)]
    lib.rt_render_params_default.argtypes = [C.POINTER(rt_render_params)]
    lib.rt_render_params_default.restype = None
    lib.rt_denoise_noalloc.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    lib.rt_reduce_accumulators.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int]
    lib.rt_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.rt_debug_alloc_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.rt_debug_philox.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_size_t, vp, vp]
    return lib
