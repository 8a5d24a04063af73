"""raytracer_amd — host-side mirror of the reference renderer API over the MI355X renderer core.

`Renderer` keeps the names and argument meaning of the reference's `class Renderer`
(reference RayTracer/renderer.h:209-238): LoadScene, BeginRender, StopRender, TraceRay,
TraceShadowRay, GetPhotonMap, GetCausticsMap — so that the parity tests read like tests of the
reference would.  All rendering goes through the C-ABI of librtcore.so (include/rtcore.h); there is
no Python or CPU implementation of the hot path in this package.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import (RT_HIT_BACK, RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK, RT_SAMPLER_PCG_HALTON, RT_SAMPLER_PHILOX,
                   default_params, rt_counters, rt_framebuffer, rt_hit, rt_photon, rt_profile, rt_ray,
                   rt_render_params, rt_scene_desc)

__all__ = ["Scene", "Renderer", "RenderImage", "default_params", "rays_from_arrays", "hits_to_arrays",
           "RT_HIT_FRONT", "RT_HIT_BACK", "RT_HIT_FRONT_AND_BACK", "RT_SAMPLER_PHILOX", "RT_SAMPLER_PCG_HALTON"]

RAY_DTYPE = np.dtype([("p", np.float32, 3), ("dir", np.float32, 3)])
HIT_DTYPE = np.dtype([("p", np.float32, 3), ("z", np.float32), ("N", np.float32, 3), ("uvw", np.float32, 3),
                      ("node", np.int32), ("face", np.int32), ("light", np.int32), ("front", np.int32)])
BVH_NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("a", np.uint32), ("bmax", np.float32, 3), ("b", np.uint32)])
PHOTON_DTYPE = np.dtype([("position", np.float32, 3), ("power", np.float32), ("dir", np.float32, 3),
                         ("color", np.uint8, 3), ("plane", np.uint8)])
assert RAY_DTYPE.itemsize == C.sizeof(rt_ray) and HIT_DTYPE.itemsize == C.sizeof(rt_hit)
assert PHOTON_DTYPE.itemsize == C.sizeof(rt_photon) == 32


def rays_from_arrays(p, d):
    r = np.zeros(len(p), dtype=RAY_DTYPE)
    r["p"] = np.asarray(p, dtype=np.float32)
    r["dir"] = np.asarray(d, dtype=np.float32)
    return r


def hits_to_arrays(h):
    return {k: h[k] for k in HIT_DTYPE.names}


class Scene:
    """A loaded, flattened scene (host memory).  Mirrors what Renderer::LoadScene leaves behind
    (reference RayTracer/xmlload.cpp:42-77): Scene + Camera."""

    def __init__(self, handle):
        self._host = _ffi.load_host()
        if not handle:
            raise RuntimeError("scene load failed: " + self._host.rth_last_error().decode())
        self._h = C.c_void_p(handle)
        self.desc_ptr = self._host.rth_scene_desc(self._h)
        self.desc = self.desc_ptr.contents

    # host_bvh=False: the meshes come without a tree and Renderer.LoadScene (rt_scene_commit) builds it on the device
    # (a per-call flag of the loader: nothing process-wide is switched)
    @classmethod
    def load(cls, path, asset_dir=None, host_bvh=True):
        host = _ffi.load_host()
        flags = 0 if host_bvh else _ffi.RTH_LOAD_NO_BVH
        return cls(host.rth_scene_load_file_ex(os.fsencode(path), os.fsencode(asset_dir) if asset_dir else None, flags))

    @classmethod
    def from_string(cls, xml, asset_dir=None, host_bvh=True):
        host = _ffi.load_host()
        flags = 0 if host_bvh else _ffi.RTH_LOAD_NO_BVH
        return cls(host.rth_scene_load_string_ex(xml.encode(), os.fsencode(asset_dir) if asset_dir else None, flags))

    @classmethod
    def stress(cls, seed=7, n_instances=2048, subdiv=4, ground_res=1024, width=3840, height=2160, host_bvh=True):
        host = _ffi.load_host()
        flags = 0 if host_bvh else _ffi.RTH_LOAD_NO_BVH
        return cls(host.rth_scene_generate_stress_ex(seed, n_instances, subdiv, ground_res, width, height, flags))

    def with_bvh(self, trees):
        """A view of this scene whose meshes carry other trees: `trees[mesh_id]` = (nodes, elements) as
        Renderer.GetMeshBVH returns them, or None for "no tree, build it on the device"."""
        return _SceneView(self, trees)

    @property
    def warnings(self):
        return self._host.rth_scene_warnings(self._h)

    def node_name(self, i):
        s = self._host.rth_scene_node_name(self._h, i)
        return s.decode() if s is not None else None

    @property
    def width(self):
        return self.desc.camera.width

    @property
    def height(self):
        return self.desc.camera.height

    def n_triangles(self):
        return sum(self.desc.meshes[i].n_faces for i in range(self.desc.n_meshes))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._host.rth_scene_free(self._h)
                self._h = None
        except Exception:
            pass


class _SceneView:
    """Scene.with_bvh: the parent's flattened scene with the BVH fields of some meshes replaced (host memory of both is
    kept alive by this object)."""

    def __init__(self, parent, trees):
        self._parent = parent
        self._host = parent._host
        self._keep = []
        n = parent.desc.n_meshes
        self._meshes = (_ffi.rt_mesh * max(n, 1))()
        for i in range(n):
            C.memmove(C.byref(self._meshes[i]), C.byref(parent.desc.meshes[i]), C.sizeof(_ffi.rt_mesh))
            if i in trees:
                m = self._meshes[i]
                if trees[i] is None:
                    m.bvh = None
                    m.bvh_elements = None
                    m.n_bvh_nodes = m.n_bvh_elements = 0
                else:
                    nodes = np.ascontiguousarray(trees[i][0], dtype=BVH_NODE_DTYPE)
                    elems = np.ascontiguousarray(trees[i][1], dtype=np.uint32)
                    self._keep += [nodes, elems]
                    m.bvh = C.cast(nodes.ctypes.data, C.POINTER(_ffi.rt_bvh_node))
                    m.bvh_elements = C.cast(elems.ctypes.data, C.POINTER(C.c_uint32))
                    m.n_bvh_nodes, m.n_bvh_elements = len(nodes), len(elems)
        self.desc = rt_scene_desc()
        C.memmove(C.byref(self.desc), C.byref(parent.desc), C.sizeof(rt_scene_desc))
        self.desc.meshes = C.cast(self._meshes, C.POINTER(_ffi.rt_mesh))
        self.desc_ptr = C.pointer(self.desc)

    width = property(lambda self: self.desc.camera.width)
    height = property(lambda self: self.desc.camera.height)
    warnings = property(lambda self: self._parent.warnings)

    def node_name(self, i):
        return self._parent.node_name(i)

    def n_triangles(self):
        return self._parent.n_triangles()

    def with_bvh(self, trees):
        return _SceneView(self, trees)


class RenderImage:
    """Output buffers (reference RenderImage, RayTracer/renderer.h:35-103)."""

    def __init__(self, w, h, pinned_lib=None):
        """pinned_lib: the loaded librtcore — the arrays then live in page-locked host memory (rt_alloc_pinned), which the
        device-to-host copies at the end of rt_render fill without a staging copy."""
        self.width, self.height = w, h
        self._pinned = []
        self._lib = pinned_lib

        def arr(shape, dtype, fill=0):
            if pinned_lib is None:
                return np.full(shape, fill, dtype)
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            ptr = C.c_void_p()
            if pinned_lib.rt_alloc_pinned(n, C.byref(ptr)) != 0:
                raise RuntimeError("rt_alloc_pinned failed")
            self._pinned.append(ptr)
            a = np.frombuffer((C.c_char * n).from_address(ptr.value), dtype=dtype).reshape(shape)
            a[...] = fill
            return a

        self.img = arr((h, w, 3), np.uint8)
        self.radiance = arr((h, w, 4), np.float32)
        self.sample_count = arr((h, w), np.int32)
        self.hit_node = arr((h, w), np.int32, -1)
        self.hit_face = arr((h, w), np.int32, -1)
        self.zbuffer = arr((h, w), np.float32)       # RenderImage::zbuffer (renderer.h:39)
        self.normal = arr((h, w, 3), np.float32)
        self.albedo = arr((h, w, 3), np.float32)

    def release(self):
        """Pinned images only: hands the page-locked memory back (the arrays must not be used afterwards)."""
        for name in ("img", "radiance", "sample_count", "hit_node", "hit_face", "zbuffer", "normal", "albedo"):
            if self._pinned:
                setattr(self, name, np.array(getattr(self, name)))   # detach: plain copies stay usable
        for ptr in self._pinned:
            self._lib.rt_free_pinned(ptr)
        self._pinned = []

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def framebuffer(self):
        fb = rt_framebuffer()
        fb.rgb8 = self.img.ctypes.data
        fb.radiance = self.radiance.ctypes.data
        fb.sample_count = self.sample_count.ctypes.data
        fb.hit_node = self.hit_node.ctypes.data
        fb.hit_face = self.hit_face.ctypes.data
        fb.depth = self.zbuffer.ctypes.data
        fb.normal = self.normal.ctypes.data
        fb.albedo = self.albedo.ctypes.data
        fb.on_device = 0
        return fb

    def SaveImage(self, path):
        host = _ffi.load_host()
        return host.rth_write_png(os.fsencode(path), self.img.ctypes.data, self.width, self.height, 3) == 0


class Renderer:
    """Drop-in for the reference's RayTracer : Renderer (RayTracer/raytracer.h:14-61) on one MI355X."""

    def __init__(self, device_id=0, _lib=None, _ctx=None):
        # _lib: tests pass the host-emulation build of the same sources (tests/_emu); the package itself only
        # ever loads librtcore.so and raises if it is missing.
        self._lib = _lib if _lib is not None else _ffi.load_core()
        ctx = _ctx if _ctx is not None else C.c_void_p()
        if _ctx is None:
            rc = self._lib.rt_context_create(device_id, C.byref(ctx))
            if rc != 0:
                raise RuntimeError(f"rt_context_create failed ({rc})")
        self._ctx = ctx
        self.scene = None
        self.renderImage = None
        self.params = default_params()

    @classmethod
    def create_multi(cls, device_ids, _lib=None):
        """One renderer per GPU of a single-process host (rt_context_create_multi: all contexts or none)."""
        lib = _lib if _lib is not None else _ffi.load_core()
        ids = (C.c_int * len(device_ids))(*device_ids)
        out = (C.c_void_p * len(device_ids))()
        rc = lib.rt_context_create_multi(len(device_ids), ids, out)
        if rc != 0:
            raise RuntimeError(f"rt_context_create_multi failed ({rc})")
        return [cls(d, _lib=lib, _ctx=C.c_void_p(out[i])) for i, d in enumerate(device_ids)]

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.rt_last_error(self._ctx)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- reference API -----------------------------------------------------------------
    def LoadScene(self, path_or_scene, asset_dir=None):
        self.scene = path_or_scene if isinstance(path_or_scene, (Scene, _SceneView)) else Scene.load(path_or_scene, asset_dir)
        self._check(self._lib.rt_scene_commit(self._ctx, self.scene.desc_ptr), "rt_scene_commit")
        return True

    def GetMeshBVH(self, mesh_id):
        """(nodes, elements) of a committed mesh's BVH in rt_mesh form — the tree it came with or the one built on the
        device for a mesh without one (reference: BVHTriMesh, RayTracer/objects.h:60-67)."""
        nn, ne = C.c_uint32(), C.c_uint32()
        self._check(self._lib.rt_scene_mesh_bvh(self._ctx, mesh_id, None, 0, None, 0, C.byref(nn), C.byref(ne)), "rt_scene_mesh_bvh")
        nodes = np.zeros(nn.value, dtype=BVH_NODE_DTYPE)
        elems = np.zeros(ne.value, dtype=np.uint32)
        self._check(self._lib.rt_scene_mesh_bvh(self._ctx, mesh_id, nodes.ctypes.data, nn.value, elems.ctypes.data, ne.value,
                                                C.byref(nn), C.byref(ne)), "rt_scene_mesh_bvh")
        return nodes, elems

    def _dims(self, params):
        w = params.width if params.width > 0 else self.scene.width
        h = params.height if params.height > 0 else self.scene.height
        return w, h

    def _join(self):
        """A render still in flight writes into the current renderImage's buffers: wait for it before they are replaced."""
        self._lib.rt_render_wait(self._ctx)   # (its status belongs to that render's own StopRender)

    def BeginRender(self, params=None):
        params = params or self.params
        w, h = self._dims(params)
        self._join()
        self.renderImage = RenderImage(w, h)
        self._fb = self.renderImage.framebuffer()
        self._check(self._lib.rt_render_begin(self._ctx, C.byref(params), C.byref(self._fb)), "rt_render_begin")

    def GetNumRenderedPixels(self):
        done, fin = C.c_int64(), C.c_int()
        self._check(self._lib.rt_render_poll(self._ctx, C.byref(done), C.byref(fin)), "rt_render_poll")
        return done.value

    def IsRenderDone(self):
        done, fin = C.c_int64(), C.c_int()
        self._check(self._lib.rt_render_poll(self._ctx, C.byref(done), C.byref(fin)), "rt_render_poll")
        return bool(fin.value)

    def StopRender(self):
        self._check(self._lib.rt_render_wait(self._ctx), "rt_render_wait")
        return self.renderImage

    def Render(self, params=None, pinned=False):
        """Blocking BeginRender + StopRender.  pinned: the output images in page-locked host memory (rt_alloc_pinned)."""
        params = params or self.params
        w, h = self._dims(params)
        self._join()
        self.renderImage = RenderImage(w, h, self._lib if pinned else None)
        fb = self.renderImage.framebuffer()
        self._check(self._lib.rt_render(self._ctx, C.byref(params), C.byref(fb)), "rt_render")
        return self.renderImage

    def TraceRay(self, rays, hitSide=RT_HIT_FRONT_AND_BACK):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        self._check(self._lib.rt_trace_closest(self._ctx, rays.ctypes.data, len(rays), hitSide, hits.ctypes.data),
                    "rt_trace_closest")
        return hits

    def TraceShadowRay(self, rays, t_max):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        t_max = np.ascontiguousarray(t_max, dtype=np.float32)
        occ = np.zeros(len(rays), dtype=np.uint8)
        self._check(self._lib.rt_trace_any(self._ctx, rays.ctypes.data, t_max.ctypes.data, len(rays), occ.ctypes.data),
                    "rt_trace_any")
        return occ.astype(bool)

    def BuildPhotonMaps(self, params=None):
        params = params or self.params
        self._check(self._lib.rt_photons_build(self._ctx, C.byref(params)), "rt_photons_build")

    def _photons(self, which, capacity=1 << 22):
        out = np.zeros(capacity, dtype=PHOTON_DTYPE)
        n = C.c_int32()
        self._check(self._lib.rt_photons_download(self._ctx, which, out.ctypes.data, capacity, C.byref(n)),
                    "rt_photons_download")
        return out[:n.value].copy()

    def GetPhotonTree(self, which, capacity=1 << 22):
        """The balanced map in heap order (node i + 1 at index i): PhotonMap::GetPhotons() after PrepareForIrradianceEstimation."""
        out = np.zeros(capacity, dtype=PHOTON_DTYPE)
        n = C.c_int32()
        self._check(self._lib.rt_photons_download_tree(self._ctx, which, out.ctypes.data, capacity, C.byref(n)),
                    "rt_photons_download_tree")
        return out[:n.value].copy()

    def GetPhotonMap(self):
        return self._photons(0)

    def GetCausticsMap(self):
        return self._photons(1)

    def SetPhotonMap(self, which, photons):
        """Caller-supplied photons (insertion order, powers scaled), then PrepareForIrradianceEstimation."""
        photons = np.ascontiguousarray(photons, dtype=PHOTON_DTYPE)
        self._check(self._lib.rt_photons_set(self._ctx, which, photons.ctypes.data, len(photons)), "rt_photons_set")

    def EstimateIrradiance(self, which, P, N, radius=3.0, ellipticity=1.0, k=128):
        P = np.ascontiguousarray(P, dtype=np.float32)
        N = np.ascontiguousarray(N, dtype=np.float32)
        out = np.zeros_like(P)
        self._check(self._lib.rt_photon_gather(self._ctx, which, P.ctypes.data, N.ctypes.data, len(P), radius,
                                               ellipticity, k, out.ctypes.data), "rt_photon_gather")
        return out

    def Denoise(self, rgb, albedo=None, normal=None):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w = rgb.shape[:2]
        out = np.zeros_like(rgb)
        a = np.ascontiguousarray(albedo, np.float32) if albedo is not None else None
        n = np.ascontiguousarray(normal, np.float32) if normal is not None else None
        self._check(self._lib.rt_denoise(self._ctx, rgb.ctypes.data, out.ctypes.data, w, h,
                                         a.ctypes.data if a is not None else None,
                                         n.ctypes.data if n is not None else None), "rt_denoise")
        return out

    def denoise_noalloc(self, d_rgb_in, d_rgb_out, width, height, d_albedo=0, d_normal=0):
        """rt_denoise_noalloc: device pointers (ints), no allocation; see include/rtcore.h."""
        self._check(self._lib.rt_denoise_noalloc(self._ctx, C.c_void_p(d_rgb_in), C.c_void_p(d_rgb_out), width, height,
                                                 C.c_void_p(d_albedo or None), C.c_void_p(d_normal or None)), "rt_denoise_noalloc")

    # -- multi-GPU / measurement helpers ---------------------------------------------------
    def render_accumulate(self, params, d_accum_ptr):
        self._check(self._lib.rt_render_accumulate(self._ctx, C.byref(params), C.c_void_p(d_accum_ptr)),
                    "rt_render_accumulate")

    def resolve(self, params, d_accum_ptr):
        w, h = self._dims(params)
        self._join()
        self.renderImage = RenderImage(w, h)
        fb = self.renderImage.framebuffer()
        self._check(self._lib.rt_resolve(self._ctx, C.byref(params), C.c_void_p(d_accum_ptr), C.byref(fb)), "rt_resolve")
        return self.renderImage

    def reduce_accumulators(self, nccl_comm, d_accum_ptr, width, height, root=0, nccl_reduce_fn=None):
        """rt_reduce_accumulators: in-place RCCL sum-reduce of a device accumulator (for hosts without torch.distributed)."""
        self._check(self._lib.rt_reduce_accumulators(self._ctx, C.c_void_p(nccl_comm), C.c_void_p(nccl_reduce_fn), C.c_void_p(d_accum_ptr),
                                                     width, height, root), "rt_reduce_accumulators")

    def set_option(self, name, value):
        """Developer knobs of this context (rt_set_option): tlas_min, gather_hints, frame_gb, fan_scale, timing."""
        self._check(self._lib.rt_set_option(self._ctx, name.encode(), float(value)), "rt_set_option")

    def debug_philox(self, counters, key_lo, key_hi):
        """rt_debug_philox: Philox4x32-10 of (n, 4) uint32 counters on the device -> (words (n, 4) uint32, units (n, 4) float32)."""
        counters = np.ascontiguousarray(counters, dtype=np.uint32).reshape(-1, 4)
        words = np.zeros_like(counters)
        units = np.zeros(counters.shape, np.float32)
        self._check(self._lib.rt_debug_philox(self._ctx, counters.ctypes.data, key_lo, key_hi, len(counters), words.ctypes.data,
                                              units.ctypes.data), "rt_debug_philox")
        return words, units

    def alloc_count(self):
        n = C.c_uint64()
        self._check(self._lib.rt_debug_alloc_count(self._ctx, C.byref(n)), "rt_debug_alloc_count")
        return n.value

    def counters(self, enable=None, reset=False):
        if enable is not None:
            self._check(self._lib.rt_counters_enable(self._ctx, int(enable)), "rt_counters_enable")
        if reset:
            self._check(self._lib.rt_counters_reset(self._ctx), "rt_counters_reset")
        c = rt_counters()
        self._check(self._lib.rt_counters_get(self._ctx, C.byref(c)), "rt_counters_get")
        return c.as_dict()

    def profile(self, enable=None, reset=False):
        if enable is not None:
            self._check(self._lib.rt_profile_enable(self._ctx, int(enable)), "rt_profile_enable")
        if reset:
            self._check(self._lib.rt_profile_reset(self._ctx), "rt_profile_reset")
        p = rt_profile()
        self._check(self._lib.rt_profile_get(self._ctx, C.byref(p)), "rt_profile_get")
        return {n: {"ms": p.ms[i], "launches": int(p.launches[i]), "items": int(p.items[i])}
                for i, n in enumerate(_ffi.RT_K_NAMES)}

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.rt_context_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
